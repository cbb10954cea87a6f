// Scaled fp16 hi/lo arithmetic on the gfx950 matrix cores (round 5: every hand-written contraction of the hot path).
//
// An fp32 value x is carried as two fp16 numbers of x s (s a power of two, exact): hi = fp16(x s), lo = fp16(x s - hi)
// (round to nearest even; x s - hi is exact in fp32).  11 + 11 significand bits: |x s - hi - lo| <= max(2^-22 |x s|, 2^-25)
// -- the second bound is the subnormal floor of fp16 (gfx950's MFMA does not flush fp16 subnormals).  A product is taken as
//     a b  ~=  a_hi b_hi + a_lo b_hi + a_hi b_lo                        (the dropped a_lo b_lo term is <= 2^-22 |a b|)
// with three v_mfma_f32_16x16x32_f16 (the bf16 instruction's rate, fp32 accumulation; the 22-bit partial products are exact in
// the accumulator): a few 2^-23 per product, i.e. BELOW what the fp32 accumulation of a K >= 64 dot product leaves --
// fp32-equivalent (tests/test_gpu_parity.py::test_*_vs_float64: error against float64 within 4 x plain fp32's), where the bf16
// split of rounds 2-4 (bf16_dev.h: 8 + 8 bits) was good to 2^-16.  What fp16 lacks is exponent range (5 bits): every operand
// gets a power-of-two scale that puts its largest magnitude near 2^14 -- per row where rows are independent outputs (weights:
// per output channel; activations: per tile row), per tensor / per tile where the index is contracted over, or a fixed one where
// the operand is bounded by construction (gate activations, probabilities, GRU states: |x| <= 1 -> 2^14).  An element smaller
// than 2^-17 of its scale group's maximum keeps an ABSOLUTE error <= 2^-39 of that maximum.  The accumulator is multiplied by
// the inverse scales (exact) in the epilogue.
//
// RULE: only a power of two may multiply a value in front of a split.  hipcc (fp-contract=fast, the HIP default) turns
// `fp16(x * c)` into a single-rounding v_fma_mix for the lo part's subtrahend while the stored hi part is the fp32 product
// converted (two roundings): with an inexact product the two disagree by an fp16 ulp now and then and hi + lo is off by 2^-11 of
// that element (found in round 5 with the softmax scale folded into Q: scores good to 6e-5 instead of 1e-7).  A power of two makes
// the product exact and both paths agree; any other factor (softmax scale, dropout scale) rides on the accumulator's epilogue
// factor.  split2h additionally takes its inputs through an empty asm, so that hi and lo are both derived from the SAME fp32
// register whatever the caller multiplied.
#pragma once
#include "bf16_dev.h"

namespace hopmi {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// the three-term product of (scaled) split operands; same issue order as bf16_dev.h: the two cross terms first
__device__ __forceinline__ f32x4 mfma_h3(u32x4 a_hi, u32x4 a_lo, u32x4 b_hi, u32x4 b_lo, f32x4 c) {
  c = mfma_f16(a_lo, b_hi, c);
  c = mfma_f16(a_hi, b_lo, c);
  return mfma_f16(a_hi, b_hi, c);
}

// ... when a lo part is known to be zero (an operand that was bf16 to begin with -- 8 significand bits fit fp16's 11 whenever
// the scaled value is a normal fp16 number): its term is not issued
template <bool A_LO, bool B_LO>
__device__ __forceinline__ f32x4 mfma_h(u32x4 a_hi, u32x4 a_lo, u32x4 b_hi, u32x4 b_lo, f32x4 c) {
  if (A_LO) c = mfma_f16(a_lo, b_hi, c);
  if (B_LO) c = mfma_f16(a_hi, b_lo, c);
  return mfma_f16(a_hi, b_hi, c);
}

// two (already scaled) floats -> {packed fp16 hi pair, packed fp16 lo pair}; element 0 in the low half (memory order)
// (`line`: the caller's source line, for the diagnostic build's status word -- common.h, split_check)
__device__ __forceinline__ u32x2 split2h(float a, float b, int line = __builtin_LINE()) {
  // (the values as fp32 registers: whatever produced them is rounded to fp32 first -- see RULE above; costs no instruction)
  asm("" : "+v"(a), "+v"(b));
  const f16x2 hi = {(_Float16)a, (_Float16)b};
  split_check(a, hi[0], line);
  split_check(b, hi[1], line);
  const f16x2 lo = {(_Float16)(a - (float)hi[0]), (_Float16)(b - (float)hi[1])};
  return u32x2{__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo)};
}

__device__ __forceinline__ Split4 split4h(float a, float b, float c, float d, int line = __builtin_LINE()) {
  const u32x2 p = split2h(a, b, line), r = split2h(c, d, line);
  return Split4{u32x2{p[0], r[0]}, u32x2{p[1], r[1]}};
}

__device__ __forceinline__ Split8 split8h(float4 a, float4 b, int line = __builtin_LINE()) {
  const u32x2 p0 = split2h(a.x, a.y, line), p1 = split2h(a.z, a.w, line), p2 = split2h(b.x, b.y, line), p3 = split2h(b.z, b.w, line);
  return Split8{u32x4{p0[0], p1[0], p2[0], p3[0]}, u32x4{p0[1], p1[1], p2[1], p3[1]}};
}
__device__ __forceinline__ Split8 split8h(float4 a, float4 b, float s, int line = __builtin_LINE()) {
  return split8h(make_float4(a.x * s, a.y * s, a.z * s, a.w * s), make_float4(b.x * s, b.y * s, b.z * s, b.w * s), line);
}

// packed fp16 pair -> the two floats
__device__ __forceinline__ float h_lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
__device__ __forceinline__ float h_hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }

// (hi + lo) * inv of 4 consecutive values
__device__ __forceinline__ float4 join4h(u32x2 hi, u32x2 lo, float inv) {
  return make_float4((h_lo(hi[0]) + h_lo(lo[0])) * inv, (h_hi(hi[0]) + h_hi(lo[0])) * inv, (h_lo(hi[1]) + h_lo(lo[1])) * inv,
                     (h_hi(hi[1]) + h_hi(lo[1])) * inv);
}

// power-of-two scale (as a float) for a group whose largest magnitude is `m` (>= 0; NaN / infinity / zero -> 1): m * s in [2^14, 2^15)
__device__ __forceinline__ float scale_for_absmax(float m) { return __uint_as_float(scale_bits_for_max(__float_as_uint(m) & 0x7fffffffu)); }
__device__ __forceinline__ float inv_pow2(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }
__device__ __forceinline__ float absmax4(float m, float4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
constexpr float H_UNIT_SCALE = 16384.f;            // the fixed scale of operands bounded by 1 in magnitude (gates, probabilities, GRU states)
constexpr float H_UNIT_INV = 1.f / 16384.f;

// ---- per-layer WaveNet weight images (hopmi_wn_prepare_weights): the fragment layout of bf16_dev.h with fp16 hi / lo parts of
// W[o][:] s_o, one power-of-two scale per output channel o and matrix, followed by the inverse scales:
//   floats [0, 64) 1 / s of the filter conv's rows, [64, 128) of the gate conv's, [128, 192) of Wm's
constexpr int WIMG_SCALE_UNITS = 3 * 64 * 4 / 16;                          // 48 x 16 B
constexpr int WIMGH_UNITS = WIMG_UNITS + WIMG_SCALE_UNITS;                 // 7216 x 16 B per layer

}  // namespace hopmi
