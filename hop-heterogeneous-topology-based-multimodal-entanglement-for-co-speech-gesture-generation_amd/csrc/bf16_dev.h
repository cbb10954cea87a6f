// Split-bf16 arithmetic on the gfx950 matrix cores: an fp32 value x is carried as two bf16 numbers, hi = bf16(x) and
// lo = bf16(x - hi), so that hi + lo = x up to 2^-17 |x|; a product of two such numbers is taken as
//     a b  ~=  a_hi b_hi + a_lo b_hi + a_hi b_lo                       (the dropped a_lo b_lo term is < 2^-16 |a b|)
// with three v_mfma_f32_16x16x32_bf16 (fp32 accumulation), which run at 16x the rate of the exact-fp32
// v_mfma_f32_16x16x4_f32: 5.3x the fp32 matrix rate at ~1.5e-5 relative accuracy per product, same exponent range as
// fp32 (unlike an fp16 split).  Measured against the fp32 oracle in tests/test_gpu_parity.py (1e-3 bar).
#pragma once
#include "common.h"

namespace hopmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// v_mfma_f32_16x16x32_bf16: D = A(16x32) B(32x16) + C.  Lane l = 16 q + n supplies A[i = n][k = 8 q + e] and
// B[k = 8 q + e][j = n] in element e = 0..7 of its operand; holds D[i = 4 q + r][j = n] in register r.
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the three-term product of split operands
__device__ __forceinline__ f32x4 mfma_split3(u32x4 a_hi, u32x4 a_lo, u32x4 b_hi, u32x4 b_lo, f32x4 c) {
  c = mfma_bf16(a_lo, b_hi, c);
  c = mfma_bf16(a_hi, b_lo, c);
  return mfma_bf16(a_hi, b_hi, c);
}

// the same product when a lo part is known to be zero (an operand that was bf16 to begin with): its term is not issued;
// <true, true> issues mfma_split3's three terms in mfma_split3's order
template <bool A_LO, bool B_LO>
__device__ __forceinline__ f32x4 mfma_split(u32x4 a_hi, u32x4 a_lo, u32x4 b_hi, u32x4 b_lo, f32x4 c) {
  if (A_LO) c = mfma_bf16(a_lo, b_hi, c);
  if (B_LO) c = mfma_bf16(a_hi, b_lo, c);
  return mfma_bf16(a_hi, b_hi, c);
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even)
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// two floats -> {packed hi pair, packed lo pair}; element 0 in the low half (memory order)
__device__ __forceinline__ u32x2 split2(float a, float b) {
  const unsigned hi = pk_bf16(a, b);
  const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
  return u32x2{hi, pk_bf16(a - ah, b - bh)};
}

struct Split4 { u32x2 hi, lo; };                                      // 4 consecutive values (8 B + 8 B)
__device__ __forceinline__ Split4 split4(float a, float b, float c, float d) {
  const u32x2 p = split2(a, b), r = split2(c, d);
  return Split4{u32x2{p[0], r[0]}, u32x2{p[1], r[1]}};
}

struct Split8 { u32x4 hi, lo; };                                      // 8 consecutive values: one MFMA operand each
__device__ __forceinline__ Split8 split8(float4 a, float4 b) {
  const u32x2 p0 = split2(a.x, a.y), p1 = split2(a.z, a.w), p2 = split2(b.x, b.y), p3 = split2(b.z, b.w);
  return Split8{u32x4{p0[0], p1[0], p2[0], p3[0]}, u32x4{p0[1], p1[1], p2[1], p3[1]}};
}

// packed bf16 pair -> the two floats
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// hi + lo of 4 consecutive values
__device__ __forceinline__ float4 join4(u32x2 hi, u32x2 lo) {
  return make_float4(bf_lo(hi[0]) + bf_lo(lo[0]), bf_hi(hi[0]) + bf_hi(lo[0]), bf_lo(hi[1]) + bf_lo(lo[1]), bf_hi(hi[1]) + bf_hi(lo[1]));
}

// LDS images of split operands: rows of bf16, row strides (in 16-bit units) chosen so that the 16-byte operand reads of
// a wave (lane (n, q) reads 8 values at row n, column 8 q + 32 ks) are bank-conflict free: stride / 8 = 10 (mod 16)
// puts the 16 lanes of every ds_read_b128 service group on 16 distinct 16-byte bank slots.
constexpr int RS = 80;            // [rows][64]  activations (r^ at one tap): 160-byte rows
constexpr int HS = 208;           // [rows][192] u | uA1 | uA2: 416-byte rows

// ---- per-layer weight images (built once per forward pass by hopmi_wn_prepare_weights) ----------------------
// MFMA A-operand fragments, split, exactly as a wave loads them (one 16-byte unit per lane, 1 KiB per wave-instruction):
//   TCN:  unit ((((w*2 + gate)*4 + ks)*2 + part)*64 + lane):  W_gate[o = 16 w + n][c = 32 (ks&1) + 8 q + e][tap = ks>>1]
//   Wm :  WIMG_TCN_UNITS + (((w*6 + ks)*2 + part)*64 + lane): Wm[o = 16 w + n][k = 32 ks + 8 q + e]
// part 0 = hi, 1 = lo; lane = 16 q + n.
constexpr int WIMG_TCN_UNITS = 4 * 2 * 4 * 2 * 64;     // 4096
constexpr int WIMG_WM_UNITS = 4 * 6 * 2 * 64;          // 3072
constexpr int WIMG_UNITS = WIMG_TCN_UNITS + WIMG_WM_UNITS;   // 7168 x 16 B = 112 KiB per layer

}  // namespace hopmi
