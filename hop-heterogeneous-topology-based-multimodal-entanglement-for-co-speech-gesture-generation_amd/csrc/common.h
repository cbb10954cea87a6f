// Shared device/host helpers for libhopmi (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "hopmi.h"

namespace hopmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: exact-f32 matrix FMA, D = A(16x4) * B(4x16) + C.
// Lane l = 16*q + i supplies A[i][k=q] and B[k=q][j=i]; holds D[4*q + r][j=i] in reg r.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int C = HOPMI_C;        // 64 channels
constexpr int K3 = 3 * C;         // 192 = [x | xA1 | xA2]
constexpr int LDH = K3 + 4;       // LDS row stride of an [rows][192] tile: 16-B aligned rows (b128 operand
                                  // reads), 196 = 4 (mod 64) so 16 consecutive rows land on 16 distinct
                                  // 16-B bank slots
constexpr int LDD = C + 4;        // same for a [rows][64] tile (68 = 4 mod 64)

void set_error(const char* fmt, ...);
int check_launch(const char* what);
int env_int(const char* name, int dflt);   // cached environment knob (api.hip)

inline int ceil_to(int x, int m) { return (x + m - 1) / m * m; }

// ---- operand scales of the fp16 hi/lo GEMM form (gemm.hip; also written by the kernels that produce its A operand)
// power-of-two scale for a row / tensor whose largest magnitude has the bits `m` (sign cleared): max * s in [2^14, 2^15); 1 for a
// non-finite one (a NaN or infinity then travels through the products as itself); the LARGEST scale (2^123) for an all-zero /
// subnormal one -- "no magnitude to protect": whoever takes the minimum over a set of rows' scales (gemm_tn.hip: the operand's one
// scale when the rows are the contraction index) is then not held back by its empty rows
__device__ __forceinline__ unsigned scale_bits_for_max(unsigned m) {
  const int e = (int)(m >> 23);
  const int sb = e == 255 ? 127 : (e == 0 ? 250 : min(268 - e, 250));
  return (unsigned)sb << 23;
}
__device__ __forceinline__ float inv_scale(unsigned scale_bits) { return __uint_as_float((254u << 23) - scale_bits); }
__device__ __forceinline__ unsigned abs_bits_max4(unsigned m, float4 v) {
  return max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
             max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
}
// lane 0 of a wave that has reduced `m` over the row writes the row's scale pair: scales[row] = s, scales[M + row] = 1 / s
__device__ __forceinline__ void store_row_scale(float* __restrict__ scales, int M, int row, unsigned m) {
  const unsigned sb = scale_bits_for_max(m);
  scales[row] = __uint_as_float(sb);
  scales[M + row] = inv_scale(sb);
}
// ---- diagnostic build only (-DHOPMI_CHECK_SPLIT: `make dbg` -> libhopmi_dbg.so, loaded through HOPMI_LIB): every conversion of a
// scaled fp32 value to its fp16 hi part reports into a device status buffer when the hi part is infinity / NaN --
//   words [0..3]: the INPUT was finite, i.e. the operand's scale let a value past fp16's range (the scale scheme of that site is
//                 wrong for this data): [0] = (file id << 16 | source line) of the first such conversion in stream order,
//                 [1] = how many, [2] = bits of the first offending (scaled) fp32 value;
//   words [4..7]: the input was already infinity / NaN (where a non-finite value first ENTERED a split), same layout.
// File ids: 1 gemm.hip, 2 gemm_tn.hip, 3 elementwise.hip, 4 attn.hip, 5 bert_attn.hip, 6 gru.hip, 7 wavenet.hip,
// 8 wavenet_stack.hip.  The buffer is registered per translation unit (hopmi_debug_set_split_status_<file>); ops.split_status()
// reads it.  The production build compiles the check away (and carries no such symbol).
#ifdef HOPMI_CHECK_SPLIT
#ifndef HOPMI_FILE_ID
#define HOPMI_FILE_ID 0                      // (a translation unit without fp16 splits)
#endif
static __device__ unsigned* g_split_status = nullptr;
__device__ __forceinline__ void split_check(float x, _Float16 hi, int line) {
  const unsigned short hb = __builtin_bit_cast(unsigned short, hi);
  if ((hb & 0x7c00u) != 0x7c00u) return;
  unsigned* w = g_split_status;
  if (w == nullptr) return;
  w += (fabsf(x) <= 3.4028235e38f) ? 0 : 4;
  if (atomicCAS(&w[0], 0u, ((unsigned)HOPMI_FILE_ID << 16) | (unsigned)line) == 0u) w[2] = __float_as_uint(x);
  atomicAdd(&w[1], 1u);
}
#define HOPMI_SPLIT_STATUS_SETTER(name)                                                                             \
  extern "C" int hopmi_debug_set_split_status_##name(unsigned* p) {                                                 \
    return hipMemcpyToSymbol(HIP_SYMBOL(hopmi::g_split_status), &p, sizeof(p)) == hipSuccess ? 0 : -1;               \
  }
#else
__device__ __forceinline__ void split_check(float, _Float16, int) {}
#define HOPMI_SPLIT_STATUS_SETTER(name)
#endif

__device__ __forceinline__ unsigned wave_max_u32(unsigned m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  return m;
}

}  // namespace hopmi
