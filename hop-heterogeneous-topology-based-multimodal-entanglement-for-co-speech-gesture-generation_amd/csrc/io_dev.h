// Typed 4-element loads / stores of activation tensors: fp32 (16 bytes) or bf16 (8 bytes, widened to / rounded from fp32
// in registers -- all arithmetic stays fp32).  The `_dt` entry points of elementwise.hip / bert_attn.hip select the storage
// type of the tensors that sit between two GEMMs (dtype 0 = fp32, 1 = bf16): under bf16 autocast the GEMM outputs are read
// and the GEMM inputs written in bf16 directly, which removes the cast kernels around every operator and halves their traffic.
#pragma once
#include "common.h"

namespace hopmi {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float4 ld4(const __bf16* p) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const u2 w = *reinterpret_cast<const u2*>(p);
  return make_float4(__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16),
                     __uint_as_float(w[1] & 0xffff0000u));
}
__device__ __forceinline__ void st4(__bf16* p, float4 v) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const bf2 a = {(__bf16)v.x, (__bf16)v.y}, b = {(__bf16)v.z, (__bf16)v.w};       // round to nearest even (v_cvt_pk_bf16_f32)
  *reinterpret_cast<u2*>(p) = u2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}

__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }

constexpr int HOPMI_F32 = 0, HOPMI_BF16 = 1;

}  // namespace hopmi
