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

}  // namespace hopmi
