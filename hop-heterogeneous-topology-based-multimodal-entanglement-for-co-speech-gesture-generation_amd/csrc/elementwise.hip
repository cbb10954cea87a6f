// Fused element-wise epilogues of the frozen BERT block on the HOP path (reference: HOP.py:204 calls HF
// BertModel; the ops are transformers' BertSelfOutput / BertOutput / BertIntermediate):
//
//   bias_gelu        out = gelu_erf(x + b)                                  (BertIntermediate)
//   bias_drop_res_ln out = LayerNorm(dropout(x + b) + residual) * g + beta  (BertSelfOutput, BertOutput; eps 1e-12)
//
// and their backward w.r.t. the ACTIVATIONS only (the LLM is frozen, HOP.py:90-91, but gradients flow through
// it to the align / reprogramming / mapping layers).  Both are pure HBM streamers: 16-B loads, one pass.
// LayerNorm rows (D <= 1024, D % 4 == 0; 768 for BERT-base) are one wave each: the row lives in registers,
// mean/variance by DPP + cross-row shuffles, so x is read once and out written once.  Dropout uses the same
// stateless hash as the attention kernel (seed, row, column) so the backward regenerates the mask.
#define HOPMI_FILE_ID 3          // (diagnostic build: common.h, split_check)
#include "f16_dev.h"
#include "io_dev.h"

namespace hopmi {

__device__ __forceinline__ unsigned ew_hash(unsigned seed, unsigned row, unsigned col) {
  unsigned x = seed ^ (row * 0x9E3779B1u) ^ (col * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ float gelu_erf_(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad_(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * v * v);
  return cdf + v * pdf;
}

// x [M][N] (N % 4 == 0), bias [N]
template <typename T>
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                            T* __restrict__ out, size_t n4, int N4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = ld4(x + 4 * i);
    const float4 b = reinterpret_cast<const float4*>(bias)[i % N4];
    st4(out + 4 * i, make_float4(gelu_erf_(v.x + b.x), gelu_erf_(v.y + b.y), gelu_erf_(v.z + b.z), gelu_erf_(v.w + b.w)));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bias_gelu_bwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                            const T* __restrict__ dy, T* __restrict__ dx, size_t n4, int N4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = ld4(x + 4 * i);
    const float4 b = reinterpret_cast<const float4*>(bias)[i % N4];
    const float4 g = ld4(dy + 4 * i);
    st4(dx + 4 * i, make_float4(g.x * gelu_erf_grad_(v.x + b.x), g.y * gelu_erf_grad_(v.y + b.y),
                                g.z * gelu_erf_grad_(v.z + b.z), g.w * gelu_erf_grad_(v.w + b.w)));
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

constexpr int LN_MAX4 = 4;            // float4 per lane: D <= 64 * 4 * 4 = 1024

// 2-norm of the row a wave holds in registers, rounded UP (it feeds the a-priori bound of the image-emitting GEMM epilogue behind
// this operator, gemm.hip: |out| <= ||row|| ||w|| + |b|), given the row's largest magnitude `amax` (bits, reduced over the wave).
// The squares are taken of the row times the power of two that puts `amax` near 2^14: a plain sum of squares underflows to ZERO for
// a row whose elements are below ~1e-19 -- gradient rows behind a saturated GRU are 1e-30 and smaller -- and a zero bound tells the
// epilogue "nothing to protect": it scaled such a row's (tiny, non-zero) products by 2^123, past fp16's range, and wrote infinity
// hi parts / NaN lo parts into the operand image (round 5's NaN in the bench regime: found with the diagnostic build's status word,
// row_norm = 0.0 against a product of 9.6e-33).  Scaled, the sum is exact in range: (2^15)^2 x 1024 elements = 2^40.
__device__ __forceinline__ float ln_row_norm(const float4 (&v)[LN_MAX4], int D4, int lane, unsigned amax) {
  const unsigned sb = scale_bits_for_max(amax);
  const float sc = __uint_as_float(sb);
  float nrm = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX4; ++k) {
    if (lane + 64 * k < D4) {
      const float a = v[k].x * sc, b = v[k].y * sc, c = v[k].z * sc, d = v[k].w * sc;
      nrm += a * a + b * b + c * c + d * d;
    }
  }
  return sqrtf(wave_sum(nrm)) * 1.0000005f * inv_scale(sb);
}

// The row a wave holds in registers, times its power-of-two scale, as the fp16 hi / lo images [2][M][D] that hopmi_gemm_f16x2_ab
// reads by LDS-DMA (hopmi_rows_image_f16's layout): the GEMM behind this operator then needs neither the split in its k-loop nor a
// pass of its own over the activations (round 5).
__device__ __forceinline__ void ln_store_image(unsigned* __restrict__ image, const float4 (&v)[LN_MAX4], int M, int D4, int row, int lane, float sc) {
  // tile-blocked layout (gemm.hip f16_blk): ((row / 128) * KB + k / 32) * 4096 + (row % 128) * 32 + k % 32 halves, part images of
  // ceil128(M) rows
  _Float16* hi = reinterpret_cast<_Float16*>(image);
  _Float16* lo = hi + (size_t)((M + 127) / 128 * 128) * D4 * 4;
  const int KB = D4 >> 3;
  const size_t base = (size_t)(row >> 7) * KB * 4096 + (size_t)(row & 127) * 32;
#pragma unroll
  for (int k = 0; k < LN_MAX4; ++k) {
    const int c4 = lane + 64 * k;
    if (c4 < D4) {
      const Split4 sp = split4h(v[k].x * sc, v[k].y * sc, v[k].z * sc, v[k].w * sc);
      const size_t at = base + (size_t)(c4 >> 3) * 4096 + 4 * (c4 & 7);
      *reinterpret_cast<u32x2*>(hi + at) = sp.hi;
      *reinterpret_cast<u32x2*>(lo + at) = sp.lo;
    }
  }
}

// one wave per row: z = dropout(x + bias) + res[row % res_rows] ; out = (z - mean) * rstd * gamma + beta
// saves z's normalised form xhat and rstd for the backward.
// `x` is typed (the GEMM output); `out` stays fp32 (it is the next block's residual) and `out_t` (nullable) receives the same
// values in x's type: the next GEMM's input.
template <typename T>
__global__ __launch_bounds__(256) void bias_drop_res_ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                                   const float* __restrict__ res, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ out,
                                                                   T* __restrict__ out_t,
                                                                   float* __restrict__ xhat, float* __restrict__ rstd_out, int M,
                                                                   int D, int res_rows, float eps, unsigned drop_thresh,
                                                                   float drop_scale, unsigned seed, const unsigned* __restrict__ seed_dev,
                                                                   float* __restrict__ row_scales, unsigned* __restrict__ image,
                                                                   float* __restrict__ row_norms) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int D4 = D >> 2;
  unsigned amax = 0;                            // max |out| of the row (row_scales: the next GEMM's fp16-form operand scale; row_norms)
  float4 z[LN_MAX4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX4; ++k) {
    const int c4 = lane + 64 * k;
    z[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < D4) {
      float4 v = ld4(x + ((size_t)row * D4 + c4) * 4);
      const float4 b = reinterpret_cast<const float4*>(bias)[c4];
      const float4 r = reinterpret_cast<const float4*>(res)[(size_t)(row % res_rows) * D4 + c4];
      v = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
      if (drop_thresh) {
        const unsigned col = 4 * c4;
        v.x = ew_hash(seed, row, col) >= drop_thresh ? v.x * drop_scale : 0.f;
        v.y = ew_hash(seed, row, col + 1) >= drop_thresh ? v.y * drop_scale : 0.f;
        v.z = ew_hash(seed, row, col + 2) >= drop_thresh ? v.z * drop_scale : 0.f;
        v.w = ew_hash(seed, row, col + 3) >= drop_thresh ? v.w * drop_scale : 0.f;
      }
      z[k] = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
      s += z[k].x + z[k].y + z[k].z + z[k].w;
    }
  }
  const float mean = wave_sum(s) / D;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX4; ++k) {
    if (lane + 64 * k < D4) {
      const float a = z[k].x - mean, b = z[k].y - mean, c = z[k].z - mean, d = z[k].w - mean;
      s2 += a * a + b * b + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(s2) / D + eps);
#pragma unroll
  for (int k = 0; k < LN_MAX4; ++k) {
    const int c4 = lane + 64 * k;
    if (c4 < D4) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c4], be = reinterpret_cast<const float4*>(beta)[c4];
      const float4 h = make_float4((z[k].x - mean) * rstd, (z[k].y - mean) * rstd, (z[k].z - mean) * rstd, (z[k].w - mean) * rstd);
      const float4 o = make_float4(h.x * g.x + be.x, h.y * g.y + be.y, h.z * g.z + be.z, h.w * g.w + be.w);
      reinterpret_cast<float4*>(out)[(size_t)row * D4 + c4] = o;
      if (out_t != nullptr) st4(out_t + ((size_t)row * D4 + c4) * 4, o);
      if (xhat != nullptr) reinterpret_cast<float4*>(xhat)[(size_t)row * D4 + c4] = h;
      amax = abs_bits_max4(amax, o);
      z[k] = o;                                 // (kept for the image and the norm below)
    }
  }
  if (rstd_out != nullptr && lane == 0) rstd_out[row] = rstd;
  if (row_scales != nullptr || row_norms != nullptr) amax = wave_max_u32(amax);
  if (row_scales != nullptr) {
    if (lane == 0) store_row_scale(row_scales, M, row, amax);
    if (image != nullptr) ln_store_image(image, z, M, D4, row, lane, __uint_as_float(scale_bits_for_max(amax)));
  }
  if (row_norms != nullptr) {
    const float nrm = ln_row_norm(z, D4, lane, amax);
    if (lane == 0) row_norms[row] = nrm;
  }
}

// dz = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)), dxh = dout * gamma ; dres = dz ; dx = dz o dropout mask
// `dout_t` (nullable): the gradient that arrived through the typed copy of the output, added to `dout`; `dx` is typed.
template <typename T>
__global__ __launch_bounds__(256) void bias_drop_res_ln_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ dout_t,
                                                                   const float* __restrict__ xhat,
                                                                   const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                                   T* __restrict__ dx, float* __restrict__ dres, int M, int D,
                                                                   unsigned drop_thresh, float drop_scale, unsigned seed, const unsigned* __restrict__ seed_dev,
                                                                   float* __restrict__ row_scales, unsigned* __restrict__ image,
                                                                   float* __restrict__ row_norms) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int D4 = D >> 2;
  unsigned amax = 0;                            // max |dx| of the row (row_scales: the next GEMM's fp16-form operand scale; row_norms)
  float4 dh[LN_MAX4], xh[LN_MAX4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX4; ++k) {
    const int c4 = lane + 64 * k;
    dh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    xh[k] = dh[k];
    if (c4 < D4) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
      float4 d = reinterpret_cast<const float4*>(dout)[(size_t)row * D4 + c4];
      if (dout_t != nullptr) {
        const float4 e = ld4(dout_t + ((size_t)row * D4 + c4) * 4);
        d = make_float4(d.x + e.x, d.y + e.y, d.z + e.z, d.w + e.w);
      }
      xh[k] = reinterpret_cast<const float4*>(xhat)[(size_t)row * D4 + c4];
      dh[k] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
      s1 += dh[k].x + dh[k].y + dh[k].z + dh[k].w;
      s2 += dh[k].x * xh[k].x + dh[k].y * xh[k].y + dh[k].z * xh[k].z + dh[k].w * xh[k].w;
    }
  }
  const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D, rstd = rstd_in[row];
#pragma unroll
  for (int k = 0; k < LN_MAX4; ++k) {
    const int c4 = lane + 64 * k;
    if (c4 < D4) {
      float4 dz = make_float4(rstd * (dh[k].x - m1 - xh[k].x * m2), rstd * (dh[k].y - m1 - xh[k].y * m2),
                              rstd * (dh[k].z - m1 - xh[k].z * m2), rstd * (dh[k].w - m1 - xh[k].w * m2));
      reinterpret_cast<float4*>(dres)[(size_t)row * D4 + c4] = dz;
      if (drop_thresh) {
        const unsigned col = 4 * c4;
        dz.x = ew_hash(seed, row, col) >= drop_thresh ? dz.x * drop_scale : 0.f;
        dz.y = ew_hash(seed, row, col + 1) >= drop_thresh ? dz.y * drop_scale : 0.f;
        dz.z = ew_hash(seed, row, col + 2) >= drop_thresh ? dz.z * drop_scale : 0.f;
        dz.w = ew_hash(seed, row, col + 3) >= drop_thresh ? dz.w * drop_scale : 0.f;
      }
      st4(dx + ((size_t)row * D4 + c4) * 4, dz);
      amax = abs_bits_max4(amax, dz);
      dh[k] = dz;                               // (kept for the image and the norm below)
    }
  }
  if (row_scales != nullptr || row_norms != nullptr) amax = wave_max_u32(amax);
  if (row_scales != nullptr) {
    if (lane == 0) store_row_scale(row_scales, M, row, amax);
    if (image != nullptr) ln_store_image(image, dh, M, D4, row, lane, __uint_as_float(scale_bits_for_max(amax)));
  }
  if (row_norms != nullptr) {
    const float nrm = ln_row_norm(dh, D4, lane, amax);
    if (lane == 0) row_norms[row] = nrm;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// BatchNorm1d of the discriminator's pre_conv (multimodal_context_net.py:226-234) on channels-last rows x [M = B*T][C <= 64]:
// training-mode batch statistics over the M rows, running-statistics update, affine map -- and its backward -- as ONE launch
// each.  The tensors are tiny (M = 4096 rows of 8 or 16 channels): as tensor operations the layer was ~12 launches forward and
// ~15 backward at the launch floor; here one workgroup walks the 256 KB twice.  Fixed summation order (thread-strided rows, then
// the row groups in index order, in double): bitwise reproducible.
constexpr int BN_T = 1024;

__global__ __launch_bounds__(BN_T) void bn_cl_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ rmean,
                                                         float* __restrict__ rvar, float* __restrict__ y, float* __restrict__ save,
                                                         int M, int C, float eps, float momentum, int training) {
  __shared__ double red[2][BN_T];
  __shared__ float sc_sh[2][64];
  const int tid = threadIdx.x, R = BN_T / C, r = tid / C, c = tid - r * C;
  const bool live = r < R;
  if (training) {
    double s1 = 0.0, s2 = 0.0;
    if (live)
      for (int m0 = r; m0 < M; m0 += 8 * R) {       // 8 rows in flight per thread (one workgroup: latency, not bandwidth)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (m0 + u * R < M) ? x[(size_t)(m0 + u * R) * C + c] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += (double)v[u]; s2 += (double)v[u] * (double)v[u]; }
      }
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    if (tid < C) {
      double a = 0.0, b = 0.0;
      for (int k = 0; k < R; ++k) { a += red[0][k * C + tid]; b += red[1][k * C + tid]; }
      const double mean = a / M;
      double var = b / M - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)eps));
      const float scale = gamma[tid] * rstd;
      sc_sh[0][tid] = scale;
      sc_sh[1][tid] = beta[tid] - (float)mean * scale;
      if (save != nullptr) { save[tid] = (float)mean; save[C + tid] = rstd; }
      if (rmean != nullptr) {
        rmean[tid] = (1.f - momentum) * rmean[tid] + momentum * (float)mean;
        rvar[tid] = (1.f - momentum) * rvar[tid] + momentum * (float)(var * ((double)M / (double)(M > 1 ? M - 1 : 1)));
      }
    }
  } else if (tid < C) {
    const float scale = gamma[tid] * rsqrtf(rvar[tid] + eps);
    sc_sh[0][tid] = scale;
    sc_sh[1][tid] = beta[tid] - rmean[tid] * scale;
  }
  __syncthreads();
  if (y != nullptr && live) {
    const float scale = sc_sh[0][c], shift = sc_sh[1][c];
    for (int m0 = r; m0 < M; m0 += 8 * R) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (m0 + u * R < M) ? x[(size_t)(m0 + u * R) * C + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (m0 + u * R < M) y[(size_t)(m0 + u * R) * C + c] = v[u] * scale + shift;
    }
  }
}

// dx = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)),  dgamma = sum dy xhat,  dbeta = sum dy   (xhat = (x - mean) rstd)
__global__ __launch_bounds__(BN_T) void bn_cl_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ gamma, const float* __restrict__ save,
                                                         float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         int M, int C) {
  __shared__ double red[2][BN_T];
  __shared__ float coef[3][64];
  const int tid = threadIdx.x, R = BN_T / C, r = tid / C, c = tid - r * C;
  const bool live = r < R;
  const float mean = live ? save[c] : 0.f, rstd = live ? save[C + c] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  if (live)
    for (int m0 = r; m0 < M; m0 += 8 * R) {
      float g[8], v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool in = m0 + u * R < M;
        g[u] = in ? dy[(size_t)(m0 + u * R) * C + c] : 0.f;
        v[u] = in ? x[(size_t)(m0 + u * R) * C + c] : mean;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s1 += (double)g[u]; s2 += (double)g[u] * (double)((v[u] - mean) * rstd); }
    }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  if (tid < C) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < R; ++k) { a += red[0][k * C + tid]; b += red[1][k * C + tid]; }
    dbeta[tid] = (float)a;
    dgamma[tid] = (float)b;
    coef[0][tid] = gamma[tid] * save[C + tid];
    coef[1][tid] = (float)(a / M);
    coef[2][tid] = (float)(b / M);
  }
  __syncthreads();
  if (dx != nullptr && live) {
    const float k0 = coef[0][c], k1 = coef[1][c], k2 = coef[2][c];
    for (int m0 = r; m0 < M; m0 += 8 * R) {
      float g[8], v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool in = m0 + u * R < M;
        g[u] = in ? dy[(size_t)(m0 + u * R) * C + c] : 0.f;
        v[u] = in ? x[(size_t)(m0 + u * R) * C + c] : mean;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (m0 + u * R < M) dx[(size_t)(m0 + u * R) * C + c] = k0 * (g[u] - k1 - (v[u] - mean) * rstd * k2);
    }
  }
}

static int ew_check(int M, int N, const char* what) {
  if (M <= 0 || N <= 0 || (N & 3)) { set_error("%s: need M > 0, N > 0, N %% 4 == 0 (M=%d N=%d)", what, M, N); return HOPMI_EINVAL; }
  return HOPMI_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Column sums of a row-major [M][N] matrix: the bias gradient db = sum over rows of dY of every trainable linear layer
// (torch.nn.functional.linear's backward; the library's generic reduction runs these at ~1.3 TB/s: 28 us for the
// 4352 x 2100 fp32 gradients of the GRU input projections, 25-32 of them per step).  Workgroup = 128 rows x 32 VEC columns:
// 32 column lanes x 8 row lanes, every thread's 16 row loads are independent (all in flight), the 8 row lanes are added in
// LDS in a fixed order, the row chunks by a second small launch in a fixed order -- bitwise reproducible.
constexpr int CS_ROWS = 128;

template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int M, int N, float* __restrict__ part) {
  __shared__ float red[8][32 * VEC];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + tx) * VEC;
  const int r0 = blockIdx.y * CS_ROWS;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  if (col < N) {
#pragma unroll
    for (int i = 0; i < CS_ROWS / 8; ++i) {
      const int r = r0 + ty + 8 * i;
      if (r < M) {
        const T* p = x + (size_t)r * N + col;
        if (VEC == 4) {
          const float4 t = ld4(p);
          acc[0] += t.x; acc[1 % VEC] += t.y; acc[2 % VEC] += t.z; acc[3 % VEC] += t.w;
        } else {
          acc[0] += (float)*p;
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) red[ty][tx * VEC + v] = acc[v];
  __syncthreads();
  if (ty == 0 && col < N) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float s = red[0][tx * VEC + v];
#pragma unroll
      for (int k = 1; k < 8; ++k) s += red[k][tx * VEC + v];
      part[(size_t)blockIdx.y * N + col + v] = s;
    }
  }
}

// chunk partials -> out: 32 columns x 8 chunk lanes per workgroup; a lane adds its chunks k = ty, ty + 8, ... in order (loads
// issued four at a time: a single serial chain of 34 dependent loads took 8.5 us), the 8 lanes are added in order in LDS
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nchunk, int N, float* __restrict__ out) {
  __shared__ float red[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  float s = 0.f;
  if (c < N) {
    int k = ty;
    for (; k + 24 < nchunk; k += 32) {
      const float a0 = part[(size_t)k * N + c], a1 = part[(size_t)(k + 8) * N + c];
      const float a2 = part[(size_t)(k + 16) * N + c], a3 = part[(size_t)(k + 24) * N + c];
      s = (((s + a0) + a1) + a2) + a3;
    }
    for (; k < nchunk; k += 8) s += part[(size_t)k * N + c];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < N) {
    float t = red[0][tx];
#pragma unroll
    for (int j = 1; j < 8; ++j) t += red[j][tx];
    out[c] = t;
  }
}

}  // namespace hopmi

using namespace hopmi;

static int ew_dtype_ok(const char* what, int dtype) {
  if (dtype != HOPMI_F32 && dtype != HOPMI_BF16) { set_error("%s: dtype %d (0 = fp32, 1 = bf16)", what, dtype); return HOPMI_EINVAL; }
  return HOPMI_OK;
}

extern "C" int hopmi_bias_gelu_fwd_dt(const void* x, const float* bias, void* out, int M, int N, int dtype, void* stream) {
  if (int e = ew_check(M, N, "hopmi_bias_gelu_fwd")) return e;
  if (int e = ew_dtype_ok("hopmi_bias_gelu_fwd_dt", dtype)) return e;
  if (!x || !bias || !out) { set_error("hopmi_bias_gelu_fwd: null pointer argument"); return HOPMI_EINVAL; }
  const size_t n4 = (size_t)M * N / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == HOPMI_BF16)
    hipLaunchKernelGGL(bias_gelu_fwd_kernel<__bf16>, dim3(grid), dim3(256), 0, st, static_cast<const __bf16*>(x), bias, static_cast<__bf16*>(out), n4, N / 4);
  else
    hipLaunchKernelGGL(bias_gelu_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(x), bias, static_cast<float*>(out), n4, N / 4);
  return check_launch("hopmi_bias_gelu_fwd");
}

extern "C" int hopmi_bias_gelu_bwd_dt(const void* x, const float* bias, const void* dy, void* dx, int M, int N, int dtype, void* stream) {
  if (int e = ew_check(M, N, "hopmi_bias_gelu_bwd")) return e;
  if (int e = ew_dtype_ok("hopmi_bias_gelu_bwd_dt", dtype)) return e;
  if (!x || !bias || !dy || !dx) { set_error("hopmi_bias_gelu_bwd: null pointer argument"); return HOPMI_EINVAL; }
  const size_t n4 = (size_t)M * N / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == HOPMI_BF16)
    hipLaunchKernelGGL(bias_gelu_bwd_kernel<__bf16>, dim3(grid), dim3(256), 0, st, static_cast<const __bf16*>(x), bias,
                       static_cast<const __bf16*>(dy), static_cast<__bf16*>(dx), n4, N / 4);
  else
    hipLaunchKernelGGL(bias_gelu_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(x), bias,
                       static_cast<const float*>(dy), static_cast<float*>(dx), n4, N / 4);
  return check_launch("hopmi_bias_gelu_bwd");
}

extern "C" int hopmi_bias_gelu_fwd(const float* x, const float* bias, float* out, int M, int N, void* stream) {
  return hopmi_bias_gelu_fwd_dt(x, bias, out, M, N, HOPMI_F32, stream);
}

extern "C" int hopmi_bias_gelu_bwd(const float* x, const float* bias, const float* dy, float* dx, int M, int N, void* stream) {
  return hopmi_bias_gelu_bwd_dt(x, bias, dy, dx, M, N, HOPMI_F32, stream);
}

extern "C" int hopmi_bias_dropout_residual_layernorm_fwd_im(const void* x, const float* bias, const float* res, int res_rows,
                                                            const float* gamma, const float* beta, float* out, void* out_t, float* xhat,
                                                            float* rstd, float* row_scales, void* image, float* row_norms, int M, int D,
                                                            float eps, float p_drop, unsigned seed, const unsigned* seed_dev, int dtype,
                                                            void* stream) {
  if (int e = ew_check(M, D, "hopmi_bias_dropout_residual_layernorm_fwd")) return e;
  if (image != nullptr && (!row_scales || D % 32)) { set_error("hopmi_bias_dropout_residual_layernorm_fwd_im: the image needs row_scales and D %% 32 == 0"); return HOPMI_EINVAL; }
  if (int e = ew_dtype_ok("hopmi_bias_dropout_residual_layernorm_fwd_dt", dtype)) return e;
  if (!x || !bias || !res || !gamma || !beta || !out) { set_error("hopmi_bias_dropout_residual_layernorm_fwd: null pointer argument"); return HOPMI_EINVAL; }
  if (D > 256 * LN_MAX4 || res_rows <= 0 || !(p_drop >= 0.f && p_drop < 1.f)) {
    set_error("hopmi_bias_dropout_residual_layernorm_fwd: D=%d (max %d), res_rows=%d, p_drop=%f", D, 256 * LN_MAX4, res_rows, p_drop);
    return HOPMI_EINVAL;
  }
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == HOPMI_BF16)
    hipLaunchKernelGGL(bias_drop_res_ln_fwd_kernel<__bf16>, dim3((M + 3) / 4), dim3(256), 0, st, static_cast<const __bf16*>(x), bias, res,
                       gamma, beta, out, static_cast<__bf16*>(out_t), xhat, rstd, M, D, res_rows, eps, thresh, dscale, seed, seed_dev, row_scales,
                       static_cast<unsigned*>(image), row_norms);
  else
    hipLaunchKernelGGL(bias_drop_res_ln_fwd_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, st, static_cast<const float*>(x), bias, res,
                       gamma, beta, out, static_cast<float*>(out_t), xhat, rstd, M, D, res_rows, eps, thresh, dscale, seed, seed_dev, row_scales,
                       static_cast<unsigned*>(image), row_norms);
  return check_launch("hopmi_bias_dropout_residual_layernorm_fwd");
}

extern "C" int hopmi_bias_dropout_residual_layernorm_fwd_rs(const void* x, const float* bias, const float* res, int res_rows,
                                                            const float* gamma, const float* beta, float* out, void* out_t, float* xhat,
                                                            float* rstd, float* row_scales, int M, int D, float eps, float p_drop,
                                                            unsigned seed, const unsigned* seed_dev, int dtype, void* stream) {
  return hopmi_bias_dropout_residual_layernorm_fwd_im(x, bias, res, res_rows, gamma, beta, out, out_t, xhat, rstd, row_scales, nullptr, nullptr, M, D,
                                                      eps, p_drop, seed, seed_dev, dtype, stream);
}

extern "C" int hopmi_bias_dropout_residual_layernorm_fwd_dt(const void* x, const float* bias, const float* res, int res_rows,
                                                            const float* gamma, const float* beta, float* out, void* out_t, float* xhat,
                                                            float* rstd, int M, int D, float eps, float p_drop, unsigned seed,
                                                            const unsigned* seed_dev, int dtype, void* stream) {
  return hopmi_bias_dropout_residual_layernorm_fwd_rs(x, bias, res, res_rows, gamma, beta, out, out_t, xhat, rstd, nullptr, M, D, eps, p_drop,
                                                      seed, seed_dev, dtype, stream);
}

extern "C" int hopmi_bias_dropout_residual_layernorm_bwd_im(const float* dout, const void* dout_t, const float* xhat, const float* rstd,
                                                            const float* gamma, void* dx, float* dres, float* row_scales, void* image,
                                                            float* row_norms, int M, int D, float p_drop, unsigned seed,
                                                            const unsigned* seed_dev, int dtype, void* stream) {
  if (int e = ew_check(M, D, "hopmi_bias_dropout_residual_layernorm_bwd")) return e;
  if (image != nullptr && (!row_scales || D % 32)) { set_error("hopmi_bias_dropout_residual_layernorm_bwd_im: the image needs row_scales and D %% 32 == 0"); return HOPMI_EINVAL; }
  if (int e = ew_dtype_ok("hopmi_bias_dropout_residual_layernorm_bwd_dt", dtype)) return e;
  if (!dout || !xhat || !rstd || !gamma || !dx || !dres) { set_error("hopmi_bias_dropout_residual_layernorm_bwd: null pointer argument"); return HOPMI_EINVAL; }
  if (D > 256 * LN_MAX4 || !(p_drop >= 0.f && p_drop < 1.f)) { set_error("hopmi_bias_dropout_residual_layernorm_bwd: D=%d p_drop=%f", D, p_drop); return HOPMI_EINVAL; }
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == HOPMI_BF16)
    hipLaunchKernelGGL(bias_drop_res_ln_bwd_kernel<__bf16>, dim3((M + 3) / 4), dim3(256), 0, st, dout, static_cast<const __bf16*>(dout_t), xhat,
                       rstd, gamma, static_cast<__bf16*>(dx), dres, M, D, thresh, dscale, seed, seed_dev, row_scales, static_cast<unsigned*>(image), row_norms);
  else
    hipLaunchKernelGGL(bias_drop_res_ln_bwd_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, st, dout, static_cast<const float*>(dout_t), xhat,
                       rstd, gamma, static_cast<float*>(dx), dres, M, D, thresh, dscale, seed, seed_dev, row_scales, static_cast<unsigned*>(image), row_norms);
  return check_launch("hopmi_bias_dropout_residual_layernorm_bwd");
}

extern "C" int hopmi_bias_dropout_residual_layernorm_bwd_rs(const float* dout, const void* dout_t, const float* xhat, const float* rstd,
                                                            const float* gamma, void* dx, float* dres, float* row_scales, int M, int D,
                                                            float p_drop, unsigned seed, const unsigned* seed_dev, int dtype, void* stream) {
  return hopmi_bias_dropout_residual_layernorm_bwd_im(dout, dout_t, xhat, rstd, gamma, dx, dres, row_scales, nullptr, nullptr, M, D, p_drop, seed,
                                                      seed_dev, dtype, stream);
}

extern "C" int hopmi_bias_dropout_residual_layernorm_bwd_dt(const float* dout, const void* dout_t, const float* xhat, const float* rstd,
                                                            const float* gamma, void* dx, float* dres, int M, int D, float p_drop,
                                                            unsigned seed, const unsigned* seed_dev, int dtype, void* stream) {
  return hopmi_bias_dropout_residual_layernorm_bwd_rs(dout, dout_t, xhat, rstd, gamma, dx, dres, nullptr, M, D, p_drop, seed, seed_dev, dtype,
                                                      stream);
}

extern "C" int hopmi_bias_dropout_residual_layernorm_fwd(const float* x, const float* bias, const float* res, int res_rows,
                                                         const float* gamma, const float* beta, float* out, float* xhat,
                                                         float* rstd, int M, int D, float eps, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream) {
  return hopmi_bias_dropout_residual_layernorm_fwd_dt(x, bias, res, res_rows, gamma, beta, out, nullptr, xhat, rstd, M, D, eps, p_drop, seed,
                                                      seed_dev, HOPMI_F32, stream);
}

extern "C" int hopmi_bias_dropout_residual_layernorm_bwd(const float* dout, const float* xhat, const float* rstd,
                                                         const float* gamma, float* dx, float* dres, int M, int D,
                                                         float p_drop, unsigned seed, const unsigned* seed_dev, void* stream) {
  return hopmi_bias_dropout_residual_layernorm_bwd_dt(dout, nullptr, xhat, rstd, gamma, dx, dres, M, D, p_drop, seed, seed_dev, HOPMI_F32, stream);
}

extern "C" size_t hopmi_colsum_ws_floats(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  const int nchunk = (M + CS_ROWS - 1) / CS_ROWS;
  return nchunk > 1 ? (size_t)nchunk * N : 0;
}

extern "C" int hopmi_colsum(const void* x, int dtype, int M, int N, float* out, float* ws, void* stream) {
  if (M <= 0 || N <= 0 || (M + CS_ROWS - 1) / CS_ROWS > 65535) { set_error("hopmi_colsum: bad sizes M=%d N=%d", M, N); return HOPMI_EINVAL; }
  if (int e = ew_dtype_ok("hopmi_colsum", dtype)) return e;
  const int nchunk = (M + CS_ROWS - 1) / CS_ROWS;
  if (!x || !out || (nchunk > 1 && !ws)) { set_error("hopmi_colsum: null pointer argument"); return HOPMI_EINVAL; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = nchunk > 1 ? ws : out;
  const size_t esz = dtype == HOPMI_BF16 ? 2 : 4;
  const bool vec = N % 4 == 0 && reinterpret_cast<uintptr_t>(x) % (4 * esz) == 0;
  const dim3 grid((unsigned)((N + (vec ? 128 : 32) - 1) / (vec ? 128 : 32)), (unsigned)nchunk);
  if (dtype == HOPMI_BF16) {
    if (vec) hipLaunchKernelGGL((colsum_partial_kernel<__bf16, 4>), grid, dim3(256), 0, st, static_cast<const __bf16*>(x), M, N, part);
    else hipLaunchKernelGGL((colsum_partial_kernel<__bf16, 1>), grid, dim3(256), 0, st, static_cast<const __bf16*>(x), M, N, part);
  } else {
    if (vec) hipLaunchKernelGGL((colsum_partial_kernel<float, 4>), grid, dim3(256), 0, st, static_cast<const float*>(x), M, N, part);
    else hipLaunchKernelGGL((colsum_partial_kernel<float, 1>), grid, dim3(256), 0, st, static_cast<const float*>(x), M, N, part);
  }
  if (nchunk > 1) hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 31) / 32), dim3(256), 0, st, part, nchunk, N, out);
  return check_launch("hopmi_colsum");
}


extern "C" int hopmi_bn_cl_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                               float* y, float* save_mean_rstd, int M, int C, float eps, float momentum, int training, void* stream) {
  if (M <= 0 || C <= 0 || C > 64) { set_error("hopmi_bn_cl_fwd: need M > 0 and 0 < C <= 64 (M=%d C=%d)", M, C); return HOPMI_EINVAL; }
  if (!x || !gamma || !beta || (!training && (!running_mean || !running_var))) {
    set_error("hopmi_bn_cl_fwd: null pointer argument");
    return HOPMI_EINVAL;
  }
  hipLaunchKernelGGL(bn_cl_fwd_kernel, dim3(1), dim3(BN_T), 0, static_cast<hipStream_t>(stream), x, gamma, beta, running_mean,
                     running_var, y, save_mean_rstd, M, C, eps, momentum, training);
  return check_launch("hopmi_bn_cl_fwd");
}

extern "C" int hopmi_bn_cl_bwd(const float* x, const float* dy, const float* gamma, const float* save_mean_rstd, float* dx,
                               float* dgamma, float* dbeta, int M, int C, void* stream) {
  if (M <= 0 || C <= 0 || C > 64) { set_error("hopmi_bn_cl_bwd: need M > 0 and 0 < C <= 64 (M=%d C=%d)", M, C); return HOPMI_EINVAL; }
  if (!x || !dy || !gamma || !save_mean_rstd || !dgamma || !dbeta) { set_error("hopmi_bn_cl_bwd: null pointer argument"); return HOPMI_EINVAL; }
  hipLaunchKernelGGL(bn_cl_bwd_kernel, dim3(1), dim3(BN_T), 0, static_cast<hipStream_t>(stream), x, dy, gamma, save_mean_rstd, dx, dgamma,
                     dbeta, M, C);
  return check_launch("hopmi_bn_cl_bwd");
}

HOPMI_SPLIT_STATUS_SETTER(elementwise)
