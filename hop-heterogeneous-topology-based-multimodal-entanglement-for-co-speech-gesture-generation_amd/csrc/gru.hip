// Bidirectional GRU layer recurrence of the HOP pose decoder (reference: model/HOP.py:166-167,248
// nn.GRU(…, hidden 350, 4 layers, bidirectional); the same cell serves the discriminator,
// model/multimodal_context_net.py:236-237,257), forward and back-propagation through time.
//
// Split of the work (torch.nn.GRU semantics, gate order r, z, n):
//   * the input projections gi[b][t][dir][3H] = x W_ih^T + b_ih for ALL time steps are one big GEMM
//     done by the caller (hipBLASLt) -- they have no sequential dependency;
//   * the sequential part -- gh = h_{t-1} W_hh^T + b_hh, the gates, h_t -- runs here, both directions together,
//     in one of two schedules behind the same entry point:
//       - persistent (whenever every workgroup of the launch can be resident at once): ONE launch per layer; see the block
//         comment in front of gru_fwd_persistent_kernel (recurrent weights as scaled fp16 hi/lo MFMA fragments in registers --
//         three-term products, fp32-equivalent, f16_dev.h --, 16 x 32 tiles, h_t handed over through y itself: fill pattern +
//         re-load, no counters);
//       - per-step: one launch per time step (a dependent launch boundary costs ~1.5 us on MI355X), the T launches
//         enqueued from C in one ABI call, exact-fp32 MFMA.
// Decomposition of a per-step launch: workgroup (jb, bb, dir) owns hidden units [16 jb, 16 jb + 16) of batch rows
// [32 bb, 32 bb + 32): out[32 x (3 gates x 16)] = h_prev[32 x H] W_slice^T on exact-fp32 MFMA 16x16x4.
// Both operand panels (32 rows of h_prev, 48 rows of W_hh) are streamed HBM/L2 -> LDS with row-contiguous
// 8-B loads (512 B per wave instruction; per-lane fragment loads would touch 64 cache lines each), K is
// zero-padded to a multiple of 64 and split over the 4 waves in 16-wide chunks read back as ds_read_b128
// MFMA operands; the 4 partial tiles are summed through LDS and the gate math is the epilogue.
// h_prev is read from the layer output y itself.
#include <mutex>
#include <utility>
#include <vector>

#define HOPMI_FILE_ID 6          // (diagnostic build: common.h, split_check)
#include "attn_dev.h"
#include "f16_dev.h"

namespace hopmi {

constexpr int GRU_BM = 32;     // batch rows per workgroup (2 MFMA row tiles)
constexpr int GRU_NU = 16;     // hidden units per workgroup (one MFMA column tile per gate)
constexpr int RED_LD = 49;     // LDS stride of the [32][48] partial tiles

// gate non-linearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32): absolute error ~1e-7, far inside the 1e-3 bar;
// the libm expf / tanhf made the gate epilogue 30 % of a recurrence step (tools/probes/gru_stamps.py)
// (__builtin_amdgcn_rcpf IS v_rcp_f32, 1 ulp; __frcp_rn expands to the ten-instruction IEEE division sequence)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

// XCD-aware work mapping (speed only; any placement is correct).  Workgroups are dealt round-robin over
// the 8 XCDs by linear id, and each XCD has a private 4 MiB L2.  A "slice" = (unit block jb, direction):
// its 67 KB of recurrent weights are needed by all batch groups of that slice at EVERY time step, so all
// workgroups of a slice get the same id % 8: an XCD then touches only ~1/8 of W_hh and keeps it in its L2
// across the T launches of a layer instead of re-fetching all 2.9 MB per step.
struct GruWork { int jb, bb, d; bool valid; };
__device__ __forceinline__ GruWork gru_decode(int nJ, int nbb) {
  const int lin = blockIdx.x, xcd = lin & 7, sidx = lin >> 3;
  const int slice = xcd + 8 * (sidx / nbb);
  GruWork wk;
  wk.bb = sidx % nbb;
  wk.valid = slice < 2 * nJ;
  wk.jb = slice % nJ;
  wk.d = slice / nJ;
  return wk;
}

// Stage `nrows` rows of K (even) floats into LDS rows of stride ldk (floats), zero-padding [K, KP).
// Row r of the panel comes from src_of(r) (nullptr -> all-zero row).  One wave per row at a time,
// lane l takes the float2 at k = 2l, 2l + 128, ...: every load instruction covers 512 contiguous bytes.
// All loads of a wave's rows are issued before the first LDS store (one memory round trip).
template <int MAXROWS_PER_WAVE, int MAXK2, int NW = 4, typename SrcOf>
__device__ __forceinline__ void stage_rows(float* dst, int ldk, int nrows, int K, int KP, SrcOf src_of, int w, int lane) {
  // rows w, w+NW, w+2NW, ... (NW waves per workgroup); per row MAXK2 float2 slots per lane
  float2 v[MAXROWS_PER_WAVE][MAXK2];
#pragma unroll
  for (int rr = 0; rr < MAXROWS_PER_WAVE; ++rr) {
    const int r = w + NW * rr;
    const float* src = src_of(min(r, nrows - 1));
    const bool row_ok = (r < nrows) && (src != nullptr);
    const float* sp = src ? src : dst;            // never dereferenced when !row_ok (clamped below)
#pragma unroll
    for (int c = 0; c < MAXK2; ++c) {
      const int k = 2 * lane + 128 * c;
      const bool ok = row_ok && (k < K);
      float2 t = make_float2(0.f, 0.f);
      if (src) t = *reinterpret_cast<const float2*>(sp + min(k, K - 2));
      v[rr][c] = ok ? t : make_float2(0.f, 0.f);
    }
  }
#pragma unroll
  for (int rr = 0; rr < MAXROWS_PER_WAVE; ++rr) {
    const int r = w + NW * rr;
#pragma unroll
    for (int c = 0; c < MAXK2; ++c) {
      const int k = 2 * lane + 128 * c;
      if (r < nrows && k < KP) *reinterpret_cast<float2*>(dst + r * ldk + k) = v[rr][c];
    }
  }
}

// acc[mt][g] += As[16mt + i][k] * Ws[16g + i][k] over this wave's k chunks [16 c0, 16 (c0 + nc)).
template <int NG>
__device__ __forceinline__ void panel_mfma(f32x4 (&acc)[2][NG], const float* As, const float* Ws, int ldk,
                                           int c0, int nc, int q, int i) {
  const float* ap = As + i * ldk + 4 * q;          // A[i = row][k = 16c + 4q + e]
  const float* wp = Ws + i * ldk + 4 * q;          // B[k][j = unit]
  for (int c = c0; c < c0 + nc; ++c) {
    float4 a[2], wv[NG];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const float4*>(ap + 16 * mt * ldk + 16 * c);
#pragma unroll
    for (int g = 0; g < NG; ++g) wv[g] = *reinterpret_cast<const float4*>(wp + 16 * g * ldk + 16 * c);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        acc[mt][g] = mfma16(a[mt].x, wv[g].x, acc[mt][g]);
        acc[mt][g] = mfma16(a[mt].y, wv[g].y, acc[mt][g]);
        acc[mt][g] = mfma16(a[mt].z, wv[g].z, acc[mt][g]);
        acc[mt][g] = mfma16(a[mt].w, wv[g].w, acc[mt][g]);
      }
  }
}

// same for ONE 16-row tile: acc[g] += As[i][k] * Ws[16g + i][k] (As already points at the tile's first row).
// The operand reads of chunk c+1 are issued before the MFMAs of chunk c (register double buffer): the trip count is
// a runtime value, which the compiler would not unroll / pipeline by itself.
template <int NG>
__device__ __forceinline__ void panel_mfma_1(f32x4 (&acc)[NG], const float* As, const float* Ws, int ldk, int c0, int nc, int q, int i) {
  const float* ap = As + i * ldk + 4 * q + 16 * c0;
  const float* wp = Ws + i * ldk + 4 * q + 16 * c0;
  float4 a = *reinterpret_cast<const float4*>(ap);
  float4 wv[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) wv[g] = *reinterpret_cast<const float4*>(wp + 16 * g * ldk);
  for (int cc = 0; cc < nc; ++cc) {
    const int cn = 16 * min(cc + 1, nc - 1);         // (the last iteration re-reads its own chunk: harmless)
    const float4 an = *reinterpret_cast<const float4*>(ap + cn);
    float4 wn[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) wn[g] = *reinterpret_cast<const float4*>(wp + 16 * g * ldk + cn);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      acc[g] = mfma16(a.x, wv[g].x, acc[g]);
      acc[g] = mfma16(a.y, wv[g].y, acc[g]);
      acc[g] = mfma16(a.z, wv[g].z, acc[g]);
      acc[g] = mfma16(a.w, wv[g].w, acc[g]);
    }
    a = an;
#pragma unroll
    for (int g = 0; g < NG; ++g) wv[g] = wn[g];
  }
}

// one time step `s` of both directions (dir 0 walks t = s, dir 1 walks t = T-1-s).
// MAXK2 = ceil(KP / 128) float2 slots per lane per row (KP = H padded to a multiple of 64).
template <int MAXK2, typename TG>
__global__ __launch_bounds__(256) void gru_fwd_step_kernel(const TG* __restrict__ gi, const float* __restrict__ whh,
                                                           const float* __restrict__ bhh, float* __restrict__ y,
                                                           float* __restrict__ gates, int B, int T, int H, int KP, int s) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GruWork wk = gru_decode((H + GRU_NU - 1) / GRU_NU, (B + GRU_BM - 1) / GRU_BM);
  if (!wk.valid) return;
  const int ldk = KP + 4;
  float* As = smem;                                // [32][ldk]  h_prev rows
  float* Ws = As + GRU_BM * ldk;                   // [48][ldk]  W_hh rows of the 3 gates x 16 units
  float* red = Ws + 3 * GRU_NU * ldk;              // [4][32][RED_LD]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, i = lane & 15;
  const int d = wk.d, j0 = wk.jb * GRU_NU, b0 = wk.bb * GRU_BM;
  const int t = d ? T - 1 - s : s, tp = d ? t + 1 : t - 1;
  const size_t ystride = (size_t)2 * H;

  // epilogue operands, fetched up front (clamped, unconditional) so their latency hides under the staging
  const int jj = tid & 15, j = j0 + jj, jc = min(j, H - 1);
  float e_gi[2][3], e_bhh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) e_bhh[g] = bhh[(d * 3 + g) * H + jc];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int bc = min(b0 + (tid >> 4) + 16 * pass, B - 1);
    const TG* gip = gi + (((size_t)bc * T + t) * 2 + d) * 3 * H + jc;
#pragma unroll
    for (int g = 0; g < 3; ++g) e_gi[pass][g] = (float)gip[g * H];
  }

  if (s > 0) {
    stage_rows<GRU_BM / 4, MAXK2>(As, ldk, GRU_BM, H, KP, [&](int r) -> const float* {
      const int b = b0 + r;
      return b < B ? y + ((size_t)b * T + tp) * ystride + d * H : nullptr;
    }, w, lane);
    stage_rows<3 * GRU_NU / 4, MAXK2>(Ws, ldk, 3 * GRU_NU, H, KP, [&](int r) -> const float* {
      const int g = r >> 4, ju = j0 + (r & 15);
      return ju < H ? whh + ((size_t)(d * 3 + g) * H + ju) * H : nullptr;
    }, w, lane);
    __syncthreads();
    f32x4 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[mt][g] = {0.f, 0.f, 0.f, 0.f};
    const int nc = KP >> 6;                                     // 16-wide k chunks per wave
    panel_mfma<3>(acc, As, Ws, ldk, w * nc, nc, q, i);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * GRU_BM + 16 * mt + 4 * q + r) * RED_LD + 16 * g + i] = acc[mt][g][r];
  }
  __syncthreads();

  // gate epilogue: thread -> (row, unit), two passes of 16 rows
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = (tid >> 4) + 16 * pass, b = b0 + row;
    if (b < B && j < H) {
      float gh[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float v = e_bhh[g];
        if (s > 0) {
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) v += red[(ww * GRU_BM + row) * RED_LD + 16 * g + jj];
        }
        gh[g] = v;
      }
      const float r = sigmoidf_(e_gi[pass][0] + gh[0]);
      const float z = sigmoidf_(e_gi[pass][1] + gh[1]);
      const float n = tanhf_(e_gi[pass][2] + r * gh[2]);
      const float hp = (s > 0) ? As[row * ldk + j] : 0.f;       // h_prev[b][j] is in the staged panel
      y[((size_t)b * T + t) * ystride + d * H + j] = (1.f - z) * n + z * hp;
      float* gp = gates + (((size_t)b * T + t) * 2 + d) * 4 * H + j;
      gp[0] = r; gp[H] = z; gp[2 * H] = n; gp[3 * H] = gh[2];
    }
  }
}

// BPTT step `s` (s = 0 is the LAST step the forward processed).  Per direction, in forward processing
// order p (dir 0: p = t, dir 1: p = T-1-t):
//   D_p   = dy_p + Dz_{p+1} + dgh_{p+1} W_hh          (gradient w.r.t. h_p; Dz_{p+1} = D_{p+1} * z_{p+1})
//   dn    = D_p (1 - z) (1 - n^2);  dz = D_p (h_{p-1} - n) z (1 - z);  dr = dn * hn * r (1 - r)
//   dgi_p = [dr, dz, dn];  dgh_p = [dr, dz, dn * r];  Dz_p = D_p * z
// The workgroup owns units [16 jb, +16) of rows [32 bb, +32): the contraction over the 3H gates of step
// p+1 is done gate block by gate block (K = H each): dgh rows and rows of W_hh^T (whhT[dir][unit][3H],
// transposed by the caller) are both contiguous.
template <int MAXK2, typename TG>
__global__ __launch_bounds__(256) void gru_bwd_step_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ gates, const float* __restrict__ whhT,
                                                           TG* __restrict__ dgi, float* __restrict__ dgh,
                                                           float* __restrict__ dhz, int B, int T, int H, int KP, int s) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GruWork wk = gru_decode((H + GRU_NU - 1) / GRU_NU, (B + GRU_BM - 1) / GRU_BM);
  if (!wk.valid) return;
  const int ldk = KP + 4;
  float* As = smem;                                // [32][ldk]  dgh rows of one gate block
  float* Ws = As + GRU_BM * ldk;                   // [16][ldk]  W_hh^T rows of the 16 units, same gate block
  float* red = Ws + GRU_NU * ldk;                  // [4][32][RED_LD]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, i = lane & 15;
  const int d = wk.d, j0 = wk.jb * GRU_NU, b0 = wk.bb * GRU_BM;
  const int t = d ? s : T - 1 - s;                 // time index processed at BPTT step s
  const int tn = d ? t - 1 : t + 1;                // the step the forward processed right after t
  const int tp = d ? t + 1 : t - 1;                // ... and right before t (source of h_prev)
  const int K = 3 * H;
  const float* dhz_in = dhz + ((size_t)((s + 1) & 1) * 2 + d) * B * H;
  float* dhz_out = dhz + ((size_t)(s & 1) * 2 + d) * B * H;

  // epilogue operands, fetched up front (clamped, unconditional)
  const int jj = tid & 15, j = j0 + jj, jc = min(j, H - 1);
  float e_dy[2], e_dhz[2], e_g[2][4], e_hp[2];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int bc = min(b0 + (tid >> 4) + 16 * pass, B - 1);
    e_dy[pass] = dy[((size_t)bc * T + t) * 2 * H + d * H + jc];
    e_dhz[pass] = dhz_in[(size_t)bc * H + jc];                     // garbage at s == 0: not used there
    const float* gp = gates + (((size_t)bc * T + t) * 2 + d) * 4 * H + jc;
#pragma unroll
    for (int g = 0; g < 4; ++g) e_g[pass][g] = gp[g * H];
    const int tpc = min(max(tp, 0), T - 1);
    e_hp[pass] = y[((size_t)bc * T + tpc) * 2 * H + d * H + jc];
  }

  if (s > 0) {
    f32x4 acc[2][1];
    acc[0][0] = {0.f, 0.f, 0.f, 0.f};
    acc[1][0] = {0.f, 0.f, 0.f, 0.f};
    const int nc = KP >> 6;
    for (int g = 0; g < 3; ++g) {
      if (g) __syncthreads();                      // previous gate block consumed
      stage_rows<GRU_BM / 4, MAXK2>(As, ldk, GRU_BM, H, KP, [&](int r) -> const float* {
        const int b = b0 + r;
        return b < B ? dgh + (((size_t)b * T + tn) * 2 + d) * K + g * H : nullptr;
      }, w, lane);
      stage_rows<GRU_NU / 4, MAXK2>(Ws, ldk, GRU_NU, H, KP, [&](int r) -> const float* {
        const int ju = j0 + r;
        return ju < H ? whhT + ((size_t)d * H + ju) * K + g * H : nullptr;
      }, w, lane);
      __syncthreads();
      panel_mfma<1>(acc, As, Ws, ldk, w * nc, nc, q, i);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(w * GRU_BM + 16 * mt + 4 * q + r) * RED_LD + i] = acc[mt][0][r];
  }
  __syncthreads();

#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = (tid >> 4) + 16 * pass, b = b0 + row;
    if (b < B && j < H) {
      float D = e_dy[pass];
      if (s > 0) {
        D += e_dhz[pass];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) D += red[(ww * GRU_BM + row) * RED_LD + jj];
      }
      const float r = e_g[pass][0], z = e_g[pass][1], n = e_g[pass][2], hn = e_g[pass][3];
      const float hp = (s < T - 1) ? e_hp[pass] : 0.f;
      const float dn = D * (1.f - z) * (1.f - n * n);
      const float dz = D * (hp - n) * z * (1.f - z);
      const float dr = dn * hn * r * (1.f - r);
      const size_t o = (((size_t)b * T + t) * 2 + d) * K + j;
      dgi[o] = (TG)dr; dgi[o + H] = (TG)dz; dgi[o + 2 * H] = (TG)dn;
      dgh[o] = dr; dgh[o + H] = dz; dgh[o + 2 * H] = dn * r;
      dhz_out[(size_t)b * H + j] = D * z;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Persistent schedule: ONE launch per layer.  Workgroup (group = (direction, 16 batch rows), unit block jb of 32 hidden
// units) walks all T steps itself; the nJ = ceil(H / 32) workgroups of a group exchange h_t (forward) / dgh_t (backward)
// through the layer's own output arrays.
//   * The recurrent weights never touch LDS: every wave keeps the MFMA B-operand fragments of its (unit half, K quarter)
//     in registers for the whole launch, split into two bf16 parts once at kernel start (72 VGPRs at H = 350).
//   * The product runs on v_mfma_f32_16x16x32_f16 with both operands carried as scaled fp16 hi + lo pairs (three terms,
//     f16_dev.h: 22 significand bits per operand, fp32 accumulation -- fp32-equivalent; rounds 2-4: bf16 pairs, 2^-16): 27 MFMAs
//     of 16 cycles per wave and step instead of the 72 exact-fp32 MFMAs of 32 cycles that were 32 % of a step
//     (tools/probes/gru_stamps.py).  Scales (powers of two): the recurrent weights one per (gate, unit) row, found once at kernel
//     start (lane maximum -> the row's 4 lanes -> the 4 K-quarter waves through LDS; the backward: one per unit over the three
//     gate blocks it contracts over); the state h, bounded by 1: 2^14; in the backward the handed-off dgh rows one per staging
//     wave (its two batch rows x three gate blocks) from the maximum of what the wave loaded: a row's scale is the same in
//     all gate blocks, so it leaves per accumulator row in the epilogue.
//   * Hand-off without counters: the hand-off array is filled with a NaN bit pattern no computation produces before the
//     kernel starts; a producer simply stores its values (sc1: write-through), a consumer loads the rows it needs with sc1
//     loads (never served from a stale cache) and re-loads whatever still reads as the pattern.  One memory round trip
//     per step instead of three (drain stores + arrive, poll the counter, load the panel).
//   * 16 x 32 tiles instead of 32 x 16: each handed-off row is read by half as many workgroups (the exchange is
//     bandwidth-bound: every workgroup of a group reads the whole group's rows; 22x -> 11x read amplification).
// Requirements: the whole grid is resident at once (host checks grid <= #CUs).  Every spin is bounded: on timeout the wave
// raises the status word, stops waiting for the rest of the launch and runs to completion, so the launch always drains
// (results are then garbage and the host raises on the status word).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 ld_sc1_f2(const float* p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)(v & 0xFFFFFFFFull)), __uint_as_float((unsigned)(v >> 32)));
}
__device__ __forceinline__ void st_sc1_f(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int GP_BM = 16;                          // batch rows per persistent workgroup (one MFMA row tile)
constexpr int GP_NU = 32;                          // hidden units per persistent workgroup (two MFMA column tiles)
constexpr int GP_RED_F = 100;                      // LDS stride of the forward partial tiles  [16][3 x 32]
constexpr int GP_RED_B = 36;                       // ... of the backward partial tiles        [16][32]
constexpr unsigned GRU_SENTINEL = 0x7FC5A3E1u;     // "not written yet": a NaN payload neither the ALUs nor torch generate
constexpr int GRU_POLL_LIMIT = 1 << 21;            // re-load rounds before a wave gives up (seconds)

// Workgroup -> (direction, batch group, unit block).  Linear ids are dealt round-robin over the 8 XCDs; all workgroups of
// a group (the ones that exchange rows) take the same id % 8, so a hand-off stays inside one XCD's L2 when the dispatcher
// places workgroups that way (speed only: the hand-off is correct for any placement).
struct GpWork { int d, bb, jb; bool valid; };
__device__ __forceinline__ GpWork gp_decode(int nJ, int nbb) {
  const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
  const int gsel = k / nJ, group = xcd + 8 * gsel;
  GpWork wk;
  wk.jb = k - gsel * nJ;
  wk.valid = group < 2 * nbb;
  wk.d = group / nbb;
  wk.bb = group - wk.d * nbb;
  return wk;
}
static int gp_grid(int nJ, int nbb) { return 8 * ((2 * nbb + 7) / 8) * nJ; }

__device__ __forceinline__ bool gru_unwritten(float2 v) {
  return __float_as_uint(v.x) == GRU_SENTINEL || __float_as_uint(v.y) == GRU_SENTINEL;
}

// Diagnostic build only (-DHOPMI_STAMPS, tools/probes/gru_stamps.py): s_memtime stamps of one mid-sequence step.
#ifdef HOPMI_STAMPS
static __device__ long long* g_gru_stamps = nullptr;
#define GRU_STAMP(slot)                                                                          \
  do {                                                                                           \
    if (s == T / 2) {                                                                            \
      unsigned long long t_;                                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
      if (g_gru_stamps && threadIdx.x == 0) g_gru_stamps[blockIdx.x * 8 + (slot)] = (long long)t_; \
    }                                                                                            \
  } while (0)
extern "C" int hopmi_debug_set_stamps_gru(long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gru_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#else
#define GRU_STAMP(slot) do { } while (0)
#endif

// Load NSEG row segments of K floats (segment m from src(m); nullptr -> zeros): lane l takes the float2 at k = 2 l + 128 c.
// Values that still hold the fill pattern are loaded again until they do not (bounded).
template <int NSEG, int MAXK2, typename SrcOf>
__device__ __forceinline__ void poll_segments(float2 (&v)[NSEG][MAXK2], SrcOf src, int K, int lane, bool& dead, int* status) {
  bool pend = false;
#pragma unroll
  for (int m = 0; m < NSEG; ++m) {
    const float* p = src(m);
#pragma unroll
    for (int c = 0; c < MAXK2; ++c) {
      const int k = 2 * lane + 128 * c;
      float2 t = make_float2(0.f, 0.f);
      if (p != nullptr && k < K) t = ld_sc1_f2(p + k);
      v[m][c] = t;
      pend |= gru_unwritten(t);
    }
  }
  int spins = 0;
  while (!dead && __ballot(pend) != 0ull) {                            // wave-uniform
    if (++spins > GRU_POLL_LIMIT) {
      dead = true;
      if (lane == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
    pend = false;
#pragma unroll
    for (int m = 0; m < NSEG; ++m) {
      const float* p = src(m);
#pragma unroll
      for (int c = 0; c < MAXK2; ++c) {
        if (gru_unwritten(v[m][c])) {                                   // (never true for a zero-filled slot)
          v[m][c] = ld_sc1_f2(p + 2 * lane + 128 * c);
          pend |= gru_unwritten(v[m][c]);
        }
      }
    }
  }
}

// commit one loaded segment, times the power of two `sc`, to LDS row `row` of a split panel (hi image, lo image; WS2 32-bit words per row)
template <int MAXK2>
__device__ __forceinline__ void commit_split_row(unsigned* hi, unsigned* lo, int ws2, int row, const float2 (&v)[MAXK2], int lane, float sc) {
#pragma unroll
  for (int c = 0; c < MAXK2; ++c) {
    const u32x2 p = split2h(v[c].x * sc, v[c].y * sc);
    hi[row * ws2 + lane + 64 * c] = p[0];
    lo[row * ws2 + lane + 64 * c] = p[1];
  }
}

// the 8 values W[k0 .. k0 + 8) of a weight row (nullptr row or k >= K -> 0)
__device__ __forceinline__ void load_w8(const float* row, int k0, int K, float4& a, float4& b) {
  float2 f[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = k0 + 2 * e;
    f[e] = (row != nullptr && k < K) ? *reinterpret_cast<const float2*>(row + k) : make_float2(0.f, 0.f);
  }
  a = make_float4(f[0].x, f[0].y, f[1].x, f[1].y);
  b = make_float4(f[2].x, f[2].y, f[3].x, f[3].y);
}
__device__ __forceinline__ float absmax_w8(float m, const float* row, int k0, int K) {
  float4 a, b;
  load_w8(row, k0, K, a, b);
  return absmax4(absmax4(m, a), b);
}
// ... times the power of two `sc`, as one split MFMA operand
__device__ __forceinline__ Split8 load_w_frag(const float* row, int k0, int K, float sc) {
  float4 a, b;
  load_w8(row, k0, K, a, b);
  return split8h(a, b, sc);
}

// Resident recurrent-weight fragments of a persistent-kernel wave (unit 16 ch + i of unit block j0, K quarter kq, 3 gates), scaled by
// one power of two per (gate, unit) ROW: the row's maximum is taken over this lane's elements, its 4 lanes (q) and the 4 K-quarter
// waves of the unit half (through `xch`: [8 waves][3][16] floats of LDS, two barriers; every lane ends with its row's scale).
// `wrow(g)`: the row of gate g.  Returns the inverse scales in inv[3].
// JOINT: one scale per unit over all three gate blocks (the backward contracts over them into ONE accumulator).
template <int MAXK2, bool JOINT = false, typename RowOf>
__device__ __forceinline__ void load_resident_w(u32x4 (&wh)[3][MAXK2], u32x4 (&wl)[3][MAXK2], float (&inv)[3], RowOf wrow, int H, int kq,
                                                int ch, int q, int i, int w, float* xch) {
  // ONE pass over the rows (a lane's loads are row-strided: 64 cache lines per wave instruction -- the second pass of the first
  // version cost 15 us per launch): the raw values wait in registers (24 MAXK2 floats) for the row maxima
  float4 ra[3][MAXK2], rb[3][MAXK2];
  float m[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float* row = wrow(g);
    float mm = 0.f;
#pragma unroll
    for (int kk = 0; kk < MAXK2; ++kk) {
      load_w8(row, (kq * MAXK2 + kk) * 32 + 8 * q, H, ra[g][kk], rb[g][kk]);
      mm = absmax4(absmax4(mm, ra[g][kk]), rb[g][kk]);
    }
    mm = fmaxf(mm, __shfl_xor(mm, 16));
    mm = fmaxf(mm, __shfl_xor(mm, 32));
    m[g] = mm;
    if (q == 0) xch[(w * 3 + g) * 16 + i] = mm;
  }
  __syncthreads();
  float mj = 0.f;
  if (JOINT) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) mj = fmaxf(mj, xch[((ch * 4 + k2) * 3 + g) * 16 + i]);
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    float mm = JOINT ? mj : m[g];
    if (!JOINT) {
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) mm = fmaxf(mm, xch[((ch * 4 + k2) * 3 + g) * 16 + i]);
    }
    const float sc = scale_for_absmax(mm);
    inv[g] = inv_pow2(sc);
#pragma unroll
    for (int kk = 0; kk < MAXK2; ++kk) {
      const Split8 f = split8h(ra[g][kk], rb[g][kk], sc);
      wh[g][kk] = f.hi; wl[g][kk] = f.lo;
    }
  }
  __syncthreads();                                                     // (xch is the partial-tile region: free again)
}

// KP = 128 MAXK2 >= H: K padded so that each of the 4 K quarters is MAXK2 MFMA steps of 32.
// 8 waves: wave w owns K quarter (w & 3) of unit half (w >> 2); the gate epilogue is one element per thread and row tile.
// MR = 16-row tiles per workgroup (round 6).  MR = 1 is the kernel of rounds 2-5.  MR = 2 (32 batch rows per workgroup, the same
// resident weight fragments serve both tiles) is for a batch that would need more workgroups than the chip holds at 16 rows --
// the decoder of a training step's TWO generator forwards run as one launch of 2 B rows (ops.gru_bidirectional_pair: same
// weights; a time step is a hand-off round trip of ~2 us around ~0.5 us of arithmetic, so the second tile rides nearly free).
template <int MAXK2, typename TG, int MR = 1>
__global__ __launch_bounds__(512) void gru_fwd_persistent_kernel(const TG* __restrict__ gi, const float* __restrict__ whh,
                                                                 const float* __restrict__ bhh, float* y,
                                                                 float* __restrict__ gates, int* status, int B, int T,
                                                                 int H, int nJ, int nbb, const TG* __restrict__ gi2, int B1) {
  // (gi2 / B1: batch rows >= B1 take their input projections from gi2[row - B1] -- hopmi_gru_fwd_pair_dt: two forwards' decoder
  // inputs that live in two tensors; a plain call passes B1 = B)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KP = 128 * MAXK2, WS2 = (KP + 48) / 2;                 // 32-bit words per LDS row: 16-byte reads conflict-free
  constexpr int BMR = GP_BM * MR;                                      // batch rows per workgroup
  unsigned* Ahi = reinterpret_cast<unsigned*>(smem);                   // [16 MR][WS2]  h_prev rows, fp16 hi parts
  unsigned* Alo = Ahi + BMR * WS2;                                     //               ... lo parts
  float* red = reinterpret_cast<float*>(Alo + BMR * WS2);              // [4 K quarters][16 MR][GP_RED_F]
  const GpWork wk = gp_decode(nJ, nbb);
  if (!wk.valid) return;
  const int d = wk.d, bb = wk.bb, jb = wk.jb;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, i = lane & 15;
  const int kq = w & 3, ch = w >> 2;
  const int j0 = jb * GP_NU, b0 = bb * BMR;
  const size_t ystride = (size_t)2 * H;
  const int jj = tid & 31, j = j0 + jj, jc = min(j, H - 1);
  const int row = tid >> 5;                                            // the thread's elements in the gate epilogue: rows row + 16 mr

  u32x4 wh[3][MAXK2], wl[3][MAXK2];                                    // resident W_hh fragments of this wave
  float e_inv[3];                                                      // accumulator -> gh of the EPILOGUE element's unit: (1 / s_W[g][unit]) 2^-14
  {
    const int ju = j0 + 16 * ch + i;
    float inv[3];
    load_resident_w<MAXK2>(wh, wl, inv, [&](int g) -> const float* { return ju < H ? whh + ((size_t)(d * 3 + g) * H + ju) * H : nullptr; },
                           H, kq, ch, q, i, w, red);
    // the epilogue walks (row, unit jj = tid & 31): hand the inverse scales from the fragment layout (unit 16 ch + i) over through LDS
    if (kq == 0 && q == 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g) red[g * GP_NU + 16 * ch + i] = inv[g] * H_UNIT_INV;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 3; ++g) e_inv[g] = red[g * GP_NU + jj];
    __syncthreads();
  }
  float e_bhh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) e_bhh[g] = bhh[(d * 3 + g) * H + jc];
  float h_own[MR];                                                     // this thread's h_{t-1}[b][j], one per row tile
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) h_own[mr] = 0.f;
  bool dead = false;

  for (int s = 0; s < T; ++s) {
    const int t = d ? T - 1 - s : s, tp = d ? t + 1 : t - 1;
    GRU_STAMP(0);
    float e_gi[MR][3];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      const int bc = min(b0 + row + GP_BM * mr, B - 1);
      const TG* gip = (bc < B1 ? gi + (size_t)bc * T * 6 * H : gi2 + (size_t)(bc - B1) * T * 6 * H) + ((size_t)t * 2 + d) * 3 * H + jc;
#pragma unroll
      for (int g = 0; g < 3; ++g) e_gi[mr][g] = (float)gip[g * H];
    }
    if (s > 0) {
      float2 v[2 * MR][MAXK2];                                         // wave w stages rows w + 8 m
      poll_segments<2 * MR, MAXK2>(v, [&](int m) -> const float* {
        const int br = b0 + w + 8 * m;
        return br < B ? y + ((size_t)br * T + tp) * ystride + d * H : nullptr;
      }, H, lane, dead, status);
      GRU_STAMP(1);
#pragma unroll
      for (int m = 0; m < 2 * MR; ++m) commit_split_row<MAXK2>(Ahi, Alo, WS2, w + 8 * m, v[m], lane, H_UNIT_SCALE);   // |h| <= 1
      __syncthreads();
      GRU_STAMP(2);
      f32x4 acc[MR][3];
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[mr][g] = {0.f, 0.f, 0.f, 0.f};
      const unsigned* ahp = Ahi + i * WS2 + 4 * q + kq * MAXK2 * 16;
      const unsigned* alp = Alo + i * WS2 + 4 * q + kq * MAXK2 * 16;
#pragma unroll
      for (int kk = 0; kk < MAXK2; ++kk) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
          const u32x4 ah = *reinterpret_cast<const u32x4*>(ahp + GP_BM * mr * WS2 + 16 * kk);
          const u32x4 al = *reinterpret_cast<const u32x4*>(alp + GP_BM * mr * WS2 + 16 * kk);
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[mr][g] = mfma_h3(ah, al, wh[g][kk], wl[g][kk], acc[mr][g]);
        }
      }
      GRU_STAMP(3);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(kq * BMR + GP_BM * mr + 4 * q + r) * GP_RED_F + 32 * g + 16 * ch + i] = acc[mr][g][r];
      __syncthreads();
      GRU_STAMP(4);
    }
    // gate epilogue, branch-free over the row tiles (the stores alone are predicated): with the whole body under `if (valid)` the two
    // tiles of MR = 2 ran one after the other -- 2 350 cycles against 1 130 for one (stamps) -- instead of sharing their LDS and
    // transcendental latencies.  Rows / units past the ends compute on clamped inputs and store nothing.
    float gh[MR][3];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      const int rw = row + GP_BM * mr;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float a = e_bhh[g];
        if (s > 0) {
          float p = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) p += red[(ww * BMR + rw) * GP_RED_F + 32 * g + jj];
          a += p * e_inv[g];
        }
        gh[mr][g] = a;
      }
    }
    float gr[MR], gz[MR], gn[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      gr[mr] = sigmoidf_(e_gi[mr][0] + gh[mr][0]);
      gz[mr] = sigmoidf_(e_gi[mr][1] + gh[mr][1]);
    }
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      gn[mr] = tanhf_(e_gi[mr][2] + gr[mr] * gh[mr][2]);
      h_own[mr] = (1.f - gz[mr]) * gn[mr] + gz[mr] * h_own[mr];
    }
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      const int b = b0 + row + GP_BM * mr;
      if (b < B && j < H) {
        st_sc1_f(y + ((size_t)b * T + t) * ystride + d * H + j, h_own[mr]);   // handed off: write-through
        float* gp = gates + (((size_t)b * T + t) * 2 + d) * 4 * H + j;
        gp[0] = gr[mr]; gp[H] = gz[mr]; gp[2 * H] = gn[mr]; gp[3 * H] = gh[mr][2];
      }
    }
    GRU_STAMP(5);
    // no trailing barrier: the next step's panel commit follows this step's second barrier (all MFMA reads done), and its
    // partial-tile writes follow the next step's first barrier (all epilogue reads of `red` done)
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Small hidden sizes (the discriminator's GRU: hidden 64, multimodal_context_net.py:236-237): W_hh of one direction
// (3 x 64 x 64) fits the registers of ONE workgroup, so nothing has to cross workgroups: workgroup = (16 batch rows,
// direction), all T steps; wave w owns units [16 w, 16 w + 16) of the three gates (24 resident split fragments);
// h_{t-1} goes through a double-buffered fp32 LDS panel (one barrier per step), h_t of the wave's own units stays in
// registers for the z h_{t-1} term.  A step is 18 MFMAs per wave + the gate math; measured 1.3 us per step (36 us per
// layer at T = 28) against the 1.7-2.1 us of cross-workgroup hand-off the persistent kernel pays per step at this size.
// The input projections of the next GS_AHEAD steps are kept in flight.  (Measured and without effect on the 1.3 us: a raw
// s_barrier behind lgkmcnt(0) instead of __syncthreads(), and a chunked double buffer of the projections that keeps every
// memory wait out of the loop body -- the step is not waiting on memory; where the time goes is the next thing to stamp.)
// ------------------------------------------------------------------------------------------------------------------
constexpr int GS_H = 64, GS_LD = 68, GS_AHEAD = 4;

template <typename TG>
__global__ __launch_bounds__(256) void gru_fwd_small_kernel(const TG* __restrict__ gi, const float* __restrict__ whh,
                                                            const float* __restrict__ bhh, float* __restrict__ y,
                                                            float* __restrict__ gates, int B, int T) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][16][GS_LD];
  constexpr int H = GS_H;
  const int d = blockIdx.x & 1, b0 = (blockIdx.x >> 1) * 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, i = lane & 15;
  const int j = 16 * w + i;                                            // this lane's unit (C layout column)
  u32x4 wh[3][2], wl[3][2];
  float e_inv[3];                                                      // accumulator -> gh: (1 / s_W[g][unit]) 2^-14
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float* wrow = whh + ((size_t)(d * 3 + g) * H + j) * H;       // the whole row sits in this unit's 4 lanes (q)
    float m = absmax_w8(absmax_w8(0.f, wrow, 8 * q, H), wrow, 32 + 8 * q, H);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    const float sc = scale_for_absmax(m);
    e_inv[g] = inv_pow2(sc) * H_UNIT_INV;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const Split8 f = load_w_frag(wrow, 32 * ks + 8 * q, H, sc);
      wh[g][ks] = f.hi; wl[g][ks] = f.lo;
    }
  }
  float e_bhh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) e_bhh[g] = bhh[(d * 3 + g) * H + j];
  int brow[4];                                                         // batch rows of this lane's 4 accumulator rows (clamped)
#pragma unroll
  for (int r = 0; r < 4; ++r) brow[r] = min(b0 + 4 * q + r, B - 1);
  float h_own[4] = {0.f, 0.f, 0.f, 0.f};
  float gbuf[GS_AHEAD][4][3];                                          // input projections of the steps in flight
  auto fetch = [&](int slot, int s) {
    const int t = d ? T - 1 - s : s;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const TG* gip = gi + (((size_t)brow[r] * T + t) * 2 + d) * 3 * H + j;
#pragma unroll
      for (int g = 0; g < 3; ++g) gbuf[slot][r][g] = (float)gip[g * H];
    }
  };
#pragma unroll
  for (int u = 0; u < GS_AHEAD; ++u)
    if (u < T) fetch(u, u);

  for (int s0 = 0; s0 < T; s0 += GS_AHEAD) {
#pragma unroll
    for (int u = 0; u < GS_AHEAD; ++u) {
      const int s = s0 + u;
      if (s < T) {                                                     // (uniform over the workgroup)
        const int t = d ? T - 1 - s : s;
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
          const float* hp = &hbuf[(s - 1) & 1][i][8 * q];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const Split8 a = split8h(*reinterpret_cast<const float4*>(hp + 32 * ks), *reinterpret_cast<const float4*>(hp + 32 * ks + 4), H_UNIT_SCALE);
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = mfma_h3(a.hi, a.lo, wh[g][ks], wl[g][ks], acc[g]);
          }
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[g][r] = acc[g][r] * e_inv[g] + e_bhh[g];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gr = sigmoidf_(gbuf[u][r][0] + acc[0][r]);
          const float gz = sigmoidf_(gbuf[u][r][1] + acc[1][r]);
          const float gn = tanhf_(gbuf[u][r][2] + gr * acc[2][r]);
          h_own[r] = (1.f - gz) * gn + gz * h_own[r];
          hbuf[s & 1][4 * q + r][j] = h_own[r];
          const int b = b0 + 4 * q + r;
          if (b < B) {
            y[((size_t)b * T + t) * 2 * H + d * H + j] = h_own[r];
            float* gp = gates + (((size_t)b * T + t) * 2 + d) * 4 * H + j;
            gp[0] = gr; gp[H] = gz; gp[2 * H] = gn; gp[3 * H] = acc[2][r];
          }
        }
        if (s + GS_AHEAD < T) fetch(u, s + GS_AHEAD);
        __syncthreads();                                               // h_s visible; everyone is done with h_{s-1}'s buffer
      }
    }
  }
}

// Persistent BPTT: same structure.  The workgroup's W_hh^T fragments (K = 3 gate blocks of H) stay in registers, the
// dgh rows of all three gate blocks are staged at once (83 KB of LDS at H = 350), dgi and the D z carry stay private.
template <int MAXK2, typename TG>
__global__ __launch_bounds__(512) void gru_bwd_persistent_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ gates, const float* __restrict__ whhT,
                                                                 TG* __restrict__ dgi, float* dgh, int* status, int B, int T,
                                                                 int H, int nJ, int nbb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KP = 128 * MAXK2, WS2 = (KP + 48) / 2;
  unsigned* Ahi = reinterpret_cast<unsigned*>(smem);                   // [3 gate blocks][16][WS2]  dgh rows, hi parts
  unsigned* Alo = Ahi + 3 * GP_BM * WS2;
  float* red = reinterpret_cast<float*>(Alo + 3 * GP_BM * WS2);        // [4 K quarters][16][GP_RED_B]
  float* SCA = red + 4 * GP_BM * GP_RED_B;                             // [16] inverse scale of the staged dgh rows (rows w, w + 8: wave w's)
  const GpWork wk = gp_decode(nJ, nbb);
  if (!wk.valid) return;
  const int d = wk.d, bb = wk.bb, jb = wk.jb;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, i = lane & 15;
  const int kq = w & 3, ch = w >> 2;
  const int j0 = jb * GP_NU, b0 = bb * GP_BM;
  const int K = 3 * H;
  const int jj = tid & 31, j = j0 + jj, jc = min(j, H - 1);
  const int row = tid >> 5, b = b0 + row, bc = min(b, B - 1);

  u32x4 wh[3][MAXK2], wl[3][MAXK2];
  float iw[3];                                                         // 1 / s of W_hh^T[unit 16 ch + i][:] (one scale over the 3 gate blocks: iw[0])
  {
    const int ju = j0 + 16 * ch + i;
    load_resident_w<MAXK2, true>(wh, wl, iw, [&](int g) -> const float* { return ju < H ? whhT + ((size_t)d * H + ju) * K + g * H : nullptr; },
                                 H, kq, ch, q, i, w, red);
  }
  float dhz_own = 0.f;                                                 // D_{p+1} z_{p+1} of this thread's element
  bool dead = false;

  for (int s = 0; s < T; ++s) {
    const int t = d ? s : T - 1 - s;
    const int tn = d ? t - 1 : t + 1, tp = d ? t + 1 : t - 1;
    GRU_STAMP(0);
    float e_dy, e_g[4], e_hp;
    {
      e_dy = dy[((size_t)bc * T + t) * 2 * H + d * H + jc];
      const float* gp = gates + (((size_t)bc * T + t) * 2 + d) * 4 * H + jc;
#pragma unroll
      for (int g = 0; g < 4; ++g) e_g[g] = gp[g * H];
      const int tpc = min(max(tp, 0), T - 1);
      e_hp = y[((size_t)bc * T + tpc) * 2 * H + d * H + jc];
    }
    if (s > 0) {
      float2 v[6][MAXK2];                                              // wave w stages segments w + 8 m = (gate block, row)
      poll_segments<6, MAXK2>(v, [&](int m) -> const float* {
        const int idx = w + 8 * m, g = idx >> 4, br = b0 + (idx & 15);
        return br < B ? dgh + (((size_t)br * T + tn) * 2 + d) * K + g * H : nullptr;
      }, H, lane, dead, status);
      GRU_STAMP(1);
      {
        // ONE power-of-two scale for the six segments this wave stages (rows w and w + 8 of the three gate blocks): the maximum
        // of what it just loaded -- one wave reduction per step.  A row's scale is then the same in all three gate blocks, so the
        // contraction over them stays ONE accumulator chain and the scale leaves in the epilogue per accumulator row.  (A scale per
        // segment with one accumulator per gate block cost 0.8 us per step: six dependent wave reductions on the critical path
        // between the hand-off's arrival and the commit.)
        float mm = 0.f;
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int c = 0; c < MAXK2; ++c) mm = fmaxf(mm, fmaxf(fabsf(v[m][c].x), fabsf(v[m][c].y)));
        const float sc = scale_for_absmax(wave_max_nonneg(mm));
#pragma unroll
        for (int m = 0; m < 6; ++m) commit_split_row<MAXK2>(Ahi, Alo, WS2, w + 8 * m, v[m], lane, sc);
        if (lane < 2) SCA[w + 8 * lane] = inv_pow2(sc);
      }
      __syncthreads();
      GRU_STAMP(2);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const unsigned* ahp = Ahi + (g * GP_BM + i) * WS2 + 4 * q + kq * MAXK2 * 16;
        const unsigned* alp = Alo + (g * GP_BM + i) * WS2 + 4 * q + kq * MAXK2 * 16;
#pragma unroll
        for (int kk = 0; kk < MAXK2; ++kk) {
          const u32x4 ah = *reinterpret_cast<const u32x4*>(ahp + 16 * kk);
          const u32x4 al = *reinterpret_cast<const u32x4*>(alp + 16 * kk);
          acc = mfma_h3(ah, al, wh[g][kk], wl[g][kk], acc);
        }
      }
      {
        const float4 ia = *reinterpret_cast<const float4*>(SCA + 4 * q);       // rows 4 q + r of the accumulator
        acc[0] *= ia.x * iw[0]; acc[1] *= ia.y * iw[0]; acc[2] *= ia.z * iw[0]; acc[3] *= ia.w * iw[0];
      }
      GRU_STAMP(3);
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(kq * GP_BM + 4 * q + r) * GP_RED_B + 16 * ch + i] = acc[r];
      __syncthreads();
      GRU_STAMP(4);
    }
    if (b < B && j < H) {
      float D = e_dy;
      if (s > 0) {
        D += dhz_own;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) D += red[(ww * GP_BM + row) * GP_RED_B + jj];
      }
      const float r = e_g[0], z = e_g[1], n = e_g[2], hn = e_g[3];
      const float hp = (s < T - 1) ? e_hp : 0.f;
      const float dn = D * (1.f - z) * (1.f - n * n);
      const float dz = D * (hp - n) * z * (1.f - z);
      const float dr = dn * hn * r * (1.f - r);
      const size_t o = (((size_t)b * T + t) * 2 + d) * K + j;
      st_sc1_f(dgh + o, dr); st_sc1_f(dgh + o + H, dz); st_sc1_f(dgh + o + 2 * H, dn * r);   // handed off
      dgi[o] = (TG)dr; dgi[o + H] = (TG)dz; dgi[o + 2 * H] = (TG)dn;
      dhz_own = D * z;
    }
    GRU_STAMP(5);
  }
}

// The same for the backward (BPTT of one direction, reference: multimodal_context_net.py:236-237 through autograd): workgroup =
// (16 batch rows, direction), all T steps, nothing crosses workgroups.  Per step, with the step processed before it being the
// direction's NEXT time step:   D = dy_t + D' z' + [dr' | dz' | dn' r'] W_hh   (the last term: 16 x 192 panel in LDS times the
// wave's resident split fragments of W_hh^T, 18 MFMAs),   dn = D (1 - z)(1 - n^2),  dz = D (h_{t-1} - n) z (1 - z),
// dr = dn hn r (1 - r);  dgi = (dr, dz, dn),  dgh = (dr, dz, dn r) (what the weight-gradient GEMMs and the next step read).
// Wave w owns units [16 w, 16 w + 16); a lane's four accumulator rows are its four batch rows, so D' z' stays in registers.
// The step's inputs (dy, the four saved gate values, h_{t-1}) of the next GS_AHEAD steps are kept in flight.
template <typename TG>
__global__ __launch_bounds__(256) void gru_bwd_small_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ gates, const float* __restrict__ whhT,
                                                            TG* __restrict__ dgi, float* __restrict__ dgh, int B, int T) {
  constexpr int H = GS_H, K = 3 * GS_H, LDP = 3 * GS_H + 4;             // panel rows of 196 floats: 16-byte aligned, 4 (mod 64)
  __shared__ __attribute__((aligned(16))) float pbuf[2][16][LDP];
  const int d = blockIdx.x & 1, b0 = (blockIdx.x >> 1) * 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, i = lane & 15;
  const int j = 16 * w + i;                                            // this lane's unit (C layout column)
  u32x4 wh[3][2], wl[3][2];
  float iw;                                                            // 1 / s of this unit's row of W_hh^T (all 3H columns: one accumulator)
  {
    const float* wrow0 = whhT + ((size_t)d * H + j) * K;               // W_hh^T: row = unit, columns = the 3H recurrent pre-activations
    float m = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) m = absmax_w8(absmax_w8(m, wrow0 + g * H, 8 * q, H), wrow0 + g * H, 32 + 8 * q, H);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    const float sc = scale_for_absmax(m);
    iw = inv_pow2(sc);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const Split8 f = load_w_frag(wrow0 + g * H, 32 * ks + 8 * q, H, sc);
        wh[g][ks] = f.hi; wl[g][ks] = f.lo;
      }
  }
  int brow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) brow[r] = min(b0 + 4 * q + r, B - 1);
  float dhz_own[4] = {0.f, 0.f, 0.f, 0.f};
  float in_dy[GS_AHEAD][4], in_g[GS_AHEAD][4][4], in_hp[GS_AHEAD][4];
  auto fetch = [&](int slot, int s) {
    const int t = d ? s : T - 1 - s;
    const int tp = d ? t + 1 : t - 1, tpc = min(max(tp, 0), T - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      in_dy[slot][r] = dy[((size_t)brow[r] * T + t) * 2 * H + d * H + j];
      const float* gp = gates + (((size_t)brow[r] * T + t) * 2 + d) * 4 * H + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) in_g[slot][r][g] = gp[g * H];
      in_hp[slot][r] = y[((size_t)brow[r] * T + tpc) * 2 * H + d * H + j];
    }
  };
#pragma unroll
  for (int u = 0; u < GS_AHEAD; ++u)
    if (u < T) fetch(u, u);

  for (int s0 = 0; s0 < T; s0 += GS_AHEAD) {
#pragma unroll
    for (int u = 0; u < GS_AHEAD; ++u) {
      const int s = s0 + u;
      if (s < T) {                                                     // (uniform over the workgroup)
        const int t = d ? s : T - 1 - s;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
          const float* pp = &pbuf[(s - 1) & 1][i][8 * q];
          // the wave reads the whole 16 x 192 panel: ONE power-of-two scale from its maximum (every wave finds the same number)
          float4 pa[3][2], pb[3][2];
          float mm = 0.f;
#pragma unroll
          for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              pa[g][ks] = *reinterpret_cast<const float4*>(pp + g * H + 32 * ks);
              pb[g][ks] = *reinterpret_cast<const float4*>(pp + g * H + 32 * ks + 4);
              mm = absmax4(absmax4(mm, pa[g][ks]), pb[g][ks]);
            }
          const float sc = scale_for_absmax(wave_max_nonneg(mm));
#pragma unroll
          for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const Split8 a = split8h(pa[g][ks], pb[g][ks], sc);
              acc = mfma_h3(a.hi, a.lo, wh[g][ks], wl[g][ks], acc);
            }
          const float k = inv_pow2(sc) * iw;
          acc[0] *= k; acc[1] *= k; acc[2] *= k; acc[3] *= k;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float D = in_dy[u][r] + (s > 0 ? dhz_own[r] + acc[r] : 0.f);
          const float gr = in_g[u][r][0], gz = in_g[u][r][1], gn = in_g[u][r][2], hn = in_g[u][r][3];
          const float hp = (s < T - 1) ? in_hp[u][r] : 0.f;
          const float dn = D * (1.f - gz) * (1.f - gn * gn);
          const float dz = D * (hp - gn) * gz * (1.f - gz);
          const float dr = dn * hn * gr * (1.f - gr);
          float* pr = &pbuf[s & 1][4 * q + r][j];
          pr[0] = dr; pr[H] = dz; pr[2 * H] = dn * gr;
          const int b = b0 + 4 * q + r;
          if (b < B) {
            const size_t o = (((size_t)b * T + t) * 2 + d) * K + j;
            dgh[o] = dr; dgh[o + H] = dz; dgh[o + 2 * H] = dn * gr;
            dgi[o] = (TG)dr; dgi[o + H] = (TG)dz; dgi[o + 2 * H] = (TG)dn;
          }
          dhz_own[r] = D * gz;
        }
        if (s + GS_AHEAD < T) fetch(u, s + GS_AHEAD);
        __syncthreads();                                               // this step's panel visible; everyone is done with the other
      }
    }
  }
}

// Before a persistent launch: zero the status word and fill the hand-off array with the "not written yet" pattern.  A
// kernel, not hipMemsetAsync: inside a replayed hipGraph a memset node was observed to land AFTER the dependent persistent
// kernel had started; kernel -> kernel ordering holds in graphs and eagerly alike.
__global__ __launch_bounds__(256) void gru_prepare_kernel(int* __restrict__ ws, int nws, u32x2* __restrict__ fill, size_t npairs) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (gid < (size_t)nws) ws[gid] = 0;
  for (size_t k = gid; k < npairs; k += stride) fill[k] = u32x2{GRU_SENTINEL, GRU_SENTINEL};
}

static void gru_prepare(void* ws, size_t ws_bytes, float* handoff, size_t n_floats, hipStream_t st) {
  const int nws = (int)(ws_bytes / sizeof(int));
  const size_t npairs = n_floats / 2;                                   // (n_floats is even: H is)
  const size_t want = (npairs > (size_t)nws ? npairs : (size_t)nws) / 256 + 1;
  const int blocks = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(gru_prepare_kernel, dim3(blocks), dim3(256), 0, st, static_cast<int*>(ws), nws,
                     reinterpret_cast<u32x2*>(handoff), npairs);
}

// The two operands the backward of a layer needs besides what the forward saved, in ONE launch (they were a transpose copy, a
// fill and two strided copies per layer): whhT[d][h][g] = whh[d][g][h] for the recurrence's backward (dh = dgh . W_hh reads W_hh
// along g), and hprev[b][t][d][:] = the state in front of step t of direction d -- y shifted one step along the direction's
// processing order, zero at its first step -- for dW_hh = sum dgh^T hprev.  Blocks [0, n_shift) shift, the rest transpose 32 x 32
// tiles through LDS.
__global__ __launch_bounds__(256) void gru_bwd_operands_kernel(const float* __restrict__ y, const float* __restrict__ whh,
                                                               float* __restrict__ hprev, float* __restrict__ whhT, int B, int T, int H,
                                                               int n_shift) {
  __shared__ float tile[32][33];
  const int bid = blockIdx.x;
  if (bid < n_shift) {
    const size_t n2 = (size_t)B * T * H;             // float2 elements of (B, T, 2, H)
    const int H2 = H / 2;
    for (size_t i = (size_t)bid * 256 + threadIdx.x; i < n2; i += (size_t)n_shift * 256) {
      const int c = (int)(i % H2);
      const size_t r = i / H2;                       // (b, t, d)
      const int d = (int)(r & 1);
      const size_t bt = r >> 1;
      const int t = (int)(bt % T);
      const bool live = d == 0 ? t > 0 : t + 1 < T;
      float2 v = make_float2(0.f, 0.f);
      if (live) v = reinterpret_cast<const float2*>(y + (d == 0 ? bt - 1 : bt + 1) * 2 * H + (size_t)d * H)[c];
      reinterpret_cast<float2*>(hprev + bt * 2 * H + (size_t)d * H)[c] = v;
    }
    return;
  }
  // whh (2, 3H, H) -> whhT (2, H, 3H)
  const int G = 3 * H, tg = (G + 31) / 32, th = (H + 31) / 32;
  int tb = bid - n_shift;
  const int d = tb / (tg * th);
  tb -= d * tg * th;
  const int g0 = (tb / th) * 32, h0 = (tb % th) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  const float* src = whh + (size_t)d * G * H;
  float* dst = whhT + (size_t)d * G * H;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int g = g0 + ty + 8 * k, h = h0 + tx;
    tile[ty + 8 * k][tx] = (g < G && h < H) ? src[(size_t)g * H + h] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int h = h0 + ty + 8 * k, g = g0 + tx;
    if (h < H && g < G) dst[(size_t)h * G + g] = tile[tx][ty + 8 * k];
  }
}

static int gru_validate(const void* const* ptrs, int n, int B, int T, int H) {
  for (int i = 0; i < n; ++i)
    if (!ptrs[i]) { set_error("hopmi_gru: null pointer argument #%d", i); return HOPMI_EINVAL; }
  if (B <= 0 || T <= 0) { set_error("hopmi_gru: B=%d T=%d must be > 0", B, T); return HOPMI_EINVAL; }
  if (H < 2 || (H & 1) || H > 384) { set_error("hopmi_gru: hidden size H=%d must be even and in [2,384] (LDS panel budget)", H); return HOPMI_EINVAL; }
  return HOPMI_OK;
}

template <int MAXK2, typename TG>
static void launch_gru_fwd(dim3 grid, size_t lds, hipStream_t st, const TG* gi, const float* whh, const float* bhh,
                           float* y, float* gates, int B, int T, int H, int KP) {
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL((gru_fwd_step_kernel<MAXK2, TG>), grid, dim3(256), lds, st, gi, whh, bhh, y, gates, B, T, H, KP, s);
}

template <int MAXK2, typename TG>
static void launch_gru_bwd(dim3 grid, size_t lds, hipStream_t st, const float* dy, const float* y, const float* gates,
                           const float* whhT, TG* dgi, float* dgh, float* ws, int B, int T, int H, int KP) {
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL((gru_bwd_step_kernel<MAXK2, TG>), grid, dim3(256), lds, st, dy, y, gates, whhT, dgi, dgh, ws, B, T, H, KP, s);
}

}  // namespace hopmi

using namespace hopmi;

static int gru_num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 1;
  }
  return cus;
}


// How many workgroups of a persistent kernel the device holds at once: the runtime's occupancy answer for this kernel, block
// size and LDS request, times the CUs (cached per kernel; queried outside any stream capture: the first call of a shape is an
// eager one).  These 512-thread, LDS-heavy kernels are admitted once or twice per CU by LDS and registers, far from the 7-8-per-CU
// edge where the query reads one high (MI355X_MICROARCH.md, residency): the answer is used as it stands, capped at 2.
static int gru_resident_capacity(const void* fn, size_t lds) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> cache;
  std::lock_guard<std::mutex> lk(mu);
  for (const auto& e : cache)
    if (e.first == fn) return e.second;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, lds) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    per_cu = 1;                                      // the launch itself would fail if not even one fitted
  }
  if (per_cu > 2) per_cu = 2;
  const int cap = per_cu * gru_num_cus();
  cache.emplace_back(fn, cap);
  return cap;
}

template <bool FWD, typename TG, int MR = 1>
static int gru_persistent_capacity(int H) {
  const int mk = (H + 127) / 128;
#define HOPMI_GP_CAP(MK_)                                                                                                    \
  {                                                                                                                          \
    constexpr int WS2 = (128 * MK_ + 48) / 2;                                                                                \
    if (FWD && MR == 2)                                                                                                      \
      return gru_resident_capacity(reinterpret_cast<const void*>(&gru_fwd_persistent_kernel<MK_, TG, 2>),                     \
                                   (size_t)2 * 2 * GP_BM * WS2 * sizeof(unsigned) + (size_t)4 * 2 * GP_BM * GP_RED_F * sizeof(float));  \
    if (FWD)                                                                                                                 \
      return gru_resident_capacity(reinterpret_cast<const void*>(&gru_fwd_persistent_kernel<MK_, TG>),                        \
                                   (size_t)2 * GP_BM * WS2 * sizeof(unsigned) + (size_t)4 * GP_BM * GP_RED_F * sizeof(float));  \
    return gru_resident_capacity(reinterpret_cast<const void*>(&gru_bwd_persistent_kernel<MK_, TG>),                          \
                                 (size_t)2 * 3 * GP_BM * WS2 * sizeof(unsigned) + (size_t)(4 * GP_BM * GP_RED_B + GP_BM) * sizeof(float)); \
  }
  if (mk <= 1) HOPMI_GP_CAP(1)
  if (mk == 2) HOPMI_GP_CAP(2)
  HOPMI_GP_CAP(3)
#undef HOPMI_GP_CAP
}

// persistent path iff the caller gave a workspace (the host side withholds it when it cannot promise that nothing else
// competes for the CUs: HOPMI_GRU_PERSISTENT=0, an RCCL exchange in flight, a shared device) and every workgroup can be
// resident at once (occupancy query x CUs)
template <bool FWD, typename TG>
static bool gru_persistent_ok(int B, int H, const void* ws) {
  const int nJ = (H + GP_NU - 1) / GP_NU, nbb = (B + GP_BM - 1) / GP_BM;
  return ws != nullptr && gp_grid(nJ, nbb) <= gru_persistent_capacity<FWD, TG>(H);
}
// the forward's 32-row form (MR = 2): for a batch whose 16-row tiling does not fit the chip
template <typename TG>
static bool gru_persistent_ok_mr2(int B, int H, const void* ws) {
  const int nJ = (H + GP_NU - 1) / GP_NU, nbb = (B + 2 * GP_BM - 1) / (2 * GP_BM);
  return ws != nullptr && env_int("HOPMI_GRU_MR2", 1) != 0 && gp_grid(nJ, nbb) <= gru_persistent_capacity<true, TG, 2>(H);
}

extern "C" size_t hopmi_gru_ws_bytes(int B, int T, int H) {
  if (B <= 0 || T <= 0 || H <= 0) return 0;
  return (size_t)32 * sizeof(int);                               // status word at int index [size - 16]; the rest is spare
}

static int* gru_status_word(void* ws, int B, int T, int H) {
  return static_cast<int*>(ws) + hopmi_gru_ws_bytes(B, T, H) / sizeof(int) - 16;
}

template <int MAXK2, typename TG, int MR = 1>
static void launch_gru_fwd_persistent(int grid, hipStream_t st, const TG* gi, const float* whh, const float* bhh, float* y,
                                      float* gates, int* status, int B, int T, int H, int nJ, int nbb, const TG* gi2, int B1) {
  constexpr int WS2 = (128 * MAXK2 + 48) / 2;
  const size_t lds = (size_t)2 * MR * GP_BM * WS2 * sizeof(unsigned) + (size_t)4 * MR * GP_BM * GP_RED_F * sizeof(float);
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_fwd_persistent_kernel<MAXK2, TG, MR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
  hipLaunchKernelGGL((gru_fwd_persistent_kernel<MAXK2, TG, MR>), dim3(grid), dim3(512), lds, st, gi, whh, bhh, y, gates, status, B, T, H,
                     nJ, nbb, gi2, B1);
}

template <typename TG>
static int gru_fwd_impl(const TG* gi, const float* whh, const float* bhh, float* y, float* gates, void* ws, int B, int T, int H,
                        void* stream, const TG* gi2 = nullptr, int B1 = -1) {
  const void* ptrs[] = {gi, whh, bhh, y, gates};
  if (int e = gru_validate(ptrs, 5, B, T, H)) return e;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool pair = gi2 != nullptr;
  if (!pair) { gi2 = gi; B1 = B; }
  if (pair && (B1 < 1 || B1 >= B)) { set_error("hopmi_gru_fwd_pair: B1=%d must lie inside (0, B=%d)", B1, B); return HOPMI_EINVAL; }
  // a pair that cannot run as ONE persistent launch runs as two plain calls (the forms below read one gi tensor)
  if (pair && !((H != GS_H || env_int("HOPMI_GRU_SMALL", 1) == 0) && (gru_persistent_ok<true, TG>(B, H, ws) || gru_persistent_ok_mr2<TG>(B, H, ws)))) {
    if (int e = gru_fwd_impl<TG>(gi, whh, bhh, y, gates, ws, B1, T, H, stream)) return e;
    return gru_fwd_impl<TG>(gi2, whh, bhh, y + (size_t)B1 * T * 2 * H, gates + (size_t)B1 * T * 2 * 4 * H, ws, B - B1, T, H, stream);
  }
  if (H == GS_H && env_int("HOPMI_GRU_SMALL", 1) != 0) {               // one workgroup per (16 rows, direction): no hand-off
    if (ws != nullptr) gru_prepare(ws, hopmi_gru_ws_bytes(B, T, H), nullptr, 0, st);                      // status = 0
    hipLaunchKernelGGL((gru_fwd_small_kernel<TG>), dim3(2 * ((B + 15) / 16)), dim3(256), 0, st, gi, whh, bhh, y, gates, B, T);
    return check_launch("hopmi_gru_fwd(small)");
  }
  if (gru_persistent_ok<true, TG>(B, H, ws)) {
    const int nJ = (H + GP_NU - 1) / GP_NU, nbb = (B + GP_BM - 1) / GP_BM;
    int* status = gru_status_word(ws, B, T, H);
    gru_prepare(ws, hopmi_gru_ws_bytes(B, T, H), y, (size_t)B * T * 2 * H, st);
    const int grid = gp_grid(nJ, nbb);
    switch ((H + 127) / 128) {
      case 1: launch_gru_fwd_persistent<1, TG>(grid, st, gi, whh, bhh, y, gates, status, B, T, H, nJ, nbb, gi2, B1); break;
      case 2: launch_gru_fwd_persistent<2, TG>(grid, st, gi, whh, bhh, y, gates, status, B, T, H, nJ, nbb, gi2, B1); break;
      default: launch_gru_fwd_persistent<3, TG>(grid, st, gi, whh, bhh, y, gates, status, B, T, H, nJ, nbb, gi2, B1); break;
    }
    return check_launch("hopmi_gru_fwd(persistent)");
  }
  if (gru_persistent_ok_mr2<TG>(B, H, ws)) {                           // 32 batch rows per workgroup (round 6)
    const int nJ = (H + GP_NU - 1) / GP_NU, nbb = (B + 2 * GP_BM - 1) / (2 * GP_BM);
    int* status = gru_status_word(ws, B, T, H);
    gru_prepare(ws, hopmi_gru_ws_bytes(B, T, H), y, (size_t)B * T * 2 * H, st);
    const int grid = gp_grid(nJ, nbb);
    switch ((H + 127) / 128) {
      case 1: launch_gru_fwd_persistent<1, TG, 2>(grid, st, gi, whh, bhh, y, gates, status, B, T, H, nJ, nbb, gi2, B1); break;
      case 2: launch_gru_fwd_persistent<2, TG, 2>(grid, st, gi, whh, bhh, y, gates, status, B, T, H, nJ, nbb, gi2, B1); break;
      default: launch_gru_fwd_persistent<3, TG, 2>(grid, st, gi, whh, bhh, y, gates, status, B, T, H, nJ, nbb, gi2, B1); break;
    }
    return check_launch("hopmi_gru_fwd(persistent, 32 rows)");
  }
  if (ws != nullptr) gru_prepare(ws, hopmi_gru_ws_bytes(B, T, H), nullptr, 0, st);                        // status = 0
  const int nJ = (H + GRU_NU - 1) / GRU_NU, nbb = (B + GRU_BM - 1) / GRU_BM;
  const dim3 grid(8 * ((2 * nJ + 7) / 8) * nbb, 1, 1);
  const int KP = ceil_to(H, 64);
  const size_t lds = ((size_t)(GRU_BM + 3 * GRU_NU) * (KP + 4) + 4 * GRU_BM * RED_LD) * sizeof(float);
  switch ((KP + 127) / 128) {
    case 1: launch_gru_fwd<1, TG>(grid, lds, st, gi, whh, bhh, y, gates, B, T, H, KP); break;
    case 2: launch_gru_fwd<2, TG>(grid, lds, st, gi, whh, bhh, y, gates, B, T, H, KP); break;
    case 3: launch_gru_fwd<3, TG>(grid, lds, st, gi, whh, bhh, y, gates, B, T, H, KP); break;
    default: launch_gru_fwd<4, TG>(grid, lds, st, gi, whh, bhh, y, gates, B, T, H, KP); break;
  }
  return check_launch("hopmi_gru_fwd");
}

extern "C" int hopmi_gru_fwd_dt(const void* gi, int gi_dtype, const float* whh, const float* bhh, float* y, float* gates, void* ws,
                                int B, int T, int H, void* stream) {
  if (gi_dtype == 1) return gru_fwd_impl(static_cast<const __bf16*>(gi), whh, bhh, y, gates, ws, B, T, H, stream);
  if (gi_dtype == 0) return gru_fwd_impl(static_cast<const float*>(gi), whh, bhh, y, gates, ws, B, T, H, stream);
  set_error("hopmi_gru_fwd_dt: gi_dtype %d (0 = fp32, 1 = bf16)", gi_dtype);
  return HOPMI_EINVAL;
}

extern "C" int hopmi_gru_fwd_pair_dt(const void* gi1, const void* gi2, int B1, int gi_dtype, const float* whh, const float* bhh, float* y,
                                     float* gates, void* ws, int B, int T, int H, void* stream) {
  if (!gi2) { set_error("hopmi_gru_fwd_pair_dt: null gi2"); return HOPMI_EINVAL; }
  if (gi_dtype == 1)
    return gru_fwd_impl(static_cast<const __bf16*>(gi1), whh, bhh, y, gates, ws, B, T, H, stream, static_cast<const __bf16*>(gi2), B1);
  if (gi_dtype == 0) return gru_fwd_impl(static_cast<const float*>(gi1), whh, bhh, y, gates, ws, B, T, H, stream, static_cast<const float*>(gi2), B1);
  set_error("hopmi_gru_fwd_pair_dt: gi_dtype %d (0 = fp32, 1 = bf16)", gi_dtype);
  return HOPMI_EINVAL;
}

extern "C" int hopmi_gru_fwd(const float* gi, const float* whh, const float* bhh, float* y, float* gates, void* ws,
                             int B, int T, int H, void* stream) {
  return gru_fwd_impl(gi, whh, bhh, y, gates, ws, B, T, H, stream);
}

extern "C" size_t hopmi_gru_bwd_ws_floats(int B, int H) {
  return (B > 0 && H > 0) ? (size_t)4 * B * H : 0;
}

template <int MAXK2, typename TG>
static void launch_gru_bwd_persistent(int grid, hipStream_t st, const float* dy, const float* y, const float* gates,
                                      const float* whhT, TG* dgi, float* dgh, int* status, int B, int T, int H, int nJ,
                                      int nbb) {
  constexpr int WS2 = (128 * MAXK2 + 48) / 2;
  const size_t lds = (size_t)2 * 3 * GP_BM * WS2 * sizeof(unsigned) + (size_t)(4 * GP_BM * GP_RED_B + GP_BM) * sizeof(float);
  hipLaunchKernelGGL((gru_bwd_persistent_kernel<MAXK2, TG>), dim3(grid), dim3(512), lds, st, dy, y, gates, whhT, dgi, dgh, status, B,
                     T, H, nJ, nbb);
}

template <typename TG>
static int gru_bwd_impl(const float* dy, const float* y, const float* gates, const float* whhT, TG* dgi, float* dgh, float* ws,
                        void* ws2, int B, int T, int H, void* stream) {
  const void* ptrs[] = {dy, y, gates, whhT, dgi, dgh, ws};
  if (int e = gru_validate(ptrs, 7, B, T, H)) return e;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (H == GS_H && env_int("HOPMI_GRU_SMALL", 1) != 0) {               // one workgroup per (16 rows, direction): no hand-off
    if (ws2 != nullptr) gru_prepare(ws2, hopmi_gru_ws_bytes(B, T, H), nullptr, 0, st);                    // status = 0
    hipLaunchKernelGGL((gru_bwd_small_kernel<TG>), dim3(2 * ((B + 15) / 16)), dim3(256), 0, st, dy, y, gates, whhT, dgi, dgh, B, T);
    return check_launch("hopmi_gru_bwd(small)");
  }
  if (gru_persistent_ok<false, TG>(B, H, ws2)) {
    const int nJ = (H + GP_NU - 1) / GP_NU, nbb = (B + GP_BM - 1) / GP_BM;
    int* status = gru_status_word(ws2, B, T, H);
    gru_prepare(ws2, hopmi_gru_ws_bytes(B, T, H), dgh, (size_t)B * T * 2 * 3 * H, st);
    const int grid = gp_grid(nJ, nbb);
    switch ((H + 127) / 128) {
      case 1: launch_gru_bwd_persistent<1, TG>(grid, st, dy, y, gates, whhT, dgi, dgh, status, B, T, H, nJ, nbb); break;
      case 2: launch_gru_bwd_persistent<2, TG>(grid, st, dy, y, gates, whhT, dgi, dgh, status, B, T, H, nJ, nbb); break;
      default: launch_gru_bwd_persistent<3, TG>(grid, st, dy, y, gates, whhT, dgi, dgh, status, B, T, H, nJ, nbb); break;
    }
    return check_launch("hopmi_gru_bwd(persistent)");
  }
  if (ws2 != nullptr) gru_prepare(ws2, hopmi_gru_ws_bytes(B, T, H), nullptr, 0, st);                      // status = 0
  const int nJ = (H + GRU_NU - 1) / GRU_NU, nbb = (B + GRU_BM - 1) / GRU_BM;
  const dim3 grid(8 * ((2 * nJ + 7) / 8) * nbb, 1, 1);
  const int KP = ceil_to(H, 64);
  const size_t lds = ((size_t)(GRU_BM + GRU_NU) * (KP + 4) + 4 * GRU_BM * RED_LD) * sizeof(float);
  switch ((KP + 127) / 128) {
    case 1: launch_gru_bwd<1, TG>(grid, lds, st, dy, y, gates, whhT, dgi, dgh, ws, B, T, H, KP); break;
    case 2: launch_gru_bwd<2, TG>(grid, lds, st, dy, y, gates, whhT, dgi, dgh, ws, B, T, H, KP); break;
    case 3: launch_gru_bwd<3, TG>(grid, lds, st, dy, y, gates, whhT, dgi, dgh, ws, B, T, H, KP); break;
    default: launch_gru_bwd<4, TG>(grid, lds, st, dy, y, gates, whhT, dgi, dgh, ws, B, T, H, KP); break;
  }
  return check_launch("hopmi_gru_bwd");
}

extern "C" int hopmi_gru_bwd_dt(const float* dy, const float* y, const float* gates, const float* whhT, void* dgi, int dgi_dtype,
                                float* dgh, float* ws, void* ws2, int B, int T, int H, void* stream) {
  if (dgi_dtype == 1) return gru_bwd_impl(dy, y, gates, whhT, static_cast<__bf16*>(dgi), dgh, ws, ws2, B, T, H, stream);
  if (dgi_dtype == 0) return gru_bwd_impl(dy, y, gates, whhT, static_cast<float*>(dgi), dgh, ws, ws2, B, T, H, stream);
  set_error("hopmi_gru_bwd_dt: dgi_dtype %d (0 = fp32, 1 = bf16)", dgi_dtype);
  return HOPMI_EINVAL;
}

extern "C" int hopmi_gru_bwd_operands(const float* y, const float* whh, float* hprev, float* whhT, int B, int T, int H, void* stream) {
  const void* ptrs[] = {y, whh, hprev, whhT};
  if (int e = gru_validate(ptrs, 4, B, T, H)) return e;
  const size_t n2 = (size_t)B * T * H;
  const int n_shift = (int)((n2 + 256 * 8 - 1) / (256 * 8) < 2048 ? (n2 + 256 * 8 - 1) / (256 * 8) : 2048);
  const int n_tr = 2 * ((3 * H + 31) / 32) * ((H + 31) / 32);
  hipLaunchKernelGGL(gru_bwd_operands_kernel, dim3(n_shift + n_tr), dim3(256), 0, static_cast<hipStream_t>(stream), y, whh, hprev, whhT,
                     B, T, H, n_shift);
  return check_launch("hopmi_gru_bwd_operands");
}

extern "C" int hopmi_gru_bwd(const float* dy, const float* y, const float* gates, const float* whhT,
                             float* dgi, float* dgh, float* ws, void* ws2, int B, int T, int H, void* stream) {
  return gru_bwd_impl(dy, y, gates, whhT, dgi, dgh, ws, ws2, B, T, H, stream);
}

HOPMI_SPLIT_STATUS_SETTER(gru)
