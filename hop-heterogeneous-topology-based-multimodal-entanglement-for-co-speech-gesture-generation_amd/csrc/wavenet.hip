// One fused WaveNet layer of the HOP graph-wavenet block (reference: model/gwnet.py:181-237), forward:
//
//   r^      = BN_{i-1}(y_{i-1})                      applied ON LOAD as r^ = y*scale + shift (identity for layer 0)
//   u       = tanh(Wf0 r^[t] + Wf1 r^[t+d] + bf) * sigmoid(Wg0 r^[t] + Wg1 r^[t+d] + bg)     gwnet.py:186-200
//   utail   = u[last 4 frames]                       the only part of the skip path that reaches the output
//   y_i     = Wm.[u; uA1; uA2] + bm + r^[t+d]        graph conv + residual, gwnet.py:224-233 (pre-BN)
//   stats   = per-channel sum / sum of squares of y_i (BatchNorm2d batch statistics, gwnet.py:237)
//
// replacing ~15 aten launches and 10 activation round trips per layer by one kernel: the activation tile
// is read once (two time taps), everything between lives in LDS/registers, y_i is written once.
//
// Work decomposition: the output rows (clip b, frame t', node v) are flat, row = slab*V + v with
// slab = b*T_out + t'; a workgroup (4 waves) takes tiles of S consecutive output slabs (<= 80 rows) and
// walks tiles persistently, keeping in registers its 16-output-channel slices of the TCN weights
// (4 x 64 x 16) and of Wm (192 x 16).  Per tile: the two tap panels R0 = r^[t'], R1 = r^[t'+d] go
// HBM -> registers -> (normalise) -> LDS; TCN on exact-fp32 MFMA with A operands read as ds_read_b128; the
// gate; node mix on MFMA; channel contraction on MFMA; epilogue adds bias + residual (from the R1 panel),
// stores y and accumulates the BatchNorm partial sums.  A tiny second kernel turns the per-workgroup
// partials into mean / rstd / running stats / the next layer's scale+shift in a fixed order (reproducible).
#include <hip/hip_ext.h>

#include "wn_dev.h"

namespace hopmi {

// gate non-linearities on the hardware exp (v_exp_f32): absolute error ~1e-7, far inside the 1e-3 bar
__device__ __forceinline__ float sigmoid_(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 2.f * __frcp_rn(1.f + __expf(-2.f * x)) - 1.f; }

// per-thread description of the NIT rows this thread streams in a tile (the idx -> row map is the same
// for the load phase and the skip-tail store phase)
template <int NIT>
struct RowMap {
  int in0[NIT];     // float4 index of the tap-0 source row start (+ c4), clamped into the tensor
  int tail[NIT];    // float4 index into utail (+ c4) or -1
  bool ok[NIT];     // row < R
};

// weight slices of wave w (16 output channels), K permuted as k = 16i + 4q + e, read straight from the Conv2d
// layout [out][in][1][tap]: the 8 floats at (o*64 + c)*2 hold both taps of 4 consecutive input channels
__device__ __forceinline__ void load_tcn_slices(float4 (&wt)[2][2][4], const float* __restrict__ wf, const float* __restrict__ wg,
                                                int w, int q, int j, int off) {
#pragma unroll
  for (int gate = 0; gate < 2; ++gate) {
    const float4* wp = reinterpret_cast<const float4*>((gate ? wg : wf) + (size_t)((16 * w + j) * C + 4 * q) * 2 + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v0 = wp[8 * i], v1 = wp[8 * i + 1];             // (c0t0 c0t1 c1t0 c1t1) (c2t0 c2t1 c3t0 c3t1)
      wt[gate][0][i] = make_float4(v0.x, v0.z, v1.x, v1.z);
      wt[gate][1][i] = make_float4(v0.y, v0.w, v1.y, v1.w);
    }
  }
}

__device__ __forceinline__ void load_wm_slice(float4 (&wreg)[12], const float* __restrict__ Wm, int w, int q, int j, int off) {
  const float4* wp = reinterpret_cast<const float4*>(Wm + (size_t)(16 * w + j) * K3 + 4 * q + off);
#pragma unroll
  for (int i = 0; i < 12; ++i) wreg[i] = wp[4 * i];
}

// HOIST = the workgroup walks several tiles: weight slices are loaded once and stay in registers.
// !HOIST = one tile per workgroup: the TCN slices are loaded behind the tile's activation loads and the Wm
// slice after the TCN phase, which starts the activation stream earlier (8 % faster at V=9, B=128).
template <int MT, bool HOIST>
__global__ __launch_bounds__(256) void wn_layer_fwd_kernel(const float* __restrict__ xin, const float* __restrict__ scsh,
                                                           const float* __restrict__ wf, const float* __restrict__ wg,
                                                           const float* __restrict__ bfp, const float* __restrict__ bgp,
                                                           const float* __restrict__ prep, const float* __restrict__ Wm,
                                                           const float* __restrict__ bm, float* __restrict__ y,
                                                           float* __restrict__ fs, float* __restrict__ utail,
                                                           float* __restrict__ stats_part, LayerGeom L, int do_gcn, int utail_ld4) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GcnGeom& g = L.g;
  constexpr int NIT = rows_nit(MT);
  float* R0 = smem;                                // [rows_lds][LDD]  r^ at frame t'
  float* R1 = R0 + g.rows_lds * LDD;               // [rows_lds][LDD]  r^ at frame t'+d
  float* Hc = R1 + g.rows_lds * LDD;               // [rows_lds][LDH]  u | uA1 | uA2
  float* AT = Hc + g.rows_lds * LDH;               // [KP][ldA]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int V = g.V, c4 = tid & 15;
  const int shift4 = L.d * V * 16;                 // tap-1 row offset in float4 units

  // MFMA products are taken transposed (D[i = channel][j = row]): a lane holds 4 consecutive channels
  // 16w + 4q + r of one row, so gate outputs, saved gates and y move as 16-byte LDS / global accesses
  const float4 bf4 = *reinterpret_cast<const float4*>(bfp + 16 * w + 4 * q);
  const float4 bg4 = *reinterpret_cast<const float4*>(bgp + 16 * w + 4 * q);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (do_gcn) {
    bias4 = *reinterpret_cast<const float4*>(bm + 16 * w + 4 * q);
    PrepRegs mr;
    prep_issue(mr, prep, g.KP * g.ldA, tid);
    prep_commit(AT, mr, g.KP * g.ldA, tid);
  }
  float4 wt[2][2][4];                              // [gate f/g][tap][i]
  float4 wreg[12];
  if (HOIST) {
    load_tcn_slices(wt, wf, wg, w, q, j, 0);
    if (do_gcn) load_wm_slice(wreg, Wm, w, q, j, 0);
  }
  const float4 sc4 = reinterpret_cast<const float4*>(scsh)[c4];
  const float4 sh4 = reinterpret_cast<const float4*>(scsh + C)[c4];
  f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};   // BatchNorm partial sums of channels 16w + 4q + r
  // the 4 padding rows behind the tile are read by the node mix's K padding (times zero): keep them finite
  for (int idx = tid; idx < 4 * C; idx += 256) Hc[(16 * MT + idx / C) * LDH + idx % C] = 0.f;

  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    HOPMI_STAMP(0);
    const int slab0 = tile * g.S;
    const int nsl = min(g.S, L.n_slabs - slab0);
    const int R = nsl * V;
    const size_t orow0 = (size_t)slab0 * V;        // first flat output row of the tile

    // ---- phase 0: stream both tap panels, normalise, commit to LDS --------------------------------
    RowMap<NIT> rm;
    RowRegs<NIT> x0r, x1r;
    {
      const float4* src4 = reinterpret_cast<const float4*>(xin);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = (tid >> 4) + 16 * it;
        const int rc = min(row, R - 1);
        const int s = (int)((rc + 0.5f) * L.invV);
        const int v = rc - s * V;
        const int slab = slab0 + s;
        const int b = (int)((slab + 0.5f) * L.invT);
        const int tp = slab - b * L.T_out;
        rm.ok[it] = row < R;
        rm.in0[it] = ((b * L.T_in + tp) * V + v) * 16 + c4;
        rm.tail[it] = (rm.ok[it] && tp >= L.T_out - 4) ? ((b * 4 + tp - (L.T_out - 4)) * V + v) * utail_ld4 + c4 : -1;
        x0r.v[it] = src4[rm.in0[it]];
        x1r.v[it] = src4[rm.in0[it] + shift4];
      }
    }
    int woff = 0;
    if (!HOIST) {
      asm volatile("" : "+v"(woff));               // keeps the weight loads inside the tile loop
      load_tcn_slices(wt, wf, wg, w, q, j, woff);
    }
    HOPMI_STAMP(1);
    {
      __syncthreads();                             // previous tile's LDS fully consumed
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = (tid >> 4) + 16 * it;
        if (row < g.rows_lds) {
          float4 a = x0r.v[it], b2 = x1r.v[it];
          a = make_float4(a.x * sc4.x + sh4.x, a.y * sc4.y + sh4.y, a.z * sc4.z + sh4.z, a.w * sc4.w + sh4.w);
          b2 = make_float4(b2.x * sc4.x + sh4.x, b2.y * sc4.y + sh4.y, b2.z * sc4.z + sh4.z, b2.w * sc4.w + sh4.w);
          if (!rm.ok[it]) { a = make_float4(0.f, 0.f, 0.f, 0.f); b2 = a; }
          *reinterpret_cast<float4*>(R0 + row * LDD + 4 * c4) = a;
          *reinterpret_cast<float4*>(R1 + row * LDD + 4 * c4) = b2;
        }
      }
    }
    __syncthreads();
    HOPMI_STAMP(2);

    // ---- phase 1: gated TCN (two taps, two gates) on MFMA, gate, u -> LDS --------------------------
    {
      f32x4 af[MT], ag[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { af[mt] = {0.f, 0.f, 0.f, 0.f}; ag[mt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int tap = 0; tap < 2; ++tap) {
        const float* ra = (tap ? R1 : R0) + j * LDD + 4 * q;      // A[i = row][k = 16i + 4q + e]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float4 a[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(ra + 16 * mt * LDD + 16 * i);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            af[mt] = mfma16(wt[0][tap][i].x, a[mt].x, af[mt]);
            ag[mt] = mfma16(wt[1][tap][i].x, a[mt].x, ag[mt]);
            af[mt] = mfma16(wt[0][tap][i].y, a[mt].y, af[mt]);
            ag[mt] = mfma16(wt[1][tap][i].y, a[mt].y, ag[mt]);
            af[mt] = mfma16(wt[0][tap][i].z, a[mt].z, af[mt]);
            ag[mt] = mfma16(wt[1][tap][i].z, a[mt].z, ag[mt]);
            af[mt] = mfma16(wt[0][tap][i].w, a[mt].w, af[mt]);
            ag[mt] = mfma16(wt[1][tap][i].w, a[mt].w, ag[mt]);
          }
        }
      }
      HOPMI_STAMP(3);
      // af[mt][r] = filter pre-activation of row 16mt + j, channel 16w + 4q + r
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = 16 * mt + j;
        const float4 f = make_float4(tanh_(af[mt][0] + bf4.x), tanh_(af[mt][1] + bf4.y), tanh_(af[mt][2] + bf4.z), tanh_(af[mt][3] + bf4.w));
        const float4 sg = make_float4(sigmoid_(ag[mt][0] + bg4.x), sigmoid_(ag[mt][1] + bg4.y), sigmoid_(ag[mt][2] + bg4.z),
                                      sigmoid_(ag[mt][3] + bg4.w));
        *reinterpret_cast<float4*>(Hc + row * LDH + 16 * w + 4 * q) = make_float4(f.x * sg.x, f.y * sg.y, f.z * sg.z, f.w * sg.w);
        if (fs != nullptr && row < R) {
          float* fp = fs + (orow0 + row) * (2 * C) + 16 * w + 4 * q;
          *reinterpret_cast<float4*>(fp) = f;
          *reinterpret_cast<float4*>(fp + C) = sg;
        }
      }
    }
    if (!HOIST && do_gcn) load_wm_slice(wreg, Wm, w, q, j, woff);
    __syncthreads();
    HOPMI_STAMP(4);

    // ---- skip tail: last 4 frames of u, LDS -> HBM as whole 256-B rows ------------------------------
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = (tid >> 4) + 16 * it;
      if (rm.tail[it] >= 0)
        reinterpret_cast<float4*>(utail)[rm.tail[it]] = *reinterpret_cast<const float4*>(Hc + row * LDH + 4 * c4);
    }

    if (do_gcn) {
      // ---- phase 2: node mix ---------------------------------------------------------------------
      node_mix_dispatch(Hc, AT, g, nsl, w, q, j);
      __syncthreads();
      HOPMI_STAMP(5);
      // ---- phase 3: channel contraction + bias + residual, y store, BatchNorm partial sums ------
      f32x4 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = {0.f, 0.f, 0.f, 0.f};
      const float* ha = Hc + j * LDH + 4 * q;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        float4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(ha + 16 * mt * LDH + 16 * i);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt] = mfma16(wreg[i].x, a[mt].x, acc[mt]);
          acc[mt] = mfma16(wreg[i].y, a[mt].y, acc[mt]);
          acc[mt] = mfma16(wreg[i].z, a[mt].z, acc[mt]);
          acc[mt] = mfma16(wreg[i].w, a[mt].w, acc[mt]);
        }
      }
      HOPMI_STAMP(6);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = 16 * mt + j;
        if (row < R) {
          const float4 res = *reinterpret_cast<const float4*>(R1 + row * LDD + 16 * w + 4 * q);       // gwnet.py:233
          const f32x4 yv = {acc[mt][0] + bias4.x + res.x, acc[mt][1] + bias4.y + res.y, acc[mt][2] + bias4.z + res.z,
                            acc[mt][3] + bias4.w + res.w};
          if (y != nullptr) *reinterpret_cast<f32x4*>(y + (orow0 + row) * C + 16 * w + 4 * q) = yv;
          st1 += yv;
          st2 += yv * yv;
        }
      }
    }
  }

  HOPMI_STAMP(7);
  if (stats_part != nullptr) {
    // sum over the 16 rows j of the DPP row (xor 1, 2, 4, 8), fixed order
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        st1[r] += __shfl_xor(st1[r], o);
        st2[r] += __shfl_xor(st2[r], o);
      }
    }
    if (j == 0) {
      *reinterpret_cast<f32x4*>(stats_part + blockIdx.x * 2 * C + 16 * w + 4 * q) = st1;
      *reinterpret_cast<f32x4*>(stats_part + blockIdx.x * 2 * C + C + 16 * w + 4 * q) = st2;
    }
  }
}

// BatchNorm2d training-mode finalisation (gwnet.py:237; torch semantics: biased variance normalises,
// unbiased variance feeds running_var, momentum 0.1): fixed-order sum of the per-workgroup partials.
__global__ __launch_bounds__(1024) void wn_bn_finalize_kernel(const float* __restrict__ part, int nblk, double n,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float momentum, float eps, float* __restrict__ scsh_out,
                                                              float* __restrict__ mean_rstd_out) {
  // 8 chunks x 128 columns (64 sums, 64 sums of squares): thread (chunk, col) adds the partials of the
  // workgroups b = chunk (mod 8) in a fixed order with 8 independent loads in flight; then the 8 chunk
  // sums are added in a fixed order (bitwise reproducible).
  __shared__ double red[8][2 * C];
  const int col = threadIdx.x & 127, chunk = threadIdx.x >> 7;
  double acc = 0.0;
  for (int b0 = chunk; b0 < nblk; b0 += 256) {       // 32 loads in flight: one memory round trip for up to 256 partials
    float v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int b = b0 + 8 * u;
      v[u] = (b < nblk) ? part[(size_t)min(b, nblk - 1) * 2 * C + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += (double)v[u];
  }
  red[chunk][col] = acc;
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1 += red[k][c]; s2 += red[k][C + c]; }
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    scsh_out[c] = sc;
    scsh_out[C + c] = beta[c] - (float)mean * sc;
    mean_rstd_out[c] = (float)mean;
    mean_rstd_out[C + c] = rstd;
    if (running_mean != nullptr) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// forward tiling: one workgroup per CU, tiles of up to 80 rows (two 48-row workgroups per CU measured slower)
constexpr int WN_GRID_DEFAULT = 256, WN_FWD_MAX_MT = WN_MAX_MT;
static LayerGeom make_fwd_geom(int B, int T_in, int V, int d) {
  return make_layer_geom(B, T_in, V, d, wn_env_int("HOPMI_WN_GRID", WN_GRID_DEFAULT), wn_env_int("HOPMI_WN_MAXMT", WN_FWD_MAX_MT));
}

static int wn_grid(const LayerGeom& L) {
  const int cap = wn_env_int("HOPMI_WN_GRID", WN_GRID_DEFAULT);
  return L.g.ntiles < cap ? L.g.ntiles : cap;
}

static thread_local hipEvent_t t_ev_start = nullptr, t_ev_stop = nullptr;    // hopmi_time_next_launch

template <int MT>
static void launch_wn_fwd(const float* xin, const float* scsh, const float* wf, const float* wg, const float* bf,
                          const float* bg, const float* prep,
                          const float* Wm, const float* bm, float* y, float* fs, float* utail, int utail_ld, float* part,
                          const LayerGeom& L, int do_gcn, int grid, hipStream_t st) {
  const GcnGeom& g = L.g;
  const size_t lds = ((size_t)g.rows_lds * (2 * LDD + LDH) + (size_t)g.KP * g.ldA) * sizeof(float);
  const hipEvent_t e0 = t_ev_start, e1 = t_ev_stop;      // null unless a measurement asked for this launch
  t_ev_start = t_ev_stop = nullptr;
  if (g.ntiles > grid)
    hipExtLaunchKernelGGL((wn_layer_fwd_kernel<MT, true>), dim3(grid), dim3(256), lds, st, e0, e1, 0, xin, scsh, wf, wg, bf, bg, prep,
                          Wm, bm, y, fs, utail, part, L, do_gcn, utail_ld / 4);
  else
    hipExtLaunchKernelGGL((wn_layer_fwd_kernel<MT, false>), dim3(grid), dim3(256), lds, st, e0, e1, 0, xin, scsh, wf, wg, bf, bg, prep,
                          Wm, bm, y, fs, utail, part, L, do_gcn, utail_ld / 4);
}

}  // namespace hopmi

using namespace hopmi;

#ifdef HOPMI_STAMPS
extern "C" int hopmi_debug_set_stamps_wn(long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int hopmi_time_next_launch(void* start_event, void* stop_event) {
  if ((start_event == nullptr) != (stop_event == nullptr)) { set_error("hopmi_time_next_launch: need both events or none"); return HOPMI_EINVAL; }
  t_ev_start = static_cast<hipEvent_t>(start_event);
  t_ev_stop = static_cast<hipEvent_t>(stop_event);
  return HOPMI_OK;
}

extern "C" size_t hopmi_wn_layer_ws_floats(int B, int T_in, int V, int dilation) {
  if (wn_validate(B, T_in, V, dilation)) return 0;
  const LayerGeom L = make_fwd_geom(B, T_in, V, dilation);
  return (size_t)wn_grid(L) * 2 * C;
}

extern "C" int hopmi_wn_layer_fwd(const float* xin, const float* scsh_in, const float* wf, const float* wg,
                                  const float* bf, const float* bg, const float* prep, const float* Wm, const float* bm, float* y, float* fs, float* utail,
                                  int utail_ld, float* ws, int B, int T_in, int V, int dilation, int do_gcn, void* stream) {
  if (int e = wn_validate(B, T_in, V, dilation)) return e;
  if (!xin || !scsh_in || !wf || !wg || !bf || !bg || !utail) { set_error("hopmi_wn_layer_fwd: null pointer argument"); return HOPMI_EINVAL; }
  if (utail_ld < C || (utail_ld & 3)) { set_error("hopmi_wn_layer_fwd: utail_ld=%d must be a multiple of 4 and >= 64", utail_ld); return HOPMI_EINVAL; }
  if (do_gcn && (!prep || !Wm || !bm)) { set_error("hopmi_wn_layer_fwd: do_gcn needs prep, Wm, bm"); return HOPMI_EINVAL; }
  const bool stats = ws != nullptr;
  if (stats && !do_gcn) { set_error("hopmi_wn_layer_fwd: batch statistics (ws) need do_gcn"); return HOPMI_EINVAL; }
  const LayerGeom L = make_fwd_geom(B, T_in, V, dilation);
  const int grid = wn_grid(L);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = stats ? ws : nullptr;
  switch (L.g.mtiles) {
    case 1: launch_wn_fwd<1>(xin, scsh_in, wf, wg, bf, bg, prep, Wm, bm, y, fs, utail, utail_ld, part, L, do_gcn, grid, st); break;
    case 2: launch_wn_fwd<2>(xin, scsh_in, wf, wg, bf, bg, prep, Wm, bm, y, fs, utail, utail_ld, part, L, do_gcn, grid, st); break;
    case 3: launch_wn_fwd<3>(xin, scsh_in, wf, wg, bf, bg, prep, Wm, bm, y, fs, utail, utail_ld, part, L, do_gcn, grid, st); break;
    case 4: launch_wn_fwd<4>(xin, scsh_in, wf, wg, bf, bg, prep, Wm, bm, y, fs, utail, utail_ld, part, L, do_gcn, grid, st); break;
    case 5: launch_wn_fwd<5>(xin, scsh_in, wf, wg, bf, bg, prep, Wm, bm, y, fs, utail, utail_ld, part, L, do_gcn, grid, st); break;
    default: set_error("hopmi_wn_layer_fwd: internal: %d m-tiles", L.g.mtiles); return HOPMI_EINVAL;
  }
  return check_launch("hopmi_wn_layer_fwd");
}

extern "C" int hopmi_wn_bn_finalize(const float* ws, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float momentum, float eps, float* scsh_out, float* mean_rstd_out,
                                    int B, int T_in, int V, int dilation, void* stream) {
  if (int e = wn_validate(B, T_in, V, dilation)) return e;
  if (!ws || !gamma || !beta || !scsh_out || !mean_rstd_out) { set_error("hopmi_wn_bn_finalize: null pointer argument"); return HOPMI_EINVAL; }
  const LayerGeom L = make_fwd_geom(B, T_in, V, dilation);
  hipLaunchKernelGGL(wn_bn_finalize_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), ws, wn_grid(L),
                     (double)L.n_slabs * V, gamma, beta, running_mean, running_var, momentum, eps, scsh_out, mean_rstd_out);
  return check_launch("hopmi_wn_bn_finalize");
}
