// Graph convolution of the HOP spatio-temporal block (reference: model/gwnet.py:8-46),
// forward and backward, exact fp32 on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   h[s] = [X[s] | A1^T X[s] | A2^T X[s]] Wm^T + bm          X[s]: V x 64 slab (b, t)
//
// HBM layout: activations are channels-last, x[row][64], row = slab*V + node, so a tile of
// S consecutive slabs is one contiguous, 256-B-row stream.  One workgroup (4 waves) owns a
// tile of S slabs (S*V rows, padded to 16-row MFMA tiles):
//   phase 0  stream the tile HBM -> LDS (float4 per lane, fully coalesced)
//   phase 1  node mix on the matrix cores: [A1^T;A2^T] (2V x V) times the slab, per slab,
//            result written next to X in LDS -> Hcat[row][192]
//   phase 2  channel contraction Hcat (rows x 192) x Wm^T (192 x 64); each wave owns 16
//            output channels and keeps its 192 x 16 slice of Wm in 48 VGPRs for the whole
//            tile, so the only per-MFMA operand traffic is one LDS dword per lane
//   epilogue bias (and in the backward: dX mix, dA / dWm / dbm partial sums)
// The K index of every contraction is permuted (lane quad q owns k = q*K/4 + step) so the
// register-resident weight slice is loaded with 16-B loads; both operands use the same
// permutation, which leaves the sum unchanged.
#include "gcn_dev.h"

namespace hopmi {

__global__ __launch_bounds__(256) void gcn_prepare_kernel(const float* __restrict__ A1, const float* __restrict__ A2,
                                                          float* __restrict__ prep, GcnGeom g) {
  __shared__ float colsum[256];
  const int V = g.V;
  float* AT = prep;
  float* AB = prep + g.KP * g.ldA;
  for (int idx = threadIdx.x; idx < g.KP * g.ldA; idx += 256) {
    const int k = idx / g.ldA, m = idx - k * g.ldA;
    float v = 0.f;
    if (k < V && m < 2 * V) v = (m < V) ? A1[k * V + m] : A2[k * V + (m - V)];
    AT[idx] = v;
  }
  for (int idx = threadIdx.x; idx < g.K2P * g.ldB; idx += 256) {
    const int k = idx / g.ldB, v = idx - k * g.ldB;
    float a = 0.f;
    if (v < V && k < 2 * V) a = (k < V) ? A1[v * V + k] : A2[v * V + (k - V)];
    AB[idx] = a;
  }
  // trailer (4 floats): the power-of-two operand scale s of the fused WaveNet kernels' contraction images [u | u A1 | u A2] and
  // its inverse.  |u| <= 1 (tanh x sigmoid) and |(u A)[w]| <= max_w sum_v |A[v][w]|: the largest column sum of |A1|, |A2| (and 1)
  // bounds every element of the images, s puts that bound in [2^14, 2^15) (f16_dev.h)
  float cs = 0.f;
  if (threadIdx.x < 2 * V) {
    const float* A = threadIdx.x < V ? A1 : A2;
    const int wcol = threadIdx.x < V ? threadIdx.x : threadIdx.x - V;
    for (int v = 0; v < V; ++v) cs += fabsf(A[v * V + wcol]);
  }
  colsum[threadIdx.x] = cs;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 1.f;
    for (int i = 0; i < 2 * V; ++i) m = fmaxf(m, colsum[i]);
    const unsigned sb = scale_bits_for_max(__float_as_uint(m));
    float* tr = prep + g.KP * g.ldA + g.K2P * g.ldB;
    tr[0] = __uint_as_float(sb);
    tr[1] = inv_scale(sb);
    tr[2] = m;
    tr[3] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// TS = storage type of x and h (float or __bf16, io_dev.h); arithmetic fp32
template <int MT, typename TS>
__global__ __launch_bounds__(256) void gcn_fwd_kernel(const TS* __restrict__ x, const float* __restrict__ prep,
                                                      const float* __restrict__ Wm,
                                                      const float* __restrict__ bm, TS* __restrict__ h,
                                                      int n_slabs, GcnGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hc = smem;                               // [rows_lds][LDH]
  float* AT = smem + g.rows_lds * LDH;            // [KP][ldA]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int V = g.V;
  const int slab0 = blockIdx.x * g.S;
  const int nsl = min(g.S, n_slabs - slab0);
  const int R = nsl * V;
  const size_t row0 = (size_t)slab0 * V;

  HOPMI_STAMP(0);
  // this wave's 192 x 16 slice of Wm^T, K permuted so that MFMA step (i, e) pairs lane quad q with
  // k = 16i + 4q + e:  wreg[i] = Wm[o = 16w + j][16i + 4q .. +3]   (one 16-B load each)
  float4 wreg[12];
  {
    const float4* wp = reinterpret_cast<const float4*>(Wm + (size_t)(16 * w + j) * K3 + 4 * q);
#pragma unroll
    for (int i = 0; i < 12; ++i) wreg[i] = wp[4 * i];
  }
  const float bias = bm[16 * w + j];

  {
    PrepRegs mr;
    RowRegs<rows_nit(MT)> xr;
    rows_issue(xr, x + row0 * C, R, tid);
    prep_issue(mr, prep, g.KP * g.ldA, tid);
    HOPMI_STAMP(1);
    rows_commit(Hc, LDH, xr, g.rows_lds, tid);
    prep_commit(AT, mr, g.KP * g.ldA, tid);
  }
  __syncthreads();
  HOPMI_STAMP(2);
  node_mix_dispatch(Hc, AT, g, nsl, w, q, j);
  __syncthreads();
  HOPMI_STAMP(3);

  // MT == g.mtiles exactly (host dispatch): no guards in the MFMA stream
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = {0.f, 0.f, 0.f, 0.f};
  const float* ha = Hc + j * LDH + 4 * q;         // A[i = row][k = 16i + 4q + e]: one b128 per (mt, i)
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    float4 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(ha + 16 * mt * LDH + 16 * i);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[mt] = mfma16(a[mt].x, wreg[i].x, acc[mt]);
      acc[mt] = mfma16(a[mt].y, wreg[i].y, acc[mt]);
      acc[mt] = mfma16(a[mt].z, wreg[i].z, acc[mt]);
      acc[mt] = mfma16(a[mt].w, wreg[i].w, acc[mt]);
    }
  }
  HOPMI_STAMP(4);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * mt + 4 * q + r;
      if (row < R) st1(h + (row0 + row) * C + 16 * w + j, acc[mt][r] + bias);
    }
  }
  HOPMI_STAMP(5);
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
constexpr int BWD_ROWS = 64;       // <= 64 rows per backward tile (LDS: two [rows][196] images + dH)
constexpr int DA_SLOTS = 5;        // ceil(3*6/4): dA accumulator tiles per wave at V <= 48

// Persistent over tiles: dWm / dbm / dA partial sums stay in registers across the block's
// tiles and are written once to part[blockIdx.x][...] (summed by gcn_bwd_reduce_kernel in a
// fixed order: reproducible, no atomics).  MT = 16-row MFMA tiles per block tile (exact).
template <int MT, typename TS>
__global__ __launch_bounds__(256) void gcn_bwd_kernel(const TS* __restrict__ x, const TS* __restrict__ dh,
                                                      const float* __restrict__ prep,
                                                      const float* __restrict__ Wm, TS* __restrict__ dx,
                                                      float* __restrict__ part, int n_slabs, GcnGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hc = smem;                                // [rows_lds][LDH]  X | X A1 | X A2
  float* Gs = Hc + g.rows_lds * LDH;               // [rows_lds][LDH]  G0 | G1 | G2 = dH Wm
  float* DH = Gs + g.rows_lds * LDH;               // [rows_lds][LDD]
  float* AT = DH + g.rows_lds * LDD;               // [KP][ldA]   forward mix matrix
  float* AB = AT + g.KP * g.ldA;                   // [K2P][ldB]  AB[k][v] = (k<V ? A1[v][k] : A2[v][k-V])
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int V = g.V;

  {
    PrepRegs mr, br;
    prep_issue(mr, prep, g.KP * g.ldA, tid);
    prep_issue(br, prep + g.KP * g.ldA, g.K2P * g.ldB, tid);
    prep_commit(AT, mr, g.KP * g.ldA, tid);
    prep_commit(AB, br, g.K2P * g.ldB, tid);
  }
  // Wm slice for G = dH Wm, K (= o) permuted as o = 16i + 4q + e:
  // wreg[b][i] = Wm[o = 16i + 4q + (0..3)][64b + 16w + j]
  float4 wreg[3][4];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* wp = Wm + (size_t)(16 * i + 4 * q) * K3 + C * b + 16 * w + j;
      wreg[b][i] = make_float4(wp[0], wp[K3], wp[2 * K3], wp[3 * K3]);
    }

  f32x4 acc_dW[4][3];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int n = 0; n < 3; ++n) acc_dW[mt][n] = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc_dA[DA_SLOTS];
#pragma unroll
  for (int sl = 0; sl < DA_SLOTS; ++sl) acc_dA[sl] = {0.f, 0.f, 0.f, 0.f};
  float acc_db = 0.f;
  const int nt_dA = g.MP >> 4;                     // column tiles of [dA1 | dA2]
  const int ntiles_dA = (g.VP >> 4) * nt_dA;

  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    const int slab0 = tile * g.S;
    const int nsl = min(g.S, n_slabs - slab0);
    const int R = nsl * V;
    const size_t row0 = (size_t)slab0 * V;

    {
      RowRegs<rows_nit(MT)> xr, dr;
      rows_issue(xr, x + row0 * C, R, tid);
      rows_issue(dr, dh + row0 * C, R, tid);
      __syncthreads();                             // previous tile fully consumed
      rows_commit(Hc, LDH, xr, g.rows_lds, tid);
      rows_commit(DH, LDD, dr, g.rows_lds, tid);
    }
    // rows >= R of the mixed columns feed the dWm contraction (times dH = 0): keep them finite
    for (int idx = tid; idx < (g.rows_lds - R) * 2 * C; idx += 256) {
      const int row = R + idx / (2 * C), c = idx % (2 * C);
      Hc[row * LDH + C + c] = 0.f;
    }
    __syncthreads();

    // (1) recompute the forward node mix -> Hcat
    node_mix_dispatch(Hc, AT, g, nsl, w, q, j);

    // (2) G = dH (rows x 64) Wm (64 x 192); wave w owns columns 64b + 16w + [0,16)
    {
      f32x4 acc[MT][3];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[mt][b] = {0.f, 0.f, 0.f, 0.f};
      const float* da = DH + j * LDD + 4 * q;      // A[i = row][k = o = 16i + 4q + e]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(da + 16 * mt * LDD + 16 * i);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            acc[mt][b] = mfma16(a[mt].x, wreg[b][i].x, acc[mt][b]);
            acc[mt][b] = mfma16(a[mt].y, wreg[b][i].y, acc[mt][b]);
            acc[mt][b] = mfma16(a[mt].z, wreg[b][i].z, acc[mt][b]);
            acc[mt][b] = mfma16(a[mt].w, wreg[b][i].w, acc[mt][b]);
          }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) Gs[(16 * mt + 4 * q + r) * LDH + C * b + 16 * w + j] = acc[mt][b][r];
    }
    __syncthreads();

    // (3) dX[s] = G0 + A1 G1 + A2 G2
    if (g.K2P == 20) dx_mix<5, TS>(Gs, AB, dx, row0, g, nsl, w, q, j);          // V = 9
    else if (g.K2P == 84) dx_mix<21, TS>(Gs, AB, dx, row0, g, nsl, w, q, j);    // V = 42
    else dx_mix<0, TS>(Gs, AB, dx, row0, g, nsl, w, q, j);

    // (4) dA{1,2}[v][w'] += sum_{s,c} X[s,v,c] G{1,2}[s,w',c]; output tiles dealt round-robin to waves;
    //     K (= c) permuted as c = 16i + 4q + e -> b128 operand reads
#pragma unroll
    for (int sl = 0; sl < DA_SLOTS; ++sl) {
      const int t = w + 4 * sl;
      if (t < ntiles_dA) {
        const int mtA = t / nt_dA, ntA = t - mtA * nt_dA;
        const int v = min(16 * mtA + j, V - 1);                  // rows >= V are discarded
        const int m = min(16 * ntA + j, 2 * V - 1);              // columns >= 2V of [dA1 | dA2] are discarded
        const int blk = (m >= V) ? 1 : 0;
        const float* xa = Hc + v * LDH + 4 * q;                              // A[i = v][k = c]
        const float* gb = Gs + (m - blk * V) * LDH + C * (1 + blk) + 4 * q;  // B[k = c][n = m]
        f32x4 acc = acc_dA[sl];
        for (int s = 0; s < nsl; ++s) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(xa + s * V * LDH + 16 * i);
            const float4 b = *reinterpret_cast<const float4*>(gb + s * V * LDH + 16 * i);
            acc = mfma16(a.x, b.x, acc);
            acc = mfma16(a.y, b.y, acc);
            acc = mfma16(a.z, b.z, acc);
            acc = mfma16(a.w, b.w, acc);
          }
        }
        acc_dA[sl] = acc;
      }
    }

    // (5) dWm[o][kk] += sum_rows dH[row][o] Hcat[row][kk]; wave w owns kk tiles 3w..3w+2
    {
      const float* da = DH + q * LDD + j;          // A[i = o][k = row = 4ks + q]
      const float* hb = Hc + q * LDH + 48 * w + j; // B[k = row][n = kk]
#pragma unroll 4
      for (int ks = 0; ks < 4 * MT; ++ks) {
        float a[4], b[3];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt] = da[4 * ks * LDD + 16 * mt];
#pragma unroll
        for (int n = 0; n < 3; ++n) b[n] = hb[4 * ks * LDH + 16 * n];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc_dW[mt][n] = mfma16(a[mt], b[n], acc_dW[mt][n]);
      }
    }

    // (6) dbm[o] += sum_rows dH[row][o]
    if (tid < C) {
      float sacc = 0.f;
      for (int row = 0; row < R; ++row) sacc += DH[row * LDD + tid];
      acc_db += sacc;
    }
  }

  // partials: [dWm 64x192][dbm 64][dA1 VxV][dA2 VxV]
  float* p = part + (size_t)blockIdx.x * (C * K3 + C + 2 * V * V);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[(16 * mt + 4 * q + r) * K3 + 48 * w + 16 * n + j] = acc_dW[mt][n][r];
  if (tid < C) p[C * K3 + tid] = acc_db;
#pragma unroll
  for (int sl = 0; sl < DA_SLOTS; ++sl) {
    const int t = w + 4 * sl;
    if (t < ntiles_dA) {
      const int mtA = t / nt_dA, ntA = t - mtA * nt_dA;
      const int m = 16 * ntA + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int v = 16 * mtA + 4 * q + r;
        if (v < V && m < 2 * V) {
          const int blk = (m >= V) ? 1 : 0;
          p[C * K3 + C + blk * V * V + v * V + (m - blk * V)] = acc_dA[sl][r];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void gcn_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int V,
                                                             float* __restrict__ dWm, float* __restrict__ dbm,
                                                             float* __restrict__ dA1, float* __restrict__ dA2) {
  const int psz = C * K3 + C + 2 * V * V;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= psz) return;
  float s = 0.f;                       // fixed order; 8 independent loads in flight per step
  for (int b0 = 0; b0 < nblk; b0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (b0 + u < nblk) ? part[(size_t)(b0 + u) * psz + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  if (i < C * K3) dWm[i] = s;
  else if (i < C * K3 + C) dbm[i - C * K3] = s;
  else if (i < C * K3 + C + V * V) dA1[i - C * K3 - C] = s;
  else dA2[i - C * K3 - C - V * V] = s;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

// Slabs per forward tile: trade MFMA-tile padding against having >= 2 workgroups per CU.
static int pick_fwd_slabs(int n_slabs, int V) {
  const int forced = env_int("HOPMI_GCN_FWD_S", 0);
  if (forced > 0 && forced * V <= 128) return forced;
  int best = 1;
  double best_score = -1.0;
  for (int S = 1; S * V <= 128; ++S) {
    const int rows = S * V, padded = ceil_to(rows, 16);
    const int ntiles = (n_slabs + S - 1) / S;
    const double score = (double)rows / padded * (ntiles >= 512 ? 1.0 : ntiles / 512.0);
    if (score >= best_score) { best_score = score; best = S; }
  }
  return best;
}

static int bwd_slabs(int V) { return BWD_ROWS / V > 0 ? BWD_ROWS / V : 1; }

static int bwd_grid(int ntiles) { return ntiles < env_int("HOPMI_GCN_BWD_GRID", 256) ? ntiles : env_int("HOPMI_GCN_BWD_GRID", 256); }

static int validate(const void* const* ptrs, int nptr, int n_slabs, int V) {
  for (int i = 0; i < nptr; ++i)
    if (!ptrs[i]) { set_error("hopmi_gcn: null pointer argument #%d", i); return HOPMI_EINVAL; }
  if (n_slabs <= 0) { set_error("hopmi_gcn: n_slabs must be > 0 (got %d)", n_slabs); return HOPMI_EINVAL; }
  if (V < 1 || V > HOPMI_MAX_NODES) { set_error("hopmi_gcn: V=%d outside [1,%d]", V, HOPMI_MAX_NODES); return HOPMI_EINVAL; }
  return HOPMI_OK;
}

template <int MT, typename TS>
static int launch_fwd(const TS* x, const float* prep, const float* Wm, const float* bm, TS* h,
                      int n_slabs, const GcnGeom& g, hipStream_t st) {
  const size_t lds = (size_t)(g.rows_lds * LDH + g.KP * g.ldA) * sizeof(float);
  // only raise the dynamic-LDS cap when a launch needs it, and only to what it needs
  static size_t attr_lds = 64 * 1024;
  if (lds > attr_lds && env_int("HOPMI_NO_LDS_ATTR", 0) == 0) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gcn_fwd_kernel<MT, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  hipLaunchKernelGGL((gcn_fwd_kernel<MT, TS>), dim3(g.ntiles), dim3(256), lds, st, x, prep, Wm, bm, h, n_slabs, g);
  return check_launch("hopmi_gcn_fwd");
}

template <int MT, typename TS>
static void launch_bwd(const TS* x, const TS* dh, const float* prep, const float* Wm, TS* dx,
                       float* ws, int n_slabs, const GcnGeom& g, int grid, size_t lds, hipStream_t st) {
  static size_t attr_lds = 64 * 1024;
  if (lds > attr_lds && env_int("HOPMI_NO_LDS_ATTR", 0) == 0) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gcn_bwd_kernel<MT, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  hipLaunchKernelGGL((gcn_bwd_kernel<MT, TS>), dim3(grid), dim3(256), lds, st, x, dh, prep, Wm, dx, ws, n_slabs, g);
}

}  // namespace hopmi

using namespace hopmi;

extern "C" size_t hopmi_gcn_prep_floats(int V) {
  if (V < 1 || V > HOPMI_MAX_NODES) return 0;
  const GcnGeom g = make_geom(1, V, 1);
  return (size_t)g.KP * g.ldA + (size_t)g.K2P * g.ldB + 4;      // + the operand-scale trailer (gcn_prepare_kernel)
}

extern "C" int hopmi_gcn_prepare(const float* A1, const float* A2, float* prep, int V, void* stream) {
  const void* ptrs[] = {A1, A2, prep};
  if (int e = validate(ptrs, 3, 1, V)) return e;
  hipLaunchKernelGGL(gcn_prepare_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), A1, A2, prep,
                     make_geom(1, V, 1));
  return check_launch("hopmi_gcn_prepare");
}

template <typename TS>
static int gcn_fwd_impl(const TS* x, const float* prep, const float* Wm, const float* bm, TS* h, int n_slabs, int V, void* stream) {
  const void* ptrs[] = {x, prep, Wm, bm, h};
  if (int e = validate(ptrs, 5, n_slabs, V)) return e;
  const GcnGeom g = make_geom(n_slabs, V, pick_fwd_slabs(n_slabs, V));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (g.mtiles) {
    case 1: return launch_fwd<1, TS>(x, prep, Wm, bm, h, n_slabs, g, st);
    case 2: return launch_fwd<2, TS>(x, prep, Wm, bm, h, n_slabs, g, st);
    case 3: return launch_fwd<3, TS>(x, prep, Wm, bm, h, n_slabs, g, st);
    case 4: return launch_fwd<4, TS>(x, prep, Wm, bm, h, n_slabs, g, st);
    case 5: return launch_fwd<5, TS>(x, prep, Wm, bm, h, n_slabs, g, st);
    case 6: return launch_fwd<6, TS>(x, prep, Wm, bm, h, n_slabs, g, st);
    case 7: return launch_fwd<7, TS>(x, prep, Wm, bm, h, n_slabs, g, st);
    case 8: return launch_fwd<8, TS>(x, prep, Wm, bm, h, n_slabs, g, st);
  }
  set_error("hopmi_gcn_fwd: internal: %d m-tiles", g.mtiles);
  return HOPMI_EINVAL;
}

extern "C" int hopmi_gcn_fwd(const float* x, const float* prep, const float* Wm, const float* bm,
                             float* h, int n_slabs, int V, void* stream) {
  return gcn_fwd_impl<float>(x, prep, Wm, bm, h, n_slabs, V, stream);
}

extern "C" int hopmi_gcn_fwd_dt(const void* x, const float* prep, const float* Wm, const float* bm, void* h, int n_slabs, int V,
                                int dtype, void* stream) {
  if (dtype == HOPMI_F32) return gcn_fwd_impl<float>(static_cast<const float*>(x), prep, Wm, bm, static_cast<float*>(h), n_slabs, V, stream);
  if (dtype == HOPMI_BF16) return gcn_fwd_impl<__bf16>(static_cast<const __bf16*>(x), prep, Wm, bm, static_cast<__bf16*>(h), n_slabs, V, stream);
  set_error("hopmi_gcn_fwd_dt: dtype %d (0 = fp32, 1 = bf16)", dtype);
  return HOPMI_EINVAL;
}

#ifdef HOPMI_STAMPS
extern "C" int hopmi_debug_set_stamps(long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" size_t hopmi_gcn_bwd_ws_floats(int n_slabs, int V) {
  if (n_slabs <= 0 || V < 1 || V > HOPMI_MAX_NODES) return 0;
  const GcnGeom g = make_geom(n_slabs, V, bwd_slabs(V));
  return (size_t)bwd_grid(g.ntiles) * (C * K3 + C + 2 * V * V);
}

template <typename TS>
static int gcn_bwd_impl(const TS* x, const TS* dh, const float* prep, const float* Wm, TS* dx, float* dA1, float* dA2, float* dWm,
                        float* dbm, float* ws, int n_slabs, int V, void* stream) {
  const void* ptrs[] = {x, dh, prep, Wm, dx, dA1, dA2, dWm, dbm, ws};
  if (int e = validate(ptrs, 10, n_slabs, V)) return e;
  const GcnGeom g = make_geom(n_slabs, V, bwd_slabs(V));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = bwd_grid(g.ntiles);
  const size_t lds = (size_t)(2 * g.rows_lds * LDH + g.rows_lds * LDD + g.KP * g.ldA + g.K2P * g.ldB) * sizeof(float);
  if (lds > 160 * 1024) { set_error("hopmi_gcn_bwd: LDS footprint %zu exceeds 160 KiB", lds); return HOPMI_EINVAL; }
  switch (g.mtiles) {
    case 1: launch_bwd<1, TS>(x, dh, prep, Wm, dx, ws, n_slabs, g, grid, lds, st); break;
    case 2: launch_bwd<2, TS>(x, dh, prep, Wm, dx, ws, n_slabs, g, grid, lds, st); break;
    case 3: launch_bwd<3, TS>(x, dh, prep, Wm, dx, ws, n_slabs, g, grid, lds, st); break;
    case 4: launch_bwd<4, TS>(x, dh, prep, Wm, dx, ws, n_slabs, g, grid, lds, st); break;
    default: set_error("hopmi_gcn_bwd: internal: %d m-tiles", g.mtiles); return HOPMI_EINVAL;
  }
  if (int e = check_launch("hopmi_gcn_bwd")) return e;
  const int psz = C * K3 + C + 2 * V * V;
  hipLaunchKernelGGL(gcn_bwd_reduce_kernel, dim3((psz + 255) / 256), dim3(256), 0, st, ws, grid, V, dWm, dbm, dA1, dA2);
  return check_launch("hopmi_gcn_bwd_reduce");
}

extern "C" int hopmi_gcn_bwd(const float* x, const float* dh, const float* prep, const float* Wm,
                             float* dx, float* dA1, float* dA2, float* dWm, float* dbm, float* ws,
                             int n_slabs, int V, void* stream) {
  return gcn_bwd_impl<float>(x, dh, prep, Wm, dx, dA1, dA2, dWm, dbm, ws, n_slabs, V, stream);
}

extern "C" int hopmi_gcn_bwd_dt(const void* x, const void* dh, const float* prep, const float* Wm, void* dx, float* dA1, float* dA2,
                                float* dWm, float* dbm, float* ws, int n_slabs, int V, int dtype, void* stream) {
  if (dtype == HOPMI_F32)
    return gcn_bwd_impl<float>(static_cast<const float*>(x), static_cast<const float*>(dh), prep, Wm, static_cast<float*>(dx), dA1, dA2,
                               dWm, dbm, ws, n_slabs, V, stream);
  if (dtype == HOPMI_BF16)
    return gcn_bwd_impl<__bf16>(static_cast<const __bf16*>(x), static_cast<const __bf16*>(dh), prep, Wm, static_cast<__bf16*>(dx), dA1,
                                dA2, dWm, dbm, ws, n_slabs, V, stream);
  set_error("hopmi_gcn_bwd_dt: dtype %d (0 = fp32, 1 = bf16)", dtype);
  return HOPMI_EINVAL;
}
