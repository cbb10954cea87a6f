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
#include "common.h"

namespace hopmi {

struct GcnGeom {
  int V;         // graph nodes
  int S;         // slabs per tile
  int mtiles;    // 16-row MFMA tiles per block tile = ceil(S*V/16)
  int rows_lds;  // 16*mtiles + 4 (K padding of the last slab reads up to 3 rows past it)
  int MP;        // ceil16(2V): rows of the stacked mix matrix [A1^T; A2^T]
  int KP;        // ceil4(V):   its K
  int ldA;       // LDS row stride of AT[KP][ldA]; == 16 (mod 32) => conflict-free operand reads
  int VP;        // ceil16(V)
  int K2P;       // ceil4(2V)
  int ldB;       // LDS row stride of AB[K2P][ldB]
  int ntiles;
};

static int stride16mod32(int n) { return (n % 32 == 16) ? n : n + 16; }

static GcnGeom make_geom(int n_slabs, int V, int S) {
  GcnGeom g;
  g.V = V;
  g.S = S;
  g.mtiles = (S * V + 15) / 16;
  g.rows_lds = 16 * g.mtiles + 4;
  g.MP = ceil_to(2 * V, 16);
  g.KP = ceil_to(V, 4);
  g.ldA = stride16mod32(g.MP);
  g.VP = ceil_to(V, 16);
  g.K2P = ceil_to(2 * V, 4);
  g.ldB = stride16mod32(g.VP);
  g.ntiles = (n_slabs + S - 1) / S;
  return g;
}

// ------------------------------------------------------------------------------------------
// shared device pieces
// ------------------------------------------------------------------------------------------

// AT[k = v][m]: m < V -> A1[v][m] (row m of A1^T), V <= m < 2V -> A2[v][m-V]; zero padded.
__device__ __forceinline__ void load_mix_matrix(float* AT, const float* __restrict__ A1, const float* __restrict__ A2,
                                                const GcnGeom& g, int tid) {
  const int V = g.V;
  for (int idx = tid; idx < g.KP * g.ldA; idx += 256) {
    const int k = idx / g.ldA, m = idx - k * g.ldA;
    float v = 0.f;
    if (k < V && m < 2 * V) v = (m < V) ? A1[k * V + m] : A2[k * V + (m - V)];
    AT[idx] = v;
  }
}

// Stream `R` rows x 64 floats from `src` into dst[row*ld + c]; rows in [R, rows_total) zeroed.
__device__ __forceinline__ void load_rows(float* dst, int ld, const float* __restrict__ src, int R, int rows_total, int tid) {
  const float4* src4 = reinterpret_cast<const float4*>(src);
  for (int idx = tid; idx < rows_total * (C / 4); idx += 256) {
    const int row = idx >> 4, c4 = idx & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < R) v = src4[(size_t)row * (C / 4) + c4];
    float* d = dst + row * ld + 4 * c4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// Node mix of the `nsl` slabs of a tile, wave `w` doing channels [16w, 16w+16):
// Hc[s*V + node][64*(1+blk) + c] = sum_v A{blk+1}[v][node] * Hc[s*V + v][c].
__device__ __forceinline__ void node_mix(float* Hc, const float* AT, const GcnGeom& g, int nsl, int w, int q, int j) {
  const int V = g.V;
  const int ksteps = g.KP >> 2, mt_n = g.MP >> 4;
  for (int s = 0; s < nsl; ++s) {
    const float* xs = Hc + (s * V + q) * LDH + 16 * w + j;       // B[k = 4ks+q][n = c]
    for (int mt = 0; mt < mt_n; ++mt) {
      const float* at = AT + q * g.ldA + 16 * mt + j;            // A[i = m][k = 4ks+q]
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < ksteps; ++ks) acc = mfma16(at[4 * ks * g.ldA], xs[4 * ks * LDH], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * mt + 4 * q + r;
        if (m < 2 * V) {
          const int blk = (m >= V) ? 1 : 0;
          Hc[(s * V + m - blk * V) * LDH + C * (1 + blk) + 16 * w + j] = acc[r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void gcn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ A1,
                                                      const float* __restrict__ A2, const float* __restrict__ Wm,
                                                      const float* __restrict__ bm, float* __restrict__ h,
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

  // this wave's 192 x 16 slice of Wm^T: wreg[ks] = Wm[o = 16w + j][k = 48q + ks]
  float wreg[48];
  {
    const float4* wp = reinterpret_cast<const float4*>(Wm + (size_t)(16 * w + j) * K3 + 48 * q);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const float4 v = wp[i];
      wreg[4 * i] = v.x; wreg[4 * i + 1] = v.y; wreg[4 * i + 2] = v.z; wreg[4 * i + 3] = v.w;
    }
  }
  const float bias = bm[16 * w + j];

  load_mix_matrix(AT, A1, A2, g, tid);
  load_rows(Hc, LDH, x + row0 * C, R, g.rows_lds, tid);
  __syncthreads();
  node_mix(Hc, AT, g, nsl, w, q, j);
  __syncthreads();

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = {0.f, 0.f, 0.f, 0.f};
  const float* ha = Hc + j * LDH + 48 * q;        // A[i = row][k = 48q + ks]
#pragma unroll
  for (int ks = 0; ks < 48; ++ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (mt < g.mtiles) acc[mt] = mfma16(ha[16 * mt * LDH + ks], wreg[ks], acc[mt]);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * mt + 4 * q + r;
      if (mt < g.mtiles && row < R) h[(row0 + row) * C + 16 * w + j] = acc[mt][r] + bias;
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
constexpr int BWD_MT = 4;          // <= 64 rows per tile
constexpr int DA_SLOTS = 5;        // ceil(3*6/4): dA accumulator tiles per wave at V <= 48

// Persistent over tiles: dWm / dbm / dA partial sums stay in registers across the block's
// tiles and are written once to part[blockIdx.x][...] (summed by gcn_bwd_reduce_kernel in a
// fixed order: reproducible, no atomics).
__global__ __launch_bounds__(256) void gcn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dh,
                                                      const float* __restrict__ A1, const float* __restrict__ A2,
                                                      const float* __restrict__ Wm, float* __restrict__ dx,
                                                      float* __restrict__ part, int n_slabs, GcnGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hc = smem;                                // [rows_lds][LDH]  X | X A1 | X A2
  float* Gs = Hc + g.rows_lds * LDH;               // [rows_lds][LDH]  G0 | G1 | G2 = dH Wm
  float* DH = Gs + g.rows_lds * LDH;               // [rows_lds][LDD]
  float* AT = DH + g.rows_lds * LDD;               // [KP][ldA]   forward mix matrix
  float* AB = AT + g.KP * g.ldA;                   // [K2P][ldB]  AB[k][v] = (k<V ? A1[v][k] : A2[v][k-V])
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int V = g.V;

  load_mix_matrix(AT, A1, A2, g, tid);
  for (int idx = tid; idx < g.K2P * g.ldB; idx += 256) {
    const int k = idx / g.ldB, v = idx - k * g.ldB;
    float a = 0.f;
    if (v < V && k < 2 * V) a = (k < V) ? A1[v * V + k] : A2[v * V + (k - V)];
    AB[idx] = a;
  }
  // Wm slice for G = dH Wm: wreg[b][ks] = Wm[o = 16q + ks][64b + 16w + j]
  float wreg[3][16];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) wreg[b][ks] = Wm[(size_t)(16 * q + ks) * K3 + C * b + 16 * w + j];

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
    const int mtiles = (R + 15) >> 4;

    __syncthreads();                               // previous tile fully consumed
    load_rows(Hc, LDH, x + row0 * C, R, g.rows_lds, tid);
    load_rows(DH, LDD, dh + row0 * C, R, g.rows_lds, tid);
    // rows >= R of the mixed columns feed the dWm contraction (times dH = 0): keep them finite
    for (int idx = tid; idx < (g.rows_lds - R) * 2 * C; idx += 256) {
      const int row = R + idx / (2 * C), c = idx % (2 * C);
      Hc[row * LDH + C + c] = 0.f;
    }
    for (int idx = tid; idx < 4 * K3; idx += 256) Gs[(16 * mtiles + idx / K3) * LDH + idx % K3] = 0.f;
    __syncthreads();

    // (1) recompute the forward node mix -> Hcat
    node_mix(Hc, AT, g, nsl, w, q, j);

    // (2) G = dH (rows x 64) Wm (64 x 192); wave w owns columns 64b + 16w + [0,16)
    {
      f32x4 acc[BWD_MT][3];
#pragma unroll
      for (int mt = 0; mt < BWD_MT; ++mt)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[mt][b] = {0.f, 0.f, 0.f, 0.f};
      const float* da = DH + j * LDD + 16 * q;     // A[i = row][k = o = 16q + ks]
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int mt = 0; mt < BWD_MT; ++mt) {
          if (mt < mtiles) {
            const float a = da[16 * mt * LDD + ks];
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[mt][b] = mfma16(a, wreg[b][ks], acc[mt][b]);
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < BWD_MT; ++mt)
        if (mt < mtiles) {
#pragma unroll
          for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) Gs[(16 * mt + 4 * q + r) * LDH + C * b + 16 * w + j] = acc[mt][b][r];
        }
    }
    __syncthreads();

    // (3) dX[s] = G0 + A1 G1 + A2 G2 ; wave w owns channels 16w + [0,16)
    {
      const int ksteps = g.K2P >> 2, mt_n = g.VP >> 4;
      for (int s = 0; s < nsl; ++s) {
        for (int mt = 0; mt < mt_n; ++mt) {
          f32x4 acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int v = 16 * mt + 4 * q + r;
            acc[r] = (v < V) ? Gs[(s * V + v) * LDH + 16 * w + j] : 0.f;
          }
          for (int ks = 0; ks < ksteps; ++ks) {
            const int k = 4 * ks + q;
            const int blk = (k >= V) ? 1 : 0;
            const int wn = (k < 2 * V) ? (k - blk * V) : 0;     // padded k: AB is zero there
            acc = mfma16(AB[k * g.ldB + 16 * mt + j], Gs[(s * V + wn) * LDH + C * (1 + blk) + 16 * w + j], acc);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int v = 16 * mt + 4 * q + r;
            if (v < V) dx[(row0 + s * V + v) * C + 16 * w + j] = acc[r];
          }
        }
      }
    }

    // (4) dA{1,2}[v][w'] += sum_{s,c} X[s,v,c] G{1,2}[s,w',c]; output tiles dealt round-robin to waves
#pragma unroll
    for (int sl = 0; sl < DA_SLOTS; ++sl) {
      const int t = w + 4 * sl;
      if (t < ntiles_dA) {
        const int mtA = t / nt_dA, ntA = t - mtA * nt_dA;
        const int v = min(16 * mtA + j, V - 1);                  // rows >= V are discarded
        int m = 16 * ntA + j;                                    // column of [dA1 | dA2]
        m = min(m, 2 * V - 1);                                   // columns >= 2V are discarded
        const int blk = (m >= V) ? 1 : 0;
        const float* xa = Hc + v * LDH + q;                      // A[i = v][k = c = 4ks + q]
        const float* gb = Gs + (m - blk * V) * LDH + C * (1 + blk) + q;   // B[k = c][n = m]
        f32x4 acc = acc_dA[sl];
        for (int s = 0; s < nsl; ++s) {
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) acc = mfma16(xa[s * V * LDH + 4 * ks], gb[s * V * LDH + 4 * ks], acc);
        }
        acc_dA[sl] = acc;
      }
    }

    // (5) dWm[o][kk] += sum_rows dH[row][o] Hcat[row][kk]; wave w owns kk tiles 3w..3w+2
    {
      const float* da = DH + q * LDD + j;          // A[i = o][k = row = 4ks + q]
      const float* hb = Hc + q * LDH + 48 * w + j; // B[k = row][n = kk]
      for (int ks = 0; ks < 4 * mtiles; ++ks) {
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
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[(size_t)b * psz + i];
  if (i < C * K3) dWm[i] = s;
  else if (i < C * K3 + C) dbm[i - C * K3] = s;
  else if (i < C * K3 + C + V * V) dA1[i - C * K3 - C] = s;
  else dA2[i - C * K3 - C - V * V] = s;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

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

static int bwd_slabs(int V) { return (16 * BWD_MT) / V > 0 ? (16 * BWD_MT) / V : 1; }

static int bwd_grid(int ntiles) { return ntiles < env_int("HOPMI_GCN_BWD_GRID", 256) ? ntiles : env_int("HOPMI_GCN_BWD_GRID", 256); }

static int validate(const void* const* ptrs, int nptr, int n_slabs, int V) {
  for (int i = 0; i < nptr; ++i)
    if (!ptrs[i]) { set_error("hopmi_gcn: null pointer argument #%d", i); return HOPMI_EINVAL; }
  if (n_slabs <= 0) { set_error("hopmi_gcn: n_slabs must be > 0 (got %d)", n_slabs); return HOPMI_EINVAL; }
  if (V < 1 || V > HOPMI_MAX_NODES) { set_error("hopmi_gcn: V=%d outside [1,%d]", V, HOPMI_MAX_NODES); return HOPMI_EINVAL; }
  return HOPMI_OK;
}

template <int MT>
static int launch_fwd(const float* x, const float* A1, const float* A2, const float* Wm, const float* bm, float* h,
                      int n_slabs, const GcnGeom& g, hipStream_t st) {
  const size_t lds = (size_t)(g.rows_lds * LDH + g.KP * g.ldA) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gcn_fwd_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(gcn_fwd_kernel<MT>, dim3(g.ntiles), dim3(256), lds, st, x, A1, A2, Wm, bm, h, n_slabs, g);
  return check_launch("hopmi_gcn_fwd");
}

}  // namespace hopmi

using namespace hopmi;

extern "C" int hopmi_gcn_fwd(const float* x, const float* A1, const float* A2, const float* Wm, const float* bm,
                             float* h, int n_slabs, int V, void* stream) {
  const void* ptrs[] = {x, A1, A2, Wm, bm, h};
  if (int e = validate(ptrs, 6, n_slabs, V)) return e;
  const GcnGeom g = make_geom(n_slabs, V, pick_fwd_slabs(n_slabs, V));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (g.mtiles <= 2) return launch_fwd<2>(x, A1, A2, Wm, bm, h, n_slabs, g, st);
  if (g.mtiles <= 4) return launch_fwd<4>(x, A1, A2, Wm, bm, h, n_slabs, g, st);
  return launch_fwd<8>(x, A1, A2, Wm, bm, h, n_slabs, g, st);
}

extern "C" size_t hopmi_gcn_bwd_ws_floats(int n_slabs, int V) {
  if (n_slabs <= 0 || V < 1 || V > HOPMI_MAX_NODES) return 0;
  const GcnGeom g = make_geom(n_slabs, V, bwd_slabs(V));
  return (size_t)bwd_grid(g.ntiles) * (C * K3 + C + 2 * V * V);
}

extern "C" int hopmi_gcn_bwd(const float* x, const float* dh, const float* A1, const float* A2, const float* Wm,
                             float* dx, float* dA1, float* dA2, float* dWm, float* dbm, float* ws,
                             int n_slabs, int V, void* stream) {
  const void* ptrs[] = {x, dh, A1, A2, Wm, dx, dA1, dA2, dWm, dbm, ws};
  if (int e = validate(ptrs, 11, n_slabs, V)) return e;
  const GcnGeom g = make_geom(n_slabs, V, bwd_slabs(V));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = bwd_grid(g.ntiles);
  const size_t lds = (size_t)(2 * g.rows_lds * LDH + g.rows_lds * LDD + g.KP * g.ldA + g.K2P * g.ldB) * sizeof(float);
  if (lds > 160 * 1024) { set_error("hopmi_gcn_bwd: LDS footprint %zu exceeds 160 KiB", lds); return HOPMI_EINVAL; }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gcn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(gcn_bwd_kernel, dim3(grid), dim3(256), lds, st, x, dh, A1, A2, Wm, dx, ws, n_slabs, g);
  if (int e = check_launch("hopmi_gcn_bwd")) return e;
  const int psz = C * K3 + C + 2 * V * V;
  hipLaunchKernelGGL(gcn_bwd_reduce_kernel, dim3((psz + 255) / 256), dim3(256), 0, st, ws, grid, V, dWm, dbm, dA1, dA2);
  return check_launch("hopmi_gcn_bwd_reduce");
}
