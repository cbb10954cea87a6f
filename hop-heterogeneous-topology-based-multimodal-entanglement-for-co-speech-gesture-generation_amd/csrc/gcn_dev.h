// Device-side building blocks shared by the graph-conv kernels (gcn.hip) and the fused WaveNet-layer
// kernels (wavenet.hip): tile geometry, prepared mix-matrix images, row streaming, node mix, dX mix.
#pragma once
#include "common.h"
#include "io_dev.h"

namespace hopmi {

// Diagnostic build only (-DHOPMI_STAMPS, tools/probes/gcn_stamps.py): per-phase s_memtime stamps of
// wave 0 of every block go to a side buffer that nothing else reads.  The product build has none.
#ifdef HOPMI_STAMPS
static __device__ long long* g_stamps = nullptr;
#define HOPMI_STAMP(slot)                                                                     \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    unsigned long long t_;                                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (g_stamps && threadIdx.x == 0) g_stamps[blockIdx.x * 8 + (slot)] = (long long)t_;      \
  } while (0)
#else
#define HOPMI_STAMP(slot) do { } while (0)
#endif

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

static inline int stride16mod32(int n) { return (n % 32 == 16) ? n : n + 16; }

static inline GcnGeom make_geom(int n_slabs, int V, int S) {
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

// Phase 0 is split into "issue every global load" and "write LDS" so that a block pays ONE memory
// round trip.  Every load is UNCONDITIONAL (clamped address, value selected afterwards): a load under
// a runtime guard makes hipcc branch around it and wait vmcnt(0) per element.

// The node-mix matrices are tiny (V x V) and the same for every workgroup and every WaveNet layer of
// a forward pass, so their zero-padded LDS images are built ONCE per pass by hopmi_gcn_prepare:
//   prep = [ AT[KP][ldA] | AB[K2P][ldB] ]
//   AT[k = v][m]: m < V -> A1[v][m] (row m of A1^T), V <= m < 2V -> A2[v][m-V]       (forward mix)
//   AB[k][v]   : k < V -> A1[v][k],                 V <= k < 2V -> A2[v][k-V]       (dX mix, backward)
// and a workgroup copies an image with at most PREP_IT coalesced 16-B loads per thread.
constexpr int PREP_IT = 6;                          // V <= 48: 48*112/4 = 1344 float4 <= 6*256
struct PrepRegs { float4 v[PREP_IT]; };

__device__ __forceinline__ void prep_issue(PrepRegs& r, const float* __restrict__ img, int nfloats, int tid) {
  const float4* src = reinterpret_cast<const float4*>(img);
  const int n4 = nfloats >> 2;
#pragma unroll
  for (int it = 0; it < PREP_IT; ++it) {
    const int idx = tid + 256 * it;
    // unconditional (clamped) load + select: a conditionally written register array ends up in scratch
    const float4 v = src[min(idx, n4 - 1)];
    r.v[it] = (idx < n4) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

__device__ __forceinline__ void prep_commit(float* dst, const PrepRegs& r, int nfloats, int tid) {
  const int n4 = nfloats >> 2;
#pragma unroll
  for (int it = 0; it < PREP_IT; ++it) {
    const int idx = tid + 256 * it;
    if (idx < n4) reinterpret_cast<float4*>(dst)[idx] = r.v[it];
  }
}

// `R` (>= 1) rows x 64 floats of `src` -> registers (NIT float4 per thread), rows >= R read as zero.
template <int NIT>
struct RowRegs { float4 v[NIT]; };

template <int NIT, typename TS>
__device__ __forceinline__ void rows_issue(RowRegs<NIT>& r, const TS* __restrict__ src, int R, int tid) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = tid + 256 * it;
    const int row = idx >> 4, c4 = idx & 15;
    const float4 v = ld4(src + 4 * (min(row, R - 1) * 16 + c4));
    r.v[it] = (row < R) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int NIT>
__device__ __forceinline__ void rows_commit(float* dst, int ld, const RowRegs<NIT>& r, int rows_total, int tid) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = tid + 256 * it;
    const int row = idx >> 4, c4 = idx & 15;
    if (row < rows_total) *reinterpret_cast<float4*>(dst + row * ld + 4 * c4) = r.v[it];
  }
}

constexpr int rows_nit(int mt) { return ((16 * mt + 4) * 16 + 255) / 256; }

// Node mix of the `nsl` slabs of a tile, wave `w` doing channels [16w, 16w+16):
// Hc[s*V + node][64*(1+blk) + c] = sum_v A{blk+1}[v][node] * Hc[s*V + v][c].
// KS = K steps ceil(V/4), MTN = 16-wide tiles ceil(2V/16) of the stacked mix matrix: the matrix stays
// in registers for all slabs and the results are stored unconditionally (stacked nodes m >= 2V of the
// padding go to a dump row in the tile's padding: columns >= 64 of row `dump_row` are never read).
template <int KS, int MTN, int SG>
__device__ __forceinline__ void node_mix_group(float* hs, float* Hc, const float (&am)[MTN][KS], const int (&woff)[MTN],
                                               const bool (&in_slab)[MTN], int V, int w, int q, int j) {
  // SG slabs at once: SG*MTN independent accumulator chains keep the matrix pipe fed while the LDS reads of
  // the group and the previous group's result writes are in flight.  The product is taken transposed,
  // D[i = channel][j = stacked node m] = sum_v X[v][channel] * A[v][m], so that a lane ends up with 4 consecutive
  // channels of one node: one 16-byte LDS write per tile instead of four 4-byte ones.
  float xb[SG][KS];
#pragma unroll
  for (int sg = 0; sg < SG; ++sg)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[sg][ks] = hs[(sg * V + 4 * ks + q) * LDH + 16 * w + j];   // A[i = c][k = v]
  f32x4 acc[SG][MTN];
#pragma unroll
  for (int sg = 0; sg < SG; ++sg)
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) acc[sg][mt] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int sg = 0; sg < SG; ++sg)
#pragma unroll
      for (int mt = 0; mt < MTN; ++mt) acc[sg][mt] = mfma16(xb[sg][ks], am[mt][ks], acc[sg][mt]);
#pragma unroll
  for (int sg = 0; sg < SG; ++sg)
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) {
      // stacked nodes m >= 2V of the padded N go to the dump row (absolute offset), the rest into slab sg
      float* dst = (in_slab[mt] ? hs + sg * V * LDH : Hc) + woff[mt];
      *reinterpret_cast<f32x4*>(dst) = acc[sg][mt];
    }
}

template <int KS, int MTN, int SG>
__device__ __forceinline__ void node_mix(float* Hc, const float* AT, const GcnGeom& g, int nsl, int dump_row,
                                         int w, int q, int j) {
  const int V = g.V;
  float am[MTN][KS];
  int woff[MTN];
  bool in_slab[MTN];
#pragma unroll
  for (int mt = 0; mt < MTN; ++mt) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) am[mt][ks] = AT[(4 * ks + q) * g.ldA + 16 * mt + j];   // B[k = v][n = m]
    const int m = 16 * mt + j;
    const int blk = (m >= V) ? 1 : 0;
    in_slab[mt] = m < 2 * V;
    woff[mt] = in_slab[mt] ? ((m - blk * V) * LDH + C * (1 + blk) + 16 * w + 4 * q) : (dump_row * LDH + C + 16 * w + 4 * q);
  }
  int s = 0;
  for (; s + SG <= nsl; s += SG) node_mix_group<KS, MTN, SG>(Hc + s * V * LDH, Hc, am, woff, in_slab, V, w, q, j);
  for (; s < nsl; ++s) node_mix_group<KS, MTN, 1>(Hc + s * V * LDH, Hc, am, woff, in_slab, V, w, q, j);
}

// generic V (runtime loops)
__device__ __forceinline__ void node_mix_generic(float* Hc, const float* AT, const GcnGeom& g, int nsl, int w, int q, int j) {
  const int V = g.V;
  const int ksteps = g.KP >> 2, mt_n = g.MP >> 4;
  for (int s = 0; s < nsl; ++s) {
    const float* xs = Hc + (s * V + q) * LDH + 16 * w + j;
    for (int mt = 0; mt < mt_n; ++mt) {
      const float* at = AT + q * g.ldA + 16 * mt + j;
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

__device__ __forceinline__ void node_mix_dispatch(float* Hc, const float* AT, const GcnGeom& g, int nsl, int w, int q, int j) {
  const int dump_row = g.rows_lds - 1;
  if (g.V == 9) node_mix<3, 2, 4>(Hc, AT, g, nsl, dump_row, w, q, j);         // TED
  else if (g.V == 42) node_mix<11, 6, 1>(Hc, AT, g, nsl, dump_row, w, q, j);  // TED-Expressive
  else node_mix_generic(Hc, AT, g, nsl, w, q, j);
}

// dX[s] = G0 + A1 G1 + A2 G2 for the slabs of a tile; wave w owns channels 16w + [0,16).
// KS = K steps (ceil(2V/4)) when known at compile time (0 = runtime loop).
template <int KS, typename TS>
__device__ __forceinline__ void dx_mix(const float* Gs, const float* AB, TS* __restrict__ dx, size_t row0,
                                       const GcnGeom& g, int nsl, int w, int q, int j) {
  const int V = g.V;
  const int ksteps = KS ? KS : (g.K2P >> 2), mt_n = g.VP >> 4;
  // lane-constant part of the B-operand address for every k step: row (node) and column block
  for (int s = 0; s < nsl; ++s) {
    float gb[KS ? KS : 1];
    if (KS) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + q;
        const int blk = (k >= V) ? 1 : 0;
        const int wn = (k < 2 * V) ? (k - blk * V) : 0;          // padded k: AB is zero there
        gb[ks] = Gs[(s * V + wn) * LDH + C * (1 + blk) + 16 * w + j];
      }
    }
    for (int mt = 0; mt < mt_n; ++mt) {
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int v = 16 * mt + 4 * q + r;
        acc[r] = (v < V) ? Gs[(s * V + v) * LDH + 16 * w + j] : 0.f;
      }
      const float* ab = AB + q * g.ldB + 16 * mt + j;            // A[i = v][k = 4ks + q]
      if (KS) {
        float aa[KS ? KS : 1];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) aa[ks] = ab[4 * ks * g.ldB];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma16(aa[ks], gb[ks], acc);
      } else {
        for (int ks = 0; ks < ksteps; ++ks) {
          const int k = 4 * ks + q;
          const int blk = (k >= V) ? 1 : 0;
          const int wn = (k < 2 * V) ? (k - blk * V) : 0;
          acc = mfma16(ab[4 * ks * g.ldB], Gs[(s * V + wn) * LDH + C * (1 + blk) + 16 * w + j], acc);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int v = 16 * mt + 4 * q + r;
        if (v < V) st1(dx + (row0 + s * V + v) * C + 16 * w + j, acc[r]);
      }
    }
  }
}

}  // namespace hopmi
