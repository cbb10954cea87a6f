// One fused WaveNet layer of the HOP graph-wavenet block, BACKWARD (autograd of model/gwnet.py:181-237).
//
// Forward of layer i (wavenet.hip):  r^ = BN_{i-1}(xin) ; u = tanh(a) sigmoid(g), [a|g] = TCN(r^[t'], r^[t'+d]) ;
//                                    y = Wm.[u;uA1;uA2] + bm + r^[t'+d] ; x^_i = BN_i(y)
// Given the gradient w.r.t. x^_i (as the two tap contributions P0n/P1n written by layer i+1's backward) this
// kernel computes, per tile of <= 48 output rows, entirely on chip:
//   dy   = BN_i backward applied ON LOAD:  dy = ca*dx^ + cb*y + ck   (coefficients from the previous launch's
//          reduction: dgamma, dbeta of BN_i folded in)
//   G    = dy Wm ;  dU = G0 + A1 G1 + A2 G2 (+ the skip-tail gradient dutail) ;  dA, dWm, dbm partial sums
//   da   = dU (1 - f^2) s ;  dg = dU f s (1 - s)              (f, s saved by the forward)
//   P0   = [da|dg] [Wf0;Wg0]            gradient w.r.t. x^_{i-1} at frame t'
//   P1   = [da|dg] [Wf1;Wg1] + dy       ... at frame t'+d (TCN tap 1 + the residual path)
//   dW_tcn, db_tcn partial sums (against the RAW xin; the BN_{i-1} scale/shift is folded in by the reduce
//          kernel: sum dAG r^ = sc * sum dAG xin + sh * sum dAG)
//   S1 = sum P, S2 = sum P * xin   the two reductions BN_{i-1}'s backward needs
// P0/P1 are separate tensors so that no two workgroups ever add into the same element (layer i-1 sums
// them on load); all parameter-gradient partials stay in accumulator registers across a workgroup's tiles and
// are summed by wn_bwd_reduce_kernel in a fixed order: bitwise reproducible, no atomics.
#include "io_dev.h"
#include "wn_dev.h"

namespace hopmi {

constexpr int WNB_MAX_MT = 3;                      // <= 48 rows per tile (LDS: 5 tile images)
constexpr int LDG = 2 * C + 4;                     // [rows][128] da|dg image, 16-B aligned rows, 132 = 4 (mod 64)
constexpr int WNB_DA_SLOTS = 5;

// per-workgroup partial layout (floats)
constexpr int PO_DWT = 0;                          // [4 = 2*tap+gate][64 o][64 c]  (against raw xin)
constexpr int PO_DBT = PO_DWT + 4 * C * C;         // [128]  column sums of [da|dg]
constexpr int PO_DWM = PO_DBT + 2 * C;             // [64][192]
constexpr int PO_DBM = PO_DWM + C * K3;            // [64]
constexpr int PO_ST = PO_DBM + C;                  // [128]  S1, S2
constexpr int PO_DA = PO_ST + 2 * C;               // [2][V][V]
__host__ __device__ constexpr int part_floats(int V) { return PO_DA + 2 * V * V; }

template <int NIT>
struct BwdRowMap {
  int in0[NIT];    // float4 index of the raw xin tap-0 row (+ c4)
  int p0[NIT];     // float4 index into P0n or -1
  int p1[NIT];     // float4 index into P1n or -1
  int tail[NIT];   // float4 index into dutail or -1
  bool ok[NIT];
};

// TS = storage type of the saved activations xin, y and of the incoming skip-tail gradient dutail (float or __bf16, io_dev.h);
// the gradients between layers (P0n / P1n in, P0 / P1 out) and all arithmetic stay fp32.
template <int MT, typename TS>
__global__ __launch_bounds__(256) void wn_layer_bwd_kernel(
    const TS* __restrict__ xin, const float* __restrict__ fs, const float* __restrict__ wf, const float* __restrict__ wg,
    const float* __restrict__ prep, const float* __restrict__ Wm, const float* __restrict__ P0n,
    const float* __restrict__ P1n, const TS* __restrict__ y, const float* __restrict__ bn_coef,
    const TS* __restrict__ dutail, float* __restrict__ P0, float* __restrict__ P1, float* __restrict__ part,
    LayerGeom L, int do_gcn, int d_next, int T_next, int dutail_ld4) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GcnGeom& g = L.g;
  constexpr int NIT = rows_nit(MT);
  float* R0 = smem;                                // [rows_lds][LDD]  RAW xin at frame t'
  float* R1 = R0 + g.rows_lds * LDD;               // [rows_lds][LDD]  RAW xin at frame t'+d
  float* DY = R1 + g.rows_lds * LDD;               // [rows_lds][LDD]  dy (zero when !do_gcn)
  float* Hc = DY + g.rows_lds * LDD;               // [rows_lds][LDH]  u | uA1 | uA2 ; later aliased by DG
  float* Gs = Hc + g.rows_lds * LDH;               // [rows_lds][LDH]  G0 (+dutail) | G1 | G2
  float* AT = Gs + g.rows_lds * LDH;               // [KP][ldA]
  float* AB = AT + g.KP * g.ldA;                   // [K2P][ldB]
  float* DG = Hc;                                  // [rows_lds][LDG]  da | dg
  int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform: per-wave bases stay in scalar registers
  int lane = tid & 63, q = lane >> 4, j = lane & 15, c4 = tid & 15;
  const int V = g.V;
  const int shift4 = L.d * V * 16;

  if (do_gcn) {
    PrepRegs mr, br;
    prep_issue(mr, prep, g.KP * g.ldA, tid);
    prep_issue(br, prep + g.KP * g.ldA, g.K2P * g.ldB, tid);
    prep_commit(AT, mr, g.KP * g.ldA, tid);
    prep_commit(AB, br, g.K2P * g.ldB, tid);
  }
  float4 ca4 = make_float4(0.f, 0.f, 0.f, 0.f), cb4 = ca4, ck4 = ca4;
  if (do_gcn) {
    ca4 = reinterpret_cast<const float4*>(bn_coef)[c4];
    cb4 = reinterpret_cast<const float4*>(bn_coef + C)[c4];
    ck4 = reinterpret_cast<const float4*>(bn_coef + 2 * C)[c4];
  }
  // the 4 padding rows behind the tile feed the node mix's K padding (times zero): keep them finite
  for (int idx = tid; idx < 4 * C; idx += 256) Hc[(16 * MT + idx / C) * LDH + idx % C] = 0.f;

  // ---- persistent partial sums ------------------------------------------------------------------
  f32x4 acc_dWt[2][8];                             // [tap][o tile]: columns c = 16w + j
#pragma unroll
  for (int tap = 0; tap < 2; ++tap)
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) acc_dWt[tap][mt] = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc_dWm[4][3];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int n = 0; n < 3; ++n) acc_dWm[mt][n] = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc_dA[WNB_DA_SLOTS];
#pragma unroll
  for (int sl = 0; sl < WNB_DA_SLOTS; ++sl) acc_dA[sl] = {0.f, 0.f, 0.f, 0.f};
  float acc_dbt = 0.f, acc_dbm = 0.f, st1 = 0.f, st2 = 0.f;
  const int nt_dA = g.MP >> 4;
  const int ntiles_dA = (g.VP >> 4) * nt_dA;

  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    // every address below derives from the lane index: hiding it from the optimiser per iteration keeps the dozens of
    // loop-invariant address registers from being hoisted out of the loop and held across all phases (they spilled)
    asm volatile("" : "+v"(tid));
    lane = tid & 63; q = lane >> 4; j = lane & 15; c4 = tid & 15;
    const int slab0 = tile * g.S;
    const int nsl = min(g.S, L.n_slabs - slab0);
    const int R = nsl * V;
    const size_t orow0 = (size_t)slab0 * V;

    // ---- phase A: stream xin (both taps, raw), dx^ (two tap contributions), y, f/s, dutail ---------
    BwdRowMap<NIT> rm;
    RowRegs<NIT> x0r, x1r, dyr, ur, tr;
    {
      const float4* f4 = reinterpret_cast<const float4*>(fs);
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
        rm.p0[it] = (do_gcn && tp < T_next) ? ((b * T_next + tp) * V + v) * 16 + c4 : -1;
        rm.p1[it] = (do_gcn && tp >= d_next) ? ((b * T_next + tp - d_next) * V + v) * 16 + c4 : -1;
        rm.tail[it] = (tp >= L.T_out - 4) ? ((b * 4 + tp - (L.T_out - 4)) * V + v) * dutail_ld4 + c4 : -1;
        x0r.v[it] = ld4(xin + 4 * (size_t)rm.in0[it]);
        x1r.v[it] = ld4(xin + 4 * (size_t)(rm.in0[it] + shift4));
        const size_t orow = orow0 + rc;
        const float4 fv = f4[orow * 32 + c4], sv = f4[orow * 32 + 16 + c4];
        ur.v[it] = make_float4(fv.x * sv.x, fv.y * sv.y, fv.z * sv.z, fv.w * sv.w);
        float4 dx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (do_gcn) {
          const float4 a = reinterpret_cast<const float4*>(P0n)[max(rm.p0[it], 0)];
          const float4 bq = reinterpret_cast<const float4*>(P1n)[max(rm.p1[it], 0)];
          const float4 yv = ld4(y + 4 * (orow * 16 + c4));
          const float m0 = rm.p0[it] >= 0 ? 1.f : 0.f, m1 = rm.p1[it] >= 0 ? 1.f : 0.f;
          dx = make_float4(ca4.x * (a.x * m0 + bq.x * m1) + cb4.x * yv.x + ck4.x, ca4.y * (a.y * m0 + bq.y * m1) + cb4.y * yv.y + ck4.y,
                           ca4.z * (a.z * m0 + bq.z * m1) + cb4.z * yv.z + ck4.z, ca4.w * (a.w * m0 + bq.w * m1) + cb4.w * yv.w + ck4.w);
        }
        dyr.v[it] = dx;
        const float4 tv = ld4(dutail + 4 * (size_t)max(rm.tail[it], 0));
        tr.v[it] = (rm.tail[it] >= 0 && rm.ok[it]) ? tv : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __syncthreads();                             // previous tile's LDS fully consumed
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = (tid >> 4) + 16 * it;
        if (row < g.rows_lds) {
          // (value selects on the components: a select between an array element and a constant aggregate makes hipcc
          // keep the whole register arrays in scratch and index them through memory)
          const float m = rm.ok[it] ? 1.f : 0.f;
          const float4 a0 = x0r.v[it], a1 = x1r.v[it], a2 = dyr.v[it], a3 = ur.v[it];
          *reinterpret_cast<float4*>(R0 + row * LDD + 4 * c4) = make_float4(m * a0.x, m * a0.y, m * a0.z, m * a0.w);
          *reinterpret_cast<float4*>(R1 + row * LDD + 4 * c4) = make_float4(m * a1.x, m * a1.y, m * a1.z, m * a1.w);
          *reinterpret_cast<float4*>(DY + row * LDD + 4 * c4) = make_float4(m * a2.x, m * a2.y, m * a2.z, m * a2.w);
          *reinterpret_cast<float4*>(Hc + row * LDH + 4 * c4) = make_float4(m * a3.x, m * a3.y, m * a3.z, m * a3.w);
          if (!do_gcn) *reinterpret_cast<float4*>(Gs + row * LDH + 4 * c4) = tr.v[it];      // dU = dutail
        }
      }
      if (do_gcn) {
        // mixed columns of the rows behind the tile feed the dWm contraction (times dy = 0): keep finite
        for (int idx = tid; idx < (g.rows_lds - R) * 2 * C; idx += 256) Hc[(R + idx / (2 * C)) * LDH + C + idx % (2 * C)] = 0.f;
      }
    }
    __syncthreads();

    if (do_gcn) {
      // ---- phase B: recompute the node mix -> Hcat ------------------------------------------------
      node_mix_dispatch(Hc, AT, g, nsl, w, q, j);
      // ---- phase C: G = dy Wm (rows x 64 . 64 x 192); wave w owns columns 64b + 16w + [0,16) -------
      {
        float4 wreg[3][4];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float* wp = Wm + (size_t)(16 * i + 4 * q) * K3 + C * b + 16 * w + j;
            wreg[b][i] = make_float4(wp[0], wp[K3], wp[2 * K3], wp[3 * K3]);
          }
        f32x4 acc[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc[mt][b] = {0.f, 0.f, 0.f, 0.f};
        const float* da = DY + j * LDD + 4 * q;
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
      // ---- phase C2: the skip-tail gradient joins dU through G0 -----------------------------------
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = (tid >> 4) + 16 * it;
        if (rm.tail[it] >= 0 && rm.ok[it]) {
          float4* gp = reinterpret_cast<float4*>(Gs + row * LDH + 4 * c4);
          const float4 a = *gp, t4 = tr.v[it];
          *gp = make_float4(a.x + t4.x, a.y + t4.y, a.z + t4.z, a.w + t4.w);
        }
      }
      // ---- phase D: dA, dWm, dbm partial sums (read Hc, G1/G2 of Gs, DY) ---------------------------
#pragma unroll
      for (int sl = 0; sl < WNB_DA_SLOTS; ++sl) {
        const int t = w + 4 * sl;
        if (t < ntiles_dA) {
          const int mtA = t / nt_dA, ntA = t - mtA * nt_dA;
          const int v = min(16 * mtA + j, V - 1);
          const int m = min(16 * ntA + j, 2 * V - 1);
          const int blk = (m >= V) ? 1 : 0;
          const float* xa = Hc + v * LDH + 4 * q;
          const float* gb = Gs + (m - blk * V) * LDH + C * (1 + blk) + 4 * q;
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
      {
        const float* da = DY + q * LDD + j;          // A[i = o][k = row = 4ks + q]
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
            for (int n = 0; n < 3; ++n) acc_dWm[mt][n] = mfma16(a[mt], b[n], acc_dWm[mt][n]);
        }
      }
      if (tid < C) {
        float sacc = 0.f;
        for (int row = 0; row < R; ++row) sacc += DY[row * LDD + tid];
        acc_dbm += sacc;
      }
      __syncthreads();                             // C2 visible; Hc no longer needed -> DG may alias it
    }

    // ---- phase E: dU -> gate backward -> [da|dg] into DG (rows >= R zero) ------------------------------
    for (int idx = tid; idx < (16 * MT - R) * 2 * C; idx += 256) DG[(R + idx / (2 * C)) * LDG + idx % (2 * C)] = 0.f;
    {
      const int mt_n = g.VP >> 4, ksteps = g.K2P >> 2;
      for (int s = 0; s < nsl; ++s) {
        for (int mt = 0; mt < mt_n; ++mt) {
          f32x4 acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int v = 16 * mt + 4 * q + r;
            acc[r] = (v < V) ? Gs[(s * V + v) * LDH + 16 * w + j] : 0.f;           // G0 + dutail
          }
          if (do_gcn) {
            const float* ab = AB + q * g.ldB + 16 * mt + j;
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
            if (v < V) {
              const int row = s * V + v;
              const float* fp = fs + (orow0 + row) * (2 * C) + 16 * w + j;
              const float f = fp[0], sg = fp[C], dU = acc[r];
              DG[row * LDG + 16 * w + j] = dU * (1.f - f * f) * sg;
              DG[row * LDG + C + 16 * w + j] = dU * f * sg * (1.f - sg);
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- phase F: P0 = DG [Wf0;Wg0], P1 = DG [Wf1;Wg1] + dy; BN_{i-1} backward reductions ---------------
    {
      f32x4 a0[MT], a1[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { a0[mt] = {0.f, 0.f, 0.f, 0.f}; a1[mt] = {0.f, 0.f, 0.f, 0.f}; }
      const float* ga = DG + j * LDG + 4 * q;        // A[i = row][k = 16i + 4q + e], k in [0,128): da then dg
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // B[k][n = c]: W[gate = i/4][o = 16(i%4) + 4q + e][c = 16w + j], both taps of an element are adjacent in
        // the Conv2d layout [out][in][1][tap]
        const float* wp = ((i >> 2) ? wg : wf) + ((size_t)(16 * (i & 3) + 4 * q) * C + 16 * w + j) * 2;
        const float2 t0 = *reinterpret_cast<const float2*>(wp), t1 = *reinterpret_cast<const float2*>(wp + 2 * C);
        const float2 t2 = *reinterpret_cast<const float2*>(wp + 4 * C), t3 = *reinterpret_cast<const float2*>(wp + 6 * C);
        const float4 b0 = make_float4(t0.x, t1.x, t2.x, t3.x);                                              // tap 0
        const float4 b1 = make_float4(t0.y, t1.y, t2.y, t3.y);                                              // tap 1
        float4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(ga + 16 * mt * LDG + 16 * i);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          a0[mt] = mfma16(a[mt].x, b0.x, a0[mt]);
          a1[mt] = mfma16(a[mt].x, b1.x, a1[mt]);
          a0[mt] = mfma16(a[mt].y, b0.y, a0[mt]);
          a1[mt] = mfma16(a[mt].y, b1.y, a1[mt]);
          a0[mt] = mfma16(a[mt].z, b0.z, a0[mt]);
          a1[mt] = mfma16(a[mt].z, b1.z, a1[mt]);
          a0[mt] = mfma16(a[mt].w, b0.w, a0[mt]);
          a1[mt] = mfma16(a[mt].w, b1.w, a1[mt]);
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + 4 * q + r;
          if (row < R) {
            const float p0 = a0[mt][r];
            const float p1 = a1[mt][r] + DY[row * LDD + 16 * w + j];
            P0[(orow0 + row) * C + 16 * w + j] = p0;
            P1[(orow0 + row) * C + 16 * w + j] = p1;
            st1 += p0 + p1;
            st2 += p0 * R0[row * LDD + 16 * w + j] + p1 * R1[row * LDD + 16 * w + j];
          }
        }
    }
    // ---- phase G: dW_tcn[tap][o][c] += sum_rows DG[row][o] xin_tap[row][c] ; db_tcn ---------------------
    {
      const float* ga = DG + q * LDG + j;            // A[i = o][k = row = 4ks + q]
      const float* r0 = R0 + q * LDD + 16 * w + j;   // B[k = row][n = c]
      const float* r1 = R1 + q * LDD + 16 * w + j;
#pragma unroll 2
      for (int ks = 0; ks < 4 * MT; ++ks) {
        float a[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) a[mt] = ga[4 * ks * LDG + 16 * mt];
        const float b0 = r0[4 * ks * LDD], b1 = r1[4 * ks * LDD];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          acc_dWt[0][mt] = mfma16(a[mt], b0, acc_dWt[0][mt]);
          acc_dWt[1][mt] = mfma16(a[mt], b1, acc_dWt[1][mt]);
        }
      }
      if (tid < 2 * C) {
        float sacc = 0.f;
        for (int row = 0; row < R; ++row) sacc += DG[row * LDG + tid];
        acc_dbt += sacc;
      }
    }
  }

  // ---- partials ------------------------------------------------------------------------------------
  float* p = part + (size_t)blockIdx.x * part_floats(V);
#pragma unroll
  for (int tap = 0; tap < 2; ++tap)
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * mt + 4 * q + r;           // 0..127: gate = o / 64
        p[PO_DWT + ((2 * tap + (o >> 6)) * C + (o & 63)) * C + 16 * w + j] = acc_dWt[tap][mt][r];
      }
  if (tid < 2 * C) p[PO_DBT + tid] = acc_dbt;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[PO_DWM + (16 * mt + 4 * q + r) * K3 + 48 * w + 16 * n + j] = acc_dWm[mt][n][r];
  if (tid < C) p[PO_DBM + tid] = acc_dbm;
  st1 += __shfl_xor(st1, 16); st1 += __shfl_xor(st1, 32);
  st2 += __shfl_xor(st2, 16); st2 += __shfl_xor(st2, 32);
  if (q == 0) { p[PO_ST + 16 * w + j] = st1; p[PO_ST + C + 16 * w + j] = st2; }
#pragma unroll
  for (int sl = 0; sl < WNB_DA_SLOTS; ++sl) {
    const int t = w + 4 * sl;
    if (t < ntiles_dA) {
      const int mtA = t / nt_dA, ntA = t - mtA * nt_dA;
      const int m = 16 * ntA + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int v = 16 * mtA + 4 * q + r;
        if (v < V && m < 2 * V) {
          const int blk = (m >= V) ? 1 : 0;
          p[PO_DA + blk * V * V + v * V + (m - blk * V)] = acc_dA[sl][r];
        }
      }
    }
  }
}

// Fixed-order sum of the per-workgroup partials.  Also:
//   * folds the BatchNorm_{i-1} affine map into dW_tcn:  dW[tap][gate][o][c] = sc[c]*raw + sh[c]*db_tcn[gate][o]
//   * turns S1, S2 into BatchNorm_{i-1}'s dgamma, dbeta and the dy-on-load coefficients for layer i-1's backward
//     (dy = ca*dx^ + cb*y + ck; see the header comment), when that BatchNorm exists (bn_prev != nullptr).
__global__ __launch_bounds__(256) void wn_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int V,
                                                            const float* __restrict__ scsh_in,      // [128] of this layer's input
                                                            const float* __restrict__ gamma_prev,   // BN_{i-1} (nullable)
                                                            const float* __restrict__ mean_rstd_prev, double n_prev,
                                                            float* __restrict__ dwf, float* __restrict__ dwg,
                                                            float* __restrict__ dbtcn,
                                                            float* __restrict__ dWm, float* __restrict__ dbm,
                                                            float* __restrict__ dA1, float* __restrict__ dA2, int acc_dA,
                                                            float* __restrict__ dgamma_prev, float* __restrict__ dbeta_prev,
                                                            float* __restrict__ coef_prev) {
  // A workgroup owns 64 consecutive columns of the partial layout; its 4 thread groups add the workgroup partials
  // b = grp (mod 4) in increasing order (8 loads in flight each), the 4 group sums are then added in a fixed order:
  // bitwise reproducible.  All section boundaries of the layout that matter here are multiples of 64.
  __shared__ float red[2][4][64];
  __shared__ float dred[256];
  const int psz = part_floats(V);
  const int tid = threadIdx.x, lc = tid & 63, grp = tid >> 6;
  const int col0 = blockIdx.x * 64, col = col0 + lc;
  const bool is_dwt = col0 < PO_DBT;                 // dW_tcn block: 64 input channels c of one (k, o)
  const bool is_st = col0 == PO_ST;                  // S1 block: the same threads also need S2 (next 64 columns)
  float s = 0.f, s2 = 0.f;
  for (int b0 = grp; b0 < nblk; b0 += 32) {
    float v[8], v2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 4 * u;
      const bool ok = b < nblk && col < psz;
      v[u] = ok ? part[(size_t)b * psz + col] : 0.f;
      v2[u] = (ok && is_st) ? part[(size_t)b * psz + col + C] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += v[u]; s2 += v2[u]; }
  }
  red[0][grp][lc] = s;
  red[1][grp][lc] = s2;
  if (is_dwt) {                                      // sum of db_tcn[gate][o] over the workgroup partials, one per thread
    const int go = (col0 >> 6) & 63, k = col0 >> 12;
    const int idx = PO_DBT + (k & 1) * C + go;
    float x = 0.f;
    for (int b = tid; b < nblk; b += 256) x += part[(size_t)b * psz + idx];
    dred[tid] = x;
  }
  __syncthreads();
  if (tid >= 64 || col >= psz) return;
  const int i = col;
  s = ((red[0][0][lc] + red[0][1][lc]) + red[0][2][lc]) + red[0][3][lc];
  if (i < PO_DBT) {                                  // dW_tcn element (k = 2*tap + gate, o, c)
    float dbt = 0.f;
    for (int t = 0; t < 256; ++t) dbt += dred[t];
    const int c = i & 63, o = (i >> 6) & 63, k = i >> 12;              // k = 2*tap + gate
    ((k & 1) ? dwg : dwf)[(o * C + c) * 2 + (k >> 1)] = scsh_in[c] * s + scsh_in[C + c] * dbt;    // Conv2d layout
  } else if (i < PO_DWM) {
    dbtcn[i - PO_DBT] = s;
  } else if (i < PO_DBM) {
    if (dWm != nullptr) dWm[i - PO_DWM] = s;
  } else if (i < PO_ST) {
    if (dbm != nullptr) dbm[i - PO_DBM] = s;
  } else if (i < PO_DA) {
    const int c = i - PO_ST;
    if (c < C && gamma_prev != nullptr) {            // thread c: S1 (this column) and S2
      const float S1 = s, S2 = ((red[1][0][lc] + red[1][1][lc]) + red[1][2][lc]) + red[1][3][lc];
      const float mean = mean_rstd_prev[c], rstd = mean_rstd_prev[C + c], gam = gamma_prev[c];
      const float dgam = rstd * (S2 - mean * S1);
      dgamma_prev[c] = dgam;
      dbeta_prev[c] = S1;
      const float sc = gam * rstd;
      const float inv_n = (float)(1.0 / n_prev);
      coef_prev[c] = sc;                                             // ca
      coef_prev[C + c] = -sc * rstd * dgam * inv_n;                   // cb
      coef_prev[2 * C + c] = -sc * S1 * inv_n + sc * mean * rstd * dgam * inv_n;   // ck
    }
  } else if (i < PO_DA + V * V) {
    if (dA1 != nullptr) dA1[i - PO_DA] = acc_dA ? dA1[i - PO_DA] + s : s;
  } else {
    if (dA2 != nullptr) dA2[i - PO_DA - V * V] = acc_dA ? dA2[i - PO_DA - V * V] + s : s;
  }
}

static_assert(PO_DBT % 64 == 0 && PO_DWM % 64 == 0 && PO_DBM % 64 == 0 && PO_ST % 64 == 0 && PO_DA % 64 == 0,
              "wn_bwd_reduce_kernel: 64-column workgroup blocks must not straddle sections that need special handling");

constexpr size_t WNB_LDS_LIMIT = 160 * 1024;        // LDS per CU on gfx950

static size_t wnb_lds_bytes(const GcnGeom& g) {
  return ((size_t)g.rows_lds * (3 * LDD + 2 * LDH) + (size_t)g.KP * g.ldA + (size_t)g.K2P * g.ldB) * sizeof(float);
}

static int wnb_grid(const LayerGeom& L) {
  const int cap = wn_env_int("HOPMI_WN_BWD_GRID", 256);
  return L.g.ntiles < cap ? L.g.ntiles : cap;
}

template <int MT, typename TS>
static void launch_wn_bwd(const TS* xin, const float* fs, const float* wf, const float* wg, const float* prep, const float* Wm,
                          const float* P0n, const float* P1n, const TS* y, const float* coef, const TS* dutail,
                          float* P0, float* P1, float* part, const LayerGeom& L, int do_gcn, int d_next, int T_next,
                          int dutail_ld, int grid, hipStream_t st) {
  const GcnGeom& g = L.g;
  const size_t lds = wnb_lds_bytes(g);
  hipLaunchKernelGGL((wn_layer_bwd_kernel<MT, TS>), dim3(grid), dim3(256), lds, st, xin, fs, wf, wg, prep, Wm, P0n, P1n, y, coef,
                     dutail, P0, P1, part, L, do_gcn, d_next, T_next, dutail_ld / 4);
}

}  // namespace hopmi

using namespace hopmi;

extern "C" size_t hopmi_wn_layer_bwd_ws_floats(int B, int T_in, int V, int dilation) {
  if (wn_validate(B, T_in, V, dilation)) return 0;
  const LayerGeom L = make_layer_geom(B, T_in, V, dilation, wn_env_int("HOPMI_WN_BWD_GRID", 256), WNB_MAX_MT);
  if (wnb_lds_bytes(L.g) > WNB_LDS_LIMIT) return 0;              // five tile images + both mix images do not fit (V > 42)
  return (size_t)wnb_grid(L) * part_floats(V);
}

template <typename TS>
static int wn_layer_bwd_impl(const TS* xin, const float* scsh_in, const float* fs, const float* wf, const float* wg,
                             const float* prep, const float* Wm, const float* P0n, const float* P1n, int d_next,
                             const TS* y, const float* bn_coef, const TS* dutail, int dutail_ld,
                             const float* gamma_prev, const float* mean_rstd_prev,
                             float* P0, float* P1, float* dwf, float* dwg, float* dbtcn, float* dWm, float* dbm,
                             float* dA1, float* dA2, int accumulate_dA, float* dgamma_prev, float* dbeta_prev,
                             float* coef_prev, float* ws,
                             int B, int T_in, int V, int dilation, int do_gcn, void* stream) {
  if (int e = wn_validate(B, T_in, V, dilation)) return e;
  if (!xin || !scsh_in || !fs || !wf || !wg || !dutail || !P0 || !P1 || !dwf || !dwg || !dbtcn || !ws) {
    set_error("hopmi_wn_layer_bwd: null pointer argument");
    return HOPMI_EINVAL;
  }
  if (do_gcn && (!prep || !Wm || !P0n || !P1n || !y || !bn_coef || !dWm || !dbm || !dA1 || !dA2)) {
    set_error("hopmi_wn_layer_bwd: do_gcn needs prep, Wm, P0n, P1n, y, bn_coef, dWm, dbm, dA1, dA2");
    return HOPMI_EINVAL;
  }
  if (gamma_prev && (!mean_rstd_prev || !dgamma_prev || !dbeta_prev || !coef_prev)) {
    set_error("hopmi_wn_layer_bwd: BatchNorm_{i-1} backward needs mean_rstd_prev, dgamma_prev, dbeta_prev, coef_prev");
    return HOPMI_EINVAL;
  }
  if (dutail_ld < C || (dutail_ld & 3)) { set_error("hopmi_wn_layer_bwd: dutail_ld=%d must be a multiple of 4 and >= 64", dutail_ld); return HOPMI_EINVAL; }
  const int T_out = T_in - dilation;
  if (do_gcn && (d_next < 1 || T_out - d_next < 4)) { set_error("hopmi_wn_layer_bwd: bad d_next=%d for T_out=%d", d_next, T_out); return HOPMI_EINVAL; }
  const LayerGeom L = make_layer_geom(B, T_in, V, dilation, wn_env_int("HOPMI_WN_BWD_GRID", 256), WNB_MAX_MT);
  if (wnb_lds_bytes(L.g) > WNB_LDS_LIMIT) {
    set_error("hopmi_wn_layer_bwd: V=%d needs %zu bytes of LDS per workgroup (limit %zu): the fused backward supports V <= 42",
              V, wnb_lds_bytes(L.g), WNB_LDS_LIMIT);
    return HOPMI_EINVAL;
  }
  const int grid = wnb_grid(L);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int T_next = T_out - d_next;
  switch (L.g.mtiles) {
    case 1: launch_wn_bwd<1, TS>(xin, fs, wf, wg, prep, Wm, P0n, P1n, y, bn_coef, dutail, P0, P1, ws, L, do_gcn, d_next, T_next, dutail_ld, grid, st); break;
    case 2: launch_wn_bwd<2, TS>(xin, fs, wf, wg, prep, Wm, P0n, P1n, y, bn_coef, dutail, P0, P1, ws, L, do_gcn, d_next, T_next, dutail_ld, grid, st); break;
    case 3: launch_wn_bwd<3, TS>(xin, fs, wf, wg, prep, Wm, P0n, P1n, y, bn_coef, dutail, P0, P1, ws, L, do_gcn, d_next, T_next, dutail_ld, grid, st); break;
    default: set_error("hopmi_wn_layer_bwd: internal: %d m-tiles", L.g.mtiles); return HOPMI_EINVAL;
  }
  if (int e = check_launch("hopmi_wn_layer_bwd")) return e;
  const int psz = part_floats(V);
  hipLaunchKernelGGL(wn_bwd_reduce_kernel, dim3((psz + 63) / 64), dim3(256), 0, st, ws, grid, V, scsh_in, gamma_prev,
                     mean_rstd_prev, (double)B * T_in * V, dwf, dwg, dbtcn, dWm, dbm, dA1, dA2, accumulate_dA, dgamma_prev, dbeta_prev,
                     coef_prev);
  return check_launch("hopmi_wn_bwd_reduce");
}

extern "C" int hopmi_wn_layer_bwd(const float* xin, const float* scsh_in, const float* fs, const float* wf, const float* wg,
                                  const float* prep, const float* Wm, const float* P0n, const float* P1n, int d_next,
                                  const float* y, const float* bn_coef, const float* dutail, int dutail_ld,
                                  const float* gamma_prev, const float* mean_rstd_prev,
                                  float* P0, float* P1, float* dwf, float* dwg, float* dbtcn, float* dWm, float* dbm,
                                  float* dA1, float* dA2, int accumulate_dA, float* dgamma_prev, float* dbeta_prev,
                                  float* coef_prev, float* ws,
                                  int B, int T_in, int V, int dilation, int do_gcn, void* stream) {
  return wn_layer_bwd_impl<float>(xin, scsh_in, fs, wf, wg, prep, Wm, P0n, P1n, d_next, y, bn_coef, dutail, dutail_ld, gamma_prev,
                                  mean_rstd_prev, P0, P1, dwf, dwg, dbtcn, dWm, dbm, dA1, dA2, accumulate_dA, dgamma_prev, dbeta_prev,
                                  coef_prev, ws, B, T_in, V, dilation, do_gcn, stream);
}

extern "C" int hopmi_wn_layer_bwd_dt(const void* xin, const float* scsh_in, const float* fs, const float* wf, const float* wg,
                                     const float* prep, const float* Wm, const float* P0n, const float* P1n, int d_next,
                                     const void* y, const float* bn_coef, const void* dutail, int dutail_ld,
                                     const float* gamma_prev, const float* mean_rstd_prev,
                                     float* P0, float* P1, float* dwf, float* dwg, float* dbtcn, float* dWm, float* dbm,
                                     float* dA1, float* dA2, int accumulate_dA, float* dgamma_prev, float* dbeta_prev,
                                     float* coef_prev, float* ws,
                                     int B, int T_in, int V, int dilation, int do_gcn, int dtype, void* stream) {
  if (dtype == HOPMI_F32)
    return wn_layer_bwd_impl<float>(static_cast<const float*>(xin), scsh_in, fs, wf, wg, prep, Wm, P0n, P1n, d_next,
                                    static_cast<const float*>(y), bn_coef, static_cast<const float*>(dutail), dutail_ld, gamma_prev,
                                    mean_rstd_prev, P0, P1, dwf, dwg, dbtcn, dWm, dbm, dA1, dA2, accumulate_dA, dgamma_prev, dbeta_prev,
                                    coef_prev, ws, B, T_in, V, dilation, do_gcn, stream);
  if (dtype == HOPMI_BF16)
    return wn_layer_bwd_impl<__bf16>(static_cast<const __bf16*>(xin), scsh_in, fs, wf, wg, prep, Wm, P0n, P1n, d_next,
                                     static_cast<const __bf16*>(y), bn_coef, static_cast<const __bf16*>(dutail), dutail_ld, gamma_prev,
                                     mean_rstd_prev, P0, P1, dwf, dwg, dbtcn, dWm, dbm, dA1, dA2, accumulate_dA, dgamma_prev, dbeta_prev,
                                     coef_prev, ws, B, T_in, V, dilation, do_gcn, stream);
  set_error("hopmi_wn_layer_bwd_dt: dtype %d (0 = fp32, 1 = bf16)", dtype);
  return HOPMI_EINVAL;
}
