// The whole WaveNet stack of the HOP graph-wavenet block (reference: model/gwnet.py:181-237, 8 layers) as ONE persistent launch,
// training mode (BatchNorm batch statistics):
//
//   for layer i:   r^ = BN_{i-1}(y_{i-1}) on load;  u = tanh(.) sigmoid(.) (gated TCN);  skip tail;  y_i = gcn(u) + r^[t+d];
//                  per-channel sum / sum of squares of y_i  ->  EXCHANGED ACROSS ALL WORKGROUPS  ->  scale / shift of BN_i
//
// Training-mode BatchNorm makes every layer a chip-wide dependency: layer i + 1 of ANY clip needs the statistics of layer i over
// ALL clips.  As eight launches that seam was a kernel boundary plus a finalisation launch per layer (8 x (10.9 + 9.8) us at
// TED / B = 128); here it is an in-launch exchange whose critical path is THREE memory hops.  Per layer every workgroup
//   1. runs its tile(s) of the layer exactly as wn_layer_fwd_kernel does (same phases, same arithmetic: three-term products of
//      scaled fp16 hi/lo operands, f16_dev.h -- fp32-equivalent), writing y_i with sc1 (write-through) stores;
//   2. publishes its 128 partial sums at once as data-tagged granules (8 bytes = {tag, fp32 bits}, ONE aligned store each:
//      the data is the flag, nothing to drain first);
//   3. the first workgroup of each group (group = workgroup index mod 8: one XCD under round-robin placement -- for speed
//      only, nothing depends on it) sweeps its group's granules until every tag matches, adds them in index order (double) and
//      publishes the group sum as granules (two per value: hi and lo float of the double);
//   4. meanwhile every workgroup drains its y stores, raises the "drained" flag of its tiles, waits for the flags of the few
//      tiles that produced the rows ITS next tile reads (neighbours: they drained at the same time) and issues that tile's
//      loads -- the activations of layer i + 1 arrive while the statistics are still on their way;
//   5. every workgroup sweeps the <= 8 group sums until every tag matches, adds them in index order and computes mean / rstd /
//      scale / shift itself: the same numbers in every workgroup, bitwise reproducible from run to run (fixed summation order
//      everywhere, no floating-point atomics).
// Hand-off forms (MI355X_MICROARCH.md, visibility): R2 granules for the statistics (tag = launch sequence number x 16 + layer + 1,
// never 0; the sequence number lives in the workspace and is advanced by workgroup 0 at the very end of a launch, so a replayed
// hipGraph sees fresh tags; no counter, nothing to reset); for y: sc1 stores, every storing wave drains, workgroup barrier,
// ONE lane's sc1 flag store; consumers: sc1 flag poll, workgroup barrier, sc1 loads of every handed-off byte.
// Requirements: the grid is resident at once (host: grid <= CUs).  Every spin is bounded: on time-out the workgroup raises the
// status word and leaves the kernel, so the launch always drains (results are then garbage, the host raises on the status word).
#include <hip/hip_ext.h>

#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#define HOPMI_FILE_ID 8          // (diagnostic build: common.h, split_check)
#include "attn_dev.h"
#include "f16_dev.h"
#include "io_dev.h"
#include "wn_dev.h"

namespace hopmi {

constexpr int STK_MAX_LAYERS = 8;
constexpr int STK_GROUPS = 8;
constexpr int STK_THREADS = 512;
// control block (ints): [0] launch sequence number, [32] status word
constexpr int STK_SYNC_INTS = 64, STK_STATUS_AT = 32;
typedef unsigned long long u64;

struct StackLayer {
  const void* xin;       // (B, T_in, V, 64) of the storage type: x0 for layer 0, y_{i-1} after
  void* y;               // (B, T_out, V, 64) or null (last layer: its output is dead, gwnet.py:240)
  const float* bf; const float* bg; const float* bm;
  const float* gamma; const float* beta;
  float* rmean; float* rvar;       // running statistics (updated in place) or null
  int T_in, T_out, d, n_slabs, S, ntiles;
  float invT, unbias;    // 1 / T_out;  n / (n - 1)
  double inv_n;          // 1 / (rows of the layer's output: B T_out V)
};

struct StackArgs {
  StackLayer L[STK_MAX_LAYERS];
  const u32x4* wimg;     // n_layers weight images (hopmi_wn_prepare_weights)
  const float* prep;     // mix-matrix images (hopmi_gcn_prepare)
  void* utail;           // (B, 4, V, utail_ld) of the storage type: layer i's skip tail at channel offset 64 i
  float* scsh_out;       // [n_layers][128]  scale | shift of BN_i
  float* mean_rstd;      // [n_layers][192]  mean | rstd | unbiased variance
  int* sync;             // STK_SYNC_INTS ints: launch sequence number, status word (zero before the first launch)
  u64* pgran;            // [n_layers][grid][128]        partial sums, granules {tag, fp32 bits}
  u64* ggran;            // [n_layers][STK_GROUPS][256]  group sums, granules (hi, lo) per value
  unsigned* yflag;       // [n_layers][max_tiles]        "the y rows of this tile have left the workgroup" (= tag)
  int max_tiles;
  int n_layers, B, V, utail_ld4, n_comb;
  int KP, ldA, MP;       // mix-matrix image geometry (GcnGeom)
  int prep_tr;           // float offset of prep's operand-scale trailer {s, 1 / s} (hopmi_gcn_prepare)
  float invV, momentum, eps;
};

#ifdef HOPMI_STAMPS
static __device__ long long* g_stk_stamps = nullptr;
#define STK_STAMP(layer, slot)                                                                              \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    unsigned long long t_;                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (g_stk_stamps && threadIdx.x == 0) g_stk_stamps[(blockIdx.x * STK_MAX_LAYERS + (layer)) * 16 + (slot)] = (long long)t_; \
  } while (0)
#else
#define STK_STAMP(layer, slot) do { } while (0)
#endif

__device__ __forceinline__ float stk_gate(float a, float g) {       // tanh(a) sigmoid(g), one reciprocal (wavenet.hip: gate_)
  const float ea = __expf(fminf(-2.f * a, 44.f)), eg = __expf(fminf(-g, 44.f));
  return (1.f - ea) * __builtin_amdgcn_rcpf((1.f + ea) * (1.f + eg));
}

typedef __attribute__((address_space(8))) void* rsrc_t;        // buffer resource (V#)
__device__ __forceinline__ auto stk_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
constexpr int AUX_SC1 = 16;                                     // gfx940+: cache-policy bit 4 = sc1 (write-through / bypass L1)

// Storage type of the activation tensors (x0, y_i, skip tails): 4 channels per access, sc1 buffer loads / stores for what is
// handed between workgroups inside the launch; all arithmetic in fp32 (bf16: y is rounded once, when it is stored).
template <typename TS> struct StkIO;
template <> struct StkIO<float> {
  typedef u32x4 raw;
  static constexpr int ES = 4;
  template <typename R> static __device__ __forceinline__ raw load(R r, int byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX_SC1); }
  static __device__ __forceinline__ float4 cvt(raw v) { return __builtin_bit_cast(float4, v); }
  template <typename R> static __device__ __forceinline__ void store(float4 v, R r, int byte_off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, AUX_SC1);
  }
};
template <> struct StkIO<__bf16> {
  typedef u32x2 raw;
  static constexpr int ES = 2;
  template <typename R> static __device__ __forceinline__ raw load(R r, int byte_off) { return __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, AUX_SC1); }
  static __device__ __forceinline__ float4 cvt(raw v) { return make_float4(bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])); }
  template <typename R> static __device__ __forceinline__ void store(float4 v, R r, int byte_off) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk_bf16(v.x, v.y), pk_bf16(v.z, v.w)}, r, byte_off, 0, AUX_SC1);
  }
};

__device__ __forceinline__ bool stk_expired(unsigned long long t0, int* status) {
  if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {              // 2 s at 100 MHz
    __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  return false;
}

// N granules per lane, re-read until every tag matches (R2: the data is the flag); v[k] = the value bits.  Bounded.
template <int N>
__device__ __forceinline__ bool stk_sweep(const u64* const (&p)[N], unsigned tag, unsigned (&v)[N], int* status, int* passes = nullptr) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    if (passes != nullptr) ++*passes;                 // (diagnostic builds only)
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const u64 x = p[k] != nullptr ? __hip_atomic_load(p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((u64)tag << 32);
      v[k] = (unsigned)x;
      ok &= (unsigned)(x >> 32) == tag;
    }
    if (ok) return true;
    if (stk_expired(t0, status)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// Node mix of the slabs s = h, h + 2, ... of a tile (as wavenet.hip: node_mix2; see there).
template <int KS, int MTN, bool HOLD>
__device__ __forceinline__ void stk_node_mix(const float* U, __bf16* Hh, __bf16* Hl, const float* AT, int V, int ldA, int nsl,
                                             int dump_row, int w, int h, int q, int j, float sh) {
  float am[HOLD ? MTN : 1][HOLD ? KS : 1];
  int woff[MTN];
#pragma unroll
  for (int mt = 0; mt < MTN; ++mt) {
    if (HOLD) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) am[mt][ks] = AT[(4 * ks + q) * ldA + 16 * mt + j];
    }
    const int m = 16 * mt + j;
    const int blk = (m >= V) ? 1 : 0;
    woff[mt] = (m < 2 * V) ? ((m - blk * V) * HS + C * (1 + blk) + 16 * w + 4 * q) : -1;
  }
  const int dump = dump_row * HS + C + 16 * w + 4 * q;
  constexpr int SGN = HOLD ? 2 : 1;
  int s = h;
  for (; s + 2 * (SGN - 1) < nsl; s += 2 * SGN) {
    float xb[SGN][KS];
#pragma unroll
    for (int sg = 0; sg < SGN; ++sg)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xb[sg][ks] = U[((s + 2 * sg) * V + 4 * ks + q) * LDD + 16 * w + j];
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) {
      f32x4 acc[SGN];
#pragma unroll
      for (int sg = 0; sg < SGN; ++sg) acc[sg] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float b = HOLD ? am[mt][ks] : AT[(4 * ks + q) * ldA + 16 * mt + j];
#pragma unroll
        for (int sg = 0; sg < SGN; ++sg) acc[sg] = mfma16(xb[sg][ks], b, acc[sg]);
      }
#pragma unroll
      for (int sg = 0; sg < SGN; ++sg) {
        const int off = woff[mt] >= 0 ? (s + 2 * sg) * V * HS + woff[mt] : dump;
        const Split4 sp = split4h(acc[sg][0] * sh, acc[sg][1] * sh, acc[sg][2] * sh, acc[sg][3] * sh);
        *reinterpret_cast<u32x2*>(Hh + off) = sp.hi;
        *reinterpret_cast<u32x2*>(Hl + off) = sp.lo;
      }
    }
  }
  for (; s < nsl; s += 2) {
    float xb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[ks] = U[(s * V + 4 * ks + q) * LDD + 16 * w + j];
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = mfma16(xb[ks], HOLD ? am[mt][ks] : AT[(4 * ks + q) * ldA + 16 * mt + j], acc);
      const int off = woff[mt] >= 0 ? s * V * HS + woff[mt] : dump;
      const Split4 sp = split4h(acc[0] * sh, acc[1] * sh, acc[2] * sh, acc[3] * sh);
      *reinterpret_cast<u32x2*>(Hh + off) = sp.hi;
      *reinterpret_cast<u32x2*>(Hl + off) = sp.lo;
    }
  }
}

__device__ __forceinline__ void stk_node_mix_generic(const float* U, __bf16* Hh, __bf16* Hl, const float* AT, int V, int ldA, int KP,
                                                     int MP, int nsl, int w, int h, int q, int j, float sh) {
  const int ksteps = KP >> 2, mt_n = MP >> 4;
  for (int s = h; s < nsl; s += 2) {
    const float* us = U + (s * V + q) * LDD + 16 * w + j;
    for (int mt = 0; mt < mt_n; ++mt) {
      const float* at = AT + q * ldA + 16 * mt + j;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < ksteps; ++ks) acc = mfma16(us[4 * ks * LDD], at[4 * ks * ldA], acc);
      const int m = 16 * mt + j;
      if (m < 2 * V) {
        const int blk = (m >= V) ? 1 : 0;
        const int off = (s * V + m - blk * V) * HS + C * (1 + blk) + 16 * w + 4 * q;
        const Split4 sp = split4h(acc[0] * sh, acc[1] * sh, acc[2] * sh, acc[3] * sh);
        *reinterpret_cast<u32x2*>(Hh + off) = sp.hi;
        *reinterpret_cast<u32x2*>(Hl + off) = sp.lo;
      }
    }
  }
}

// MT = the LARGEST tile (in 16-row MFMA tiles) of any layer of the launch: it fixes the LDS layout; a layer whose tiles are
// smaller skips the MFMA tiles it does not have (wave-uniform predicates).  16-row tile mt belongs to row half mt & 1, so the
// two halves stay balanced at every tile size.
template <int MT, typename TS>
__global__ __launch_bounds__(STK_THREADS) void wn_stack_fwd_kernel(StackArgs A) {
  typedef StkIO<TS> IO;
  constexpr int ES = IO::ES;                          // bytes per stored element
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NIT = ((16 * MT + 4) * 16 + STK_THREADS - 1) / STK_THREADS;
  constexpr int MTH = (MT + 1) / 2;
  constexpr int rows_lds = 16 * MT + 4;
  __bf16* R0h = reinterpret_cast<__bf16*>(smem);
  __bf16* R0l = R0h + rows_lds * RS;
  __bf16* R1h = R0l + rows_lds * RS;
  __bf16* R1l = R1h + rows_lds * RS;
  float* U = reinterpret_cast<float*>(R1l + rows_lds * RS);
  __bf16* Hh = reinterpret_cast<__bf16*>(U + rows_lds * LDD);
  __bf16* Hl = Hh + rows_lds * HS;
  float* AT = reinterpret_cast<float*>(Hl + rows_lds * HS);
  float* SCSH = AT + A.KP * A.ldA;                                 // [128] scale | shift of the layer being read
  int* FLAG = reinterpret_cast<int*>(SCSH + 2 * C);                // [4]
  float* GB = reinterpret_cast<float*>(FLAG + 4);                  // [n_layers][128] gamma | beta of every layer (read once: the
                                                                   // finalisation sits on the critical path of every exchange)
  float* WINV = GB + STK_MAX_LAYERS * 2 * C;                       // [n_layers][192] inverse weight scales: filter | gate | Wm (x 1 / s_h)
  float* RSI = WINV + STK_MAX_LAYERS * 3 * C;                      // [rows_lds] inverse of the tap panels' row scales
  float* RSL = RSI + rows_lds;                                     // [n_layers] uniform operand scale of layer l's input (l >= 1: behind a BatchNorm)
  // exchange scratch, aliased onto the (dead between layers) operand images Hh | Hl: 2 x rows_lds x 416 B >= 16.6 KB at MT = 1
  float* RED = reinterpret_cast<float*>(Hh);                       // [2][128] floats
  double* COMB = reinterpret_cast<double*>(RED + 4 * C);           // [4][128] doubles
  double* FIN = COMB + 4 * 2 * C;                                  // [8][128] doubles

  int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), w = wv & 3, h = wv >> 2;
  int lane = tid & 63, q = lane >> 4, j = lane & 15, c4 = tid & 15;
  const int V = A.V;
  // roles: the first G workgroups compute tiles, the last n_groups are DEDICATED combiners (one per group: they have no tile,
  // so they sweep their group's granules from the moment the layer starts and publish the group sum as soon as the slowest
  // member's row is visible -- a computing combiner reached its sweep ~5 000 cycles after its own tile)
  const int n_groups = A.n_comb, G = gridDim.x - n_groups, bid = blockIdx.x;
  const bool combiner = bid >= G;
  const int grp = combiner ? bid - G : bid % n_groups;
  const int gsz = (G - grp + n_groups - 1) / n_groups;             // computing workgroups b = grp + n_groups m, m < gsz
  int* status = A.sync + STK_STATUS_AT;

  // identity scale / shift for layer 0
  if (tid < 2 * C) SCSH[tid] = tid < C ? 1.f : 0.f;
  for (int idx = tid; idx < A.n_layers * 2 * C; idx += STK_THREADS) {
    const int l = idx / (2 * C), c = idx % (2 * C);
    GB[idx] = c < C ? A.L[l].gamma[c] : A.L[l].beta[c - C];
  }
  // operand scale of the contraction images u | uA1 | uA2 (uniform; hopmi_gcn_prepare) and the layers' inverse weight scales
  const float sh = A.prep[A.prep_tr], ish = A.prep[A.prep_tr + 1];
  for (int idx = tid; idx < A.n_layers * 3 * C; idx += STK_THREADS) {
    const int l = idx / (3 * C), c = idx % (3 * C);
    const float v = reinterpret_cast<const float*>(A.wimg + (size_t)l * WIMGH_UNITS + WIMG_UNITS)[c];
    WINV[idx] = c < 2 * C ? v : v * ish;
  }
  // The uniform operand scales of the layers behind a BatchNorm depend on the PARAMETERS only: |BN_l(y)_c| <= |gamma_c| sqrt(n - 1) +
  // |beta_c| (a deviation from the mean is at most sqrt(n - 1) biased standard deviations; rstd sqrt(var) <= 1), 1/16 on top for the
  // storage rounding of y (bf16 mode) and the fp32 evaluation.  Wave l computes layer l + 1's scale here, off every critical path.
  if (wv + 1 < A.n_layers) {
    const float g0 = A.L[wv].gamma[lane], b0 = A.L[wv].beta[lane];
    const float bnd = (fabsf(g0) * sqrtf((float)(1.0 / A.L[wv].inv_n)) + fabsf(b0)) * 1.0625f;
    const float mx = wave_max_nonneg(bnd);
    if (lane == 0) RSL[wv + 1] = scale_for_absmax(mx);
  }
  if (tid == 0) { RSL[0] = 1.f; FLAG[0] = FLAG[1] = FLAG[2] = 0; FLAG[3] = *reinterpret_cast<volatile int*>(A.sync); }   // [3]: launch sequence number
  // u rows this workgroup never writes (tiles smaller than MT, the 4 padding rows) are read by the node mix's K padding times
  // zero: they must be finite
  for (int idx = tid; idx < rows_lds * LDD; idx += STK_THREADS) U[idx] = 0.f;
  {
    // the mix image -> LDS: up to 4 independent loads per thread and round trip (V = 42: 1 232 float4 = 3 per thread)
    const int at_n4 = (A.KP * A.ldA) >> 2;
    const float4* src = reinterpret_cast<const float4*>(A.prep);
    for (int idx0 = tid; idx0 < at_n4; idx0 += 4 * STK_THREADS) {
      const float4 v0 = src[min(idx0, at_n4 - 1)], v1 = src[min(idx0 + STK_THREADS, at_n4 - 1)];
      const float4 v2 = src[min(idx0 + 2 * STK_THREADS, at_n4 - 1)], v3 = src[min(idx0 + 3 * STK_THREADS, at_n4 - 1)];
      reinterpret_cast<float4*>(AT)[idx0] = v0;
      if (idx0 + STK_THREADS < at_n4) reinterpret_cast<float4*>(AT)[idx0 + STK_THREADS] = v1;
      if (idx0 + 2 * STK_THREADS < at_n4) reinterpret_cast<float4*>(AT)[idx0 + 2 * STK_THREADS] = v2;
      if (idx0 + 3 * STK_THREADS < at_n4) reinterpret_cast<float4*>(AT)[idx0 + 3 * STK_THREADS] = v3;
    }
  }

  // weight fragments: straight from the prepared image (L2-resident), the TCN set behind the tile's commit, the graph-conv set
  // behind the TCN phase, so that neither is live with the other
  u32x4 wt[2][4][2];
  u32x4 wm[6][2];
  auto load_wt = [&](int layer) {
    const u32x4* tp = A.wimg + (size_t)layer * WIMGH_UNITS + (size_t)(w * 2) * 4 * 2 * 64 + lane;
#pragma unroll
    for (int gate = 0; gate < 2; ++gate)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int part = 0; part < 2; ++part) wt[gate][ks][part] = tp[((gate * 4 + ks) * 2 + part) * 64];
  };
  // (one k-step's fragments of both gates: issued between the commit loop's iterations, see phase 0)
  auto load_wt_ks = [&](int layer, int ks) {
    const u32x4* tp = A.wimg + (size_t)layer * WIMGH_UNITS + (size_t)(w * 2) * 4 * 2 * 64 + lane;
#pragma unroll
    for (int gate = 0; gate < 2; ++gate)
#pragma unroll
      for (int part = 0; part < 2; ++part) wt[gate][ks][part] = tp[((gate * 4 + ks) * 2 + part) * 64];
  };
  auto load_wm_ks = [&](int layer, int ks) {
    const u32x4* mp = A.wimg + (size_t)layer * WIMGH_UNITS + WIMG_TCN_UNITS + (size_t)(w * 6) * 2 * 64 + lane;
#pragma unroll
    for (int part = 0; part < 2; ++part) wm[ks][part] = mp[(ks * 2 + part) * 64];
  };
  auto load_wm = [&](int layer) {
    const u32x4* mp = A.wimg + (size_t)layer * WIMGH_UNITS + WIMG_TCN_UNITS + (size_t)(w * 6) * 2 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
      for (int part = 0; part < 2; ++part) wm[ks][part] = mp[(ks * 2 + part) * 64];
  };

  // a tile's two tap panels -> registers (sc1: rows another workgroup wrote in this launch); the tile's row map stays in
  // registers (tail[]: where the skip tail of the row goes, or -1; ok[]: row < R)
  int tail[NIT];
  bool ok[NIT];
  typename IO::raw x0r[NIT], x1r[NIT];
  auto issue_tile = [&](const StackLayer& L, int tile) {
    const int slab0 = tile * L.S;
    const int R = min(L.S, L.n_slabs - slab0) * V;
    const auto xr = stk_rsrc(L.xin, (unsigned)A.B * L.T_in * V * (64u * ES));
    const int shift4b = L.d * V * 64 * ES;           // tap-1 row offset in bytes
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = (tid >> 4) + (STK_THREADS / 16) * it;
      const int rc = min(row, R - 1);
      const int s = (int)((rc + 0.5f) * A.invV);
      const int v = rc - s * V;
      const int slab = slab0 + s;
      const int b = (int)((slab + 0.5f) * L.invT);
      const int tp = slab - b * L.T_out;
      ok[it] = row < R;
      const int in0 = (((b * L.T_in + tp) * V + v) * 16 + c4) * 4 * ES;          // byte offset
      tail[it] = (ok[it] && tp >= L.T_out - 4) ? ((b * 4 + tp - (L.T_out - 4)) * V + v) * A.utail_ld4 + c4 : -1;
      x0r[it] = IO::load(xr, in0);
      x1r[it] = IO::load(xr, in0 + shift4b);
    }
  };
  // per-layer constants (this lane's 4 channels): requested with the first tile's panels, i.e. during the previous layer's
  // exchange, so that no load is young when the layer starts (the wait in front of the weight loads would pay for it)
  float4 bf4, bg4, bias4;
  auto load_consts = [&](const StackLayer& L) {
    bf4 = *reinterpret_cast<const float4*>(L.bf + 16 * w + 4 * q);
    bg4 = *reinterpret_cast<const float4*>(L.bg + 16 * w + 4 * q);
    bias4 = *reinterpret_cast<const float4*>(L.bm + 16 * w + 4 * q);
  };
  if (!combiner && bid < A.L[0].ntiles) issue_tile(A.L[0], bid);
  load_consts(A.L[0]);
  {
    // touch every layer's descriptor now (first, middle and last word: all its cache lines): the scalar loads of a layer's
    // geometry then hit the scalar cache instead of paying a memory round trip at the head of every layer
    int warm = 0;
    for (int l = 0; l < A.n_layers; ++l) warm += (int)(size_t)A.L[l].xin + (int)(size_t)A.L[l].gamma + A.L[l].ntiles + (int)A.L[l].invT;
    if (tid == 0) FLAG[2] = warm;
  }
  __syncthreads();                                   // SCSH / FLAG[3] are read below
  const unsigned seq = (unsigned)FLAG[3];

  for (int layer = 0; layer < A.n_layers; ++layer) {
    const StackLayer& L = A.L[layer];
    STK_STAMP(layer, 0);
    const bool last_layer = layer == A.n_layers - 1;
    const unsigned tag = (seq << 4) | (unsigned)(layer + 1);
    const auto yr = stk_rsrc(L.y, L.y != nullptr ? (unsigned)A.B * L.T_out * V * (64u * ES) : 0u);   // (null y: zero records, stores dropped)
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
    TS* utail = static_cast<TS*>(A.utail) + C * layer;

    // the first tile's tap panels are in flight since the previous layer's exchange
    for (int tile = combiner ? L.ntiles : bid; tile < L.ntiles; tile += G) {
      asm volatile("" : "+v"(tid));                  // keep per-lane address math from being hoisted out of the loops
      lane = tid & 63; q = lane >> 4; j = lane & 15; c4 = tid & 15;
      const int slab0 = tile * L.S;
      const int nsl = min(L.S, L.n_slabs - slab0);
      const int R = nsl * V;
      const int nt = (R + 15) >> 4;                  // populated 16-row tiles
      const unsigned orow0 = (unsigned)slab0 * V;

      // ---- phase 0: both tap panels (in flight since before the statistics arrived for the first tile) -> normalise -> split -> LDS
      if (tile != bid) issue_tile(L, tile);          // (its producers' flags were checked with the first tile's, see the exchange)
      float4 sc4 = reinterpret_cast<const float4*>(SCSH)[c4];
      float4 sh4 = reinterpret_cast<const float4*>(SCSH + C)[c4];
      // Layers behind a BatchNorm take ONE power-of-two operand scale for the whole layer from an a-priori bound (computed in the
      // kernel's prologue from gamma / beta): |gamma_c| sqrt(n - 1) + |beta_c| bounds every normalised value, values at one
      // sigma sit 2^-7 ... 2^-9 below it -- far inside the 2^17 window in which an fp16 hi / lo pair keeps 22 bits (f16_dev.h) -- and the
      // scale folds into the scale / shift the tile is multiplied with anyway (a power of two: exact).  Layer 0 reads the start
      // conv's output, which has no such bound: per-row scales from the row maxima (a DPP row reduction per row).
      const bool uni = layer > 0;
      const float rsl = uni ? RSL[layer] : 1.f, irsl = inv_pow2(rsl);
      if (uni) {
        sc4 = make_float4(sc4.x * rsl, sc4.y * rsl, sc4.z * rsl, sc4.w * rsl);
        sh4 = make_float4(sh4.x * rsl, sh4.y * rsl, sh4.z * rsl, sh4.w * rsl);
      }
      __syncthreads();                               // previous tile's LDS fully consumed
      STK_STAMP(layer, 12);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
#ifndef STK_EXP_NO_WT
        // the TCN weight fragments (L2-resident image, 128 KiB per workgroup through the CU's 64 B/clk vector-memory path: ~2 000
        // cycles of ISSUE) ride between the commit's iterations, one k-step's four loads at a time: the conversions of iteration
        // `it` run while they are in the memory pipe.  (All 16 in front of the commit made the waves sit in their issue; all 16
        // behind it -- rounds 3-4 -- put the 2 000 cycles on the critical path of every layer.)
        if (it < 4) { load_wt_ks(layer, it); __builtin_amdgcn_sched_barrier(0); }
#endif
        const int row = (tid >> 4) + (STK_THREADS / 16) * it;
        if (row < rows_lds) {
          float4 a = IO::cvt(x0r[it]), b2 = IO::cvt(x1r[it]);
          a = make_float4(a.x * sc4.x + sh4.x, a.y * sc4.y + sh4.y, a.z * sc4.z + sh4.z, a.w * sc4.w + sh4.w);
          b2 = make_float4(b2.x * sc4.x + sh4.x, b2.y * sc4.y + sh4.y, b2.z * sc4.z + sh4.z, b2.w * sc4.w + sh4.w);
          if (!ok[it]) { a = make_float4(0.f, 0.f, 0.f, 0.f); b2 = a; }
          float rs = 1.f, irs = irsl;
          if (!uni) {
            // one power-of-two scale per output row over BOTH taps (the row's 128 values sit in the 16 lanes of a DPP row)
            rs = scale_for_absmax(row16_max(absmax4(absmax4(0.f, a), b2)));
            irs = inv_pow2(rs);
          }
          const Split4 sa = split4h(a.x * rs, a.y * rs, a.z * rs, a.w * rs), sb = split4h(b2.x * rs, b2.y * rs, b2.z * rs, b2.w * rs);
          const int off = row * RS + 4 * c4;
          *reinterpret_cast<u32x2*>(R0h + off) = sa.hi;
          *reinterpret_cast<u32x2*>(R0l + off) = sa.lo;
          *reinterpret_cast<u32x2*>(R1h + off) = sb.hi;
          *reinterpret_cast<u32x2*>(R1l + off) = sb.lo;
          if (c4 == 0) RSI[row] = irs;
        }
      }
      // the TCN weight fragments (L2-resident image, 128 KiB per workgroup through the CU's 64 B/clk vector-memory path: ~2 000
      // cycles).  Requested only here: in front of the commit the waves sit in the issue of these loads instead of committing,
      // and in front of the exchange's sweeps every sweep pass would wait for them (a load's data waits for every older one)
      STK_STAMP(layer, 13);
      __builtin_amdgcn_sched_barrier(0);
#ifdef STK_EXP_NO_WT
      if (layer == 0) load_wt(layer);                // (timing experiment: results wrong)
#else
      // (the k-steps the commit loop had no iteration for; spreading all four over the iterations measured no better: 93.8-95.8 us
      // against 94.3-95.0, V = 42 219.6 against 216.7)
#pragma unroll
      for (int ks = NIT < 4 ? NIT : 4; ks < 4; ++ks) load_wt_ks(layer, ks);
#endif
      __syncthreads();
      STK_STAMP(layer, 1);

      // ---- phase 1: gated TCN as split products, gate, u -> LDS -------------------------------------------------------
      {
        f32x4 af[MTH], ag[MTH];
#pragma unroll
        for (int i = 0; i < MTH; ++i) { af[i] = {0.f, 0.f, 0.f, 0.f}; ag[i] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const __bf16* rh = ((ks >> 1) ? R1h : R0h) + j * RS + 32 * (ks & 1) + 8 * q;
          const __bf16* rl = ((ks >> 1) ? R1l : R0l) + j * RS + 32 * (ks & 1) + 8 * q;
          u32x4 bh[MTH], bl[MTH];
#pragma unroll
          for (int i = 0; i < MTH; ++i) {
            const int mt = min(2 * i + h, MT - 1);
            bh[i] = *reinterpret_cast<const u32x4*>(rh + 16 * mt * RS);
            bl[i] = *reinterpret_cast<const u32x4*>(rl + 16 * mt * RS);
          }
#pragma unroll
          for (int i = 0; i < MTH; ++i) {
            if (2 * i + h < nt) {
              af[i] = mfma_h3(wt[0][ks][0], wt[0][ks][1], bh[i], bl[i], af[i]);
              ag[i] = mfma_h3(wt[1][ks][0], wt[1][ks][1], bh[i], bl[i], ag[i]);
            }
          }
        }
        STK_STAMP(layer, 2);
#pragma unroll
        for (int i = 0; i < MTH; ++i) {
#ifndef STK_EXP_NO_WT
          // (the graph-conv weight fragments -- 96 KiB per workgroup, ~1 500 cycles of issue -- ride between the gate loop's
          // iterations like the TCN fragments between the commit's: two k-steps per iteration, the rest behind the loop)
#pragma unroll
          for (int ks = 2 * i; ks < 2 * i + 2 && ks < 6; ++ks) load_wm_ks(layer, ks);
          __builtin_amdgcn_sched_barrier(0);
#endif
          const int mt = 2 * i + h;
          if (mt < MT && mt < nt) {
            const int row = 16 * mt + j;
            const float ir = uni ? irsl : RSI[row];          // (uniform layers: a scalar, no LDS read)
            const float4 isf4 = *reinterpret_cast<const float4*>(WINV + layer * 3 * C + 16 * w + 4 * q);
            const float4 isg4 = *reinterpret_cast<const float4*>(WINV + layer * 3 * C + C + 16 * w + 4 * q);
            // pre-activations in real units: accumulator x (1 / row scale) x (1 / output channel's weight scale) + bias
            const float4 u = make_float4(stk_gate(af[i][0] * (isf4.x * ir) + bf4.x, ag[i][0] * (isg4.x * ir) + bg4.x),
                                         stk_gate(af[i][1] * (isf4.y * ir) + bf4.y, ag[i][1] * (isg4.y * ir) + bg4.y),
                                         stk_gate(af[i][2] * (isf4.z * ir) + bf4.z, ag[i][2] * (isg4.z * ir) + bg4.z),
                                         stk_gate(af[i][3] * (isf4.w * ir) + bf4.w, ag[i][3] * (isg4.w * ir) + bg4.w));
            *reinterpret_cast<float4*>(U + row * LDD + 16 * w + 4 * q) = u;
            const Split4 su = split4h(u.x * sh, u.y * sh, u.z * sh, u.w * sh);
            *reinterpret_cast<u32x2*>(Hh + row * HS + 16 * w + 4 * q) = su.hi;
            *reinterpret_cast<u32x2*>(Hl + row * HS + 16 * w + 4 * q) = su.lo;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef STK_EXP_NO_WT
        if (layer == 0) load_wm(layer);
#else
#pragma unroll
        for (int ks = 2 * MTH < 6 ? 2 * MTH : 6; ks < 6; ++ks) load_wm_ks(layer, ks);     // (what the loop had no iteration for)
#endif
      }
      __syncthreads();
      STK_STAMP(layer, 3);

      // ---- skip tail: last 4 frames of u, LDS -> HBM as whole 256-B rows (read by later launches only: plain stores) --
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = (tid >> 4) + (STK_THREADS / 16) * it;
        if (tail[it] >= 0) st4(utail + 4 * (size_t)tail[it], *reinterpret_cast<const float4*>(U + row * LDD + 4 * c4));
      }

      // ---- phase 2: node mix (exact fp32 MFMA, K = V) -> split images ------------------------------------------------
      if (V == 9) stk_node_mix<3, 2, true>(U, Hh, Hl, AT, V, A.ldA, nsl, rows_lds - 1, w, h, q, j, sh);
      else if (V == 42) stk_node_mix<11, 6, false>(U, Hh, Hl, AT, V, A.ldA, nsl, rows_lds - 1, w, h, q, j, sh);
      else stk_node_mix_generic(U, Hh, Hl, AT, V, A.ldA, A.KP, A.MP, nsl, w, h, q, j, sh);
      __syncthreads();
      STK_STAMP(layer, 4);

      // ---- phase 3: channel contraction (K = 192) + bias + residual, y store (write-through), BatchNorm sums ---------
      f32x4 acc[MTH];
#pragma unroll
      for (int i = 0; i < MTH; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        u32x4 bh[MTH], bl[MTH];
#pragma unroll
        for (int i = 0; i < MTH; ++i) {
          const int mt = min(2 * i + h, MT - 1);
          bh[i] = *reinterpret_cast<const u32x4*>(Hh + (16 * mt + j) * HS + 32 * ks + 8 * q);
          bl[i] = *reinterpret_cast<const u32x4*>(Hl + (16 * mt + j) * HS + 32 * ks + 8 * q);
        }
#pragma unroll
        for (int i = 0; i < MTH; ++i)
          if (2 * i + h < nt) acc[i] = mfma_h3(wm[ks][0], wm[ks][1], bh[i], bl[i], acc[i]);
      }
      STK_STAMP(layer, 5);
#pragma unroll
      for (int i = 0; i < MTH; ++i) {
        const int mt = 2 * i + h;
        const int row = 16 * mt + j;
        if (mt < MT && row < R) {
          const float4 res = join4h(*reinterpret_cast<const u32x2*>(R1h + row * RS + 16 * w + 4 * q),
                                    *reinterpret_cast<const u32x2*>(R1l + row * RS + 16 * w + 4 * q), uni ? irsl : RSI[row]);
          const float4 ism4 = *reinterpret_cast<const float4*>(WINV + layer * 3 * C + 2 * C + 16 * w + 4 * q);   // (1 / s_o)(1 / s_h)
          const f32x4 yv = {acc[i][0] * ism4.x + bias4.x + res.x, acc[i][1] * ism4.y + bias4.y + res.y, acc[i][2] * ism4.z + bias4.z + res.z,
                            acc[i][3] * ism4.w + bias4.w + res.w};
          IO::store(make_float4(yv[0], yv[1], yv[2], yv[3]), yr, ((orow0 + row) * C + 16 * w + 4 * q) * ES);
          st1 += yv;
          st2 += yv * yv;
        }
      }
    }
    STK_STAMP(layer, 6);

    // ---- exchange -------------------------------------------------------------------------------------------------------
    bool alive = true;
    if (combiner) {
      // the group's rows, added in index order (members b = grp + n_groups m): slice = m mod 4 per thread, then the 4 slices
      const int col = tid & (2 * C - 1), slice = tid >> 7;
      double acc = 0.0;
      for (int m0 = slice; m0 < gsz && alive; m0 += 32) {
        const u64* p[8];
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int m = m0 + 4 * u;
          p[u] = m < gsz ? A.pgran + ((size_t)layer * G + grp + n_groups * m) * 2 * C + col : nullptr;
        }
        alive = stk_sweep<8>(p, tag, v, status);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (m0 + 4 * u < gsz) ? (double)__uint_as_float(v[u]) : 0.0;
      }
      __syncthreads();                               // (COMB may still be read by the previous layer's publication)
      COMB[slice * 2 * C + col] = acc;
      __syncthreads();
      if (tid < 2 * C) {
        const double t = ((COMB[tid] + COMB[2 * C + tid]) + COMB[4 * C + tid]) + COMB[6 * C + tid];
        const float hi = (float)t, lo = (float)(t - (double)hi);
        u64* g = A.ggran + ((size_t)layer * STK_GROUPS + grp) * 4 * C + 2 * tid;
        __hip_atomic_store(g, ((u64)tag << 32) | __float_as_uint(hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, ((u64)tag << 32) | __float_as_uint(lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      STK_STAMP(layer, 8);
      if (!alive) return;                            // (status raised)
      continue;                                      // a combiner needs no statistics: on to the next layer's rows
    }
    // this workgroup's partial row: the 16 rows j of a DPP row, then the two row halves, fixed order
#pragma unroll
    for (int r = 0; r < 4; ++r) {                    // (DPP row sums on the VALU: no ds_bpermute round trips)
      st1[r] = row16_sum(st1[r]);
      st2[r] = row16_sum(st2[r]);
    }
    __syncthreads();                                 // the images are dead: RED / FIN alias them
    if (tid == 64) FLAG[0] = 0;                      // (wave 1's lane 0, which raises it below; every wave has left the previous wait loop)
    if (j == 0) {
      *reinterpret_cast<f32x4*>(RED + h * 2 * C + 16 * w + 4 * q) = st1;
      *reinterpret_cast<f32x4*>(RED + h * 2 * C + C + 16 * w + 4 * q) = st2;
    }
    __syncthreads();
    if (tid < 2 * C) {                               // 2 waves x 64 granules: whole 512-byte lines per store instruction
      const float sum = RED[tid] + RED[2 * C + tid];
      __hip_atomic_store(A.pgran + ((size_t)layer * G + bid) * 2 * C + tid, ((u64)tag << 32) | __float_as_uint(sum), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    STK_STAMP(layer, 7);
    if (last_layer && bid != 0) break;               // nobody waits for the last layer's statistics: workgroup 0 finalises them
    if (!last_layer) {
      // y of this layer: drained by every storing wave, then the tiles' flags (ONE lane)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        for (int tile = bid; tile < L.ntiles; tile += G)
          __hip_atomic_store(A.yflag + (size_t)layer * A.max_tiles + tile, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    STK_STAMP(layer, 8);
    // the <= 8 group sums (hi + lo): two 16-byte loads per lane (two granules each), each wave instruction one contiguous KiB of a
    // group's 2 KiB, wave = group.  The first pass goes out NOW, in front of the neighbour-flag wait and of the next tile's loads
    // (a wave's loads return in order: behind them every pass would wait for the tile's 48 KB first)
    const int g = tid >> 6;
    const auto gr = stk_rsrc(A.ggran + ((size_t)layer * STK_GROUPS + (g < n_groups ? g : 0)) * 4 * C, 4 * C * 8);
    u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(gr, lane * 16, 0, AUX_SC1);            // granules 2 lane, 2 lane + 1
    u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(gr, 1024 + lane * 16, 0, AUX_SC1);     // granules 128 + 2 lane, ...
    {
      // Two things have to happen before the layer can start, in whatever order they become possible: (a) the next tile's loads go
      // out as soon as the tiles that produced its rows have drained (neighbours: they drained when this workgroup did) -- wave 1
      // polls their flags and raises FLAG[0] in LDS, every wave issues its share when it sees it; (b) the group sums arrive --
      // every wave keeps sweeping its group's KiBs meanwhile.  No barrier between the two: a wave leaves when it has done both.
      const StackLayer& N = A.L[last_layer ? layer : layer + 1];
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      volatile int* nb = FLAG;                       // [0]: 0 = waiting, 1 = neighbours drained, -1 = time-out
      if (!last_layer && wv == 1) {
        bool good = true;
        for (int tile = bid; tile < N.ntiles && good; tile += G) {
          const int s_lo = tile * N.S, s_hi = min(s_lo + N.S, N.n_slabs) - 1;
          const int b_lo = s_lo / N.T_out, b_hi = s_hi / N.T_out;
          const int p_lo = (b_lo * L.T_out + (s_lo - b_lo * N.T_out)) / L.S;
          const int p_hi = min((b_hi * L.T_out + (s_hi - b_hi * N.T_out) + N.d) / L.S, L.ntiles - 1);
          for (int p0 = p_lo; p0 <= p_hi && good; p0 += 64) {
            const unsigned* f = A.yflag + (size_t)layer * A.max_tiles + min(p0 + lane, p_hi);
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
              if (stk_expired(t0, status)) { good = false; break; }
              __builtin_amdgcn_s_sleep(1);
            }
            good = __all(good);
          }
        }
        if (lane == 0) nb[0] = good ? 1 : -1;
      }
      bool swept = false, xdone = last_layer;
#ifdef HOPMI_STAMPS
      int npass = 1;
#endif
      for (;;) {
        if (!swept) {
          swept = g >= n_groups || (v0[1] == tag && v0[3] == tag && v1[1] == tag && v1[3] == tag);          // (value, tag) pairs
          swept = __all(swept);
        }
        if (!xdone) {
          const int f = nb[0];
          if (f < 0) { alive = false; break; }
          if (f > 0) {
            if (bid < N.ntiles) issue_tile(N, bid);
            load_consts(N);
            xdone = true;
          }
        }
        if (swept && xdone) break;
        if (stk_expired(t0, status)) { alive = false; break; }
        __builtin_amdgcn_s_sleep(1);
        if (!swept) {
          v0 = __builtin_amdgcn_raw_buffer_load_b128(gr, lane * 16, 0, AUX_SC1);
          v1 = __builtin_amdgcn_raw_buffer_load_b128(gr, 1024 + lane * 16, 0, AUX_SC1);
#ifdef HOPMI_STAMPS
          ++npass;
#endif
        }
      }
      STK_STAMP(layer, 9);
#ifdef HOPMI_STAMPS
      if (g_stk_stamps && threadIdx.x == 0) g_stk_stamps[(blockIdx.x * STK_MAX_LAYERS + layer) * 16 + 11] = npass;
#endif
      STK_STAMP(layer, 14);
      float* FINF = reinterpret_cast<float*>(FIN);                   // [8 groups][256] floats: value v = (hi, lo) at 2 v, 2 v + 1
      {
        const bool live = g < n_groups;
        float* f = FINF + g * 4 * C;
        *reinterpret_cast<float2*>(f + 2 * lane) = live ? make_float2(__uint_as_float(v0[0]), __uint_as_float(v0[2])) : make_float2(0.f, 0.f);
        *reinterpret_cast<float2*>(f + 2 * C + 2 * lane) = live ? make_float2(__uint_as_float(v1[0]), __uint_as_float(v1[2])) : make_float2(0.f, 0.f);
      }
      if (!alive) FLAG[1] = 1;                       // (benign race: every writer writes 1)
      __syncthreads();
      STK_STAMP(layer, 15);
      if (tid < C) {
        // (all STK_GROUPS slots, unrolled: a dead group's slot holds zeros -- written above by its wave -- and x + 0.0 == x, so the
        // sum is the same bits as over the live groups only; with the run-time bound the loop ran as 8 dependent LDS round trips,
        // ~1 400 cycles on the critical path of every layer: now the 16 reads are in flight together)
        float2 fa[STK_GROUPS], fb[STK_GROUPS];
#pragma unroll
        for (int k = 0; k < STK_GROUPS; ++k) {
          fa[k] = *reinterpret_cast<const float2*>(FINF + k * 4 * C + 2 * tid);
          fb[k] = *reinterpret_cast<const float2*>(FINF + k * 4 * C + 2 * (C + tid));
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < STK_GROUPS; ++k) {
          s1 += (double)fa[k].x + (double)fa[k].y;
          s2 += (double)fb[k].x + (double)fb[k].y;
        }
        // mean and variance in double (E[y^2] - mean^2 cancels); everything behind them in float: 1 / sqrt by the hardware
        // rsq with one Newton step (1e-7 relative)
        const double mean = s1 * L.inv_n;
        double var = s2 * L.inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float varf = (float)var, ve = varf + A.eps;
        float rstd = __builtin_amdgcn_rsqf(ve);
        rstd = rstd * (1.5f - 0.5f * ve * rstd * rstd);
        const float sc = GB[layer * 2 * C + tid] * rstd;
        const float sh = GB[layer * 2 * C + C + tid] - (float)mean * sc;
        const float unbiased = varf * L.unbias;
        SCSH[tid] = sc;
        SCSH[C + tid] = sh;
        if (bid == 0) {                              // one writer of the layer's outputs
          float* so = A.scsh_out + layer * 2 * C;
          float* mr = A.mean_rstd + layer * 3 * C;
          so[tid] = sc;
          so[C + tid] = sh;
          mr[tid] = (float)mean;
          mr[C + tid] = rstd;
          mr[2 * C + tid] = unbiased;
          if (L.rmean != nullptr) {
            L.rmean[tid] = (1.f - A.momentum) * L.rmean[tid] + A.momentum * (float)mean;
            L.rvar[tid] = (1.f - A.momentum) * L.rvar[tid] + A.momentum * unbiased;
          }
        }
      }
      __syncthreads();
      if (FLAG[1]) return;                           // a sweep timed out (status raised)
    }
    STK_STAMP(layer, 10);
    // every workgroup has read the sequence number (it published its last-layer row after): advance it for the next launch
    if (last_layer && tid == 0) __hip_atomic_store(A.sync, (int)(seq + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static size_t stk_lds_bytes(int mt, int KP, int ldA) {
  const size_t rows_lds = 16 * mt + 4;
  return rows_lds * (4 * RS + 2 * HS) * sizeof(__bf16) + (rows_lds * (LDD + 1) + (size_t)KP * ldA + 2 * C) * sizeof(float) + 16 + STK_MAX_LAYERS * sizeof(float) +
         (size_t)STK_MAX_LAYERS * (2 + 3) * C * sizeof(float);
}

struct StackPlan {
  int mt_max, grid, n_comb, n_cu, tiles_max;     // grid = computing workgroups; + n_comb dedicated combiners
  size_t lds;
  int S[STK_MAX_LAYERS], ntiles[STK_MAX_LAYERS], mt[STK_MAX_LAYERS];
};

template <int MT>
static const void* stk_fn(int dtype) {
  return dtype == HOPMI_BF16 ? reinterpret_cast<const void*>(&wn_stack_fwd_kernel<MT, __bf16>)
                             : reinterpret_cast<const void*>(&wn_stack_fwd_kernel<MT, float>);
}

static const void* stk_fn_for(int mt, int dtype) {
  switch (mt) {
    case 1: return stk_fn<1>(dtype);
    case 2: return stk_fn<2>(dtype);
    case 3: return stk_fn<3>(dtype);
    case 4: return stk_fn<4>(dtype);
    case 5: return stk_fn<5>(dtype);
  }
  return nullptr;
}

// Geometry of every layer for a grid of `grid_target` workgroups (the per-layer launches' own rule, wn_dev.h), the largest tile,
// and the resident grid: occupancy (the runtime's answer for this kernel / block / LDS size) x CUs.  Plans are cached per
// geometry: the device / occupancy queries run once, outside any stream capture (a recorded step replays launches only).
static int stk_plan_build(int B, int T_in, int V, const int* dil, int n_layers, int dtype, StackPlan* P) {
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) {
    (void)hipGetLastError();
    set_error("hopmi_wn_stack: no device");
    return HOPMI_EINVAL;
  }
  // STK_GROUPS CUs are kept for the dedicated combiner workgroups
  const int avail = n_cu > 2 * STK_GROUPS ? n_cu - STK_GROUPS : (n_cu > 1 ? n_cu - 1 : 1);
  const int target = wn_env_int("HOPMI_WN_GRID", avail) < avail ? wn_env_int("HOPMI_WN_GRID", avail) : avail;
  int T = T_in, mt_max = 1, tiles_max = 1;
  for (int l = 0; l < n_layers; ++l) {
    if (int e = wn_validate(B, T, V, dil[l])) return e;
    if ((long long)B * T * V * 256 >= (1LL << 31)) { set_error("hopmi_wn_stack: layer tensor beyond 2 GiB (32-bit buffer offsets)"); return HOPMI_EINVAL; }
    const LayerGeom L = make_layer_geom(B, T, V, dil[l], target, WN_MAX_MT);
    P->S[l] = L.g.S; P->ntiles[l] = L.g.ntiles; P->mt[l] = L.g.mtiles;
    if (L.g.mtiles > mt_max) mt_max = L.g.mtiles;
    if (L.g.ntiles > tiles_max) tiles_max = L.g.ntiles;
    T -= dil[l];
  }
  const GcnGeom g = make_geom(1, V, 1);
  P->mt_max = mt_max;
  P->tiles_max = tiles_max;
  P->lds = stk_lds_bytes(mt_max, g.KP, g.ldA);
  P->n_cu = n_cu;
  if (P->lds > 160 * 1024) { set_error("hopmi_wn_stack: tile needs %zu bytes of LDS", P->lds); return HOPMI_EINVAL; }
  const void* fn = stk_fn_for(mt_max, dtype);
  if (fn == nullptr) { set_error("hopmi_wn_stack: internal: %d m-tiles", mt_max); return HOPMI_EINVAL; }
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, STK_THREADS, P->lds) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    set_error("hopmi_wn_stack: the kernel does not fit a CU (occupancy query)");
    return HOPMI_EINVAL;
  }
  // one workgroup per CU even where two would fit: the per-layer tiles are sized for n_cu workgroups, and a margin of
  // residency is worth more than the second workgroup (MI355X_MICROARCH.md: the query can read one high)
  P->grid = tiles_max < target ? tiles_max : target;
  P->n_comb = P->grid < STK_GROUPS ? P->grid : STK_GROUPS;
  if (P->grid + P->n_comb > n_cu) { set_error("hopmi_wn_stack: %d workgroups do not fit %d CUs", P->grid + P->n_comb, n_cu); return HOPMI_EINVAL; }
  return HOPMI_OK;
}

struct StackPlanKey { int B, T_in, V, n_layers, dtype, dil[STK_MAX_LAYERS]; };
static std::mutex g_plan_mu;
static std::vector<std::pair<StackPlanKey, StackPlan>> g_plans;

static int stk_plan(int B, int T_in, int V, const int* dil, int n_layers, int dtype, StackPlan* P) {
  if (n_layers < 1 || n_layers > STK_MAX_LAYERS || dil == nullptr) { set_error("hopmi_wn_stack: 1..%d layers", STK_MAX_LAYERS); return HOPMI_EINVAL; }
  if (dtype != HOPMI_F32 && dtype != HOPMI_BF16) { set_error("hopmi_wn_stack: dtype %d (0 = fp32, 1 = bf16)", dtype); return HOPMI_EINVAL; }
  StackPlanKey k{};
  k.B = B; k.T_in = T_in; k.V = V; k.n_layers = n_layers; k.dtype = dtype;
  for (int l = 0; l < n_layers; ++l) k.dil[l] = dil[l];
  std::lock_guard<std::mutex> lk(g_plan_mu);
  for (const auto& e : g_plans)
    if (memcmp(&e.first, &k, sizeof(k)) == 0) { *P = e.second; return HOPMI_OK; }
  if (int e = stk_plan_build(B, T_in, V, dil, n_layers, dtype, P)) return e;
  g_plans.emplace_back(k, *P);
  return HOPMI_OK;
}

hipEvent_t wn_take_timing_events(hipEvent_t* stop);     // wavenet.hip (hopmi_time_next_launch)

}  // namespace hopmi

using namespace hopmi;

#ifdef HOPMI_STAMPS
extern "C" int hopmi_debug_set_stamps_stack(long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stk_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int hopmi_wn_stack_grid(int B, int T_in, int V, const int* dilations, int n_layers) {
  StackPlan P;
  if (stk_plan(B, T_in, V, dilations, n_layers, HOPMI_F32, &P)) return 0;
  return P.grid + P.n_comb;
}

extern "C" size_t hopmi_wn_stack_ws_bytes(int B, int T_in, int V, const int* dilations, int n_layers) {
  StackPlan P;
  if (stk_plan(B, T_in, V, dilations, n_layers, HOPMI_F32, &P)) return 0;
  return (size_t)STK_SYNC_INTS * sizeof(int) + ((size_t)n_layers * P.grid * 2 * C + (size_t)n_layers * STK_GROUPS * 4 * C) * sizeof(u64) +
         (size_t)n_layers * P.tiles_max * sizeof(unsigned);
}

extern "C" int hopmi_wn_stack_fwd_dt(const void* x0, const void* wimg, const float* const* bf, const float* const* bg, const float* prep,
                                     const float* const* bm, const float* const* gamma, const float* const* beta,
                                     float* const* running_mean, float* const* running_var, float momentum, float eps, void* const* y,
                                     void* utail, int utail_ld, float* scsh_out, float* mean_rstd_out, void* ws, int B, int T_in, int V,
                                     const int* dilations, int n_layers, int dtype, void* stream) {
  StackPlan P;
  if (int e = stk_plan(B, T_in, V, dilations, n_layers, dtype, &P)) return e;
  if (!x0 || !wimg || !bf || !bg || !prep || !bm || !gamma || !beta || !y || !utail || !scsh_out || !mean_rstd_out || !ws) {
    set_error("hopmi_wn_stack_fwd: null pointer argument");
    return HOPMI_EINVAL;
  }
  if (utail_ld < C * n_layers || (utail_ld & 3)) { set_error("hopmi_wn_stack_fwd: utail_ld=%d must be a multiple of 4 and >= 64 * n_layers", utail_ld); return HOPMI_EINVAL; }
  StackArgs A{};
  int T = T_in;
  for (int l = 0; l < n_layers; ++l) {
    StackLayer& L = A.L[l];
    if (!bf[l] || !bg[l] || !bm[l] || !gamma[l] || !beta[l] || (l < n_layers - 1 && !y[l])) {
      set_error("hopmi_wn_stack_fwd: null pointer for layer %d", l);
      return HOPMI_EINVAL;
    }
    L.xin = l == 0 ? x0 : y[l - 1];
    L.y = l < n_layers - 1 ? y[l] : nullptr;
    L.bf = bf[l]; L.bg = bg[l]; L.bm = bm[l]; L.gamma = gamma[l]; L.beta = beta[l];
    L.rmean = running_mean ? running_mean[l] : nullptr;
    L.rvar = running_var ? running_var[l] : nullptr;
    if ((L.rmean == nullptr) != (L.rvar == nullptr)) { set_error("hopmi_wn_stack_fwd: running_mean / running_var of layer %d: both or none", l); return HOPMI_EINVAL; }
    L.T_in = T; L.d = dilations[l]; L.T_out = T - dilations[l];
    L.n_slabs = B * L.T_out; L.S = P.S[l]; L.ntiles = P.ntiles[l];
    L.invT = 1.0f / L.T_out;
    {
      const double n = (double)L.n_slabs * V;
      L.inv_n = 1.0 / n;
      L.unbias = n > 1.0 ? (float)(n / (n - 1.0)) : 1.0f;
    }
    T = L.T_out;
  }
  const GcnGeom g = make_geom(1, V, 1);
  A.wimg = static_cast<const u32x4*>(wimg);
  A.prep = prep;
  A.utail = utail;
  A.scsh_out = scsh_out;
  A.mean_rstd = mean_rstd_out;
  A.sync = static_cast<int*>(ws);
  A.pgran = reinterpret_cast<u64*>(A.sync + STK_SYNC_INTS);
  A.ggran = A.pgran + (size_t)n_layers * P.grid * 2 * C;
  A.yflag = reinterpret_cast<unsigned*>(A.ggran + (size_t)n_layers * STK_GROUPS * 4 * C);
  A.max_tiles = P.tiles_max;
  A.n_layers = n_layers; A.B = B; A.V = V; A.utail_ld4 = utail_ld / 4; A.n_comb = P.n_comb;
  A.KP = g.KP; A.ldA = g.ldA; A.MP = g.MP;
  A.prep_tr = g.KP * g.ldA + g.K2P * g.ldB;
  A.invV = 1.0f / V; A.momentum = momentum; A.eps = eps;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e1 = nullptr;
  const hipEvent_t e0 = wn_take_timing_events(&e1);
  const bool bf16 = dtype == HOPMI_BF16;
  switch (P.mt_max) {
#define HOPMI_STK_CASE(MT_)                                                                                                          \
    case MT_:                                                                                                                        \
      if (bf16) hipExtLaunchKernelGGL((wn_stack_fwd_kernel<MT_, __bf16>), dim3(P.grid + P.n_comb), dim3(STK_THREADS), P.lds, st, e0, e1, 0, A);   \
      else hipExtLaunchKernelGGL((wn_stack_fwd_kernel<MT_, float>), dim3(P.grid + P.n_comb), dim3(STK_THREADS), P.lds, st, e0, e1, 0, A);       \
      break;
    HOPMI_STK_CASE(1) HOPMI_STK_CASE(2) HOPMI_STK_CASE(3) HOPMI_STK_CASE(4) HOPMI_STK_CASE(5)
#undef HOPMI_STK_CASE
    default: set_error("hopmi_wn_stack_fwd: internal: %d m-tiles", P.mt_max); return HOPMI_EINVAL;
  }
  return check_launch("hopmi_wn_stack_fwd");
}

extern "C" int hopmi_wn_stack_fwd(const float* x0, const void* wimg, const float* const* bf, const float* const* bg, const float* prep,
                                  const float* const* bm, const float* const* gamma, const float* const* beta,
                                  float* const* running_mean, float* const* running_var, float momentum, float eps, float* const* y,
                                  float* utail, int utail_ld, float* scsh_out, float* mean_rstd_out, void* ws, int B, int T_in, int V,
                                  const int* dilations, int n_layers, void* stream) {
  return hopmi_wn_stack_fwd_dt(x0, wimg, bf, bg, prep, bm, gamma, beta, running_mean, running_var, momentum, eps,
                               reinterpret_cast<void* const*>(y), utail, utail_ld, scsh_out, mean_rstd_out, ws, B, T_in, V, dilations,
                               n_layers, HOPMI_F32, stream);
}

HOPMI_SPLIT_STATUS_SETTER(wavenet_stack)
