// Weight gradients on the fp16 matrix cores: C[N][K] = A[M][N]^T . B[M][K]  ("TN": both operands are ACTIVATIONS, row-major, and the
// contraction runs over their rows) -- dW = dY^T X of every nn.Linear / nn.GRU weight behind the reference's HOP.py:116-134,166-167,
// which rounds 1-4 left to the library's fp32 GEMM (1.5 ms of a 15.5 ms step).  Same arithmetic as hopmi_gemm_f16x2 (gemm.hip,
// f16_dev.h): every operand as two power-of-two-scaled fp16 parts, three MFMA terms, fp32 accumulation -- fp32-equivalent.
//
// What is different from the NT form: the MFMA wants 8 CONSECUTIVE contraction indices per lane, and here those are 8 consecutive
// ROWS of a row-major matrix.  The transposition happens on the way into LDS: a thread owns one column of the tile and eight
// consecutive rows -- eight 4-byte loads, each wave instruction one contiguous 256-byte piece of a row -- scales, splits and
// writes them as ONE 16-byte ds_write per part into the transposed images At[n][m], Bt[k][m] (rows of 32 m = 64 bytes, row stride
// 80 bytes: both the column-major stores and the fragment reads hit 16 distinct 16-byte slots per service group).
// Scales: the contraction index is the row, so rows cannot carry scales of their own; each operand gets ONE power of two, the
// smallest of its per-row scales (hopmi_row_scales / the producers' fused scales: the consumers of the same tensors as A operands of
// the NT form have them already), found by every workgroup in its prologue (M floats per operand from L2).
// Tiles 128 (n) x 128 (k), 8 waves of 64 x 32, m-steps of 32 (64-row steps -- two MFMA k-blocks per barrier pair, 74 KB of LDS -- are 3-9 %
// faster back to back and were dropped: the GAN-phase step went from 20.5 to 21.4 ms with them, reproducibly, with multi-millisecond
// gaps inside graph replays -- every kernel of the step a few per cent slower; the larger LDS footprint next to the persistent
// kernels of that phase is the suspect); the m range is SPLIT over workgroups when the output has too few
// tiles for the chip (a GRU's 1050 x 350 recurrent gradient: 27 tiles): split s writes slab s of a workspace, a second launch adds
// the slabs in index order -- bitwise reproducible, no atomics.
#define HOPMI_FILE_ID 2          // (diagnostic build: common.h, split_check)
#include "f16_dev.h"

namespace hopmi {

constexpr int TN_T = 128;          // output tile (both ways)
constexpr int TN_MS = 32;          // rows (contraction) per step
constexpr int TN_LD = 40;          // LDS row stride in halves (80 bytes)
constexpr int TN_THREADS = 512;

struct TnArgs {
  const float* A; const float* B; float* C; float* slabs;
  const float* a_rows; const float* b_rows;      // [2][M] row-scale pairs {s, 1 / s} of the operands (only s is read)
  long long batchA, batchB, batchC, batchS;      // element strides between batch members (grid.y)
  int lda, ldb, ldc;
  int M, N, K;
  int tiles_n, tiles_k, splits, steps_per_split;
  int accumulate;                                // C += (single-split form only; the slab sum takes it otherwise)
  float* colsum;                                 // nullable: [N] column sums of A (the bias gradient of the same linear), batch == 1
  float* cs_slabs;                               // ... [splits][N] partial sums when the rows are split
};

__device__ __forceinline__ float tn_tensor_scale(const float* rows, int M, float* red) {
  // the smallest per-row scale (rows of zeros carry the largest one: common.h scale_bits_for_max); 1 without scales
  const int tid = threadIdx.x;
  float mn = 3.0e38f;
  if (rows != nullptr)
    for (int m = tid; m < M; m += TN_THREADS) mn = fminf(mn, rows[m]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = mn;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < TN_THREADS / 64; ++w) r = fminf(r, red[w]);
  return rows != nullptr ? r : 1.f;
}

__global__ __launch_bounds__(TN_THREADS, 2) void gemm_f16_tn_kernel(TnArgs P) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * 2 * TN_T * TN_LD];      // [operand][part][128 rows][TN_LD]
  __shared__ float red[TN_THREADS / 64];
  int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wv >> 2, wc = wv & 3;
  const int lane = tid & 63, q = lane >> 4, i16 = lane & 15;
  const int b = blockIdx.x;
  const int tk = b % P.tiles_k, tn = (b / P.tiles_k) % P.tiles_n, split = b / (P.tiles_k * P.tiles_n);
  const int n0 = tn * TN_T, k0 = tk * TN_T;
  const float* A = P.A + (size_t)blockIdx.y * P.batchA;
  const float* B = P.B + (size_t)blockIdx.y * P.batchB;

  const float sA = tn_tensor_scale(P.a_rows, P.M, red);
  const float sB = tn_tensor_scale(P.b_rows, P.M, red);

  // staging map: column c of the tile, rows 8 o .. 8 o + 7 of the step.  Buffer loads: a row past M is past the resource's extent and
  // reads as zero (it enters the contraction: it must), a column past N / K reads whatever lies there -- it only feeds output rows /
  // columns that are not stored -- and nothing is guarded (hipcc turns a guarded load into a branch with a full wait behind it).
  // The per-thread byte offset is loop-invariant; the row of the step rides on the scalar offset.
  const int c = tid & 127, o = tid >> 7;
  const auto ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)(((size_t)(P.M - 1) * P.lda + P.N) * 4), 0x00020000);
  const auto br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (unsigned)(((size_t)(P.M - 1) * P.ldb + P.K) * 4), 0x00020000);
  const int step0 = split * P.steps_per_split;
  const int nsteps = max(0, min(P.steps_per_split, (P.M + TN_MS - 1) / TN_MS - step0));
  const unsigned a_voff = (unsigned)(((step0 * TN_MS + 8 * o) * P.lda + n0 + c) * 4);
  const unsigned b_voff = (unsigned)(((step0 * TN_MS + 8 * o) * P.ldb + k0 + c) * 4);
  _Float16* At = lds;
  _Float16* Bt = lds + 2 * TN_T * TN_LD;
  const int st_off = c * TN_LD + 8 * o;

  // One staging register set, issued one step ahead.  (Measured and dropped: a second set issued two steps ahead with counted
  // vmcnt waits -- a lone workgroup stays at 1.3 us per step either way: like the NT form the step is the MFMA phase (~1 900 cycles
  // for 192 MFMAs) plus the split / commit phase (~1 300) between two barriers, not the loads' latency; what overlaps the two is the
  // CU's second workgroup.)
  float av[8], bv[8];
  auto issue = [&](int step) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      av[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ar, a_voff, (step * TN_MS + e) * P.lda * 4, 0));
      bv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, b_voff, (step * TN_MS + e) * P.ldb * 4, 0));
    }
  };
  // (the bias gradient of the same linear is the column sum of A = dY: the workgroups of the first k-tile column add up what they
  // stage anyway -- thread (c, o) owns column c, rows 8 o .. 8 o + 7 of every step)
  const bool want_cs = P.colsum != nullptr && tk == 0;
  float asum = 0.f;
  auto commit = [&]() {
    if (want_cs) asum += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
    const Split8 sa = split8h(make_float4(av[0], av[1], av[2], av[3]), make_float4(av[4], av[5], av[6], av[7]), sA);
    const Split8 sb = split8h(make_float4(bv[0], bv[1], bv[2], bv[3]), make_float4(bv[4], bv[5], bv[6], bv[7]), sB);
    *reinterpret_cast<u32x4*>(At + st_off) = sa.hi;
    *reinterpret_cast<u32x4*>(At + TN_T * TN_LD + st_off) = sa.lo;
    *reinterpret_cast<u32x4*>(Bt + st_off) = sb.hi;
    *reinterpret_cast<u32x4*>(Bt + TN_T * TN_LD + st_off) = sb.lo;
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = {0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
    issue(0);
    commit();
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("" : "+v"(tid));                      // (keeps the loop-invariant LDS addresses from being hoisted into registers)
    const int ln = tid & 63, qq = ln >> 4, nn = ln & 15;
    const bool more = s + 1 < nsteps;
    if (more) issue(s + 1);
    const _Float16* la = At + (64 * wr + nn) * TN_LD + 8 * qq;
    const _Float16* lb = Bt + (32 * wc + nn) * TN_LD + 8 * qq;
    u32x4 bh[2], bl[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      bh[ni] = *reinterpret_cast<const u32x4*>(lb + 16 * ni * TN_LD);
      bl[ni] = *reinterpret_cast<const u32x4*>(lb + TN_T * TN_LD + 16 * ni * TN_LD);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const u32x4 ah = *reinterpret_cast<const u32x4*>(la + 16 * mi * TN_LD);
      const u32x4 al = *reinterpret_cast<const u32x4*>(la + TN_T * TN_LD + 16 * mi * TN_LD);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_h3(ah, al, bh[ni], bl[ni], acc[mi][ni]);
    }
    __syncthreads();                                   // every wave has read this step's fragments
    if (more) commit();
    __syncthreads();
  }

  // epilogue: lane (q, i16) holds rows 4 q + r (n), column i16 (k) of every 16 x 16 tile
  const float inv = inv_pow2(sA) * inv_pow2(sB);
  float* out;
  int ldo;
  if (P.splits > 1) {
    out = P.slabs + (size_t)blockIdx.y * P.batchS + (size_t)split * P.N * P.K;
    ldo = P.K;
  } else {
    out = P.C + (size_t)blockIdx.y * P.batchC;
    ldo = P.ldc;
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = k0 + 32 * wc + 16 * ni + i16;
      if (col >= P.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + 64 * wr + 16 * mi + 4 * q + r;
        if (row < P.N) {
          float* p = out + (size_t)row * ldo + col;
          const float v = acc[mi][ni][r] * inv;
          *p = (P.splits == 1 && P.accumulate) ? *p + v : v;
        }
      }
    }
  if (want_cs) {                                       // (wave-uniform; the loop's last barrier is behind every LDS read)
    float* r = reinterpret_cast<float*>(lds);
    r[o * TN_T + c] = asum;
    __syncthreads();
    if (tid < TN_T && n0 + tid < P.N) {
      const float v = (r[tid] + r[TN_T + tid]) + (r[2 * TN_T + tid] + r[3 * TN_T + tid]);
      if (P.splits > 1) P.cs_slabs[(size_t)split * P.N + n0 + tid] = v;
      else P.colsum[n0 + tid] = v;
    }
  }
}

// Round 6: the same kernel with TWO tile buffers and the commit of step s + 1 (scale, split, transposing LDS stores: ~100 vector
// instructions per thread) issued BETWEEN the MFMA groups of step s instead of behind a barrier of its own -- one barrier per step,
// and the matrix pipe runs while the SIMD's vector issue slots (an MFMA holds them for 8 of its 16 cycles) take the split.  The LDS
// rows shrink to their 64 data bytes (no pad; the 16-byte slot XOR-swizzled with the row, gemm.hip's gswz: column-major stores and
// fragment reads both conflict-free), so two buffers are 64 KB and two workgroups per CU still fit.  Same terms, same order per
// output element: bit-identical to the single-buffer form (tools/bench_linear.py --tn checks it).
__device__ __forceinline__ int tn_swz(int row) { return (0x1230 >> (4 * ((row >> 2) & 3))) & 3; }

__global__ __launch_bounds__(TN_THREADS, 2) void gemm_f16_tn_db_kernel(TnArgs P) {
  constexpr int LDR = 32;                                                           // halves per LDS row (64 bytes, swizzled)
  constexpr int BUF = 2 * 2 * TN_T * LDR;                                            // halves per tile buffer: [operand][part][128 rows][32]
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * BUF];
  __shared__ float red[TN_THREADS / 64];
  int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wv >> 2, wc = wv & 3;
  const int lane = tid & 63, q = lane >> 4, i16 = lane & 15;
  const int b = blockIdx.x;
  const int tk = b % P.tiles_k, tn = (b / P.tiles_k) % P.tiles_n, split = b / (P.tiles_k * P.tiles_n);
  const int n0 = tn * TN_T, k0 = tk * TN_T;
  const float* A = P.A + (size_t)blockIdx.y * P.batchA;
  const float* B = P.B + (size_t)blockIdx.y * P.batchB;

  const float sA = tn_tensor_scale(P.a_rows, P.M, red);
  const float sB = tn_tensor_scale(P.b_rows, P.M, red);

  // staging map: column c of the tile, rows 8 o .. 8 o + 7 of the step.  Buffer loads: a row past M is past the resource's extent and
  // reads as zero (it enters the contraction: it must), a column past N / K reads whatever lies there -- it only feeds output rows /
  // columns that are not stored -- and nothing is guarded (hipcc turns a guarded load into a branch with a full wait behind it).
  // The per-thread byte offset is loop-invariant; the row of the step rides on the scalar offset.
  const int c = tid & 127, o = tid >> 7;
  const auto ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)(((size_t)(P.M - 1) * P.lda + P.N) * 4), 0x00020000);
  const auto br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (unsigned)(((size_t)(P.M - 1) * P.ldb + P.K) * 4), 0x00020000);
  const int step0 = split * P.steps_per_split;
  const int nsteps = max(0, min(P.steps_per_split, (P.M + TN_MS - 1) / TN_MS - step0));
  const unsigned a_voff = (unsigned)(((step0 * TN_MS + 8 * o) * P.lda + n0 + c) * 4);
  const unsigned b_voff = (unsigned)(((step0 * TN_MS + 8 * o) * P.ldb + k0 + c) * 4);
  const int st_off = c * LDR + ((o ^ tn_swz(c)) << 3);

  // One staging register set, issued one step ahead.  (Measured and dropped: a second set issued two steps ahead with counted
  // vmcnt waits -- a lone workgroup stays at 1.3 us per step either way: like the NT form the step is the MFMA phase (~1 900 cycles
  // for 192 MFMAs) plus the split / commit phase (~1 300) between two barriers, not the loads' latency; what overlaps the two is the
  // CU's second workgroup.)
  float av[8], bv[8];
  auto issue = [&](int step) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      av[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ar, a_voff, (step * TN_MS + e) * P.lda * 4, 0));
      bv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, b_voff, (step * TN_MS + e) * P.ldb * 4, 0));
    }
  };
  // (the bias gradient of the same linear is the column sum of A = dY: the workgroups of the first k-tile column add up what they
  // stage anyway -- thread (c, o) owns column c, rows 8 o .. 8 o + 7 of every step)
  const bool want_cs = P.colsum != nullptr && tk == 0;
  float asum = 0.f;
  auto commit_a = [&](_Float16* buf) {
    if (want_cs) asum += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
    const Split8 sa = split8h(make_float4(av[0], av[1], av[2], av[3]), make_float4(av[4], av[5], av[6], av[7]), sA);
    *reinterpret_cast<u32x4*>(buf + st_off) = sa.hi;
    *reinterpret_cast<u32x4*>(buf + TN_T * LDR + st_off) = sa.lo;
  };
  auto commit_b = [&](_Float16* buf) {
    const Split8 sb = split8h(make_float4(bv[0], bv[1], bv[2], bv[3]), make_float4(bv[4], bv[5], bv[6], bv[7]), sB);
    *reinterpret_cast<u32x4*>(buf + 2 * TN_T * LDR + st_off) = sb.hi;
    *reinterpret_cast<u32x4*>(buf + 3 * TN_T * LDR + st_off) = sb.lo;
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = {0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
    issue(0);
    commit_a(lds);
    commit_b(lds);
    if (nsteps > 1) issue(1);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("" : "+v"(tid));                      // (keeps the loop-invariant LDS addresses from being hoisted into registers)
    const int ln = tid & 63, qq = ln >> 4, nn = ln & 15;
    const bool more = s + 1 < nsteps;
    const _Float16* cur = lds + (s & 1) * BUF;
    _Float16* nxt = lds + ((s + 1) & 1) * BUF;
    const int fsl = (qq ^ tn_swz(nn)) << 3;            // (tile rows are 16 mi + nn: the swizzle only depends on nn)
    const _Float16* la = cur + (64 * wr + nn) * LDR + fsl;
    const _Float16* lb = cur + 2 * TN_T * LDR + (32 * wc + nn) * LDR + fsl;
    u32x4 bh[2], bl[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      bh[ni] = *reinterpret_cast<const u32x4*>(lb + 16 * ni * LDR);
      bl[ni] = *reinterpret_cast<const u32x4*>(lb + TN_T * LDR + 16 * ni * LDR);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const u32x4 ah = *reinterpret_cast<const u32x4*>(la + 16 * mi * LDR);
      const u32x4 al = *reinterpret_cast<const u32x4*>(la + TN_T * LDR + 16 * mi * LDR);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_h3(ah, al, bh[ni], bl[ni], acc[mi][ni]);
      // the next step's tile (its raw values arrived during the previous step) is split and stored into the OTHER buffer while
      // this step's MFMAs run: A behind the first MFMA group, B behind the third
      if (more && mi == 0) { __builtin_amdgcn_sched_barrier(0); commit_a(nxt); __builtin_amdgcn_sched_barrier(0); }
      if (more && mi == 2) { __builtin_amdgcn_sched_barrier(0); commit_b(nxt); __builtin_amdgcn_sched_barrier(0); }
    }
    if (s + 2 < nsteps) issue(s + 2);
    __syncthreads();                                   // the other buffer is complete; every wave is done reading this one
  }

  // epilogue: lane (q, i16) holds rows 4 q + r (n), column i16 (k) of every 16 x 16 tile
  const float inv = inv_pow2(sA) * inv_pow2(sB);
  float* out;
  int ldo;
  if (P.splits > 1) {
    out = P.slabs + (size_t)blockIdx.y * P.batchS + (size_t)split * P.N * P.K;
    ldo = P.K;
  } else {
    out = P.C + (size_t)blockIdx.y * P.batchC;
    ldo = P.ldc;
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = k0 + 32 * wc + 16 * ni + i16;
      if (col >= P.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + 64 * wr + 16 * mi + 4 * q + r;
        if (row < P.N) {
          float* p = out + (size_t)row * ldo + col;
          const float v = acc[mi][ni][r] * inv;
          *p = (P.splits == 1 && P.accumulate) ? *p + v : v;
        }
      }
    }
  if (want_cs) {                                       // (wave-uniform; the loop's last barrier is behind every LDS read)
    float* r = reinterpret_cast<float*>(lds);
    r[o * TN_T + c] = asum;
    __syncthreads();
    if (tid < TN_T && n0 + tid < P.N) {
      const float v = (r[tid] + r[TN_T + tid]) + (r[2 * TN_T + tid] + r[3 * TN_T + tid]);
      if (P.splits > 1) P.cs_slabs[(size_t)split * P.N + n0 + tid] = v;
      else P.colsum[n0 + tid] = v;
    }
  }
}

// C[n][k] (+)= slab 0 + slab 1 + ... (index order)
__global__ __launch_bounds__(256) void gemm_tn_sum_kernel(const float* __restrict__ slabs, int splits, long long batchS, float* __restrict__ C,
                                                          int ldc, long long batchC, int N, int K, int accumulate,
                                                          const float* __restrict__ cs_slabs, float* __restrict__ colsum) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (colsum != nullptr && idx < (size_t)N && blockIdx.y == 0) {       // the column sums of A ride along (index order too)
    float v = cs_slabs[idx];
    for (int p = 1; p < splits; ++p) v += cs_slabs[(size_t)p * N + idx];
    colsum[idx] = v;
  }
  if (idx >= (size_t)N * K) return;
  const float* s = slabs + (size_t)blockIdx.y * batchS + idx;
  float v = s[0];
  for (int p = 1; p < splits; ++p) v += s[(size_t)p * N * K];
  const size_t n = idx / K, k = idx - n * K;
  float* c = C + (size_t)blockIdx.y * batchC + n * ldc + k;
  *c = accumulate ? *c + v : v;
}

static int tn_splits(int M, int N, int K, int batch) {
  const int tiles = ((N + TN_T - 1) / TN_T) * ((K + TN_T - 1) / TN_T) * batch;
  const int steps = (M + TN_MS - 1) / TN_MS;
  int cus = 256;
  const int forced = env_int("HOPMI_GEMM_TN_SPLITS", 0);
  if (forced > 0) return forced < steps ? forced : steps;
  // fill the chip about twice (two workgroups per CU), but keep at least 8 steps per split
  int s = (2 * cus + tiles - 1) / tiles;
  if (s > steps / 8) s = steps / 8;
  if (s < 1) s = 1;
  if (s > 32) s = 32;
  return s;
}

}  // namespace hopmi

using namespace hopmi;

extern "C" size_t hopmi_gemm_f16x2_tn_ws_floats(int M, int N, int K, int batch) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
  const int s = tn_splits(M, N, K, batch);
  return s > 1 ? (size_t)s * N * K * batch + (size_t)s * N : 0;        // (+ the column-sum slabs of hopmi_gemm_f16x2_tn_cs)
}

extern "C" int hopmi_gemm_f16x2_tn_cs(const float* A, int lda, long long batch_stride_a, const float* a_rows, const float* B, int ldb,
                                      long long batch_stride_b, const float* b_rows, float* C, int ldc, long long batch_stride_c, float* ws,
                                      int M, int N, int K, int batch, int accumulate, float* a_colsum, void* stream);

extern "C" int hopmi_gemm_f16x2_tn(const float* A, int lda, long long batch_stride_a, const float* a_rows, const float* B, int ldb,
                                   long long batch_stride_b, const float* b_rows, float* C, int ldc, long long batch_stride_c, float* ws,
                                   int M, int N, int K, int batch, int accumulate, void* stream) {
  return hopmi_gemm_f16x2_tn_cs(A, lda, batch_stride_a, a_rows, B, ldb, batch_stride_b, b_rows, C, ldc, batch_stride_c, ws, M, N, K, batch,
                                accumulate, nullptr, stream);
}

extern "C" int hopmi_gemm_f16x2_tn_cs(const float* A, int lda, long long batch_stride_a, const float* a_rows, const float* B, int ldb,
                                      long long batch_stride_b, const float* b_rows, float* C, int ldc, long long batch_stride_c, float* ws,
                                      int M, int N, int K, int batch, int accumulate, float* a_colsum, void* stream) {
  if (a_colsum && batch != 1) { set_error("hopmi_gemm_f16x2_tn_cs: the column sums are for batch == 1"); return HOPMI_EINVAL; }
  if (!A || !B || !C || !a_rows || !b_rows) { set_error("hopmi_gemm_f16x2_tn: null pointer argument (operands, their row scales, C)"); return HOPMI_EINVAL; }
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || lda < N || ldb < K || ldc < K) {
    set_error("hopmi_gemm_f16x2_tn: bad extents M=%d N=%d K=%d batch=%d lda=%d ldb=%d ldc=%d", M, N, K, batch, lda, ldb, ldc);
    return HOPMI_EINVAL;
  }
  if (((long long)M + 2 * TN_MS) * (lda > ldb ? lda : ldb) * 4 + 4 * TN_T * 4 >= (1LL << 31)) {
    set_error("hopmi_gemm_f16x2_tn: operand beyond 2 GiB (32-bit buffer offsets)");
    return HOPMI_EINVAL;
  }
  TnArgs P{};
  P.A = A; P.B = B; P.C = C; P.a_rows = a_rows; P.b_rows = b_rows;
  P.batchA = batch_stride_a; P.batchB = batch_stride_b; P.batchC = batch_stride_c;
  P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.M = M; P.N = N; P.K = K;
  P.tiles_n = (N + TN_T - 1) / TN_T; P.tiles_k = (K + TN_T - 1) / TN_T;
  P.splits = tn_splits(M, N, K, batch);
  const int steps = (M + TN_MS - 1) / TN_MS;
  P.steps_per_split = (steps + P.splits - 1) / P.splits;
  P.accumulate = accumulate;
  if (P.splits > 1) {
    if (!ws) { set_error("hopmi_gemm_f16x2_tn: this shape splits its rows %d ways and needs the workspace (hopmi_gemm_f16x2_tn_ws_floats)", P.splits); return HOPMI_EINVAL; }
    P.slabs = ws;
    P.batchS = (long long)P.splits * N * K;
    P.cs_slabs = ws + (size_t)P.splits * N * K * batch;
  }
  P.colsum = a_colsum;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // HOPMI_GEMM_TN_DB: 1 = the double-buffered form (round 6), 0 = the single-buffer form
  // (also built and measured in round 6, then removed: the same form with 16-byte loads -- threads 0-255 staging A, 256-511 B, four
  // float4 loads and eight 8-byte LDS stores per thread and step instead of sixteen 4-byte loads and four 16-byte stores --
  // bit-identical products, 128.0 against 127.1 us at the GRU shape, nothing in the step: the vector-memory instruction count is
  // not what bounds this kernel; profiles/r06_bench_tn.txt)
  if (env_int("HOPMI_GEMM_TN_DB", 1) >= 1)
    hipLaunchKernelGGL(gemm_f16_tn_db_kernel, dim3(P.tiles_n * P.tiles_k * P.splits, batch), dim3(TN_THREADS), 0, st, P);
  else
    hipLaunchKernelGGL(gemm_f16_tn_kernel, dim3(P.tiles_n * P.tiles_k * P.splits, batch), dim3(TN_THREADS), 0, st, P);
  if (int e = check_launch("hopmi_gemm_f16x2_tn")) return e;
  if (P.splits > 1) {
    hipLaunchKernelGGL(gemm_tn_sum_kernel, dim3((unsigned)(((size_t)N * K + 255) / 256), batch), dim3(256), 0, st, P.slabs, P.splits, P.batchS,
                       C, ldc, batch_stride_c, N, K, accumulate, P.cs_slabs, a_colsum);
    return check_launch("hopmi_gemm_f16x2_tn(sum)");
  }
  return HOPMI_OK;
}

HOPMI_SPLIT_STATUS_SETTER(gemm_tn)
