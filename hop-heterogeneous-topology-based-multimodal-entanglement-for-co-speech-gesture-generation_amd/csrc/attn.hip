// Audio->text "reprogramming" cross-attention of HOP (reference: model/HOP.py:289-299):
//
//   scores = einsum("blhe,she->bhls", Q, K);  A = dropout(softmax(scores / sqrt(E)));  O = einsum("bhls,she->blhe", A, V)
//
// with L = 34 audio frames, S = 1500 text prototypes shared by the whole batch, H = 8 heads, E = 128.
// The reference materialises the (B,8,34,1500) score tensor (209 MB at B = 128) three times per forward;
// here it never leaves the chip: flash-style online softmax, exact-fp32 MFMA (16x16x4) for both contractions.
//
// Rows (b, l) are flat: N = B*L query rows per head.  Workgroup = (64-row tile, head); wave w owns 16 rows
// entirely (its softmax statistics never cross waves) and keeps their Q fragments in registers.  The 64-key
// K and V chunks are streamed HBM/L2 -> registers -> LDS (register prefetch of the next chunk under the
// current chunk's MFMAs).  Linear workgroup id = tile*H + head, so with H = 8 every head lives on one XCD
// and its K/V (1.5 MB) stay in that XCD's L2 (speed only).  Dropout uses a stateless hash of
// (seed, row, head, key) so the backward can regenerate the mask.
#include "attn_dev.h"
#include "bf16_dev.h"

namespace hopmi {

constexpr int AE = 128;            // head dim (d_keys = d_ff = 128, HOP.py:119)
constexpr int AKC = 32;            // keys per chunk in the forward / dQ kernels: 43 KB of LDS => 3 workgroups per CU
constexpr int ABM = 64;            // query rows per workgroup (16 per wave)
constexpr int ALD = AE + 4;        // LDS row stride of the K / V (Q / dO) images
constexpr int APLD = AKC + 4;      // LDS row stride of a wave's P tile
constexpr int KVK = 64;            // dK/dV kernel: keys per workgroup (16 per wave)
constexpr int KVR = 32;            // dK/dV kernel: query rows per staged tile
constexpr int KVPLD = KVR + 4;     // LDS row stride of its transposed P / dS tiles
constexpr int AKT = AKC / 16;      // key tiles per chunk

// ------------------------------------------------------------------------------------------------------
// bf16 operand images (split: hi + lo, bf16_dev.h) of a [S][H][E] fp32 tensor, made once per call by
// attn_images_kernel so that the main kernels stage MFMA operands with plain 16-byte copies:
//   nat[h][part][Sp][E]   rows = keys    (B operand of a product that sums over e: lane reads 8 consecutive e)
//   tr [h][part][E][Sp]   rows = e       (B operand of a product that sums over keys: lane reads 8 consecutive keys)
// Sp = S rounded up to a multiple of 32; rows >= S are zero.
// ------------------------------------------------------------------------------------------------------
typedef unsigned short u16;

__global__ __launch_bounds__(256) void attn_images_kernel(const float* __restrict__ X, int S, int Sp, int H, float mul,
                                                          u16* __restrict__ nat, u16* __restrict__ tr) {
  __shared__ u16 th[2][32][AE + 2];                                  // [part][key][e] (+2: odd 4-byte stride for the column reads)
  const int tid = threadIdx.x, h = blockIdx.x % H, key0 = (blockIdx.x / H) * 32;
  const size_t rs = (size_t)H * AE;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int u = tid + 256 * it, k = u >> 5, c4 = u & 31;            // key row, float4 column
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (key0 + k < S) x = *reinterpret_cast<const f32x4*>(X + (size_t)(key0 + k) * rs + (size_t)h * AE + 4 * c4);
    const Split4 sp = split4(x[0] * mul, x[1] * mul, x[2] * mul, x[3] * mul);
    if (nat != nullptr) {
      *reinterpret_cast<u32x2*>(nat + ((size_t)(h * 2 + 0) * Sp + key0 + k) * AE + 4 * c4) = sp.hi;
      *reinterpret_cast<u32x2*>(nat + ((size_t)(h * 2 + 1) * Sp + key0 + k) * AE + 4 * c4) = sp.lo;
    }
    *reinterpret_cast<unsigned*>(&th[0][k][4 * c4]) = sp.hi[0]; *reinterpret_cast<unsigned*>(&th[0][k][4 * c4 + 2]) = sp.hi[1];
    *reinterpret_cast<unsigned*>(&th[1][k][4 * c4]) = sp.lo[0]; *reinterpret_cast<unsigned*>(&th[1][k][4 * c4 + 2]) = sp.lo[1];
  }
  if (tr == nullptr) return;
  __syncthreads();
  {
    const int part = tid >> 7, e = tid & 127;                         // one e-row of one part per thread: 32 keys = 64 bytes
    unsigned pk[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) pk[k2] = (unsigned)th[part][2 * k2][e] | ((unsigned)th[part][2 * k2 + 1][e] << 16);
    u32x4* dst = reinterpret_cast<u32x4*>(tr + ((size_t)(h * 2 + part) * AE + e) * Sp + key0);
#pragma unroll
    for (int v = 0; v < 4; ++v) dst[v] = u32x4{pk[4 * v], pk[4 * v + 1], pk[4 * v + 2], pk[4 * v + 3]};
  }
}

// XOR swizzle of the 16-byte slots of 64-byte LDS rows ([rows][32 x bf16]) read as MFMA operands by ds_read_b128
// (lane (n, q) reads slot q of row n): slot' = slot ^ f((row >> 2) & 3), f = 0, 3, 2, 1 puts the 16 lanes of every
// ds_read_b128 service group on 16 distinct slots of the 256-byte bank line.
__device__ __forceinline__ int swz64(int row) { return (0x1230 >> (4 * ((row >> 2) & 3))) & 3; }

// Forward.  Workgroup = (64-row tile, head), wave w owns 16 rows (softmax statistics never cross waves) and keeps their
// scaled, split Q fragments in registers.  Per 32-key chunk: the K image rows (256 B, slot ^= row & 15) and the V^T image
// rows (64 B, swz64) are copied global -> registers (one chunk ahead) -> LDS; scores = 24 MFMAs (2 key tiles x 4 k-steps
// x 3 terms), online softmax on the accumulator layout, P -> the wave's LDS tile as packed hi / lo pairs, O += 24 MFMAs.
// Score tile nt, column j is key 2 j + nt of the chunk (K rows are permuted in LDS accordingly), so a lane's two scores of
// a row are adjacent keys: one packed 32-bit store per part.
// 40 KB of LDS, <= 168 registers => 3 workgroups per CU: all 544 workgroups of the B = 128 shape are resident at once.
__global__ __launch_bounds__(256, 3) void reprog_attn_fwd_kernel(const float* __restrict__ Q, const u16* __restrict__ Knat,
                                                              const u16* __restrict__ Vtr, float* __restrict__ O,
                                                              float* __restrict__ lse, int N, int S, int Sp, int H, float scale,
                                                              unsigned drop_thresh, float drop_scale, unsigned seed, const unsigned* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Kl_ = lds;                        // [2 parts][32 rows][256 B]
  unsigned char* Vl_ = lds + 2 * 32 * 256;         // [2 parts][128 rows][64 B]
  unsigned char* Pl_ = Vl_ + 2 * 128 * 64;         // [4 waves][2 parts][16 rows][64 B]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int h = blockIdx.x % H, tile = blockIdx.x / H;
  const size_t rs = (size_t)H * AE;
  const int row_a = tile * ABM + 16 * w + j;                        // A-layout row of this lane

  u32x4 qh[4], ql[4];                                               // A[i = row][k = e = 32 ks + 8 q + x], pre-scaled
  {
    const f32x4* qp = reinterpret_cast<const f32x4*>(Q + (size_t)min(row_a, N - 1) * rs + (size_t)h * AE + 8 * q);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 a = qp[8 * ks], b = qp[8 * ks + 1];
      const Split8 f = split8(make_float4(a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale),
                              make_float4(b[0] * scale, b[1] * scale, b[2] * scale, b[3] * scale));
      qh[ks] = f.hi; ql[ks] = f.lo;
    }
  }
  f32x4 acc_o[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) acc_o[nt] = {0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -1e30f; l_run[r] = 0.f; }

  // chunk staging: 8 16-byte units per thread (K hi, K lo, V^T hi, V^T lo; two of each)
  const u16* kbase = Knat + (size_t)h * 2 * Sp * AE;
  const u16* vbase = Vtr + (size_t)h * 2 * AE * Sp;
  u32x4 pre[8];
  int dst_off[8];
#pragma unroll
  for (int part = 0; part < 2; ++part)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = tid + 256 * it;
      const int k = u >> 4, s = u & 15, lrow = (k & 1) * 16 + (k >> 1);
      dst_off[part * 2 + it] = part * 32 * 256 + lrow * 256 + ((s ^ (lrow & 15)) << 4);
      const int e = u >> 2, s4 = u & 3;
      dst_off[4 + part * 2 + it] = 2 * 32 * 256 + part * 128 * 64 + e * 64 + ((s4 ^ swz64(e)) << 4);
    }
#define HOPMI_ATTN_ISSUE(c_)                                                                                         \
  _Pragma("unroll") for (int part = 0; part < 2; ++part)                                                             \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                                               \
      const int u = tid + 256 * it;                                                                                  \
      pre[part * 2 + it] = *reinterpret_cast<const u32x4*>(kbase + ((size_t)part * Sp + (c_) * 32 + (u >> 4)) * AE + 8 * (u & 15)); \
      pre[4 + part * 2 + it] = *reinterpret_cast<const u32x4*>(vbase + ((size_t)part * AE + (u >> 2)) * Sp + (c_) * 32 + 8 * (u & 3)); \
    }
  const int nchunk = Sp / 32;
  unsigned char* Pw = Pl_ + w * 2 * 16 * 64;
  const int row_c0 = tile * ABM + 16 * w + 4 * q;
  HOPMI_ATTN_ISSUE(0)

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                                               // previous chunk's LDS images consumed
#pragma unroll
    for (int x = 0; x < 8; ++x) *reinterpret_cast<u32x4*>(lds + dst_off[x]) = pre[x];
    __syncthreads();
    if (c + 1 < nchunk) { HOPMI_ATTN_ISSUE(c + 1) }

    // ---- scores: S[16 rows][32 keys] = (scale Q) K^T ------------------------------------------------
    f32x4 acc_s[2];
    acc_s[0] = {0.f, 0.f, 0.f, 0.f}; acc_s[1] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int off = (16 * nt + j) * 256 + (((4 * ks + q) ^ j) << 4);
        const u32x4 kh = *reinterpret_cast<const u32x4*>(Kl_ + off);
        const u32x4 kl = *reinterpret_cast<const u32x4*>(Kl_ + 32 * 256 + off);
        acc_s[nt] = mfma_split3(qh[ks], ql[ks], kh, kl, acc_s[nt]);
      }
    // ---- online softmax over this chunk (lane holds rows 4q + r, keys 2j + nt) ------------------------
    const int key0 = c * 32 + 2 * j;
    float p[2][4], alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -1e30f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float sv = (key0 + nt < S) ? acc_s[nt][r] : -1e30f;
        p[nt][r] = sv;
        mx = fmaxf(mx, sv);
      }
      mx = row16_max(mx);
      const float m_new = fmaxf(m_run[r], mx);
      alpha[r] = __expf(m_run[r] - m_new);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float e = __expf(p[nt][r] - m_new);
        p[nt][r] = e;
        sum += e;
      }
      sum = row16_sum(sum);
      l_run[r] = l_run[r] * alpha[r] + sum;
      m_run[r] = m_new;
    }
    // dropout on the probabilities (HOP.py:296), P -> this wave's LDS tile, [row][key] order, hi and lo images
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pv[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        pv[nt] = p[nt][r];
        if (drop_thresh) pv[nt] = (attn_hash(seed, row_c0 + r, h, key0 + nt) >= drop_thresh) ? pv[nt] * drop_scale : 0.f;
      }
      const u32x2 sp = split2(pv[0], pv[1]);
      const int prow = 4 * q + r;
      const int off = prow * 64 + (((j >> 2) ^ swz64(prow)) << 4) + ((j & 3) << 2);
      *reinterpret_cast<unsigned*>(Pw + off) = sp[0];
      *reinterpret_cast<unsigned*>(Pw + 16 * 64 + off) = sp[1];
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_o[nt][r] *= alpha[r];
    // (the P tile is private to the wave: LDS operations of one wave complete in order, no barrier)
    // ---- O[16 rows][128] += P[16][32 keys] V[32 keys][128] --------------------------------------------
    {
      const int poff = j * 64 + ((q ^ swz64(j)) << 4);
      const u32x4 ph = *reinterpret_cast<const u32x4*>(Pw + poff);
      const u32x4 pl = *reinterpret_cast<const u32x4*>(Pw + 16 * 64 + poff);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int voff = (16 * nt + j) * 64 + ((q ^ swz64(j)) << 4);
        const u32x4 vh = *reinterpret_cast<const u32x4*>(Vl_ + voff);
        const u32x4 vl = *reinterpret_cast<const u32x4*>(Vl_ + 128 * 64 + voff);
        acc_o[nt] = mfma_split3(ph, pl, vh, vl, acc_o[nt]);
      }
    }
  }
#undef HOPMI_ATTN_ISSUE

  // ---- epilogue: normalise, store O and the log-sum-exp of every row ---------------------------------
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = tile * ABM + 16 * w + 4 * q + r;
    if (row < N) {
      const float inv = 1.f / l_run[r];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) O[(size_t)row * rs + (size_t)h * AE + 16 * nt + j] = acc_o[nt][r] * inv;
      if (j == 0) lse[(size_t)row * H + h] = m_run[r] + __logf(l_run[r]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// backward.  With P = softmax(scale * Q K^T) (recomputed from the saved log-sum-exp), M the dropout keep
// mask / (1 - p), delta[n][h] = sum_e dO O (computed by the caller):
//   dV = (P o M)^T dO        dP = (dO V^T) o M        dS = P o (dP - delta) * scale
//   dQ = dS K                dK = dS^T Q
// Two kernels so that every output element has exactly one owner (no atomics, bitwise reproducible):
//   reprog_attn_bwd_dq_kernel   grid (64-row query tile, head): like the forward, loops over key chunks
//   reprog_attn_bwd_dkv_kernel  grid (64-key chunk, head): wave w owns 16 keys (K/V rows in registers, dK/dV
//                               in accumulators) and loops over the query tiles staged through LDS
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void reprog_attn_bwd_dq_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                 const float* __restrict__ Vv, const float* __restrict__ dO,
                                                                 const float* __restrict__ lse, const float* __restrict__ delta,
                                                                 float* __restrict__ dQ, int N, int S, int H, float scale,
                                                                 unsigned drop_thresh, float drop_scale, unsigned seed, const unsigned* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = Ks + AKC * ALD;
  float* Ps = Vs + AKC * ALD;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int h = blockIdx.x % H, tile = blockIdx.x / H;
  const size_t rs = (size_t)H * AE;
  const int row_a = min(tile * ABM + 16 * w + j, N - 1);

  float4 qf[8], dof[8];
  {
    const float4* qp = reinterpret_cast<const float4*>(Q + (size_t)row_a * rs + (size_t)h * AE + 4 * q);
    const float4* dp = reinterpret_cast<const float4*>(dO + (size_t)row_a * rs + (size_t)h * AE + 4 * q);
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) { qf[ii] = qp[4 * ii]; dof[ii] = dp[4 * ii]; }
  }
  float lse_r[4], del_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(tile * ABM + 16 * w + 4 * q + r, N - 1);
    lse_r[r] = lse[(size_t)row * H + h];
    del_r[r] = delta[(size_t)row * H + h];
  }
  f32x4 acc_dq[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) acc_dq[nt] = {0.f, 0.f, 0.f, 0.f};

  const int srow = tid >> 5, sc4 = tid & 31;
  f32x4 kreg[AKC / 8], vreg[AKC / 8];          // native vectors: HIP's float4 struct arrays stayed in scratch here
#define HOPMI_ATTN_ISSUE_KV(c_)                                                                          \
  _Pragma("unroll") for (int it = 0; it < AKC / 8; ++it) {                                               \
    const int key_ = min((c_) * AKC + srow + 8 * it, S - 1);                                             \
    kreg[it] = reinterpret_cast<const f32x4*>(K + (size_t)key_ * rs + (size_t)h * AE)[sc4];              \
    vreg[it] = reinterpret_cast<const f32x4*>(Vv + (size_t)key_ * rs + (size_t)h * AE)[sc4];             \
  }
  const int nchunk = (S + AKC - 1) / AKC;
  float* Pw = Ps + w * 16 * APLD;
  const int row_c0 = tile * ABM + 16 * w + 4 * q;

  for (int c = 0; c < nchunk; ++c) {
    HOPMI_ATTN_ISSUE_KV(c)
    __syncthreads();
#pragma unroll
    for (int it = 0; it < AKC / 8; ++it) {
      *reinterpret_cast<f32x4*>(Ks + (srow + 8 * it) * ALD + 4 * sc4) = kreg[it];
      *reinterpret_cast<f32x4*>(Vs + (srow + 8 * it) * ALD + 4 * sc4) = vreg[it];
    }
    __syncthreads();

    f32x4 acc_s[AKT], acc_dp[AKT];
#pragma unroll
    for (int nt = 0; nt < AKT; ++nt) { acc_s[nt] = {0.f, 0.f, 0.f, 0.f}; acc_dp[nt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
      float4 bk[AKT], bv[AKT];
#pragma unroll
      for (int nt = 0; nt < AKT; ++nt) {
        bk[nt] = *reinterpret_cast<const float4*>(Ks + (16 * nt + j) * ALD + 16 * ii + 4 * q);
        bv[nt] = *reinterpret_cast<const float4*>(Vs + (16 * nt + j) * ALD + 16 * ii + 4 * q);
      }
#pragma unroll
      for (int nt = 0; nt < AKT; ++nt) {
        acc_s[nt] = mfma16(qf[ii].x, bk[nt].x, acc_s[nt]);
        acc_dp[nt] = mfma16(dof[ii].x, bv[nt].x, acc_dp[nt]);
        acc_s[nt] = mfma16(qf[ii].y, bk[nt].y, acc_s[nt]);
        acc_dp[nt] = mfma16(dof[ii].y, bv[nt].y, acc_dp[nt]);
        acc_s[nt] = mfma16(qf[ii].z, bk[nt].z, acc_s[nt]);
        acc_dp[nt] = mfma16(dof[ii].z, bv[nt].z, acc_dp[nt]);
        acc_s[nt] = mfma16(qf[ii].w, bk[nt].w, acc_s[nt]);
        acc_dp[nt] = mfma16(dof[ii].w, bv[nt].w, acc_dp[nt]);
      }
    }
    const int key0 = c * AKC;
#pragma unroll
    for (int nt = 0; nt < AKT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + 16 * nt + j;
        const float pr = (key < S) ? __expf(acc_s[nt][r] * scale - lse_r[r]) : 0.f;
        float dp = acc_dp[nt][r];
        if (drop_thresh) dp = (attn_hash(seed, row_c0 + r, h, key) >= drop_thresh) ? dp * drop_scale : 0.f;
        Pw[(4 * q + r) * APLD + 16 * nt + j] = pr * (dp - del_r[r]) * scale;              // dS
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AKT; ++i) {
      const float4 a = *reinterpret_cast<const float4*>(Pw + j * APLD + 16 * i + 4 * q);     // dS[row j][key]
      const float* kb = Ks + (16 * i + 4 * q) * ALD + j;                                      // B[k = key][n = e]
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        acc_dq[nt] = mfma16(a.x, kb[16 * nt], acc_dq[nt]);
        acc_dq[nt] = mfma16(a.y, kb[ALD + 16 * nt], acc_dq[nt]);
        acc_dq[nt] = mfma16(a.z, kb[2 * ALD + 16 * nt], acc_dq[nt]);
        acc_dq[nt] = mfma16(a.w, kb[3 * ALD + 16 * nt], acc_dq[nt]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = tile * ABM + 16 * w + 4 * q + r;
    if (row < N) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) dQ[(size_t)row * rs + (size_t)h * AE + 16 * nt + j] = acc_dq[nt][r];
    }
  }
}

__global__ __launch_bounds__(256) void reprog_attn_bwd_dkv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                  const float* __restrict__ Vv, const float* __restrict__ dO,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  float* __restrict__ dKp, float* __restrict__ dVp, int N, int S,
                                                                  int H, int nsplit, float scale, unsigned drop_thresh,
                                                                  float drop_scale, unsigned seed, const unsigned* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  // workgroup = (key chunk of 64, head, row split): the query-row tiles t = split, split + nsplit, ... are
  // walked here and the partial dK / dV go to slab `split` of dKp / dVp ([nsplit][S][H][E]); the caller
  // adds the slabs in a fixed order.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                                // [32 rows][ALD]
  float* Ds = Qs + KVR * ALD;                      // [32 rows][ALD]  dO
  float* Ps = Ds + KVR * ALD;                      // [4 waves][2][16][KVPLD]   (P o M)^T and dS^T tiles
  float* Ls = Ps + 4 * 2 * 16 * KVPLD;             // [32] lse, [32] delta
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int nchunk = (S + KVK - 1) / KVK;
  const int h = blockIdx.x % H, chunk = (blockIdx.x / H) % nchunk, split = blockIdx.x / (H * nchunk);
  const size_t rs = (size_t)H * AE;
  const int key_a = min(chunk * KVK + 16 * w + j, S - 1);        // A-layout key row of this lane

  float4 kf[8], vf[8];
  {
    const float4* kp = reinterpret_cast<const float4*>(K + (size_t)key_a * rs + (size_t)h * AE + 4 * q);
    const float4* vp = reinterpret_cast<const float4*>(Vv + (size_t)key_a * rs + (size_t)h * AE + 4 * q);
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) { kf[ii] = kp[4 * ii]; vf[ii] = vp[4 * ii]; }
  }
  f32x4 acc_dk[8], acc_dv[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) { acc_dk[nt] = {0.f, 0.f, 0.f, 0.f}; acc_dv[nt] = {0.f, 0.f, 0.f, 0.f}; }

  const int srow = tid >> 5, sc4 = tid & 31;
  f32x4 qreg[KVR / 8], dreg[KVR / 8];
  float lreg = 0.f, greg = 0.f;
#define HOPMI_ATTN_ISSUE_QD(t_)                                                                          \
  {                                                                                                      \
    _Pragma("unroll") for (int it = 0; it < KVR / 8; ++it) {                                             \
      const int row_ = min((t_) * KVR + srow + 8 * it, N - 1);                                           \
      qreg[it] = reinterpret_cast<const f32x4*>(Q + (size_t)row_ * rs + (size_t)h * AE)[sc4];           \
      dreg[it] = reinterpret_cast<const f32x4*>(dO + (size_t)row_ * rs + (size_t)h * AE)[sc4];          \
    }                                                                                                    \
    const int lrow_ = min((t_) * KVR + (tid & (KVR - 1)), N - 1);                                        \
    lreg = lse[(size_t)lrow_ * H + h];                                                                   \
    greg = delta[(size_t)lrow_ * H + h];                                                                 \
  }
  const int ntile = (N + KVR - 1) / KVR;
  if (split < ntile) HOPMI_ATTN_ISSUE_QD(split)
  float* Pw = Ps + w * 2 * 16 * KVPLD;             // (P o M)^T
  float* Sw = Pw + 16 * KVPLD;                     // dS^T
  const int key_c0 = chunk * KVK + 16 * w + 4 * q; // C-layout key rows 4q + r of this wave's tile

  for (int t = split; t < ntile; t += nsplit) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < KVR / 8; ++it) {
      *reinterpret_cast<f32x4*>(Qs + (srow + 8 * it) * ALD + 4 * sc4) = qreg[it];
      *reinterpret_cast<f32x4*>(Ds + (srow + 8 * it) * ALD + 4 * sc4) = dreg[it];
    }
    if (tid < KVR) { Ls[tid] = lreg; Ls[KVR + tid] = greg; }
    __syncthreads();
    if (t + nsplit < ntile) HOPMI_ATTN_ISSUE_QD(t + nsplit)

    // S^T[16 keys][32 rows] = K_w Q^T ;  dP^T = V_w dO^T
    f32x4 acc_s[2], acc_dp[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { acc_s[nt] = {0.f, 0.f, 0.f, 0.f}; acc_dp[nt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
      float4 bq[2], bd[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        bq[nt] = *reinterpret_cast<const float4*>(Qs + (16 * nt + j) * ALD + 16 * ii + 4 * q);
        bd[nt] = *reinterpret_cast<const float4*>(Ds + (16 * nt + j) * ALD + 16 * ii + 4 * q);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc_s[nt] = mfma16(kf[ii].x, bq[nt].x, acc_s[nt]);
        acc_dp[nt] = mfma16(vf[ii].x, bd[nt].x, acc_dp[nt]);
        acc_s[nt] = mfma16(kf[ii].y, bq[nt].y, acc_s[nt]);
        acc_dp[nt] = mfma16(vf[ii].y, bd[nt].y, acc_dp[nt]);
        acc_s[nt] = mfma16(kf[ii].z, bq[nt].z, acc_s[nt]);
        acc_dp[nt] = mfma16(vf[ii].z, bd[nt].z, acc_dp[nt]);
        acc_s[nt] = mfma16(kf[ii].w, bq[nt].w, acc_s[nt]);
        acc_dp[nt] = mfma16(vf[ii].w, bd[nt].w, acc_dp[nt]);
      }
    }
    // lane holds key rows 4q + r, query-row column 16 nt + j
    const int row0 = t * KVR;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int row = row0 + 16 * nt + j;
      const float lrow = Ls[16 * nt + j], drow = Ls[KVR + 16 * nt + j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key_c0 + r;
        const float pr = (key < S && row < N) ? __expf(acc_s[nt][r] * scale - lrow) : 0.f;
        float keepf = 1.f;
        if (drop_thresh) keepf = (attn_hash(seed, row, h, key) >= drop_thresh) ? drop_scale : 0.f;
        Pw[(4 * q + r) * KVPLD + 16 * nt + j] = pr * keepf;                                   // (P o M)^T
        Sw[(4 * q + r) * KVPLD + 16 * nt + j] = pr * (acc_dp[nt][r] * keepf - drow) * scale;  // dS^T
      }
    }
    __syncthreads();
    // dV_w[16 keys][128] += (P o M)^T[16][32 rows] dO[32 rows][128];  dK_w += dS^T Q
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 ap = *reinterpret_cast<const float4*>(Pw + j * KVPLD + 16 * i + 4 * q);  // A[key j][row 16i+4q+e]
      const float4 as = *reinterpret_cast<const float4*>(Sw + j * KVPLD + 16 * i + 4 * q);
      const float* db = Ds + (16 * i + 4 * q) * ALD + j;                                     // B[k = row][n = e]
      const float* qb = Qs + (16 * i + 4 * q) * ALD + j;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        acc_dv[nt] = mfma16(ap.x, db[16 * nt], acc_dv[nt]);
        acc_dk[nt] = mfma16(as.x, qb[16 * nt], acc_dk[nt]);
        acc_dv[nt] = mfma16(ap.y, db[ALD + 16 * nt], acc_dv[nt]);
        acc_dk[nt] = mfma16(as.y, qb[ALD + 16 * nt], acc_dk[nt]);
        acc_dv[nt] = mfma16(ap.z, db[2 * ALD + 16 * nt], acc_dv[nt]);
        acc_dk[nt] = mfma16(as.z, qb[2 * ALD + 16 * nt], acc_dk[nt]);
        acc_dv[nt] = mfma16(ap.w, db[3 * ALD + 16 * nt], acc_dv[nt]);
        acc_dk[nt] = mfma16(as.w, qb[3 * ALD + 16 * nt], acc_dk[nt]);
      }
    }
  }
  float* dK = dKp + (size_t)split * S * rs;
  float* dV = dVp + (size_t)split * S * rs;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = key_c0 + r;
    if (key < S) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        dK[(size_t)key * rs + (size_t)h * AE + 16 * nt + j] = acc_dk[nt][r];
        dV[(size_t)key * rs + (size_t)h * AE + 16 * nt + j] = acc_dv[nt][r];
      }
    }
  }
}

}  // namespace hopmi

using namespace hopmi;

// workspace of hopmi_reprog_attn_fwd / _bwd: the bf16 operand images of K and V (natural and transposed, hi + lo)
static int attn_sp(int S) { return (S + 31) / 32 * 32; }
extern "C" size_t hopmi_reprog_attn_ws_bytes(int S, int H, int E) {
  if (S <= 0 || H <= 0 || E != AE) return 0;
  return (size_t)4 * 2 * H * attn_sp(S) * AE * sizeof(u16);
}

extern "C" int hopmi_reprog_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, void* ws,
                                     int N, int S, int H, int E, float scale, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream) {
  if (!q || !k || !v || !o || !lse || !ws) { set_error("hopmi_reprog_attn_fwd: null pointer argument"); return HOPMI_EINVAL; }
  if (E != AE || N <= 0 || S <= 0 || H <= 0) {
    set_error("hopmi_reprog_attn_fwd: need head dim 128 and positive sizes (N=%d S=%d H=%d E=%d)", N, S, H, E);
    return HOPMI_EINVAL;
  }
  if (!(p_drop >= 0.f && p_drop < 1.f)) { set_error("hopmi_reprog_attn_fwd: p_drop=%f outside [0,1)", p_drop); return HOPMI_EINVAL; }
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int Sp = attn_sp(S);
  const size_t img = (size_t)2 * H * Sp * AE;                       // elements per image
  u16* knat = static_cast<u16*>(ws);
  u16* vtr = knat + 3 * img;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(attn_images_kernel, dim3((Sp / 32) * H), dim3(256), 0, st, k, S, Sp, H, 1.f, knat, (u16*)nullptr);
  hipLaunchKernelGGL(attn_images_kernel, dim3((Sp / 32) * H), dim3(256), 0, st, v, S, Sp, H, 1.f, (u16*)nullptr, vtr);
  if (int e = check_launch("hopmi_reprog_attn_fwd(images)")) return e;
  const int ntile = (N + ABM - 1) / ABM;
  const size_t lds = (size_t)2 * 32 * 256 + 2 * 128 * 64 + 4 * 2 * 16 * 64;
  hipLaunchKernelGGL(reprog_attn_fwd_kernel, dim3(ntile * H), dim3(256), lds, st, q, knat, vtr, o, lse, N, S, Sp, H, scale,
                     thresh, dscale, seed, seed_dev);
  return check_launch("hopmi_reprog_attn_fwd");
}

// dK/dV grid = 24 key chunks x 8 heads x splits workgroups at 2 resident per CU (512 slots): 8 splits make
// it exactly 3 full rounds at the B = 128 shape
extern "C" int hopmi_reprog_attn_bwd_splits(void) { return 8; }

extern "C" int hopmi_reprog_attn_bwd(const float* q, const float* k, const float* v, const float* d_o, const float* lse,
                                     const float* delta, float* dq, float* dk, float* dv, int N, int S, int H, int E,
                                     float scale, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream) {
  if (!q || !k || !v || !d_o || !lse || !delta || !dq || !dk || !dv) { set_error("hopmi_reprog_attn_bwd: null pointer argument"); return HOPMI_EINVAL; }
  if (E != AE || N <= 0 || S <= 0 || H <= 0) {
    set_error("hopmi_reprog_attn_bwd: need head dim 128 and positive sizes (N=%d S=%d H=%d E=%d)", N, S, H, E);
    return HOPMI_EINVAL;
  }
  if (!(p_drop >= 0.f && p_drop < 1.f)) { set_error("hopmi_reprog_attn_bwd: p_drop=%f outside [0,1)", p_drop); return HOPMI_EINVAL; }
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds_q = (size_t)(2 * AKC * ALD + 4 * 16 * APLD) * sizeof(float);
  hipLaunchKernelGGL(reprog_attn_bwd_dq_kernel, dim3(((N + ABM - 1) / ABM) * H), dim3(256), lds_q, st, q, k, v, d_o, lse, delta,
                     dq, N, S, H, scale, thresh, dscale, seed, seed_dev);
  if (int e = check_launch("hopmi_reprog_attn_bwd(dq)")) return e;
  const int nsplit = hopmi_reprog_attn_bwd_splits();
  const size_t lds_kv = (size_t)(2 * KVR * ALD + 4 * 2 * 16 * KVPLD + 2 * KVR) * sizeof(float);
  hipLaunchKernelGGL(reprog_attn_bwd_dkv_kernel, dim3(((S + KVK - 1) / KVK) * H * nsplit), dim3(256), lds_kv, st, q, k, v, d_o,
                     lse, delta, dk, dv, N, S, H, nsplit, scale, thresh, dscale, seed, seed_dev);
  return check_launch("hopmi_reprog_attn_bwd(dkv)");
}
