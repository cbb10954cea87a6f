// Audio->text "reprogramming" cross-attention of HOP (reference: model/HOP.py:289-299):
//
//   scores = einsum("blhe,she->bhls", Q, K);  A = dropout(softmax(scores / sqrt(E)));  O = einsum("bhls,she->blhe", A, V)
//
// with L = 34 audio frames, S = 1500 text prototypes shared by the whole batch, H = 8 heads, E = 128.
// The reference materialises the (B,8,34,1500) score tensor (209 MB at B = 128) three times per forward;
// here it never leaves the chip: flash-style online softmax, and all five contractions (forward: Q K^T, P V; backward:
// dO V^T, dS K, dS^T Q, P^T dO) as three-term products of scaled fp16 hi/lo operands on v_mfma_f32_16x16x32_f16 (f16_dev.h:
// 22 significand bits per operand, fp32 accumulation, fp32 softmax: fp32-equivalent; rounds 2-4 carried bf16 hi/lo pairs, 2^-16
// per product).  Operand scales (powers of two, exact): K, V -- and in the backward's image operands Q, dO -- one per (tensor, head)
// from a maximum pass (attn_absmax_kernel) that the image kernel turns into {s, 1/s}; Q and dO where they are A fragments in
// registers: one per query row; the probabilities (<= 1 / (1 - p_drop)): 2^14; dS, whose magnitude is not known in advance: one
// per accumulator row from the row's RUNNING maximum -- when it grows past a power of two the row of the accumulator is
// rescaled (exact), as the online softmax does with its running maximum.
//
// Rows (b, l) are flat: N = B*L query rows per head.  Workgroup = (64-row tile, head); wave w owns 16 rows
// entirely (its softmax statistics never cross waves) and keeps their Q fragments in registers.  The MFMA B operands are
// read from bf16 images of K / V (and, in the backward, Q / dO) that attn_images_kernel prepares per call (natural rows
// and transposed rows, hi + lo), so that staging a 32-key chunk is a set of plain 16-byte copies global -> registers -> LDS.
// Linear workgroup id = tile*H + head, so with H = 8 every head lives on one XCD and its images stay in that XCD's L2
// (speed only).  Dropout uses a stateless hash of (seed, row, head, key pair) so the backward can regenerate the mask.
//
// Storage type TIO of q / k / v / o / dO / dq (the `_dt` entry points): fp32, or bf16 as the tensors leave and enter the
// bf16 GEMMs around the operator under autocast (no cast launches; loads widened, stores rounded, arithmetic unchanged).
// A bf16 operand has no lo part (its 8 significand bits fit fp16's 11), so the terms that would multiply it are not issued
// (mfma_h<A_LO, B_LO>): the scaled Q, P and dS keep theirs.
#define HOPMI_FILE_ID 4          // (diagnostic build: common.h, split_check)
#include "attn_dev.h"
#include "f16_dev.h"
#include "io_dev.h"

namespace hopmi {

constexpr int AE = 128;            // head dim (d_keys = d_ff = 128, HOP.py:119)
constexpr int ABM = 64;            // query rows per workgroup (16 per wave)

// ------------------------------------------------------------------------------------------------------
// bf16 operand images (split: hi + lo, bf16_dev.h) of a [S][H][E] fp32 tensor, made once per call by
// attn_images_kernel so that the main kernels stage MFMA operands with plain 16-byte copies:
//   nat[h][part][Sp][E]   rows = keys    (B operand of a product that sums over e: lane reads 8 consecutive e)
//   tr [h][part][E][Sp]   rows = e       (B operand of a product that sums over keys: lane reads 8 consecutive keys)
// Sp = S rounded up to a multiple of 32; rows >= S are zero.
// ------------------------------------------------------------------------------------------------------
typedef unsigned short u16;

// ---- operand scales -------------------------------------------------------------------------------------------------------
// scale block of the workspace (behind the K / V images): sc[tensor][H][SCW] = {s, 1 / s, max |x|, 0} (tensor 0 = K, 1 = V,
// 2 = Q, 3 = dO; written by the image kernel), then the partial maxima parts[tensor][H][APARTS] (bit patterns of |x|,
// attn_absmax_kernel)
constexpr int APARTS = 16;
constexpr int ATENS = 4;
constexpr int SCW = 4;
__host__ __device__ inline size_t attn_scale_floats(int H) { return (size_t)ATENS * H * SCW + (size_t)ATENS * H * APARTS; }

struct AbsmaxArgs { const void* X[ATENS]; int rows[ATENS]; };

// partial maxima of |X[t][rows][H][E]| per (tensor, head): block = (part, head, tensor), fixed work split, no atomics
template <typename TIO>
__global__ __launch_bounds__(256) void attn_absmax_kernel(AbsmaxArgs A, int H, unsigned* __restrict__ parts) {
  __shared__ float red[4];
  const int part = blockIdx.x, h = blockIdx.y, t = blockIdx.z;
  const TIO* X = static_cast<const TIO*>(A.X[t]);
  float m = 0.f;
  if (X != nullptr) {
    const int rows = A.rows[t], per = (rows + APARTS - 1) / APARTS;
    const int r0 = part * per, r1 = min(rows, r0 + per);
    const size_t rs = (size_t)H * AE;
    // 32 float4 per row: 8 rows per pass of the block
    for (int r = r0 + (threadIdx.x >> 5); r < r1; r += 8) m = absmax4(m, ld4(X + (size_t)r * rs + (size_t)h * AE + 4 * (threadIdx.x & 31)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) parts[((size_t)t * H + h) * APARTS + part] = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

struct ImageArgs {
  const void* X[2];      // [S][H][E] tensors of this launch (blockIdx.y selects)
  u16* nat[2];           // or null
  u16* tr[2];            // or null
  int tensor[2];         // index into the scale block
};
// (Only powers of two ever multiply a value in front of a split: hipcc contracts `fp16(x * c)` into ONE rounding (v_fma_mix) for
// the lo part's subtrahend while the stored hi part comes from the fp32 product rounded twice -- with an inexact product the two
// disagree by an fp16 ulp now and then, i.e. hi + lo is off by 2^-11.  The softmax scale therefore rides on the accumulator ->
// score factor, never on Q.)

template <typename TIO>
__global__ __launch_bounds__(256) void attn_images_kernel(ImageArgs A, int S, int Sp, int H, float* __restrict__ scblock) {
  __shared__ u16 th[2][32][AE + 2];                                  // [part][key][e] (+2: odd 4-byte stride for the column reads)
  const int which = blockIdx.y;
  const TIO* X = static_cast<const TIO*>(A.X[which]);
  u16* nat = A.nat[which];
  u16* tr = A.tr[which];
  const int tid = threadIdx.x, h = blockIdx.x % H, key0 = (blockIdx.x / H) * 32;
  const size_t rs = (size_t)H * AE;
  // the (tensor, head) scale: every block derives the same number from the partial maxima; the first block of a head publishes it
  float mul;                                                           // the (tensor, head) power of two
  {
    const unsigned* parts = reinterpret_cast<const unsigned*>(scblock + (size_t)ATENS * H * SCW) + ((size_t)A.tensor[which] * H + h) * APARTS;
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < APARTS; ++i) m = fmaxf(m, __uint_as_float(parts[i]));
    const float sc = scale_for_absmax(m);
    mul = sc;
    if (key0 == 0 && tid == 0) {
      float* o = scblock + ((size_t)A.tensor[which] * H + h) * SCW;
      o[0] = sc;
      o[1] = inv_pow2(sc);
      o[2] = m;
      o[3] = 0.f;
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int u = tid + 256 * it, k = u >> 5, c4 = u & 31;            // key row, float4 column
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (key0 + k < S) x = ld4(X + (size_t)(key0 + k) * rs + (size_t)h * AE + 4 * c4);
    const Split4 sp = split4h(x.x * mul, x.y * mul, x.z * mul, x.w * mul);
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

// A-operand fragments of one query row held by the 4 lanes (q = 0..3) of column j: the row's power-of-two scale s from its 128
// values, the fragments of x s, and the inverse scales of the lane's four ACCUMULATOR rows 4 q + r (the same rows, other lanes)
template <typename TIO>
__device__ __forceinline__ void load_row_frags(const TIO* p, u32x4 (&fh)[4], u32x4 (&fl)[4], float (&inv_c)[4], int q) {
  float4 a[4], b[4];
  float m = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    a[ks] = ld4(p + 32 * ks);
    b[ks] = ld4(p + 32 * ks + 4);
    m = absmax4(absmax4(m, a[ks]), b[ks]);
  }
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  const float sc = scale_for_absmax(m);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const Split8 f = split8h(a[ks], b[ks], sc);
    fh[ks] = f.hi; fl[ks] = f.lo;
  }
  const float inv = inv_pow2(sc);
#pragma unroll
  for (int r = 0; r < 4; ++r) inv_c[r] = __shfl(inv, 4 * q + r);
}

// Running scale of ONE row of a dS tile (accumulator layout: a lane holds two columns of the row, the row spans the 16 lanes of a
// DPP row).  `sd` is the row's current power-of-two scale (start: 2^60, "as large as any value could use"); a tile whose largest
// magnitude would not fit under it (>= 2^15 after scaling) replaces it by the scale of that magnitude, and the row of the
// accumulator is taken along (x new / old: exact, and rare -- the running maximum has to double).  Returns the scale to split
// this tile's row with.
constexpr float DS_SCALE0 = 1152921504606846976.f;                  // 2^60
template <int NACC>
__device__ __forceinline__ float ds_row_scale(float d0, float d1, float& sd, f32x4 (&acc)[NACC], int r) {
  const float rm = row16_max(fmaxf(fabsf(d0), fabsf(d1)));
  if (__ballot(rm * sd >= 32768.f) != 0ull) {
    const float sn = rm * sd >= 32768.f ? scale_for_absmax(rm) : sd;
    const float ratio = sn * inv_pow2(sd);
#pragma unroll
    for (int nt = 0; nt < NACC; ++nt) acc[nt][r] *= ratio;
    sd = sn;
  }
  return sd;
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
template <typename TIO>
__global__ __launch_bounds__(256, 3) void reprog_attn_fwd_kernel(const TIO* __restrict__ Q, const u16* __restrict__ Knat,
                                                              const u16* __restrict__ Vtr, TIO* __restrict__ O,
                                                              float* __restrict__ lse, const float* __restrict__ scblock, int N, int S, int Sp,
                                                              int H, float scale,
                                                              unsigned drop_thresh, float drop_scale, unsigned seed, const unsigned* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  constexpr bool IN_LO = sizeof(TIO) == 4;        // fp32 inputs have a lo part, bf16 inputs do not
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Kl_ = lds;                        // [2 parts][32 rows][256 B]
  unsigned char* Vl_ = lds + 2 * 32 * 256;         // [2 parts][128 rows][64 B]
  unsigned char* Pl_ = Vl_ + 2 * 128 * 64;         // [4 waves][2 parts][16 rows][64 B]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int h = blockIdx.x % H, tile = blockIdx.x / H;
  const size_t rs = (size_t)H * AE;
  const int row_a = tile * ABM + 16 * w + j;                        // A-layout row of this lane

  u32x4 qh[4], ql[4];                                               // A[i = row][k = e = 32 ks + 8 q + x] x the row's power of two
  float cs[4];                                                      // accumulator -> score: softmax scale x (1 / s_q[row 4q + r]) (1 / s_K)
  load_row_frags(Q + (size_t)min(row_a, N - 1) * rs + (size_t)h * AE + 8 * q, qh, ql, cs, q);
  const float inv_k = scblock[(0 * H + h) * SCW + 1], inv_v = scblock[(1 * H + h) * SCW + 1];
#pragma unroll
  for (int r = 0; r < 4; ++r) cs[r] *= inv_k * scale;
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
  const unsigned row_c0 = (unsigned)(tile * ABM + 16 * w + 4 * q);
  HOPMI_ATTN_ISSUE(0)

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                                               // previous chunk's LDS images consumed
#pragma unroll
    for (int x = 0; x < 8; ++x) *reinterpret_cast<u32x4*>(lds + dst_off[x]) = pre[x];
    __syncthreads();
    if (c + 1 < nchunk) { HOPMI_ATTN_ISSUE(c + 1) }

    // ---- scores: S[16 rows][32 keys] = scale Q K^T --------------------------------------------------
    f32x4 acc_s[2];
    acc_s[0] = {0.f, 0.f, 0.f, 0.f}; acc_s[1] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int off = (16 * nt + j) * 256 + (((4 * ks + q) ^ j) << 4);
        const u32x4 kh = *reinterpret_cast<const u32x4*>(Kl_ + off);
        const u32x4 kl = *reinterpret_cast<const u32x4*>(Kl_ + 32 * 256 + off);
        acc_s[nt] = mfma_h<true, IN_LO>(qh[ks], ql[ks], kh, kl, acc_s[nt]);
      }
    // ---- online softmax over this chunk (lane holds rows 4q + r, keys 2j + nt) ------------------------
    const int key0 = c * 32 + 2 * j;
    const bool pad0 = key0 >= S, pad1 = key0 + 1 >= S;               // (the padded keys of the last chunk)
    float p[2][4], alpha[4];
    bool moved = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s0 = pad0 ? -1e30f : acc_s[0][r] * cs[r], s1 = pad1 ? -1e30f : acc_s[1][r] * cs[r];
      const float mx = row16_max(fmaxf(s0, s1));
      const float m_new = fmaxf(m_run[r], mx);
      moved |= m_new != m_run[r];
      alpha[r] = __expf(m_run[r] - m_new);
      p[0][r] = __expf(s0 - m_new);
      p[1][r] = __expf(s1 - m_new);
      l_run[r] = l_run[r] * alpha[r] + (p[0][r] + p[1][r]);         // this lane's keys only: summed over the row at the end
      m_run[r] = m_new;
    }
    // dropout on the probabilities (HOP.py:296), P -> this wave's LDS tile, [row][key] order, hi and lo images
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pv0 = p[0][r], pv1 = p[1][r];
      if (drop_thresh) {
        const unsigned x = attn_hash_pair(attn_rowhead(seed, row_c0 + r, (unsigned)h), (unsigned)(c * 16 + j));
        pv0 = ((x & 0xFFFFu) >= drop_thresh) ? pv0 * drop_scale : 0.f;
        pv1 = ((x >> 16) >= drop_thresh) ? pv1 * drop_scale : 0.f;
      }
      const u32x2 sp = split2h(pv0 * H_UNIT_SCALE, pv1 * H_UNIT_SCALE);      // (P o M) <= 1 / (1 - p_drop): the fixed scale 2^14
      const int prow = 4 * q + r;
      const int off = prow * 64 + (((j >> 2) ^ swz64(prow)) << 4) + ((j & 3) << 2);
      *reinterpret_cast<unsigned*>(Pw + off) = sp[0];
      *reinterpret_cast<unsigned*>(Pw + 16 * 64 + off) = sp[1];
    }
    if (__ballot(moved) != 0ull) {                                  // some row's running maximum moved: rescale O
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc_o[nt][r] *= alpha[r];
    }
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
        acc_o[nt] = mfma_h<true, IN_LO>(ph, pl, vh, vl, acc_o[nt]);
      }
    }
  }
#undef HOPMI_ATTN_ISSUE

  // ---- epilogue: normalise, store O and the log-sum-exp of every row ---------------------------------
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = tile * ABM + 16 * w + 4 * q + r;
    const float l_row = row16_sum(l_run[r]);
    if (row < N) {
      const float inv = (H_UNIT_INV * inv_v) / l_row;               // accumulator units: 2^14 s_V
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) st1(O + (size_t)row * rs + (size_t)h * AE + 16 * nt + j, acc_o[nt][r] * inv);
      if (j == 0) lse[(size_t)row * H + h] = m_run[r] + __logf(l_row);
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
// Both kernels take their B operands from the fp16 hi/lo images of attn_images_kernel and run every contraction as three
// v_mfma_f32_16x16x32_f16 terms.  The softmax scale multiplies the accumulators (scores) and, for dK = dS^T Q with
// dS = P o (dP - delta) * scale, the dK / dV kernel's epilogue.
//
// dQ kernel, per 32-key chunk: K image rows, V image rows (both 256 B, key-permuted as in the forward) and K^T image rows
// (64 B) in LDS (48 KB => 3 workgroups per CU); S and dP = 48 MFMAs; dS -> the wave's packed tile, which reuses the V
// region (one extra barrier per chunk); dQ += 24 MFMAs.
template <typename TIO>
__global__ __launch_bounds__(256, 3) void reprog_attn_bwd_dq_kernel(const TIO* __restrict__ Q, const u16* __restrict__ Knat,
                                                                 const u16* __restrict__ Vnat, const u16* __restrict__ Ktr,
                                                                 const TIO* __restrict__ dO, const float* __restrict__ lse,
                                                                 const float* __restrict__ delta, TIO* __restrict__ dQ,
                                                                 const float* __restrict__ scblock, int N,
                                                                 int S, int Sp, int H, float scale, unsigned drop_thresh,
                                                                 float drop_scale, unsigned seed, const unsigned* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  constexpr bool IN_LO = sizeof(TIO) == 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Kl_ = lds;                        // [2 parts][32 rows][256 B]
  unsigned char* Vl_ = lds + 2 * 32 * 256;         // [2 parts][32 rows][256 B]; then [4 waves][2 parts][16][64 B] dS tiles
  unsigned char* Tl_ = Vl_ + 2 * 32 * 256;         // [2 parts][128 rows][64 B]  K^T
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int h = blockIdx.x % H, tile = blockIdx.x / H;
  const size_t rs = (size_t)H * AE;
  const int row_a = min(tile * ABM + 16 * w + j, N - 1);

  u32x4 qh[4], ql[4], dh[4], dl[4];
  // (tensor, head) scales: Q and dO as the image kernel scaled them for the dK / dV kernel, K and V
  const float cs = scblock[(2 * H + h) * SCW + 1] * scblock[(0 * H + h) * SCW + 1] * scale;   // accumulator -> score: scale (1 / s_Q)(1 / s_K)
  const float cd = scblock[(3 * H + h) * SCW + 1] * scblock[(1 * H + h) * SCW + 1];      // accumulator -> dP:    (1 / s_dO)(1 / s_V)
  const float inv_k = scblock[(0 * H + h) * SCW + 1];
  float l1 = 0.f;                                                   // sum_e |dO[row][e]| of this lane's A-layout row (over its 4 lanes below)
  {
    const float kq = scblock[(2 * H + h) * SCW], kd = scblock[(3 * H + h) * SCW];
    const TIO* qp = Q + (size_t)row_a * rs + (size_t)h * AE + 8 * q;
    const TIO* dp = dO + (size_t)row_a * rs + (size_t)h * AE + 8 * q;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const Split8 f = split8h(ld4(qp + 32 * ks), ld4(qp + 32 * ks + 4), kq);
      qh[ks] = f.hi; ql[ks] = f.lo;
      const float4 da = ld4(dp + 32 * ks), db = ld4(dp + 32 * ks + 4);
      l1 += (fabsf(da.x) + fabsf(da.y)) + (fabsf(da.z) + fabsf(da.w)) + (fabsf(db.x) + fabsf(db.y)) + (fabsf(db.z) + fabsf(db.w));
      const Split8 g = split8h(da, db, kd);
      dh[ks] = g.hi; dl[ks] = g.lo;
    }
    l1 += __shfl_xor(l1, 16);
    l1 += __shfl_xor(l1, 32);
  }
  // One power-of-two scale per dS row, fixed for the whole key loop, from a bound known up front:
  //   |dS[row][key]| = P |dP keep - delta| scale  <=  (||dO_row||_1 max|V| / (1 - p_drop) + |delta_row|) scale        (P <= 1)
  // Typical elements sit ~2^-16 below it (P ~ 1 / S), still inside fp16's normal range for their lo parts; what is smaller keeps
  // an absolute error of 2^-39 of the bound (f16_dev.h) -- against a dQ that is a sum over S such terms.  (The dK / dV kernel,
  // which has the registers, follows the running maximum instead: ds_row_scale.)
  float lse_r[4], del_r[4], sds[4];
  {
    const float vmax = scblock[(1 * H + h) * SCW + 2] * drop_scale;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(tile * ABM + 16 * w + 4 * q + r, N - 1);
      lse_r[r] = lse[(size_t)row * H + h];
      del_r[r] = delta[(size_t)row * H + h];
      sds[r] = scale_for_absmax((__shfl(l1, 4 * q + r) * vmax + fabsf(del_r[r])) * scale);
    }
  }
  f32x4 acc_dq[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) acc_dq[nt] = {0.f, 0.f, 0.f, 0.f};

  const u16* kbase = Knat + (size_t)h * 2 * Sp * AE;
  const u16* vbase = Vnat + (size_t)h * 2 * Sp * AE;
  const u16* tbase = Ktr + (size_t)h * 2 * AE * Sp;
  int dst_k[2], dst_t[2];                                           // LDS offsets inside a part's image
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int u = tid + 256 * it;
    const int k = u >> 4, s = u & 15, lrow = (k & 1) * 16 + (k >> 1);
    dst_k[it] = lrow * 256 + ((s ^ (lrow & 15)) << 4);
    const int e = u >> 2, s4 = u & 3;
    dst_t[it] = e * 64 + ((s4 ^ swz64(e)) << 4);
  }
  const int nchunk = Sp / 32;
  unsigned char* Pw = Vl_ + w * 2 * 16 * 64;
  const unsigned row_c0 = (unsigned)(tile * ABM + 16 * w + 4 * q);

  for (int c = 0; c < nchunk; ++c) {
    u32x4 pre[8];                                                   // other resident workgroups cover this latency
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int u = tid + 256 * it;
        const size_t nat_off = ((size_t)part * Sp + c * 32 + (u >> 4)) * AE + 8 * (u & 15);
        pre[part * 2 + it] = *reinterpret_cast<const u32x4*>(kbase + nat_off);
        pre[4 + part * 2 + it] = *reinterpret_cast<const u32x4*>(vbase + nat_off);
      }
    __syncthreads();                                               // previous chunk's images and dS tiles consumed
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        *reinterpret_cast<u32x4*>(Kl_ + part * 32 * 256 + dst_k[it]) = pre[part * 2 + it];
        *reinterpret_cast<u32x4*>(Vl_ + part * 32 * 256 + dst_k[it]) = pre[4 + part * 2 + it];
      }
    __syncthreads();
    u32x4 pret[4];                                                  // K^T rows: in flight under the S / dP products
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int u = tid + 256 * it;
        pret[part * 2 + it] = *reinterpret_cast<const u32x4*>(tbase + ((size_t)part * AE + (u >> 2)) * Sp + c * 32 + 8 * (u & 3));
      }

    f32x4 acc_s[2], acc_dp[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { acc_s[nt] = {0.f, 0.f, 0.f, 0.f}; acc_dp[nt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int off = (16 * nt + j) * 256 + (((4 * ks + q) ^ j) << 4);
        const u32x4 kh = *reinterpret_cast<const u32x4*>(Kl_ + off);
        const u32x4 kl = *reinterpret_cast<const u32x4*>(Kl_ + 32 * 256 + off);
        acc_s[nt] = mfma_h<true, IN_LO>(qh[ks], ql[ks], kh, kl, acc_s[nt]);
        const u32x4 vh = *reinterpret_cast<const u32x4*>(Vl_ + off);
        const u32x4 vl = *reinterpret_cast<const u32x4*>(Vl_ + 32 * 256 + off);
        acc_dp[nt] = mfma_h<IN_LO, IN_LO>(dh[ks], dl[ks], vh, vl, acc_dp[nt]);
      }
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int it = 0; it < 2; ++it) *reinterpret_cast<u32x4*>(Tl_ + part * 128 * 64 + dst_t[it]) = pret[part * 2 + it];
    __syncthreads();                                               // K^T rows visible; every wave is done with the V rows: their region takes the dS tiles
    const int key0 = c * 32 + 2 * j;
    const bool tail = c == nchunk - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float ds[2];
      unsigned x = 0xFFFFFFFFu;
      if (drop_thresh) x = attn_hash_pair(attn_rowhead(seed, row_c0 + r, (unsigned)h), (unsigned)(c * 16 + j));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float pr = __expf(acc_s[nt][r] * cs - lse_r[r]);
        if (tail && key0 + nt >= S) pr = 0.f;
        float dp = acc_dp[nt][r] * cd;
        if (drop_thresh) dp = (((nt ? x >> 16 : x & 0xFFFFu)) >= drop_thresh) ? dp * drop_scale : 0.f;
        ds[nt] = pr * (dp - del_r[r]) * scale;
      }
      const u32x2 sp = split2h(ds[0] * sds[r], ds[1] * sds[r]);
      const int prow = 4 * q + r;
      const int off = prow * 64 + (((j >> 2) ^ swz64(prow)) << 4) + ((j & 3) << 2);
      *reinterpret_cast<unsigned*>(Pw + off) = sp[0];
      *reinterpret_cast<unsigned*>(Pw + 16 * 64 + off) = sp[1];
    }
    {
      const int poff = j * 64 + ((q ^ swz64(j)) << 4);
      const u32x4 ph = *reinterpret_cast<const u32x4*>(Pw + poff);
      const u32x4 pl = *reinterpret_cast<const u32x4*>(Pw + 16 * 64 + poff);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int toff = (16 * nt + j) * 64 + ((q ^ swz64(j)) << 4);
        const u32x4 th = *reinterpret_cast<const u32x4*>(Tl_ + toff);
        const u32x4 tl = *reinterpret_cast<const u32x4*>(Tl_ + 128 * 64 + toff);
        acc_dq[nt] = mfma_h<true, IN_LO>(ph, pl, th, tl, acc_dq[nt]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = tile * ABM + 16 * w + 4 * q + r;
    if (row < N) {
      const float k = inv_pow2(sds[r]) * inv_k;                     // accumulator units: s_dS[row] s_K
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) st1(dQ + (size_t)row * rs + (size_t)h * AE + 16 * nt + j, acc_dq[nt][r] * k);
    }
  }
}

// dK/dV kernel: workgroup = (16 NW-key chunk, head, row split), NW waves, wave w owns 16 keys: their K and V rows are MFMA A
// fragments in registers, dK / dV live in accumulators.  The query-row tiles t = split, split + nsplit, ... (32 rows each) are
// staged from the Q / dO images: natural rows for S^T = K Q^T and dP^T = V dO^T, transposed rows for dV += (P o M)^T dO
// and dK += dS^T Q.  The partial dK / dV go to slab `split` of dKp / dVp ([nsplit][S][H][E]); the caller adds the slabs
// in a fixed order.
// Every workgroup stages every one of its row tiles (64 KB of images per tile) whatever its number of keys, so the launch's
// L2 -> LDS traffic is (key chunks x heads) x 8.7 MB: 1.67 GB with 64-key chunks at B = 128 -- that, not the matrix work,
// set the 215 us of rounds 1-2, and the 8 row splits that filled the chip with 1 536 small workgroups wrote 8 partial slabs
// (96 MB for 12 MB of dK / dV).  NW = 8 (default): 128-key chunks halve the staging traffic, 8 waves share each staged tile,
// and 96 (chunk, head) units x 2 row splits = 192 workgroups of 512 threads fill the chip in one round with TWO slabs.
// NW = 4 (HOPMI_ATTN_DKV_WAVES=4): the previous form (64-key chunks, 8 splits, 2 workgroups per CU).
template <typename TIO, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void reprog_attn_bwd_dkv_kernel(const TIO* __restrict__ K, const TIO* __restrict__ Vv,
                                                                  const u16* __restrict__ Qnat, const u16* __restrict__ Qtr,
                                                                  const u16* __restrict__ Dnat, const u16* __restrict__ Dtr,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  float* __restrict__ dKp, float* __restrict__ dVp,
                                                                  const float* __restrict__ scblock, int N, int Np,
                                                                  int S, int H, int nsplit, float scale, unsigned drop_thresh,
                                                                  float drop_scale, unsigned seed, const unsigned* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  constexpr bool IN_LO = sizeof(TIO) == 4;
  constexpr int KVK = 16 * NW, NT = 64 * NW, NIT = 512 / NT;       // keys per workgroup, threads, staging iterations per image part
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Ql_ = lds;                        // [2 parts][32 rows][256 B]   (scaled) Q rows
  unsigned char* Dl_ = Ql_ + 2 * 32 * 256;         //                            dO rows
  unsigned char* QT_ = Dl_ + 2 * 32 * 256;         // [2 parts][128 rows][64 B]   (scaled) Q^T
  unsigned char* DT_ = QT_ + 2 * 128 * 64;         //                            dO^T
  unsigned char* Pl_ = DT_ + 2 * 128 * 64;         // [NW waves][2 tiles][2 parts][16][64 B]   (P o M)^T and dS^T
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int nchunk = (S + KVK - 1) / KVK;
  const int h = blockIdx.x % H, chunk = (blockIdx.x / H) % nchunk, split = blockIdx.x / (H * nchunk);
  const size_t rs = (size_t)H * AE;
  const int key_a = min(chunk * KVK + 16 * w + j, S - 1);        // A-layout key row of this lane

  u32x4 kh[4], kl[4], vh[4], vl[4];
  // the (tensor, head) scales of K, V (the A fragments below) and of the Q and dO images
  const float s_k = scblock[(0 * H + h) * SCW], s_v = scblock[(1 * H + h) * SCW];
  const float inv_q = scblock[(2 * H + h) * SCW + 1], inv_d = scblock[(3 * H + h) * SCW + 1];
  const float cs = scblock[(0 * H + h) * SCW + 1] * inv_q * scale, cd = scblock[(1 * H + h) * SCW + 1] * inv_d;   // accumulator -> score / dP
  {
    const TIO* kp = K + (size_t)key_a * rs + (size_t)h * AE + 8 * q;
    const TIO* vp = Vv + (size_t)key_a * rs + (size_t)h * AE + 8 * q;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const Split8 f = split8h(ld4(kp + 32 * ks), ld4(kp + 32 * ks + 4), s_k);
      kh[ks] = f.hi; kl[ks] = f.lo;
      const Split8 g = split8h(ld4(vp + 32 * ks), ld4(vp + 32 * ks + 4), s_v);
      vh[ks] = g.hi; vl[ks] = g.lo;
    }
  }
  float sds[4] = {DS_SCALE0, DS_SCALE0, DS_SCALE0, DS_SCALE0};         // running power-of-two scale of dS^T per key row (ds_row_scale)
  f32x4 acc_dk[8], acc_dv[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) { acc_dk[nt] = {0.f, 0.f, 0.f, 0.f}; acc_dv[nt] = {0.f, 0.f, 0.f, 0.f}; }

  const u16* qn = Qnat + (size_t)h * 2 * Np * AE;
  const u16* dn = Dnat + (size_t)h * 2 * Np * AE;
  const u16* qt = Qtr + (size_t)h * 2 * AE * Np;
  const u16* dt = Dtr + (size_t)h * 2 * AE * Np;
  int dst_k[NIT], dst_t[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int u = tid + NT * it;
    const int k = u >> 4, s = u & 15, lrow = (k & 1) * 16 + (k >> 1);
    dst_k[it] = lrow * 256 + ((s ^ (lrow & 15)) << 4);
    const int e = u >> 2, s4 = u & 3;
    dst_t[it] = e * 64 + ((s4 ^ swz64(e)) << 4);
  }
  const int ntile = Np / 32;
  unsigned char* Pw = Pl_ + w * 2 * 2 * 16 * 64;   // (P o M)^T: hi, lo; then dS^T: hi, lo
  const int key_c0 = chunk * KVK + 16 * w + 4 * q; // C-layout key rows 4q + r of this wave's tile

  for (int t = split; t < ntile; t += nsplit) {
    u32x4 pre[8 * NIT];
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int u = tid + NT * it;
        const size_t nat_off = ((size_t)part * Np + t * 32 + (u >> 4)) * AE + 8 * (u & 15);
        const size_t tr_off = ((size_t)part * AE + (u >> 2)) * Np + t * 32 + 8 * (u & 3);
        pre[part * NIT + it] = *reinterpret_cast<const u32x4*>(qn + nat_off);
        pre[2 * NIT + part * NIT + it] = *reinterpret_cast<const u32x4*>(dn + nat_off);
        pre[4 * NIT + part * NIT + it] = *reinterpret_cast<const u32x4*>(qt + tr_off);
        pre[6 * NIT + part * NIT + it] = *reinterpret_cast<const u32x4*>(dt + tr_off);
      }
    // lane holds key rows 4q + r, query rows row0 + 2j + nt
    const int row0 = t * 32 + 2 * j;
    float lrow[2], drow[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int rr = min(row0 + nt, N - 1);
      lrow[nt] = lse[(size_t)rr * H + h];
      drow[nt] = delta[(size_t)rr * H + h];
    }
    __syncthreads();
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        *reinterpret_cast<u32x4*>(Ql_ + part * 32 * 256 + dst_k[it]) = pre[part * NIT + it];
        *reinterpret_cast<u32x4*>(Dl_ + part * 32 * 256 + dst_k[it]) = pre[2 * NIT + part * NIT + it];
        *reinterpret_cast<u32x4*>(QT_ + part * 128 * 64 + dst_t[it]) = pre[4 * NIT + part * NIT + it];
        *reinterpret_cast<u32x4*>(DT_ + part * 128 * 64 + dst_t[it]) = pre[6 * NIT + part * NIT + it];
      }
    __syncthreads();

    // S^T[16 keys][32 rows] = scale K_w Q^T ;  dP^T = V_w dO^T
    f32x4 acc_s[2], acc_dp[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { acc_s[nt] = {0.f, 0.f, 0.f, 0.f}; acc_dp[nt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int off = (16 * nt + j) * 256 + (((4 * ks + q) ^ j) << 4);
        const u32x4 bqh = *reinterpret_cast<const u32x4*>(Ql_ + off);
        const u32x4 bql = *reinterpret_cast<const u32x4*>(Ql_ + 32 * 256 + off);
        acc_s[nt] = mfma_h<IN_LO, true>(kh[ks], kl[ks], bqh, bql, acc_s[nt]);
        const u32x4 bdh = *reinterpret_cast<const u32x4*>(Dl_ + off);
        const u32x4 bdl = *reinterpret_cast<const u32x4*>(Dl_ + 32 * 256 + off);
        acc_dp[nt] = mfma_h<IN_LO, IN_LO>(vh[ks], vl[ks], bdh, bdl, acc_dp[nt]);
      }
    unsigned xk[2][2];                                              // [row nt][key pair]: keys key_c0 + 0/1 and key_c0 + 2/3
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const unsigned rh = attn_rowhead(seed, (unsigned)(row0 + nt), (unsigned)h);
      xk[nt][0] = attn_hash_pair(rh, (unsigned)(key_c0 >> 1));
      xk[nt][1] = attn_hash_pair(rh, (unsigned)(key_c0 >> 1) + 1u);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = key_c0 + r;
      float pm[2], ds[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int row = row0 + nt;
        const float pr = (key < S && row < N) ? __expf(acc_s[nt][r] * cs - lrow[nt]) : 0.f;
        float keepf = 1.f;
        if (drop_thresh) keepf = (((r & 1) ? xk[nt][r >> 1] >> 16 : xk[nt][r >> 1] & 0xFFFFu) >= drop_thresh) ? drop_scale : 0.f;
        pm[nt] = pr * keepf;                                          // (P o M)^T
        ds[nt] = pr * (acc_dp[nt][r] * cd * keepf - drow[nt]);        // dS^T / scale (the scale joins in the epilogue)
      }
      const float sd = ds_row_scale(ds[0], ds[1], sds[r], acc_dk, r);  // (key row r of the tile: running scale over the query rows)
      const u32x2 sp = split2h(pm[0] * H_UNIT_SCALE, pm[1] * H_UNIT_SCALE), ss = split2h(ds[0] * sd, ds[1] * sd);
      const int prow = 4 * q + r;
      const int off = prow * 64 + (((j >> 2) ^ swz64(prow)) << 4) + ((j & 3) << 2);
      *reinterpret_cast<unsigned*>(Pw + off) = sp[0];
      *reinterpret_cast<unsigned*>(Pw + 16 * 64 + off) = sp[1];
      *reinterpret_cast<unsigned*>(Pw + 2 * 16 * 64 + off) = ss[0];
      *reinterpret_cast<unsigned*>(Pw + 3 * 16 * 64 + off) = ss[1];
    }
    // dV_w[16 keys][128] += (P o M)^T[16][32 rows] dO[32 rows][128];  dK_w += dS^T Q (x scale in the epilogue)    (tiles are wave-private)
    {
      const int poff = j * 64 + ((q ^ swz64(j)) << 4);
      const u32x4 ph = *reinterpret_cast<const u32x4*>(Pw + poff);
      const u32x4 pl = *reinterpret_cast<const u32x4*>(Pw + 16 * 64 + poff);
      const u32x4 sh = *reinterpret_cast<const u32x4*>(Pw + 2 * 16 * 64 + poff);
      const u32x4 sl = *reinterpret_cast<const u32x4*>(Pw + 3 * 16 * 64 + poff);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int toff = (16 * nt + j) * 64 + ((q ^ swz64(j)) << 4);
        const u32x4 dth = *reinterpret_cast<const u32x4*>(DT_ + toff);
        const u32x4 dtl = *reinterpret_cast<const u32x4*>(DT_ + 128 * 64 + toff);
        acc_dv[nt] = mfma_h<true, IN_LO>(ph, pl, dth, dtl, acc_dv[nt]);
        const u32x4 qth = *reinterpret_cast<const u32x4*>(QT_ + toff);
        const u32x4 qtl = *reinterpret_cast<const u32x4*>(QT_ + 128 * 64 + toff);
        acc_dk[nt] = mfma_h<true, true>(sh, sl, qth, qtl, acc_dk[nt]);
      }
    }
  }
  float* dK = dKp + (size_t)split * S * rs;
  float* dV = dVp + (size_t)split * S * rs;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = key_c0 + r;
    if (key < S) {
      const float kk = inv_pow2(sds[r]) * inv_q * scale, kv = H_UNIT_INV * inv_d;   // accumulator units: s_dS[key] s_Q / scale and 2^14 s_dO
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        dK[(size_t)key * rs + (size_t)h * AE + 16 * nt + j] = acc_dk[nt][r] * kk;
        dV[(size_t)key * rs + (size_t)h * AE + 16 * nt + j] = acc_dv[nt][r] * kv;
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
  return (size_t)4 * 2 * H * attn_sp(S) * AE * sizeof(u16) + attn_scale_floats(H) * sizeof(float);      // 4 images + the scale block
}

static int attn_args_ok(const char* what, int N, int S, int H, int E, float p_drop, int dtype) {
  if (E != AE || N <= 0 || S <= 0 || H <= 0) {
    set_error("%s: need head dim 128 and positive sizes (N=%d S=%d H=%d E=%d)", what, N, S, H, E);
    return HOPMI_EINVAL;
  }
  if (!(p_drop >= 0.f && p_drop < 1.f)) { set_error("%s: p_drop=%f outside [0,1)", what, p_drop); return HOPMI_EINVAL; }
  // the dropped-out probabilities (P o M) <= 1 / (1 - p_drop) are split at the FIXED scale 2^14 (forward P V, backward dV): from
  // p_drop = 0.75 on, 4 x 2^14 leaves fp16's range and the row-maximum element (P == 1) would become infinity.  The reference trains
  // with 0.1 (HOP.py:256); refuse what this kernel cannot represent instead of returning NaN.
  if (p_drop >= 0.75f) {
    set_error("%s: p_drop=%f: the fp16 hi/lo form of the dropped-out probabilities holds p_drop < 0.75 (use ops.strict_fp32 for more)", what, p_drop);
    return HOPMI_EINVAL;
  }
  if (dtype != HOPMI_F32 && dtype != HOPMI_BF16) { set_error("%s: dtype %d (0 = fp32, 1 = bf16)", what, dtype); return HOPMI_EINVAL; }
  return HOPMI_OK;
}

template <typename TIO>
static int launch_reprog_attn_fwd(const void* q_, const void* k_, const void* v_, void* o_, float* lse, void* ws, int N, int S, int H,
                                  float scale, float p_drop, unsigned seed, const unsigned* seed_dev, hipStream_t st) {
  const TIO *q = static_cast<const TIO*>(q_), *k = static_cast<const TIO*>(k_), *v = static_cast<const TIO*>(v_);
  TIO* o = static_cast<TIO*>(o_);
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 65536.0) : 0u;   // 16-bit keep fields (attn_hash_pair)
  const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int Sp = attn_sp(S);
  const size_t img = (size_t)2 * H * Sp * AE;                       // elements per image
  u16* knat = static_cast<u16*>(ws);
  u16* vtr = knat + 3 * img;
  float* scblock = reinterpret_cast<float*>(knat + 4 * img);
  unsigned* parts = reinterpret_cast<unsigned*>(scblock + (size_t)ATENS * H * SCW);
  AbsmaxArgs am{};
  am.X[0] = k; am.rows[0] = S; am.X[1] = v; am.rows[1] = S;
  hipLaunchKernelGGL(attn_absmax_kernel<TIO>, dim3(APARTS, H, 2), dim3(256), 0, st, am, H, parts);
  ImageArgs ia{};
  ia.X[0] = k; ia.nat[0] = knat; ia.tr[0] = nullptr; ia.tensor[0] = 0;
  ia.X[1] = v; ia.nat[1] = nullptr; ia.tr[1] = vtr; ia.tensor[1] = 1;
  hipLaunchKernelGGL(attn_images_kernel<TIO>, dim3((Sp / 32) * H, 2), dim3(256), 0, st, ia, S, Sp, H, scblock);
  if (int e = check_launch("hopmi_reprog_attn_fwd(images)")) return e;
  const int ntile = (N + ABM - 1) / ABM;
  const size_t lds = (size_t)2 * 32 * 256 + 2 * 128 * 64 + 4 * 2 * 16 * 64;
  hipLaunchKernelGGL(reprog_attn_fwd_kernel<TIO>, dim3(ntile * H), dim3(256), lds, st, q, knat, vtr, o, lse, scblock, N, S, Sp, H, scale,
                     thresh, dscale, seed, seed_dev);
  return check_launch("hopmi_reprog_attn_fwd");
}

extern "C" int hopmi_reprog_attn_fwd_dt(const void* q, const void* k, const void* v, void* o, int dtype, float* lse, void* ws,
                                        int N, int S, int H, int E, float scale, float p_drop, unsigned seed, const unsigned* seed_dev,
                                        void* stream) {
  if (!q || !k || !v || !o || !lse || !ws) { set_error("hopmi_reprog_attn_fwd: null pointer argument"); return HOPMI_EINVAL; }
  if (int e = attn_args_ok("hopmi_reprog_attn_fwd", N, S, H, E, p_drop, dtype)) return e;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == HOPMI_BF16 ? launch_reprog_attn_fwd<__bf16>(q, k, v, o, lse, ws, N, S, H, scale, p_drop, seed, seed_dev, st)
                             : launch_reprog_attn_fwd<float>(q, k, v, o, lse, ws, N, S, H, scale, p_drop, seed, seed_dev, st);
}

extern "C" int hopmi_reprog_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, void* ws,
                                     int N, int S, int H, int E, float scale, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream) {
  return hopmi_reprog_attn_fwd_dt(q, k, v, o, HOPMI_F32, lse, ws, N, S, H, E, scale, p_drop, seed, seed_dev, stream);
}

// dK/dV kernel form: waves per workgroup (16 keys each) and the number of query-row splits = partial slabs (see the kernel)
static int attn_dkv_waves() { return env_int("HOPMI_ATTN_DKV_WAVES", 8) == 4 ? 4 : 8; }
extern "C" int hopmi_reprog_attn_bwd_splits(void) { return attn_dkv_waves() == 4 ? 8 : env_int("HOPMI_ATTN_DKV_SPLITS", 2) > 0 ? env_int("HOPMI_ATTN_DKV_SPLITS", 2) : 2; }

static int attn_np(int N) { return (N + 31) / 32 * 32; }
extern "C" size_t hopmi_reprog_attn_bwd_ws_bytes(int N, int S, int H, int E) {
  if (N <= 0 || S <= 0 || H <= 0 || E != AE) return 0;
  return hopmi_reprog_attn_ws_bytes(S, H, E) + (size_t)4 * 2 * H * attn_np(N) * AE * sizeof(u16);
}

template <typename TIO>
static int launch_reprog_attn_bwd(const void* q_, const void* k_, const void* v_, const void* d_o_, const float* lse, const float* delta,
                                  void* dq_, float* dk, float* dv, void* ws, int N, int S, int H, float scale, float p_drop,
                                  unsigned seed, const unsigned* seed_dev, hipStream_t st) {
  const TIO *q = static_cast<const TIO*>(q_), *k = static_cast<const TIO*>(k_), *v = static_cast<const TIO*>(v_);
  const TIO* d_o = static_cast<const TIO*>(d_o_);
  TIO* dq = static_cast<TIO*>(dq_);
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 65536.0) : 0u;   // 16-bit keep fields (attn_hash_pair)
  const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int Sp = attn_sp(S), Np = attn_np(N);
  const size_t kimg = (size_t)2 * H * Sp * AE, qimg = (size_t)2 * H * Np * AE;
  u16* knat = static_cast<u16*>(ws);
  u16* ktr = knat + kimg;
  u16* vnat = ktr + kimg;
  float* scblock = reinterpret_cast<float*>(knat + 4 * kimg);
  unsigned* parts = reinterpret_cast<unsigned*>(scblock + (size_t)ATENS * H * SCW);
  u16* qnat = reinterpret_cast<u16*>(static_cast<unsigned char*>(ws) + hopmi_reprog_attn_ws_bytes(S, H, AE));
  u16* qtr = qnat + qimg;
  u16* dnat = qtr + qimg;
  u16* dtr = dnat + qimg;
  AbsmaxArgs am{};
  am.X[0] = k; am.rows[0] = S; am.X[1] = v; am.rows[1] = S; am.X[2] = q; am.rows[2] = N; am.X[3] = d_o; am.rows[3] = N;
  hipLaunchKernelGGL(attn_absmax_kernel<TIO>, dim3(APARTS, H, 4), dim3(256), 0, st, am, H, parts);
  ImageArgs ia{};
  ia.X[0] = k; ia.nat[0] = knat; ia.tr[0] = ktr; ia.tensor[0] = 0;
  ia.X[1] = v; ia.nat[1] = vnat; ia.tr[1] = nullptr; ia.tensor[1] = 1;
  hipLaunchKernelGGL(attn_images_kernel<TIO>, dim3((Sp / 32) * H, 2), dim3(256), 0, st, ia, S, Sp, H, scblock);
  ImageArgs ib{};
  ib.X[0] = q; ib.nat[0] = qnat; ib.tr[0] = qtr; ib.tensor[0] = 2;
  ib.X[1] = d_o; ib.nat[1] = dnat; ib.tr[1] = dtr; ib.tensor[1] = 3;
  hipLaunchKernelGGL(attn_images_kernel<TIO>, dim3((Np / 32) * H, 2), dim3(256), 0, st, ib, N, Np, H, scblock);
  if (int e = check_launch("hopmi_reprog_attn_bwd(images)")) return e;
  const size_t lds_q = (size_t)2 * 2 * 32 * 256 + 2 * 128 * 64;
  hipLaunchKernelGGL(reprog_attn_bwd_dq_kernel<TIO>, dim3(((N + ABM - 1) / ABM) * H), dim3(256), lds_q, st, q, knat, vnat, ktr, d_o,
                     lse, delta, dq, scblock, N, S, Sp, H, scale, thresh, dscale, seed, seed_dev);
  if (int e = check_launch("hopmi_reprog_attn_bwd(dq)")) return e;
  const int nsplit = hopmi_reprog_attn_bwd_splits();
  if (attn_dkv_waves() == 8) {
    const size_t lds_kv = (size_t)2 * 2 * 32 * 256 + 2 * 2 * 128 * 64 + 8 * 2 * 2 * 16 * 64;        // 96 KB: one workgroup per CU
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&reprog_attn_bwd_dkv_kernel<TIO, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_kv) != hipSuccess) (void)hipGetLastError();
      attr_done = true;
    }
    hipLaunchKernelGGL((reprog_attn_bwd_dkv_kernel<TIO, 8>), dim3(((S + 127) / 128) * H * nsplit), dim3(512), lds_kv, st, k, v, qnat, qtr,
                       dnat, dtr, lse, delta, dk, dv, scblock, N, Np, S, H, nsplit, scale, thresh, dscale, seed, seed_dev);
  } else {
    const size_t lds_kv = (size_t)2 * 2 * 32 * 256 + 2 * 2 * 128 * 64 + 4 * 2 * 2 * 16 * 64;
    hipLaunchKernelGGL((reprog_attn_bwd_dkv_kernel<TIO, 4>), dim3(((S + 63) / 64) * H * nsplit), dim3(256), lds_kv, st, k, v, qnat, qtr,
                       dnat, dtr, lse, delta, dk, dv, scblock, N, Np, S, H, nsplit, scale, thresh, dscale, seed, seed_dev);
  }
  return check_launch("hopmi_reprog_attn_bwd(dkv)");
}

extern "C" int hopmi_reprog_attn_bwd_dt(const void* q, const void* k, const void* v, const void* d_o, int dtype, const float* lse,
                                        const float* delta, void* dq, float* dk, float* dv, void* ws, int N, int S, int H, int E,
                                        float scale, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream) {
  if (!q || !k || !v || !d_o || !lse || !delta || !dq || !dk || !dv || !ws) { set_error("hopmi_reprog_attn_bwd: null pointer argument"); return HOPMI_EINVAL; }
  if (int e = attn_args_ok("hopmi_reprog_attn_bwd", N, S, H, E, p_drop, dtype)) return e;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == HOPMI_BF16
             ? launch_reprog_attn_bwd<__bf16>(q, k, v, d_o, lse, delta, dq, dk, dv, ws, N, S, H, scale, p_drop, seed, seed_dev, st)
             : launch_reprog_attn_bwd<float>(q, k, v, d_o, lse, delta, dq, dk, dv, ws, N, S, H, scale, p_drop, seed, seed_dev, st);
}

extern "C" int hopmi_reprog_attn_bwd(const float* q, const float* k, const float* v, const float* d_o, const float* lse,
                                     const float* delta, float* dq, float* dk, float* dv, void* ws, int N, int S, int H, int E,
                                     float scale, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream) {
  return hopmi_reprog_attn_bwd_dt(q, k, v, d_o, HOPMI_F32, lse, delta, dq, dk, dv, ws, N, S, H, E, scale, p_drop, seed, seed_dev, stream);
}

HOPMI_SPLIT_STATUS_SETTER(attn)
