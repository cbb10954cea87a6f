// Self-attention of the frozen BERT encoder that HOP runs its reprogrammed embeddings through
// (reference: model/HOP.py:204 -> transformers BertSelfAttention; 34 tokens, 12 heads x 64):
//
//   P = dropout(softmax(Q K^T / 8));  O = P V        per (clip b, head h), L <= 64 tokens
//
// The library path ran this as a generic memory-efficient attention (41 us forward / 121 us backward per layer
// for 0.45 GFLOP) plus layout copies on both sides.  Here one workgroup owns one (b, h), one wave per 16 query
// rows (forward) or per 16 rows / 16 keys (backward): Q, K, V are read straight out of the fused QKV GEMM output
// [B][L][3][H][64]; score strips stay in accumulator registers, the softmax reduces over DPP rows, O is written
// in the [B][L][H*64] layout the output projection wants, and the backward recomputes P and writes dQ, dK, dV
// straight into the [B][L][3][H][64] gradient of the QKV GEMM: no transposes, no concatenations, no saved
// probabilities, no cross-workgroup reductions (bitwise reproducible).  All contractions on exact-fp32 MFMA
// (16x16x4, outputs transposed so that every global store is 16 bytes per lane); dropout is the stateless
// (seed,row,head,key) hash of attn.hip.
#define HOPMI_FILE_ID 5          // (diagnostic build: common.h, split_check)
#include "attn_dev.h"
#include "io_dev.h"
#include "f16_dev.h"

namespace hopmi {

constexpr int BD = 64;             // head dim
constexpr int BLD = BD + 4;        // LDS row stride of the Q / K / V / dO images
constexpr int BMAXL = 64;

// rows [0, LP) x 64 floats of one of q/k/v (or d_o) of (b, h) -> LDS image [LP][BLD], rows >= L zeroed
template <int NT, typename T>
__device__ __forceinline__ void stage_head(float* dst, const T* __restrict__ src, size_t row_stride, int L, int LP, int tid) {
  for (int idx = tid; idx < LP * 16; idx += NT) {
    const int row = idx >> 4, c4 = idx & 15;
    const float4 v = ld4(src + (size_t)min(row, L - 1) * row_stride + 4 * c4);
    *reinterpret_cast<float4*>(dst + row * BLD + 4 * c4) = (row < L) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

#define HOPMI_MFMA4(acc_, a_, b_)          \
  acc_ = mfma16((a_).x, (b_).x, acc_);     \
  acc_ = mfma16((a_).y, (b_).y, acc_);     \
  acc_ = mfma16((a_).z, (b_).z, acc_);     \
  acc_ = mfma16((a_).w, (b_).w, acc_);

// C^T tile product used for every "probabilities x values"-shaped contraction:
//   out[di] (D[i = d][j = n]) += sum_k X[k][16 di + i] * T[n = j][k],   k = 0 .. 16*MT
// X: an LDS head image (row-major over k, stride BLD), T: the wave's private [16][PLD] tile (K-contiguous).
// The lane ends up with 4 consecutive d of row/key n = j: a 16-byte store.
template <int MT>
__device__ __forceinline__ void xt_product(f32x4 (&o)[4], const float* X, const float* Tw, int PLD, int q, int j) {
#pragma unroll
  for (int di = 0; di < 4; ++di) o[di] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ii = 0; ii < MT; ++ii) {
    const float4 bv = *reinterpret_cast<const float4*>(Tw + j * PLD + 16 * ii + 4 * q);
#pragma unroll
    for (int di = 0; di < 4; ++di) {
      const float* ap = X + (16 * ii + 4 * q) * BLD + 16 * di + j;
      o[di] = mfma16(ap[0], bv.x, o[di]);
      o[di] = mfma16(ap[BLD], bv.y, o[di]);
      o[di] = mfma16(ap[2 * BLD], bv.z, o[di]);
      o[di] = mfma16(ap[3 * BLD], bv.w, o[di]);
    }
  }
}

// Forward.  Workgroup = (b, h), MT = ceil(L / 16) waves; wave w owns query rows [16w, 16w + 16) end to end:
// its score strip stays in accumulator registers, the softmax reduces over the 16 lanes of a DPP row, the
// dropped-out probabilities pass through a wave-private LDS tile to become an MFMA operand.  One barrier.
// (round 5) The output handed on as the attention-output GEMM's operand IMAGE (hopmi_gemm_f16x2_ab_ep; gemm.hip f16_blk layout) beside
// the fp32 values: a row of the output spans the H workgroups of its clip, so its scale cannot come from its maximum -- it comes from a
// BOUND every one of them can compute: O = dropout(P) V is a sub-convex combination of V's rows times 1 / (1 - p), so
// |O[b, :, :]| <= max |V[b, :, :]| / (1 - p), and max |V| of the clip is the maximum of the QKV product's partial row maxima
// (c_rowmax of hopmi_gemm_f16x2(_ab_ep): [tiles][M], the V columns are tiles vt0 .. vt1 - 1) over the clip's L rows.  One scale per
// clip: typical outputs sit 2^1 ... 2^4 below it.
struct BertAttnImage { const float* v_rowmax; int vt0, vt1, M; _Float16* image; float* scales; };

template <int MT, typename T>
__global__ __launch_bounds__(64 * MT) void bert_attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, int L, int H,
                                                                unsigned thresh, float dscale, unsigned seed, const unsigned* __restrict__ seed_dev,
                                                                BertAttnImage im) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  constexpr int LP = 16 * MT, PLD = LP + 4, NT = 64 * MT;
  __shared__ unsigned s_vmax[MT];
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = Ks + LP * BLD;
  float* Pw = Vs + LP * BLD + (threadIdx.x >> 6) * 16 * PLD;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const size_t rs = (size_t)3 * H * BD;
  const T* base = qkv + (size_t)b * L * rs + (size_t)h * BD;
  stage_head<NT>(Ks, base + (size_t)H * BD, rs, L, LP, tid);
  stage_head<NT>(Vs, base + (size_t)2 * H * BD, rs, L, LP, tid);
  float4 qf[4];
  {
    const T* qp = base + (size_t)min(16 * w + j, L - 1) * rs + 4 * q;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) qf[ii] = ld4(qp + 16 * ii);
  }
  if (im.image != nullptr) {                     // the clip's max |V| from the QKV product's partial row maxima (in front of the barrier)
    unsigned m = 0u;
    const int nv = (im.vt1 - im.vt0) * L;
    for (int i = tid; i < nv; i += NT) {
      const int t = im.vt0 + i / L, r = i - (i / L) * L;
      m = max(m, __float_as_uint(im.v_rowmax[(size_t)t * im.M + (size_t)(blockIdx.x / H) * L + r]) & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((tid & 63) == 0) s_vmax[tid >> 6] = m;
  }
  __syncthreads();
  f32x4 s[MT];
#pragma unroll
  for (int nt = 0; nt < MT; ++nt) s[nt] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) {
      const float4 kb = *reinterpret_cast<const float4*>(Ks + (16 * nt + j) * BLD + 16 * ii + 4 * q);
      HOPMI_MFMA4(s[nt], qf[ii], kb)
    }
  // s[nt][r] = scores of row 16w + 4q + r, key 16nt + j
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float m = -1e30f;
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) {
      s[nt][r] = (16 * nt + j < L) ? s[nt][r] * 0.125f : -1e30f;
      m = fmaxf(m, s[nt][r]);
    }
    m = row16_max(m);
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) {
      s[nt][r] = (16 * nt + j < L) ? __expf(s[nt][r] - m) : 0.f;
      sum += s[nt][r];
    }
    const float inv = dscale / row16_sum(sum);
    const unsigned row = (unsigned)(b * L + 16 * w + 4 * q + r);
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) {
      const bool keep = thresh == 0u || attn_hash(seed, row, (unsigned)h, (unsigned)(16 * nt + j)) >= thresh;
      Pw[(4 * q + r) * PLD + 16 * nt + j] = keep ? s[nt][r] * inv : 0.f;
    }
  }
  f32x4 o[4];
  xt_product<MT>(o, Vs, Pw, PLD, q, j);
  const int row = 16 * w + j;
  if (row < L) {
    T* op = out + ((size_t)(b * L + row) * H + h) * BD + 4 * q;
#pragma unroll
    for (int di = 0; di < 4; ++di) st4(op + 16 * di, make_float4(o[di][0], o[di][1], o[di][2], o[di][3]));
  }
  if (im.image != nullptr) {
    unsigned m = s_vmax[0];
#pragma unroll
    for (int k = 1; k < MT; ++k) m = max(m, s_vmax[k]);
    // (1 / (1 - p) <= 2 and the fp32 accumulation's rounding: the bound times 1.001)
    const unsigned sb = scale_bits_for_max(__float_as_uint(__uint_as_float(m) * dscale * 1.001f) & 0x7fffffffu);
    const float sc = __uint_as_float(sb);
    if (row < L) {
      const int D = H * BD, grow = b * L + row;
      _Float16* hi = im.image;
      _Float16* lo = hi + (size_t)((im.M + 127) / 128 * 128) * D;
#pragma unroll
      for (int di = 0; di < 4; ++di) {
        const Split4 sp = split4h(o[di][0] * sc, o[di][1] * sc, o[di][2] * sc, o[di][3] * sc);
        const size_t at = ((size_t)(grow >> 7) * (D >> 5) + ((h * BD + 16 * di + 4 * q) >> 5)) * 4096 + (size_t)(grow & 127) * 32 +
                          ((h * BD + 16 * di + 4 * q) & 31);
        *reinterpret_cast<u32x2*>(hi + at) = sp.hi;
        *reinterpret_cast<u32x2*>(lo + at) = sp.lo;
      }
    }
    if (h == 0 && tid < L) { im.scales[b * L + tid] = sc; im.scales[im.M + b * L + tid] = inv_scale(sb); }
  }
}

// Backward, two passes over the same LDS images of Q, K, V, dO:
//   pass 1, wave = 16 query rows: scores, softmax statistics (kept in LDS for pass 2), dPd = dO V^T,
//           delta = rowsum(Pd * dPd), dS, dQ = dS K / 8;
//   pass 2, wave = 16 keys: recomputes its columns of P and dPd (as transposed strips, from the statistics),
//           dV = Pd^T dO, dK = dS^T Q / 8.
// Every output element has one owner wave and a fixed summation order: no atomics, bitwise reproducible.
template <int MT, typename T>
__global__ __launch_bounds__(64 * MT) void bert_attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                T* __restrict__ dqkv, int L, int H, unsigned thresh,
                                                                float dscale, unsigned seed, const unsigned* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // device-side stream position (hipGraph replays advance it)
  constexpr int LP = 16 * MT, PLD = LP + 4, NT = 64 * MT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Ks = Qs + LP * BLD;
  float* Vs = Ks + LP * BLD;
  float* Gs = Vs + LP * BLD;                       // dO
  float* m_s = Gs + LP * BLD;                      // row max of the scaled scores
  float* il_s = m_s + LP;                          // 1 / row sum
  float* dl_s = il_s + LP;                         // delta
  float* Tw = dl_s + LP + (threadIdx.x >> 6) * 16 * PLD;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const size_t rs = (size_t)3 * H * BD;
  const T* base = qkv + (size_t)b * L * rs + (size_t)h * BD;
  stage_head<NT>(Qs, base, rs, L, LP, tid);
  stage_head<NT>(Ks, base + (size_t)H * BD, rs, L, LP, tid);
  stage_head<NT>(Vs, base + (size_t)2 * H * BD, rs, L, LP, tid);
  stage_head<NT>(Gs, dout + (size_t)b * L * H * BD + (size_t)h * BD, (size_t)H * BD, L, LP, tid);
  __syncthreads();
  T* dst = dqkv + (size_t)b * L * rs + (size_t)h * BD;

  // ---------------- pass 1: rows 16w .. 16w + 15 ----------------------------------------------------------
  {
    f32x4 s[MT], d[MT];
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) { s[nt] = {0.f, 0.f, 0.f, 0.f}; d[nt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const float4 qa = *reinterpret_cast<const float4*>(Qs + (16 * w + j) * BLD + 16 * ii + 4 * q);
      const float4 ga = *reinterpret_cast<const float4*>(Gs + (16 * w + j) * BLD + 16 * ii + 4 * q);
#pragma unroll
      for (int nt = 0; nt < MT; ++nt) {
        const float4 kb = *reinterpret_cast<const float4*>(Ks + (16 * nt + j) * BLD + 16 * ii + 4 * q);
        const float4 vb = *reinterpret_cast<const float4*>(Vs + (16 * nt + j) * BLD + 16 * ii + 4 * q);
        HOPMI_MFMA4(s[nt], qa, kb)
        HOPMI_MFMA4(d[nt], ga, vb)
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m = -1e30f;
#pragma unroll
      for (int nt = 0; nt < MT; ++nt) {
        s[nt][r] = (16 * nt + j < L) ? s[nt][r] * 0.125f : -1e30f;
        m = fmaxf(m, s[nt][r]);
      }
      m = row16_max(m);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < MT; ++nt) {
        s[nt][r] = (16 * nt + j < L) ? __expf(s[nt][r] - m) : 0.f;
        sum += s[nt][r];
      }
      const float il = 1.f / row16_sum(sum);
      const int rl = 16 * w + 4 * q + r;
      const unsigned row = (unsigned)(b * L + rl);
      float pd[MT], dl = 0.f;
#pragma unroll
      for (int nt = 0; nt < MT; ++nt) {
        const bool keep = thresh == 0u || attn_hash(seed, row, (unsigned)h, (unsigned)(16 * nt + j)) >= thresh;
        s[nt][r] *= il;                                              // P
        pd[nt] = keep ? s[nt][r] * dscale : 0.f;                     // dropped-out P
        dl += pd[nt] * d[nt][r];
      }
      dl = row16_sum(dl);
      if (j == 0) { m_s[rl] = m; il_s[rl] = il; dl_s[rl] = dl; }
#pragma unroll
      for (int nt = 0; nt < MT; ++nt) Tw[(4 * q + r) * PLD + 16 * nt + j] = (pd[nt] * d[nt][r] - s[nt][r] * dl) * 0.125f;   // dS / 8
    }
    f32x4 o[4];
    xt_product<MT>(o, Ks, Tw, PLD, q, j);                            // dQ^T = K^T dS^T
    const int row = 16 * w + j;
    if (row < L) {
      T* op = dst + (size_t)row * rs + 4 * q;
#pragma unroll
      for (int di = 0; di < 4; ++di) st4(op + 16 * di, make_float4(o[di][0], o[di][1], o[di][2], o[di][3]));
    }
  }
  __syncthreads();                                                   // statistics of every row visible

  // ---------------- pass 2: keys 16w .. 16w + 15 ----------------------------------------------------------
  {
    f32x4 st[MT], dt[MT];
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) { st[nt] = {0.f, 0.f, 0.f, 0.f}; dt[nt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const float4 ka = *reinterpret_cast<const float4*>(Ks + (16 * w + j) * BLD + 16 * ii + 4 * q);
      const float4 va = *reinterpret_cast<const float4*>(Vs + (16 * w + j) * BLD + 16 * ii + 4 * q);
#pragma unroll
      for (int nt = 0; nt < MT; ++nt) {
        const float4 qb = *reinterpret_cast<const float4*>(Qs + (16 * nt + j) * BLD + 16 * ii + 4 * q);
        const float4 gb = *reinterpret_cast<const float4*>(Gs + (16 * nt + j) * BLD + 16 * ii + 4 * q);
        HOPMI_MFMA4(st[nt], ka, qb)
        HOPMI_MFMA4(dt[nt], va, gb)
      }
    }
    // st[nt][r] = score of (row 16nt + j, key 16w + 4q + r); dt likewise for dPd
    float ds[MT][4];
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) {
      const int rl = 16 * nt + j;
      const float m = m_s[rl], il = il_s[rl], dl = dl_s[rl];
      const unsigned row = (unsigned)(b * L + rl);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * w + 4 * q + r;
        const float p = (rl < L && key < L) ? __expf(st[nt][r] * 0.125f - m) * il : 0.f;
        const bool keep = thresh == 0u || attn_hash(seed, row, (unsigned)h, (unsigned)key) >= thresh;
        const float pd = keep ? p * dscale : 0.f;
        ds[nt][r] = (pd * dt[nt][r] - p * dl) * 0.125f;
        Tw[(4 * q + r) * PLD + rl] = pd;                             // [key in strip][row]
      }
    }
    const int key = 16 * w + j;
    f32x4 o[4];
    xt_product<MT>(o, Gs, Tw, PLD, q, j);                            // dV^T = dO^T Pd
    if (key < L) {
      T* op = dst + (size_t)key * rs + 2 * H * BD + 4 * q;
#pragma unroll
      for (int di = 0; di < 4; ++di) st4(op + 16 * di, make_float4(o[di][0], o[di][1], o[di][2], o[di][3]));
    }
#pragma unroll
    for (int nt = 0; nt < MT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Tw[(4 * q + r) * PLD + 16 * nt + j] = ds[nt][r];
    xt_product<MT>(o, Qs, Tw, PLD, q, j);                            // dK^T = Q^T dS / 8
    if (key < L) {
      T* op = dst + (size_t)key * rs + H * BD + 4 * q;
#pragma unroll
      for (int di = 0; di < 4; ++di) st4(op + 16 * di, make_float4(o[di][0], o[di][1], o[di][2], o[di][3]));
    }
  }
}

static int bert_attn_validate(const char* what, int B, int L, int H, float p_drop) {
  if (B <= 0 || L < 1 || L > BMAXL || H < 1 || !(p_drop >= 0.f && p_drop < 1.f)) {
    set_error("%s: bad arguments B=%d L=%d (1..%d) H=%d p_drop=%g", what, B, L, BMAXL, H, (double)p_drop);
    return HOPMI_EINVAL;
  }
  if ((long long)B * H > 0x7fffffffLL || (long long)B * L > 0x7fffffffLL) {
    set_error("%s: B*H or B*L overflows the grid / row index", what);
    return HOPMI_EINVAL;
  }
  return HOPMI_OK;
}

}  // namespace hopmi

using namespace hopmi;

template <typename T>
static void launch_bert_attn_fwd(const void* qkv, void* out, int B, int L, int H, float p_drop, unsigned seed, const unsigned* seed_dev,
                                 hipStream_t st, BertAttnImage im = BertAttnImage{nullptr, 0, 0, 0, nullptr, nullptr}) {
  const int MT = (L + 15) / 16, LP = 16 * MT;
  const size_t lds = ((size_t)2 * LP * BLD + (size_t)LP * (LP + 4)) * sizeof(float);
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const float dscale = 1.f / (1.f - p_drop);
  const T* x = static_cast<const T*>(qkv);
  T* o = static_cast<T*>(out);
  switch (MT) {
    case 1: hipLaunchKernelGGL((bert_attn_fwd_kernel<1, T>), dim3(B * H), dim3(64), lds, st, x, o, L, H, thresh, dscale, seed, seed_dev, im); break;
    case 2: hipLaunchKernelGGL((bert_attn_fwd_kernel<2, T>), dim3(B * H), dim3(128), lds, st, x, o, L, H, thresh, dscale, seed, seed_dev, im); break;
    case 3: hipLaunchKernelGGL((bert_attn_fwd_kernel<3, T>), dim3(B * H), dim3(192), lds, st, x, o, L, H, thresh, dscale, seed, seed_dev, im); break;
    default: hipLaunchKernelGGL((bert_attn_fwd_kernel<4, T>), dim3(B * H), dim3(256), lds, st, x, o, L, H, thresh, dscale, seed, seed_dev, im); break;
  }
}

template <typename T>
static void launch_bert_attn_bwd(const void* qkv, const void* d_out, void* dqkv, int B, int L, int H, float p_drop, unsigned seed,
                                 const unsigned* seed_dev, hipStream_t st) {
  const int MT = (L + 15) / 16, LP = 16 * MT;
  const size_t lds = ((size_t)4 * LP * BLD + 3 * (size_t)LP + (size_t)LP * (LP + 4)) * sizeof(float);
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const float dscale = 1.f / (1.f - p_drop);
  const T* x = static_cast<const T*>(qkv);
  const T* g = static_cast<const T*>(d_out);
  T* d = static_cast<T*>(dqkv);
  switch (MT) {
    case 1: hipLaunchKernelGGL((bert_attn_bwd_kernel<1, T>), dim3(B * H), dim3(64), lds, st, x, g, d, L, H, thresh, dscale, seed, seed_dev); break;
    case 2: hipLaunchKernelGGL((bert_attn_bwd_kernel<2, T>), dim3(B * H), dim3(128), lds, st, x, g, d, L, H, thresh, dscale, seed, seed_dev); break;
    case 3: hipLaunchKernelGGL((bert_attn_bwd_kernel<3, T>), dim3(B * H), dim3(192), lds, st, x, g, d, L, H, thresh, dscale, seed, seed_dev); break;
    default: hipLaunchKernelGGL((bert_attn_bwd_kernel<4, T>), dim3(B * H), dim3(256), lds, st, x, g, d, L, H, thresh, dscale, seed, seed_dev); break;
  }
}

static int bert_dtype_ok(const char* what, int dtype) {
  if (dtype != HOPMI_F32 && dtype != HOPMI_BF16) { set_error("%s: dtype %d (0 = fp32, 1 = bf16)", what, dtype); return HOPMI_EINVAL; }
  return HOPMI_OK;
}

extern "C" int hopmi_bert_attn_fwd_dt(const void* qkv, void* out, int B, int L, int H, float p_drop, unsigned seed, const unsigned* seed_dev,
                                      int dtype, void* stream) {
  if (int e = bert_attn_validate("hopmi_bert_attn_fwd", B, L, H, p_drop)) return e;
  if (int e = bert_dtype_ok("hopmi_bert_attn_fwd_dt", dtype)) return e;
  if (!qkv || !out) { set_error("hopmi_bert_attn_fwd: null pointer argument"); return HOPMI_EINVAL; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == HOPMI_BF16) launch_bert_attn_fwd<__bf16>(qkv, out, B, L, H, p_drop, seed, seed_dev, st);
  else launch_bert_attn_fwd<float>(qkv, out, B, L, H, p_drop, seed, seed_dev, st);
  return check_launch("hopmi_bert_attn_fwd");
}

extern "C" int hopmi_bert_attn_fwd_im(const float* qkv, float* out, const float* v_rowmax, int vt0, int vt1, void* image, float* scales, int B,
                                      int L, int H, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream) {
  if (int e = bert_attn_validate("hopmi_bert_attn_fwd_im", B, L, H, p_drop)) return e;
  if (!qkv || !out || !v_rowmax || !image || !scales || vt0 < 0 || vt1 <= vt0 || (H * BD) % 32) {
    set_error("hopmi_bert_attn_fwd_im: null pointer argument / V tile range [%d, %d) / H * 64 %% 32", vt0, vt1);
    return HOPMI_EINVAL;
  }
  launch_bert_attn_fwd<float>(qkv, out, B, L, H, p_drop, seed, seed_dev, static_cast<hipStream_t>(stream),
                              BertAttnImage{v_rowmax, vt0, vt1, B * L, static_cast<_Float16*>(image), scales});
  return check_launch("hopmi_bert_attn_fwd_im");
}

extern "C" int hopmi_bert_attn_bwd_dt(const void* qkv, const void* d_out, void* dqkv, int B, int L, int H, float p_drop, unsigned seed,
                                      const unsigned* seed_dev, int dtype, void* stream) {
  if (int e = bert_attn_validate("hopmi_bert_attn_bwd", B, L, H, p_drop)) return e;
  if (int e = bert_dtype_ok("hopmi_bert_attn_bwd_dt", dtype)) return e;
  if (!qkv || !d_out || !dqkv) { set_error("hopmi_bert_attn_bwd: null pointer argument"); return HOPMI_EINVAL; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == HOPMI_BF16) launch_bert_attn_bwd<__bf16>(qkv, d_out, dqkv, B, L, H, p_drop, seed, seed_dev, st);
  else launch_bert_attn_bwd<float>(qkv, d_out, dqkv, B, L, H, p_drop, seed, seed_dev, st);
  return check_launch("hopmi_bert_attn_bwd");
}

extern "C" int hopmi_bert_attn_fwd(const float* qkv, float* out, int B, int L, int H, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream) {
  return hopmi_bert_attn_fwd_dt(qkv, out, B, L, H, p_drop, seed, seed_dev, HOPMI_F32, stream);
}

extern "C" int hopmi_bert_attn_bwd(const float* qkv, const float* d_out, float* dqkv, int B, int L, int H, float p_drop,
                                   unsigned seed, const unsigned* seed_dev, void* stream) {
  return hopmi_bert_attn_bwd_dt(qkv, d_out, dqkv, B, L, H, p_drop, seed, seed_dev, HOPMI_F32, stream);
}

HOPMI_SPLIT_STATUS_SETTER(bert_attn)
