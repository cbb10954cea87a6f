// Self-attention of the frozen BERT encoder that HOP runs its reprogrammed embeddings through
// (reference: model/HOP.py:204 -> transformers BertSelfAttention; 34 tokens, 12 heads x 64):
//
//   P = dropout(softmax(Q K^T / 8));  O = P V        per (clip b, head h), L <= 64 tokens
//
// The library path ran this as a generic memory-efficient attention (40 us forward / 121 us backward per
// layer for 0.45 GFLOP) plus layout copies on both sides.  Here one workgroup owns one (b, h): Q, K, V
// (L x 64 each) are read straight out of the fused QKV GEMM output [B][L][3][H][64], everything lives in LDS,
// O is written in the [B][L][H*64] layout the output projection wants, and the backward recomputes P and
// writes dQ, dK, dV straight into the [B][L][3][H][64] gradient of the QKV GEMM: no transposes, no
// concatenations, no saved probabilities, no cross-workgroup reductions (bitwise reproducible).
// All contractions on exact-fp32 MFMA (16x16x4); dropout is the stateless (seed,row,head,key) hash of attn.hip.
#include "common.h"

namespace hopmi {

constexpr int BD = 64;             // head dim
constexpr int BLD = BD + 4;        // LDS row stride of the Q / K / V / dO images
constexpr int BMAXL = 64;

__device__ __forceinline__ unsigned bert_hash(unsigned seed, unsigned row, unsigned head, unsigned key) {
  unsigned x = seed ^ (row * 0x9E3779B1u) ^ (key * 0x85EBCA77u) ^ (head * 0xC2B2AE3Du);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// rows [0, LP) x 64 floats of one of q/k/v (or d_o) of (b, h) -> LDS image [LP][BLD], rows >= L zeroed
__device__ __forceinline__ void stage_head(float* dst, const float* __restrict__ src, size_t row_stride, int L, int LP, int tid) {
  for (int idx = tid; idx < LP * 16; idx += 256) {
    const int row = idx >> 4, c4 = idx & 15;
    const float4 v = reinterpret_cast<const float4*>(src + (size_t)min(row, L - 1) * row_stride)[c4];
    *reinterpret_cast<float4*>(dst + row * BLD + 4 * c4) = (row < L) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// One 16x16 output tile D = sum_k A[i][k] B[k][j] over K = 4*ksteps, both operands K-contiguous in LDS:
// A[i][k] = a[i*lda + k], B[k][j] = b[j*ldb + k].  K is visited in the permuted order k = 16ii + 4q + e so
// that each lane's four consecutive k are one 16-byte LDS read (K must be a multiple of 16).
__device__ __forceinline__ f32x4 tile_kk(const float* a, int lda, const float* b, int ldb, int K, int q, int j, f32x4 acc) {
  const float* ap = a + j * lda + 4 * q;
  const float* bp = b + j * ldb + 4 * q;
  for (int ii = 0; ii < K / 16; ++ii) {
    const float4 av = *reinterpret_cast<const float4*>(ap + 16 * ii);
    const float4 bv = *reinterpret_cast<const float4*>(bp + 16 * ii);
    acc = mfma16(av.x, bv.x, acc);
    acc = mfma16(av.y, bv.y, acc);
    acc = mfma16(av.z, bv.z, acc);
    acc = mfma16(av.w, bv.w, acc);
  }
  return acc;
}

// D = sum_k A[i][k] B[k][j] with A K-contiguous (a[i*lda + k]) and B row-major over k (b[k*ldb + j])
__device__ __forceinline__ f32x4 tile_kn(const float* a, int lda, const float* b, int ldb, int K, int q, int j, f32x4 acc) {
  const float* ap = a + j * lda + 4 * q;
  for (int ii = 0; ii < K / 16; ++ii) {
    const float4 av = *reinterpret_cast<const float4*>(ap + 16 * ii);
    const float* bp = b + (16 * ii + 4 * q) * ldb + j;
    acc = mfma16(av.x, bp[0], acc);
    acc = mfma16(av.y, bp[ldb], acc);
    acc = mfma16(av.z, bp[2 * ldb], acc);
    acc = mfma16(av.w, bp[3 * ldb], acc);
  }
  return acc;
}

// D = sum_k A[i][k] B[k][j] with both operands row-major over k: A[i][k] = a[k*lda + i], B[k][j] = b[k*ldb + j]
__device__ __forceinline__ f32x4 tile_nn(const float* a, int lda, const float* b, int ldb, int K, int q, int j, f32x4 acc) {
  for (int ks = 0; ks < K / 4; ++ks) {
    const int k = 4 * ks + q;
    acc = mfma16(a[k * lda + j], b[k * ldb + j], acc);
  }
  return acc;
}

// scores -> probabilities in place: Ss[row][key] (pre-scaled scores) -> softmax over key < L; optionally also
// the dropped-out copy Pd = P * keep / (1 - p).  Wave w handles rows w, w+4, ...; lane = key.
__device__ __forceinline__ void softmax_rows(float* Ss, float* Pd, int lds, int L, int LP, int w, int lane, unsigned seed,
                                             unsigned row0, unsigned head, unsigned thresh, float dscale) {
  for (int row = w; row < LP; row += 4) {
    const bool in = lane < L && row < L;
    const float s = in ? Ss[row * lds + lane] : -1e30f;
    const float m = wave_max(s);
    const float e = in ? __expf(s - m) : 0.f;
    const float sum = wave_sum(e);
    const float p = in ? e / sum : 0.f;
    if (lane < LP) {
      Ss[row * lds + lane] = p;
      if (Pd != nullptr) {
        const bool keep = thresh == 0u || bert_hash(seed, row0 + row, head, lane) >= thresh;
        Pd[row * lds + lane] = keep ? p * dscale : 0.f;
      }
    }
  }
}

__global__ __launch_bounds__(256) void bert_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int H,
                                                            unsigned thresh, float dscale, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = (L + 15) & ~15, lds = LP + 4;
  float* Qs = smem;
  float* Ks = Qs + LP * BLD;
  float* Vs = Ks + LP * BLD;
  float* Ss = Vs + LP * BLD;                       // [LP][lds] scores -> probabilities
  float* Pd = Ss + LP * lds;                       // [LP][lds] dropped-out probabilities
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const size_t rs = (size_t)3 * H * BD;
  const float* base = qkv + (size_t)b * L * rs + (size_t)h * BD;
  stage_head(Qs, base, rs, L, LP, tid);
  stage_head(Ks, base + (size_t)H * BD, rs, L, LP, tid);
  stage_head(Vs, base + (size_t)2 * H * BD, rs, L, LP, tid);
  __syncthreads();
  const int MT = LP >> 4;
  // S = Q K^T / 8
  for (int t = w; t < MT * MT; t += 4) {
    const int mi = t / MT, ni = t % MT;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_kk(Qs + 16 * mi * BLD, BLD, Ks + 16 * ni * BLD, BLD, BD, q, j, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ss[(16 * mi + 4 * q + r) * lds + 16 * ni + j] = acc[r] * 0.125f;
  }
  __syncthreads();
  softmax_rows(Ss, Pd, lds, L, LP, w, lane, seed, (unsigned)(b * L), (unsigned)h, thresh, dscale);
  __syncthreads();
  // O^T = V^T Pd^T: D[i = d][j = row] = sum_key V[key][d] Pd[row][key]  -> lane holds 4 consecutive d of one row
  for (int t = w; t < 4 * MT; t += 4) {
    const int di = t / MT, mi = t % MT;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // A[i][k] = Vs[k][16di + i] (row-major over k), B[k][j] = Pd[(16mi + j)][k] (K-contiguous)
    const float* bp = Pd + (16 * mi + j) * lds + 4 * q;
    for (int ii = 0; ii < LP / 16; ++ii) {
      const float4 bv = *reinterpret_cast<const float4*>(bp + 16 * ii);
      const float* ap = Vs + (16 * ii + 4 * q) * BLD + 16 * di + j;
      acc = mfma16(ap[0], bv.x, acc);
      acc = mfma16(ap[BLD], bv.y, acc);
      acc = mfma16(ap[2 * BLD], bv.z, acc);
      acc = mfma16(ap[3 * BLD], bv.w, acc);
    }
    const int row = 16 * mi + j;
    if (row < L)
      *reinterpret_cast<float4*>(out + ((size_t)(b * L + row) * H + h) * BD + 16 * di + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

__global__ __launch_bounds__(256) void bert_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                            float* __restrict__ dqkv, int L, int H, unsigned thresh, float dscale,
                                                            unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = (L + 15) & ~15, lds = LP + 4;
  float* Qs = smem;
  float* Ks = Qs + LP * BLD;
  float* Vs = Ks + LP * BLD;
  float* Gs = Vs + LP * BLD;                       // dO
  float* Ss = Gs + LP * BLD;                       // P, later dS
  float* Pd = Ss + LP * lds;                       // dropped-out P
  float* Ds = Pd + LP * lds;                       // dP
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const size_t rs = (size_t)3 * H * BD;
  const float* base = qkv + (size_t)b * L * rs + (size_t)h * BD;
  stage_head(Qs, base, rs, L, LP, tid);
  stage_head(Ks, base + (size_t)H * BD, rs, L, LP, tid);
  stage_head(Vs, base + (size_t)2 * H * BD, rs, L, LP, tid);
  stage_head(Gs, dout + (size_t)b * L * H * BD + (size_t)h * BD, (size_t)H * BD, L, LP, tid);
  __syncthreads();
  const int MT = LP >> 4;
  // S = Q K^T / 8 and dPd = dO V^T (both K-contiguous over d)
  for (int t = w; t < 2 * MT * MT; t += 4) {
    const int which = t / (MT * MT), tt = t % (MT * MT);
    const int mi = tt / MT, ni = tt % MT;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (which == 0) {
      acc = tile_kk(Qs + 16 * mi * BLD, BLD, Ks + 16 * ni * BLD, BLD, BD, q, j, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ss[(16 * mi + 4 * q + r) * lds + 16 * ni + j] = acc[r] * 0.125f;
    } else {
      acc = tile_kk(Gs + 16 * mi * BLD, BLD, Vs + 16 * ni * BLD, BLD, BD, q, j, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ds[(16 * mi + 4 * q + r) * lds + 16 * ni + j] = acc[r];
    }
  }
  __syncthreads();
  softmax_rows(Ss, Pd, lds, L, LP, w, lane, seed, (unsigned)(b * L), (unsigned)h, thresh, dscale);
  __syncthreads();
  float* dst = dqkv + (size_t)b * L * rs + (size_t)h * BD;
  // dV = Pd^T dO: D[i = key][j = d] = sum_row Pd[row][key] dO[row][d]   (both row-major over k = row)
  for (int t = w; t < 4 * MT; t += 4) {
    const int mi = t / 4, di = t % 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile_nn(Pd + 16 * mi, lds, Gs + 16 * di, BLD, LP, q, j, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * mi + 4 * q + r;
      if (key < L) dst[(size_t)key * rs + 2 * H * BD + 16 * di + j] = acc[r];
    }
  }
  // dS = P * (keep/(1-p) * dPd - rowsum(P * keep/(1-p) * dPd)) / 8: since Pd = P * keep/(1-p), P*dP = Pd*dPd
  for (int row = w; row < LP; row += 4) {
    const float pd = lane < LP ? Pd[row * lds + lane] : 0.f;
    const float p = lane < LP ? Ss[row * lds + lane] : 0.f;
    const float dpd = lane < LP ? Ds[row * lds + lane] : 0.f;
    const float delta = wave_sum(pd * dpd);
    if (lane < LP) Ss[row * lds + lane] = (pd * dpd - p * delta) * 0.125f;
  }
  __syncthreads();
  // dQ = dS K: D[i = row][j = d] = sum_key dS[row][key] K[key][d];  dK = dS^T Q: D[i = key][j = d] = sum_row dS[row][key] Q[row][d]
  for (int t = w; t < 8 * MT; t += 4) {
    const int which = t / (4 * MT), tt = t % (4 * MT);
    const int mi = tt / 4, di = tt % 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (which == 0) acc = tile_kn(Ss + 16 * mi * lds, lds, Ks + 16 * di, BLD, LP, q, j, acc);
    else acc = tile_nn(Ss + 16 * mi, lds, Qs + 16 * di, BLD, LP, q, j, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * mi + 4 * q + r;
      if (row < L) dst[(size_t)row * rs + which * H * BD + 16 * di + j] = acc[r];
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

extern "C" int hopmi_bert_attn_fwd(const float* qkv, float* out, int B, int L, int H, float p_drop, unsigned seed, void* stream) {
  if (int e = bert_attn_validate("hopmi_bert_attn_fwd", B, L, H, p_drop)) return e;
  if (!qkv || !out) { set_error("hopmi_bert_attn_fwd: null pointer argument"); return HOPMI_EINVAL; }
  const int LP = (L + 15) & ~15;
  const size_t lds = ((size_t)3 * LP * BLD + 2 * (size_t)LP * (LP + 4)) * sizeof(float);
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  hipLaunchKernelGGL(bert_attn_fwd_kernel, dim3(B * H), dim3(256), lds, static_cast<hipStream_t>(stream), qkv, out, L, H, thresh,
                     1.f / (1.f - p_drop), seed);
  return check_launch("hopmi_bert_attn_fwd");
}

extern "C" int hopmi_bert_attn_bwd(const float* qkv, const float* d_out, float* dqkv, int B, int L, int H, float p_drop,
                                   unsigned seed, void* stream) {
  if (int e = bert_attn_validate("hopmi_bert_attn_bwd", B, L, H, p_drop)) return e;
  if (!qkv || !d_out || !dqkv) { set_error("hopmi_bert_attn_bwd: null pointer argument"); return HOPMI_EINVAL; }
  const int LP = (L + 15) & ~15;
  const size_t lds = ((size_t)4 * LP * BLD + 3 * (size_t)LP * (LP + 4)) * sizeof(float);
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  hipLaunchKernelGGL(bert_attn_bwd_kernel, dim3(B * H), dim3(256), lds, static_cast<hipStream_t>(stream), qkv, d_out, dqkv, L, H,
                     thresh, 1.f / (1.f - p_drop), seed);
  return check_launch("hopmi_bert_attn_bwd");
}
