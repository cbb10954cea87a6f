// fp32 GEMM against frozen weights on the bf16 matrix cores: C[M][N] = A[M][K] . Bt[N][K]^T (+ bias[N]).
//
// The frozen BERT (HOP.py:90-91,204) is 70 % of the step's FLOPs: ~1.1 TFLOP of fp32 GEMMs per training step against
// weights that never change.  The exact-fp32 MFMA peaks at 157 TFLOP/s (the library reaches 115-140 on these shapes); the
// bf16 MFMA is 16x faster per instruction.  Here every fp32 operand is carried as NP bf16 parts (bf16_dev.h):
//   NP = 3:  x = x0 + x1 + x2 holds all 24 significand bits; a.b = sum over i + j <= 2 of a_i b_j  (6 MFMAs per product block);
//            the dropped terms are < 2^-24 |a b| -- the rounding error of an fp32 multiply -- so the GEMM is fp32-EQUIVALENT
//            (measured against float64: same error as the library's fp32 GEMM, tests/test_gpu_parity.py)
//   NP = 2:  x = x0 + x1 (2^-17), three terms: 2^-16-class products at twice the rate (the WaveNet kernels' arithmetic)
// with fp32 accumulation in the MFMA.  The weights are split ONCE (hopmi_gemm_split_prepare: they are frozen, also their
// transpose for the activation gradient dX = dY . W); the activations are split on the fly when a tile is committed to LDS.
//
// Workgroup = 128 x 128 output tile, 8 waves of 64 x 32 (4 x 2 MFMA tiles, 32 accumulator registers: two waves per SIMD, so
// one wave's LDS reads / splits / commits run beside its partner's MFMAs), K steps of 32 (one v_mfma_f32_16x16x32_bf16
// k-step); the next tile's global loads are issued before the MFMAs of the current one and committed behind them.  Two
// forms, chosen by the tile count: DB (two LDS buffers, one barrier per step, 144 KiB: one workgroup per CU) when the tiles
// fit the chip about once; !DB (one buffer, two barriers per step, 72 KiB: TWO workgroups per CU that cover each other's
// commit phases and halve the tile-count quantisation) for the larger grids.  LDS rows are 96 bytes (64 data + 32 pad): the 16 lanes
// of every ds_read_b128 service group then hit 16 distinct 16-byte bank slots.  Tiles are enumerated XCD-aware: the tiles
// of one XCD walk one panel of A against consecutive panels of B, so both stay in that XCD's L2.
#define HOPMI_FILE_ID 1          // (diagnostic build: common.h, split_check)
#include "bf16_dev.h"

#include <cstdint>

namespace hopmi {

#ifdef HOPMI_STAMPS
static __device__ long long* g_gemm_stamps = nullptr;
// stamps of wave 0 of every workgroup at k-step nk / 2: [block][8]
#define GEMM_STAMP(slot)                                                                           \
  do {                                                                                             \
    if (kt == nk / 2 && wv == 0) {                                                                 \
      unsigned long long t_;                                                                       \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
      if (g_gemm_stamps && (threadIdx.x & 63) == 0) g_gemm_stamps[blockIdx.x * 8 + (slot)] = (long long)t_; \
    }                                                                                              \
  } while (0)
#else
#define GEMM_STAMP(slot) do { } while (0)
#endif

constexpr int GN = 128, GK = 32;                   // (the tile's M extent BM is a template parameter: 128 or 64)
constexpr int GLD = 48;                            // LDS row stride in bf16 units (96 bytes)

template <int NP>
struct SplitN { u32x2 p[NP]; };                    // NP packed pairs for two consecutive values

template <int NP>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[NP]) {
  float ra = a, rb = b;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const unsigned h = pk_bf16(ra, rb);
    out[i] = h;
    if (i + 1 < NP) { ra -= bf_lo(h); rb -= bf_hi(h); }
  }
}

// ---- the fp16 hi/lo form (round 4): TWO parts per operand, THREE terms, fp32-equivalent ------------------------------------
// x s = h + l with h = fp16(x s), l = fp16(x s - h): 11 + 11 significand bits, |x s - h - l| <= 2^-22 |x s|; the product is taken as
// h_a h_b + h_a l_b + l_a h_b (the dropped l_a l_b is <= 2^-22 |a b|) on v_mfma_f32_16x16x32_f16 (the bf16 instruction's rate; the
// 22-bit products are exact in the fp32 accumulator).  Per product that is a few 2^-23 -- below what the fp32 ACCUMULATION of a
// K >= 128 dot product leaves, measured: the same error against float64 as the six-term bf16 form and the library's fp32 GEMM
// (tests/test_gpu_parity.py::test_gemm_split_vs_float64) at HALF the matrix instructions.  fp16 has 5 exponent bits, so every operand
// is scaled by a power of two s (exact) that puts its largest magnitude in [2^14, 2^15): the lo part of every element within 2^-17
// of the maximum is a normal fp16 number; smaller elements carry an absolute error <= 2^-39 of the maximum.  The weights' scale is
// found when their image is prepared (frozen: once; one scale per tensor); the activations get one scale PER ROW (an output
// element only ever sees one row of A, so rows of very different magnitude -- gradient rows -- do not share a window), written by
// hopmi_row_scales (one wave per row) or by whichever kernel produced A.  The epilogue multiplies the accumulators by
// 2^-(s_a[row] + s_b) (exact).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// two (already scaled) floats -> packed fp16 hi pair, packed fp16 lo pair (round to nearest even; x - hi is exact in fp32)
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned (&out)[2], int line = __builtin_LINE()) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 hi = {(_Float16)a, (_Float16)b};
  split_check(a, hi[0], line);
  split_check(b, hi[1], line);
  const h2 lo = {(_Float16)(a - (float)hi[0]), (_Float16)(b - (float)hi[1])};
  out[0] = __builtin_bit_cast(unsigned, hi);
  out[1] = __builtin_bit_cast(unsigned, lo);
}
constexpr int F16_PARTS = 16;                      // the `parts` code of this form at the C ABI ("two fp16 parts")
// Image of a [N][K] weight for this form: hi and lo fp16 images [2][Np][Kp], Np = N rounded up to the 128-column tile, Kp = K rounded
// up to the 32-wide k-step (pad rows / columns are zero: the kernel reads them unguarded), then Np floats: 1 / s_n, the inverse of
// row n's power-of-two scale (every output column has its own: weight rows -- and the columns of W that become the rows of the
// W^T image -- may differ by orders of magnitude).
__host__ __device__ inline int f16_np(int N) { return (N + GN - 1) / GN * GN; }
__host__ __device__ inline int f16_kp(int K) { return (K + GK - 1) / GK * GK; }
// TILE-BLOCKED image layout (round 5).  An operand tile of a k-step is 128 rows x 32 halves = 128 x 64 bytes.  In a row-major image
// those 64 bytes are half of a 128-byte line of a K*2-byte row, 256 separate lines per operand and k-step: staging ALONE (LDS-DMA, no
// MFMA) ran at 11 TB/s -- 43 us of the 69 us QKV product (tools/probes/dma_layout.hip).  Blocked, a tile's 8 KiB are contiguous:
// 17-21 TB/s (28 us).  Element (row r, column k) of one part image with KB = Kp / 32 k-steps per row block:
//     ((r / 128) * KB + k / 32) * 4096 + (r % 128) * 32 + k % 32           [halves]
// rows padded to whole 128-row blocks.  Used by every fp16 hi/lo image: the weights' (hopmi_gemm_f16x2_prepare), the activations'
// (hopmi_rows_image_f16, the LayerNorm kernels' image outputs).
__host__ __device__ inline size_t f16_blk(int r, int k, int KB) { return ((size_t)(r >> 7) * KB + (k >> 5)) * 4096 + (size_t)(r & 127) * 32 + (k & 31); }

// per-row power-of-two scales of A [M][K] (K % 4 == 0): out[row] = s, out[M + row] = 1 / s; one wave per row
__global__ __launch_bounds__(256) void row_scales_kernel(const float* __restrict__ A, int M, int K, float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float4* src = reinterpret_cast<const float4*>(A + (size_t)row * K);
  unsigned m = 0;
  for (int i = lane; i < K / 4; i += 64) m = abs_bits_max4(m, src[i]);
  m = wave_max_u32(m);
  if (lane == 0) store_row_scale(out, M, row, m);
}

// A [M][K] (K % 4 == 0) -> the A operand of the LDS-DMA form: scaled fp16 hi / lo images [2][M][K] and the scale pairs [2][M]; one
// wave per row (maximum, then split: the row is read twice, the second time from the caches)
__global__ __launch_bounds__(256) void f16_rows_image_kernel(const float* __restrict__ A, int M, int K, unsigned* __restrict__ img,
                                                             float* __restrict__ scales) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float4* src = reinterpret_cast<const float4*>(A + (size_t)row * K);
  unsigned m = 0;
  for (int i = lane; i < K / 4; i += 64) m = abs_bits_max4(m, src[i]);
  m = wave_max_u32(m);
  const unsigned sb = scale_bits_for_max(m);
  const float sc = __uint_as_float(sb);
  if (lane == 0) { scales[row] = sc; scales[M + row] = inv_scale(sb); }
  // (tile-blocked layout, K % 32 == 0; the part images are Mp = ceil128(M) rows)
  _Float16* hi = reinterpret_cast<_Float16*>(img);
  _Float16* lo = hi + (size_t)((M + 127) / 128 * 128) * K;
  const int KB = K >> 5;
  for (int i = lane; i < K / 4; i += 64) {
    const float4 v = src[i];
    unsigned p0[2], p1[2];
    split_pair_f16(v.x * sc, v.y * sc, p0);
    split_pair_f16(v.z * sc, v.w * sc, p1);
    const size_t at = f16_blk(row, 4 * i, KB);
    *reinterpret_cast<u32x2*>(hi + at) = u32x2{p0[0], p1[0]};
    *reinterpret_cast<u32x2*>(lo + at) = u32x2{p0[1], p1[1]};
  }
}

// ... for any even K and 8-byte aligned rows (the mapping layer's weight: K = 30522): 8-byte loads, columns padded with zeros to Kp =
// K rounded up to the 32-wide k-step.  Workgroup = 4 rows: pass 1 one wave per row (maximum); pass 2 every wave instruction covers
// the 4 rows x one 32-column block -- 4 x 64 contiguous bytes of the tile-blocked image, whole 128-byte lines (a wave per row would
// write 64-byte pieces whose neighbours belong to other waves: 242 us for the 183 MB mapping weight, now ~100)
__global__ __launch_bounds__(256) void f16_rows_image_ragged_kernel(const float* __restrict__ A, int M, int K, int Kp, unsigned* __restrict__ img,
                                                                    float* __restrict__ scales) {
  __shared__ float s_sc[4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = blockIdx.x * 4;
  {
    const int row = row0 + wv;
    unsigned m = 0;
    if (row < M) {
      const float2* src = reinterpret_cast<const float2*>(A + (size_t)row * K);
      for (int i = lane; i < K / 2; i += 64) {
        const float2 v = src[i];
        m = max(m, max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu));
      }
      m = wave_max_u32(m);
    }
    const unsigned sb = scale_bits_for_max(m);
    if (lane == 0) {
      s_sc[wv] = __uint_as_float(sb);
      if (row < M) { scales[row] = __uint_as_float(sb); scales[M + row] = inv_scale(sb); }
    }
  }
  __syncthreads();
  _Float16* hi = reinterpret_cast<_Float16*>(img);
  _Float16* lo = hi + (size_t)((M + 127) / 128 * 128) * Kp;
  const int KB = Kp >> 5;
  const int r = lane >> 4, c2 = (lane & 15) * 2;         // lane -> (row of the four, column pair of the block)
  const int row = row0 + r;
  if (row >= M) return;
  const float sc = s_sc[r];
  const float* src = A + (size_t)row * K;
  for (int kb = wv; kb < KB; kb += 4) {
    const int k = 32 * kb + c2;
    unsigned parts[2] = {0u, 0u};
    if (k < K) {
      const float2 v = *reinterpret_cast<const float2*>(src + k);
      split_pair_f16(v.x * sc, v.y * sc, parts);
    }
    const size_t at = f16_blk(row, k, KB);
    *reinterpret_cast<unsigned*>(hi + at) = parts[0];
    *reinterpret_cast<unsigned*>(lo + at) = parts[1];
  }
}

// C[m][n] = slab 0 + slab 1 + ... (index order) + row_bias[m]: the split-K partial products of the LDS-DMA form
__global__ __launch_bounds__(256) void gemm_splitk_sum_kernel(const float* __restrict__ slabs, int splits, const float* __restrict__ row_bias,
                                                              float* __restrict__ C, int M, int N) {
  const size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const size_t total = (size_t)M * N;
  if (idx >= total) return;
  float4 v = *reinterpret_cast<const float4*>(slabs + idx);
  for (int p = 1; p < splits; ++p) {
    const float4 t = *reinterpret_cast<const float4*>(slabs + (size_t)p * total + idx);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  if (row_bias != nullptr) {
    const float b = row_bias[idx / N];                // (N % 4 == 0: the four values share a row)
    v.x += b; v.y += b; v.z += b; v.w += b;
  }
  *reinterpret_cast<float4*>(C + idx) = v;
}

// W [N][K] (row-major, K % 2 == 0) -> its image: one wave per image row n < Np (two passes over the row: maximum, then split)
__global__ __launch_bounds__(256) void f16_prepare_rows_kernel(const float* __restrict__ W, int N, int K, int Np, int Kp,
                                                               unsigned* __restrict__ img, float* __restrict__ inv_out) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= Np) return;
  _Float16* hi = reinterpret_cast<_Float16*>(img);                     // (tile-blocked layout)
  _Float16* lo = hi + (size_t)Np * Kp;
  const int KB = Kp >> 5;
  if (n >= N) {
    for (int i = lane; i < Kp / 2; i += 64) {
      const size_t at = f16_blk(n, 2 * i, KB);
      *reinterpret_cast<unsigned*>(hi + at) = 0u;
      *reinterpret_cast<unsigned*>(lo + at) = 0u;
    }
    if (lane == 0) inv_out[n] = 1.f;
    return;
  }
  const float2* src = reinterpret_cast<const float2*>(W + (size_t)n * K);
  unsigned m = 0;
  for (int i = lane; i < K / 2; i += 64) {
    const float2 v = src[i];
    m = max(m, max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu));
  }
  m = wave_max_u32(m);
  const unsigned sb = scale_bits_for_max(m);
  const float sc = __uint_as_float(sb);
  if (lane == 0) inv_out[n] = inv_scale(sb);
  for (int i = lane; i < Kp / 2; i += 64) {
    unsigned parts[2] = {0u, 0u};
    if (i < K / 2) {
      const float2 v = src[i];
      split_pair_f16(v.x * sc, v.y * sc, parts);
    }
    const size_t at = f16_blk(n, 2 * i, KB);
    *reinterpret_cast<unsigned*>(hi + at) = parts[0];
    *reinterpret_cast<unsigned*>(lo + at) = parts[1];
  }
}

// weights -> NP bf16 part images [NP][N][K] (row-major, the MFMA B-operand's 8 consecutive k are contiguous)
template <int NP>
__global__ __launch_bounds__(256) void gemm_split_prepare_kernel(const float* __restrict__ W, size_t n, unsigned* __restrict__ img) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // pair index
  if (2 * i >= n) return;
  const float2 v = reinterpret_cast<const float2*>(W)[i];
  unsigned parts[NP];
  split_pair<NP>(v.x, v.y, parts);
#pragma unroll
  for (int p = 0; p < NP; ++p) img[(size_t)p * (n / 2) + i] = parts[p];
}

constexpr int GT = 512;                            // threads: 8 waves as 2 (M) x 4 (N)

// Epilogue forms (hopmi_gemm_split_ep): the BertIntermediate activation and its derivative ride on the product that feeds /
// follows them instead of a launch of their own that re-reads and re-writes the M x 3072 tensor.  The same expressions as
// elementwise.hip's bias_gelu kernels on the same fp32 values: bit-identical to GEMM + hopmi_bias_gelu_fwd / _bwd.
//   EP_BIAS       C = acc + bias
//   EP_GELU       h = acc + bias;  C = gelu_erf(h);  C2 = h when asked for (the backward's operand)
//   EP_GELU_GRAD  C = (acc + bias) * gelu_erf'(aux)            (aux = the h of the forward, same shape as C)
enum { EP_BIAS = 0, EP_GELU = 1, EP_GELU_GRAD = 2 };
__device__ __forceinline__ float gemm_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gemm_gelu_grad(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * v * v);
  return cdf + v * pdf;
}

// BM = 64 halves the tile (8 waves of 32 x 32) for shapes whose 128-row tiling leaves CUs idle or quantises badly.
template <int NP, bool DB, int BM, bool F16 = false, bool RAGGED = false>
__global__ __launch_bounds__(GT, DB ? 1 : 2) void gemm_split_kernel(const float* __restrict__ A, const __bf16* __restrict__ Bimg,
                                                         const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K,
                                                         int tiles_m, int tiles_n, int ep, float* __restrict__ C2,
                                                         const float* __restrict__ aux, const float* __restrict__ a_rows, int a_parts,
                                                         float* __restrict__ c_rowmax) {
  static_assert(!F16 || NP == 2, "the fp16 form carries two parts");
  // fp16 form: this tile's per-row activation scales (s, 1 / s) and, when asked for, the per-row maxima of what it writes
  __shared__ float s_scale[F16 ? BM : 1], s_inv[F16 ? BM : 1];
  __shared__ unsigned s_cmax[F16 ? BM : 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // [buffer 2][A: part NP x BM rows | B: part NP x 128 rows][GLD]
  __bf16* lds = reinterpret_cast<__bf16*>(smem_raw);
  constexpr int MI = BM / 32;                      // 16-row MFMA tiles per wave (wave rows = BM / 2)
  constexpr int PART_A = BM * GLD, PART_B = GN * GLD;        // one part image of an operand (bf16 units)
  constexpr int OPER = NP * PART_A;                // offset of the B images inside a buffer
  constexpr int BUFSZ = NP * (PART_A + PART_B), BUF = DB ? BUFSZ : 0;
  int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wv >> 2, wc = wv & 3;
  const int lane = tid & 63, q = lane >> 4, n = lane & 15;

  // XCD-aware tile order: workgroups b and b + 8 share an XCD (round-robin dispatch); give each XCD a contiguous run of tiles,
  // tiles of a run ordered n fastest, so an XCD re-reads one A panel and streams neighbouring B panels through its own L2
  const int ntiles = tiles_m * tiles_n;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  const int per = ntiles >> 3, rem = ntiles & 7;
  const int tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
  if (idx >= per + (xcd < rem ? 1 : 0)) return;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * GN;
  // fp16 form: N and K are the LOGICAL extents; the weight image is padded to Np x Kp (zeros) and carries Np inverse row scales
  // behind the two part images; the activations' per-row scales are a_rows[row] (s) and a_rows[M + row] (1 / s)
  const int Np = F16 ? f16_np(N) : N, Kp = F16 ? f16_kp(K) : K;
  const float* inv_b = reinterpret_cast<const float*>(Bimg + (size_t)NP * Np * Kp);

  // staging maps.  B (128 rows x 64 bytes per part): thread t moves 16 bytes of row (t >> 2), k = 8 (t & 3) .. +7.
  // A (BM rows x 128 bytes): BM = 128: the 32 bytes of row (t >> 2) at k = 8 (t & 3); BM = 64: 16 bytes of row (t >> 3) at
  // k = 4 (t & 7).
  constexpr int AF4 = BM / 64;                     // float4 loads of A per thread and k step
  const int brow = tid >> 2, bq = tid & 3;
  const int arow = (BM == 128) ? (tid >> 2) : (tid >> 3);
  const int acol = (BM == 128) ? 8 * (tid & 3) : 4 * (tid & 7);
  const float* a_src = A + (size_t)min(m0 + arow, M - 1) * K + acol;
  if (F16) {
    // a_parts == 0: a_rows = [2][M] scale pairs (hopmi_row_scales / the LayerNorm kernels); a_parts = P > 0: a_rows = [P][M] partial
    // row maxima of |A| as the kernel that produced A left them (one per workgroup that held a piece of the row): reduced here
    if (tid < BM) {
      const int row = min(m0 + tid, M - 1);
      float sc, iv;
      if (a_parts == 0) {
        sc = a_rows[row];
        iv = a_rows[M + row];
      } else {
        unsigned mx = 0;
        for (int pp = 0; pp < a_parts; ++pp) mx = max(mx, __float_as_uint(a_rows[(size_t)pp * M + row]) & 0x7fffffffu);
        const unsigned sb = scale_bits_for_max(mx);
        sc = __uint_as_float(sb);
        iv = inv_scale(sb);
      }
      s_scale[tid] = sc;
      s_inv[tid] = iv;
      s_cmax[tid] = 0u;
    }
    __syncthreads();
  }
  const float sa = F16 ? s_scale[arow] : 1.f;
  // (fp16 form: the weight image is tile-blocked, f16_blk: tile (tn, kt) is 8 KiB contiguous; bf16 forms: row-major)
  const __bf16* b_src = F16 ? Bimg + (size_t)tn * (Kp >> 5) * 4096 + brow * 32 + 8 * bq : Bimg + (size_t)(n0 + brow) * Kp + 8 * bq;
  const size_t b_part = (size_t)Np * Kp;
  const int a_off = arow * GLD + acol, b_off = brow * GLD + 8 * bq;

  float4 a_st[AF4];
  u32x4 b_st[NP];
  auto issue = [&](int k0) {
#ifdef HOPMI_GEMM_EXP_NOLOAD
    if (k0 > 0) return;                              // (timing experiment: results wrong)
#endif
#pragma unroll
    for (int i = 0; i < AF4; ++i) {
      // (fp16 form: K may end inside the last k-step -- the image's pad columns are zero, the activations' must not be read)
      if (RAGGED && k0 + acol + 4 * i >= K) a_st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      else a_st[i] = reinterpret_cast<const float4*>(a_src + k0)[i];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) b_st[p] = *reinterpret_cast<const u32x4*>(b_src + p * b_part + (F16 ? (size_t)(k0 >> 5) * 4096 : (size_t)k0));
  };
  auto commit = [&](int buf) {
    __bf16* la = lds + buf * BUF + a_off;
    __bf16* lb = lds + buf * BUF + OPER + b_off;
    unsigned parts[2 * AF4][NP];
#pragma unroll
    for (int i = 0; i < AF4; ++i) {
      if constexpr (F16) {
        split_pair_f16(a_st[i].x * sa, a_st[i].y * sa, parts[2 * i]);
        split_pair_f16(a_st[i].z * sa, a_st[i].w * sa, parts[2 * i + 1]);
      } else {
        split_pair<NP>(a_st[i].x, a_st[i].y, parts[2 * i]);
        split_pair<NP>(a_st[i].z, a_st[i].w, parts[2 * i + 1]);
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (BM == 128) *reinterpret_cast<u32x4*>(la + p * PART_A) = u32x4{parts[0][p], parts[1][p], parts[2][p], parts[3][p]};
      else *reinterpret_cast<u32x2*>(la + p * PART_A) = u32x2{parts[0][p], parts[1][p]};
      *reinterpret_cast<u32x4*>(lb + p * PART_B) = b_st[p];
    }
  };

  f32x4 acc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = {0.f, 0.f, 0.f, 0.f};

  const int nk = Kp / GK;
  issue(0);
  commit(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    // (the lane index is hidden from the optimiser per iteration: keeps loop-invariant LDS addresses from being hoisted into
    // dozens of registers held across the whole loop)
    asm volatile("" : "+v"(tid));
    GEMM_STAMP(0);
    const int ln = tid & 63, qq = ln >> 4, nn = ln & 15;
    const bool more = kt + 1 < nk;
    // DB: the loop body is branch-free (the last step re-stages its own tile into the idle buffer), so that the splits and
    // LDS stores of the next tile can be scheduled between this tile's MFMAs instead of behind them
    if (DB) issue(min(kt + 1, nk - 1) * GK);
    else if (more) issue((kt + 1) * GK);
    const __bf16* la = lds + (DB ? (kt & 1) : 0) * BUF + ((BM / 2) * wr + nn) * GLD + 8 * qq;
    const __bf16* lb = lds + (DB ? (kt & 1) : 0) * BUF + OPER + (32 * wc + nn) * GLD + 8 * qq;
    u32x4 af[MI][NP];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int p = 0; p < NP; ++p) af[mi][p] = *reinterpret_cast<const u32x4*>(la + p * PART_A + 16 * mi * GLD);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      u32x4 bf[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) bf[p] = *reinterpret_cast<const u32x4*>(lb + p * PART_B + 16 * ni * GLD);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        f32x4 c = acc[mi][ni];
        // smallest terms first: i + j = NP - 1, ..., 0
#pragma unroll
        for (int s = NP - 1; s >= 0; --s)
#pragma unroll
          for (int i = 0; i <= s; ++i) c = F16 ? mfma_f16(af[mi][i], bf[s - i], c) : mfma_bf16(af[mi][i], bf[s - i], c);
        acc[mi][ni] = c;
      }
    }
    GEMM_STAMP(1);
    if (DB) {
      commit((kt + 1) & 1);
#ifndef HOPMI_GEMM_NO_SCHED
      // order for the scheduler: next tile's global loads first, this tile's fragment reads, a run of MFMAs (the loads
      // land meanwhile), then the splits (2 VALU per MFMA) and the LDS stores (1 per 2 MFMAs) between the remaining MFMAs
      constexpr int NMFMA = MI * 2 * (NP * (NP + 1) / 2), NVALU_SLOTS = 16, NLEAD = NMFMA - NVALU_SLOTS - 4 * NP;
      __builtin_amdgcn_sched_group_barrier(0x020, AF4 + NP, 0);
      if (NLEAD > 0) __builtin_amdgcn_sched_group_barrier(0x008, NLEAD, 0);
#pragma unroll
      for (int g = 0; g < NVALU_SLOTS; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
#pragma unroll
      for (int g = 0; g < 2 * NP; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
#endif
    } else {
      __syncthreads();                               // one buffer: every wave has read this step's fragments
      GEMM_STAMP(2);
      if (more) commit(0);
    }
    GEMM_STAMP(3);
    __syncthreads();
    GEMM_STAMP(4);
  }

  // epilogue: lane (q, n) holds rows 4q + r, column n of every 16 x 16 tile
  unsigned rmx[F16 ? MI : 1][4];                     // (fp16 form, c_rowmax: max |value written| per row this lane touches)
#pragma unroll
  for (int mi = 0; mi < (F16 ? MI : 1); ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) rmx[mi][r] = 0u;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + 32 * wc + 16 * ni + n;
    if (F16 && col >= N) continue;                   // (pad columns of the last column tile)
    const float bv = bias != nullptr ? bias[col] : 0.f;
    const float out_scale_b = F16 ? inv_b[col] : 1.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = (BM / 2) * wr + 16 * mi + 4 * q + r, row = m0 + rl;
        if (row < M) {
          const size_t at = (size_t)row * N + col;
          const float h = F16 ? acc[mi][ni][r] * s_inv[rl] * out_scale_b + bv : acc[mi][ni][r] + bv;
          float out;
          if (ep == EP_BIAS) out = h;
          else if (ep == EP_GELU) {
            if (C2 != nullptr) C2[at] = h;
            out = gemm_gelu(h);
          } else out = h * gemm_gelu_grad(aux[at]);
          C[at] = out;
          if (F16) rmx[mi][r] = max(rmx[mi][r], __float_as_uint(out) & 0x7fffffffu);
        }
      }
  }
  if (F16 && c_rowmax != nullptr) {
    // the 16 lanes that share q hold 16 columns each of the same rows; then the four column waves through LDS; one row of
    // c_rowmax [tiles_n][M] per column tile: the consuming GEMM reduces the tiles_n partial maxima in its prologue
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsigned v = rmx[mi][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
        if (n == 0) atomicMax(&s_cmax[(BM / 2) * wr + 16 * mi + 4 * q + r], v);
      }
    __syncthreads();
    if (tid < BM && m0 + tid < M) c_rowmax[(size_t)tn * M + m0 + tid] = __uint_as_float(s_cmax[tid]);
  }
}

template <int NP, bool DB, int BM, bool F16 = false, bool RAGGED = false>
static void launch_gemm_variant(const float* A, const void* Bimg, const float* bias, float* C, int M, int N, int K, hipStream_t st,
                                int ep, float* C2, const float* aux, const float* a_rows = nullptr, int a_parts = 0,
                                float* c_rowmax = nullptr) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + GN - 1) / GN;
  const size_t lds = (size_t)(DB ? 2 : 1) * NP * (BM + GN) * GLD * sizeof(__bf16);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<NP, DB, BM, F16, RAGGED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
  const int ntiles = tiles_m * tiles_n;
  const int grid = ((ntiles + 7) / 8) * 8;                // every XCD gets the same number of slots; surplus ones return at once
  hipLaunchKernelGGL((gemm_split_kernel<NP, DB, BM, F16, RAGGED>), dim3(grid), dim3(GT), lds, st, A, static_cast<const __bf16*>(Bimg), bias, C, M, N,
                     K, tiles_m, tiles_n, ep, C2, aux, a_rows, a_parts, c_rowmax);
}

// ------------------------------------------------------------------------------------------------------------------
// Both operands as prepared part images (A: hopmi_gemm_split_prepare on the activations, or a producer kernel that writes
// the image itself): nothing is split in the kernel and nothing passes through registers on its way to LDS -- every tile is
// staged by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, 16 rows x 64 B of one part image).  The split
// kernel above spends 36 % of a k-step committing the next tile (tools/probes/gemm_stamps.py: 1 900 cycles MFMA phase,
// 1 280 commit, 320 barrier); here the commit is six DMA instructions per wave.  LDS rows are 64 B, unpadded (the DMA
// writes lane-linearly), with the 16-byte slots XOR-swizzled through the SOURCE address and the fragment reads
// (slot ^ f((row >> 2) & 3), f = 0, 3, 2, 1: conflict-free ds_read_b128).  NBUF tile buffers: the DMA of tile kt + NBUF - 1
// is issued while tile kt is multiplied; a counted s_waitcnt leaves the youngest tiles in flight across the raw barrier.
__device__ __forceinline__ int gswz(int row) { return (0x1230 >> (4 * ((row >> 2) & 3))) & 3; }

// F16: the fp16 hi/lo form (NP = 2) -- A image [2][M][K] of scaled fp16 parts + a_rows [2][M] scale pairs as hopmi_rows_image_f16
// writes them, B image padded to Np rows with its Np inverse row scales behind it (f16_np / hopmi_gemm_f16x2_prepare); N may be
// ragged (store guard), K % 32 == 0.  The same three terms in the same order as gemm_split_kernel<2, ., ., true>: bit-identical.
// The product handed on as the NEXT fp16-form GEMM's operand image instead of (or beside) fp32 values: `image` = tile-blocked hi / lo
// images of C times one power of two per row, `scales` its [2][M] pairs.  The row's scale has to be known before any column tile is
// written and the row's maximum is not (it spans the other column tiles' workgroups): it is taken from an a-priori BOUND,
// |C[row][:]| <= row_norm[row] * mul + add -- Cauchy-Schwarz on the product (row_norm = the 2-norm of the A operand's row, written by
// the LayerNorm kernel that produced it; mul = the largest 2-norm of a row of the weight, times the epilogue's Lipschitz constant;
// add = the largest |bias|).  Typical elements sit 2^3 ... 2^6 below such a bound, well inside the 2^17 window in which their lo parts
// stay normal fp16 numbers (f16_dev.h).
// (HOPMI_AB_EXP: compile-time timing experiments on the k-loop -- wrong results, same schedule; tools/probes/ab_limits.py)
#ifndef HOPMI_AB_EXP
#define HOPMI_AB_EXP 0
#endif
struct AbImageOut { _Float16* image; float* scales; const float* row_norm; float mul, add; };

// BM = 64 (fp16 form): 64-row tiles (8 waves of 32 x 32) for shapes whose 128-row tiling leaves most CUs idle (N = 768 at M = 2176:
// 102 tiles -> 204).
// WV = waves per workgroup (round 6).  The k-loop is LDS-BANDWIDTH bound, not matrix-pipe bound: per k-step a 64 x 128 tile's 8 waves
// of 32 x 32 read 8 x 8 KB of fragments and the DMA writes 24 KB -- 88 KB through the CU's 128 B/clk LDS = 704 clocks against 386 clocks
// of MFMA (3 terms x 64 x 128 x 32 x 2 flops at 4 069 flops/clk/CU): the matrix pipe cannot be busy more than 0.55 of the time, measured
// 0.26.  Four waves of 32 x 64 read 4 x 12 KB for the same tile (an A fragment serves four column tiles instead of two): 72 KB, 576
// clocks, with three workgroups per CU (48 KB of LDS each) instead of two.  Same terms in the same order per output element: bit-identical.
template <int NP, int NBUF, bool F16 = false, int BM = 128, int WV = 8>
__global__ __launch_bounds__(64 * WV, 1) void gemm_split_ab_kernel(const __bf16* __restrict__ Aimg, const __bf16* __restrict__ Bimg,
                                                             const float* __restrict__ bias, float* __restrict__ C, int M, int N,
                                                             int K, int tiles_m, int tiles_n, const float* __restrict__ a_rows, int ep,
                                                             float* __restrict__ C2, const float* __restrict__ aux,
                                                             float* __restrict__ c_rowmax, AbImageOut io) {
  static_assert(BM == 128 || (BM == 64 && F16), "64-row tiles: fp16 form only");
  static_assert(WV == 8 || (WV == 4 && BM == 64 && F16), "4-wave workgroups: 64-row tiles of the fp16 form");
  constexpr int WC = WV / 2;                       // wave columns (wave rows: always 2)
  constexpr int NI = GN / (16 * WC);               // 16-column MFMA tiles per wave: 2 (8 waves) or 4 (4 waves)
  constexpr int BCH = 8 / WV;                      // 16-row chunks of the B tile every wave stages per part
  __shared__ unsigned s_cmax[F16 ? BM : 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int IMG = 128 * 64;                    // bytes of one part image of the B operand tile
  constexpr int IMGA = BM * 64;                    // ... of the A operand tile
  constexpr int MI = BM / 32;                      // 16-row MFMA tiles per wave (wave rows = BM / 2)
  constexpr int BUFB = NP * (IMGA + IMG);          // [A parts | B parts]
  int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wv / WC, wc = wv % WC;
  const int lane = tid & 63, q = lane >> 4, n = lane & 15;
  const int ntiles = tiles_m * tiles_n;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  const int per = ntiles >> 3, rem = ntiles & 7;
  const int tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
  if (idx >= per + (xcd < rem ? 1 : 0)) return;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * GN;

  // staging: wave wv moves rows 16 wv .. +15 of every (operand, part) image; lane -> (row, LDS slot); source slot swizzled
  const int srow = 16 * wv + (lane >> 2), sslot = (lane & 3) ^ gswz(lane >> 2);
  // (fp16 form: both images tile-blocked, f16_blk -- a k-step's tile is 8 KiB contiguous, a wave instruction one contiguous KiB;
  // bf16 forms: row-major)
  const int Np = F16 ? f16_np(N) : N;
  const int arow = min(m0 + srow, M - 1);
  // (a 64-row tile is half of a 128-row block of the image: rows (m0 & 127) .. + 63)
  const __bf16* a_src0 = F16 ? Aimg + (size_t)(m0 >> 7) * (K >> 5) * 4096 + (arow - (m0 & ~127)) * 32 + 8 * sslot : Aimg + (size_t)arow * K + 8 * sslot;
  const __bf16* b_src0 = F16 ? Bimg + (size_t)tn * (K >> 5) * 4096 + srow * 32 + 8 * sslot : Bimg + (size_t)(n0 + srow) * K + 8 * sslot;
  const __bf16 *a_src = a_src0, *b_src = b_src0;
  const size_t a_part = F16 ? (size_t)((M + 127) / 128) * 128 * K : (size_t)M * K, b_part = (size_t)Np * K;
  const size_t kstride = F16 ? 4096 : GK;
  const float* inv_b = reinterpret_cast<const float*>(Bimg + (size_t)NP * Np * K);
  auto stage = [&](int kt) {
    unsigned char* dst = smem_raw + (kt % NBUF) * BUFB + wv * 1024;
    const size_t k0 = kt * kstride;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (BM == 128 || wv < 4)                         // (64-row tiles: waves 0-3 move the A rows; wave-uniform)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src + p * a_part + k0),
                                         (__attribute__((address_space(3))) void*)(dst + p * IMGA), 16, 0, 0);
#pragma unroll
      for (int c = 0; c < BCH; ++c)                    // (chunk wv + WV c: 16 rows = 512 halves of the blocked image, 1 KiB of LDS further on)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src + p * b_part + k0 + (size_t)c * WV * 512),
                                         (__attribute__((address_space(3))) void*)(dst + NP * IMGA + p * IMG + c * WV * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = {0.f, 0.f, 0.f, 0.f};
  // (64-row tiles: waves 4-7 issue half the DMA instructions of waves 0-3; the counted waits below assume 2 NP per tile, which
  // over-waits for those waves -- harmless)
  constexpr int PER_TILE = 2 * NP;
  static_assert(WV == 8 || NBUF == 2, "4-wave workgroups: the counted waits below are written for two tile buffers (vmcnt(0))");

  // split-K (gridDim.y > 1; hopmi_gemm_f16x2_ab_splitk): slab y multiplies k-steps [nk_all y / S, nk_all (y + 1) / S) into C + y M N
  const int nk_all = K / GK, ksp = gridDim.y, ky = blockIdx.y;
  const int kt0 = (int)((long long)nk_all * ky / ksp), nk = (int)((long long)nk_all * (ky + 1) / ksp) - kt0;
  a_src += (size_t)kt0 * kstride;
  b_src += (size_t)kt0 * kstride;
  if (C != nullptr) C += (size_t)ky * M * N;
#pragma unroll
  for (int pre = 0; pre < NBUF - 1; ++pre)
    if (pre < nk) stage(pre);
  if (nk > NBUF - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BM == 128 ? PER_TILE : NP) * (NBUF - 2)) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int fa_off = ((BM / 2) * wr + n) * 64 + ((q ^ gswz(n)) << 4);          // (the swizzle only depends on n: row = 16 k + n)
  const int fb_off = NP * IMGA + (16 * NI * wc + n) * 64 + ((q ^ gswz(n)) << 4);
  for (int kt = 0; kt < nk; ++kt) {
    const bool ahead = kt + NBUF - 1 < nk;
#if HOPMI_AB_EXP == 1                                  // (timing experiment: no DMA in the k-loop)
    if (ahead && K < 0) stage(kt + NBUF - 1);
#else
    if (ahead) stage(kt + NBUF - 1);
#endif
    const unsigned char* buf = smem_raw + (kt % NBUF) * BUFB;
    u32x4 af[MI][NP];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#if HOPMI_AB_EXP == 3                                  // (timing experiment: no fragment reads)
        af[mi][p] = u32x4{(unsigned)(lane + kt), (unsigned)mi, (unsigned)p, 0x3c003c00u};
#else
        af[mi][p] = *reinterpret_cast<const u32x4*>(buf + fa_off + p * IMGA + mi * 16 * 64);
#endif
      }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      u32x4 bf[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#if HOPMI_AB_EXP == 3
        bf[p] = u32x4{(unsigned)(lane ^ kt), (unsigned)ni, (unsigned)p, 0x3c003c00u};
#else
        bf[p] = *reinterpret_cast<const u32x4*>(buf + fb_off + p * IMG + ni * 16 * 64);
#endif
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        f32x4 c = acc[mi][ni];
#if HOPMI_AB_EXP == 2                                  // (timing experiment: no matrix instructions)
#pragma unroll
        for (int p = 0; p < NP; ++p) c[p] += __uint_as_float(af[mi][p][0] ^ bf[p][1]);
#else
#pragma unroll
        for (int s = NP - 1; s >= 0; --s)
#pragma unroll
          for (int i = 0; i <= s; ++i) c = F16 ? mfma_f16(af[mi][i], bf[s - i], c) : mfma_bf16(af[mi][i], bf[s - i], c);
#endif
        acc[mi][ni] = c;
      }
    }
    // tile kt + 1 must have landed (the NBUF - 2 youngest tiles may stay in flight); every wave is done reading tile kt
    if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BM == 128 ? PER_TILE : NP) * (NBUF - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // epilogue (the split form's, gemm_split_kernel: bias / GELU (+ the pre-activation) / GELU gradient; the row maxima of what is
  // written per column tile for the GEMM that consumes C)
  if (F16 && c_rowmax != nullptr && tid < BM) s_cmax[tid] = 0u;
  if (F16 && c_rowmax != nullptr) __syncthreads();
  unsigned rmx[F16 ? MI : 1][4];
#pragma unroll
  for (int mi = 0; mi < (F16 ? MI : 1); ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) rmx[mi][r] = 0u;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + 16 * NI * wc + 16 * ni + n;
    if (F16 && col >= N) continue;                   // (pad columns of the last column tile)
    const float bv = bias != nullptr ? bias[col] : 0.f;
    const float sb = F16 ? inv_b[col] : 1.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (BM / 2) * wr + 16 * mi + 4 * q + r;
        if (row < M) {
          const size_t at = (size_t)row * N + col;
          const float h = F16 ? acc[mi][ni][r] * a_rows[M + row] * sb + bv : acc[mi][ni][r] + bv;
          float out;
          if (!F16 || ep == EP_BIAS) out = h;
          else if (ep == EP_GELU) {
            if (C2 != nullptr) C2[at] = h;
            out = gemm_gelu(h);
          } else out = h * gemm_gelu_grad(aux[at]);
          if (C != nullptr) C[at] = out;
          if (F16) rmx[mi][r] = max(rmx[mi][r], __float_as_uint(out) & 0x7fffffffu);
          if (F16 && io.image != nullptr) {
            // (2-byte stores: the 16 lanes of a quarter write 32 contiguous bytes of the row's 64-byte tile segment)
            const unsigned sbits = scale_bits_for_max(__float_as_uint(io.row_norm[row] * io.mul + io.add) & 0x7fffffffu);
            const float xs = out * __uint_as_float(sbits);
            const _Float16 hi = (_Float16)xs;
            split_check(xs, hi, __LINE__);
#ifdef HOPMI_CHECK_SPLIT
            // (diagnostic build: the first violation of the a-priori bound leaves its coordinates in words [8..15])
            if ((__builtin_bit_cast(unsigned short, hi) & 0x7c00u) == 0x7c00u && fabsf(xs) <= 3.4028235e38f && g_split_status != nullptr &&
                atomicCAS(&g_split_status[8], 0u, 1u) == 0u) {
              unsigned* w = g_split_status;
              w[9] = (unsigned)row; w[10] = (unsigned)col; w[11] = __float_as_uint(io.row_norm[row]); w[12] = __float_as_uint(io.mul);
              w[13] = __float_as_uint(io.add); w[14] = __float_as_uint(out); w[15] = (unsigned)ep;
            }
#endif
            // (the image this writes is a ROWS image of an [M][N] operand: hopmi_rows_image_f16_bytes, f16_kp(N) = N columns -- the
            // layout its consumer, this kernel's A-operand staging, reads; not the weight images' 128-padded width)
            const size_t ia = f16_blk(row, col, f16_kp(N) >> 5);
            io.image[ia] = hi;
            io.image[(size_t)((M + 127) / 128) * 128 * f16_kp(N) + ia] = (_Float16)(xs - (float)hi);
            if (tn == 0 && ni == 0 && n == 0) { io.scales[row] = __uint_as_float(sbits); io.scales[M + row] = inv_scale(sbits); }
          }
        }
      }
  }
  if (F16 && c_rowmax != nullptr) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsigned v = rmx[mi][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
        if (n == 0) atomicMax(&s_cmax[(BM / 2) * wr + 16 * mi + 4 * q + r], v);
      }
    __syncthreads();
    if (tid < BM && m0 + tid < M) c_rowmax[(size_t)tn * M + m0 + tid] = __uint_as_float(s_cmax[tid]);
  }
}

template <int NP, int NBUF, bool F16 = false, int BM = 128, int WV = 8>
static void launch_gemm_ab(const void* Aimg, const void* Bimg, const float* bias, float* C, int M, int N, int K, hipStream_t st,
                           const float* a_rows = nullptr, int ep = EP_BIAS, float* C2 = nullptr, const float* aux = nullptr,
                           float* c_rowmax = nullptr, AbImageOut io = AbImageOut{nullptr, nullptr, nullptr, 0.f, 0.f}) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + GN - 1) / GN;
  const size_t lds = (size_t)NBUF * NP * (BM + 128) * 64;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_ab_kernel<NP, NBUF, F16, BM, WV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
  const int ntiles = tiles_m * tiles_n;
  const int grid = ((ntiles + 7) / 8) * 8;
  hipLaunchKernelGGL((gemm_split_ab_kernel<NP, NBUF, F16, BM, WV>), dim3(grid), dim3(64 * WV), lds, st, static_cast<const __bf16*>(Aimg),
                     static_cast<const __bf16*>(Bimg), bias, C, M, N, K, tiles_m, tiles_n, a_rows, ep, C2, aux, c_rowmax, io);
}

// Tile choice by the number of 128 x 128 tiles, measured at the frozen BERT's shapes for M = 4352 (TED, batch 128) and M = 2176
// (TED-Expressive, batch 64) (tools/bench_gemm.py [--m M], HOPMI_GEMM_TILE forces a form): double-buffered with one workgroup
// per CU when the tiles cover the chip about once (160..256 tiles, e.g. N = 768 at M = 4352: 204), single-buffered with two
// workgroups per CU above that (306, 408, 612, 816 tiles), and 64-row tiles when 128-row tiles would leave most CUs idle
// (N = 768 at M = 2176: 102 tiles -> 204; 38 -> 29.5 us, 129 -> 103 us; at 204+ tiles the 64-row form loses 5-15 %).
// Also measured and dropped: 4 waves of 64 x 64 with two workgroups per CU (-8...-30 %), a second fragment register set read
// one step ahead (-12...-30 %).
template <int NP, bool F16 = false>
static int launch_gemm_split(const float* A, const void* Bimg, const float* bias, float* C, int M, int N, int K, hipStream_t st,
                             int ep = EP_BIAS, float* C2 = nullptr, const float* aux = nullptr, const float* a_rows = nullptr,
                             int a_parts = 0, float* c_rowmax = nullptr) {
  const int t128 = ((M + 127) / 128) * ((N + GN - 1) / GN);
  const int forced = env_int("HOPMI_GEMM_TILE", 0);       // diagnostics: 1 = 128/DB, 2 = 128/!DB, 3 = 64/!DB
  const int mode = (forced >= 1 && forced <= 3) ? forced : (t128 < 160 ? 3 : (t128 <= 256 ? 1 : 2));
  if (F16 && (K % GK) != 0) {                             // K ends inside the last k-step: the guarded A loads
    if (mode == 1) launch_gemm_variant<NP, true, 128, F16, F16>(A, Bimg, bias, C, M, N, K, st, ep, C2, aux, a_rows, a_parts, c_rowmax);
    else if (mode == 2) launch_gemm_variant<NP, false, 128, F16, F16>(A, Bimg, bias, C, M, N, K, st, ep, C2, aux, a_rows, a_parts, c_rowmax);
    else launch_gemm_variant<NP, false, 64, F16, F16>(A, Bimg, bias, C, M, N, K, st, ep, C2, aux, a_rows, a_parts, c_rowmax);
    return check_launch("hopmi_gemm_split");
  }
  if (mode == 1) launch_gemm_variant<NP, true, 128, F16>(A, Bimg, bias, C, M, N, K, st, ep, C2, aux, a_rows, a_parts, c_rowmax);
  else if (mode == 2) launch_gemm_variant<NP, false, 128, F16>(A, Bimg, bias, C, M, N, K, st, ep, C2, aux, a_rows, a_parts, c_rowmax);
  else launch_gemm_variant<NP, false, 64, F16>(A, Bimg, bias, C, M, N, K, st, ep, C2, aux, a_rows, a_parts, c_rowmax);
  return check_launch("hopmi_gemm_split");
}

}  // namespace hopmi

using namespace hopmi;

#ifdef HOPMI_STAMPS
extern "C" int hopmi_debug_set_stamps_gemm(long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t hopmi_gemm_f16x2_image_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return (size_t)2 * f16_np(N) * f16_kp(K) * sizeof(_Float16) + (size_t)f16_np(N) * sizeof(float);
}

extern "C" int hopmi_gemm_f16x2_prepare(const float* W, int N, int K, void* image, void* stream) {
  if (!W || !image || N <= 0 || K <= 0 || (K & 1) || (reinterpret_cast<uintptr_t>(W) & 7)) {
    set_error("hopmi_gemm_f16x2_prepare: need W [N][K] (8-byte aligned), image, even K (N=%d K=%d)", N, K);
    return HOPMI_EINVAL;
  }
  const int Np = f16_np(N), Kp = f16_kp(K);
  float* inv = reinterpret_cast<float*>(static_cast<unsigned char*>(image) + (size_t)2 * Np * Kp * sizeof(_Float16));
  hipLaunchKernelGGL(f16_prepare_rows_kernel, dim3((Np + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), W, N, K, Np, Kp,
                     static_cast<unsigned*>(image), inv);
  return check_launch("hopmi_gemm_f16x2_prepare");
}

extern "C" size_t hopmi_gemm_split_image_bytes(int N, int K, int parts) {
  if (N > 0 && K > 0 && parts == F16_PARTS) return hopmi_gemm_f16x2_image_bytes(N, K);
  return (N > 0 && K > 0 && (parts == 2 || parts == 3)) ? (size_t)parts * N * K * sizeof(__bf16) : 0;
}

extern "C" int hopmi_row_scales(const float* A, int M, int K, float* scales, void* stream) {
  if (!A || !scales || M <= 0 || K <= 0 || (K & 3) || (reinterpret_cast<uintptr_t>(A) & 15)) {
    set_error("hopmi_row_scales: need A (16-byte aligned), scales [2 M] and K %% 4 == 0 (M=%d K=%d)", M, K);
    return HOPMI_EINVAL;
  }
  hipLaunchKernelGGL(row_scales_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), A, M, K, scales);
  return check_launch("hopmi_row_scales");
}

extern "C" size_t hopmi_rows_image_f16_bytes(int M, int K) {
  // (K % 32 == 0: the image has K columns; any other even K: f16_kp(K) columns, the pad zero)
  return (M > 0 && K > 0 && K % 2 == 0) ? (size_t)2 * ((M + 127) / 128 * 128) * f16_kp(K) * sizeof(_Float16) : 0;
}

extern "C" int hopmi_rows_image_f16(const float* A, int M, int K, void* image, float* scales, void* stream) {
  if (!A || !image || !scales || M <= 0 || K <= 0 || (K & 1) || (reinterpret_cast<uintptr_t>(A) & 7)) {
    set_error("hopmi_rows_image_f16: need A (8-byte aligned), image (hopmi_rows_image_f16_bytes), scales [2 M], even K (M=%d K=%d)", M, K);
    return HOPMI_EINVAL;
  }
  if (K % GK == 0 && !(reinterpret_cast<uintptr_t>(A) & 15))
    hipLaunchKernelGGL(f16_rows_image_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), A, M, K,
                       static_cast<unsigned*>(image), scales);
  else
    hipLaunchKernelGGL(f16_rows_image_ragged_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), A, M, K, f16_kp(K),
                       static_cast<unsigned*>(image), scales);
  return check_launch("hopmi_rows_image_f16");
}

// Split-K of the LDS-DMA form for a product with few tiles and a huge contraction (the mapping layer's S = W_map E + b: 1500 x 768
// outputs, K = 30522): C = A Bt^T + row_bias[:, None], A and Bt as images with f16_kp(K) columns.  `splits` slabs of partial products in
// `workspace`, added in index order by a second launch (bitwise reproducible).
static int ab_splits(int M, int N, int K) {
  const int forced = env_int("HOPMI_GEMM_AB_SPLITS", 0);
  const int nk = f16_kp(K) / GK;
  const int tiles = ((M + 63) / 64) * ((N + GN - 1) / GN);
  int sp = forced > 0 ? forced : (768 + tiles - 1) / tiles;          // about one round of three workgroups per CU
  if (sp > nk / 16) sp = nk / 16;                                    // at least 16 k-steps per slab
  if (sp < 1) sp = 1;
  if (sp > 64) sp = 64;
  return sp;
}

extern "C" size_t hopmi_gemm_f16x2_ab_splitk_ws_floats(int M, int N, int K) {
  return (M > 0 && N > 0 && K > 0) ? (size_t)ab_splits(M, N, K) * M * N : 0;
}

extern "C" int hopmi_gemm_f16x2_ab_splitk(const void* Aimage, const float* a_scales, const void* Bimage, const float* row_bias, float* C,
                                          int M, int N, int K, float* workspace, void* stream) {
  if (!Aimage || !a_scales || !Bimage || !C || !workspace) { set_error("hopmi_gemm_f16x2_ab_splitk: null pointer argument"); return HOPMI_EINVAL; }
  if (M <= 0 || N <= 0 || K <= 0 || (K & 1) || (N & 3)) {
    set_error("hopmi_gemm_f16x2_ab_splitk: need even K and N %% 4 == 0 (M=%d N=%d K=%d)", M, N, K);
    return HOPMI_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int Kp = f16_kp(K), sp = ab_splits(M, N, K);
  {
    constexpr int BM = 64;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + GN - 1) / GN;
    const size_t lds = (size_t)2 * 2 * (BM + 128) * 64;
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_ab_kernel<2, 2, true, BM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024) != hipSuccess)
        (void)hipGetLastError();
      attr_done = true;
    }
    const int grid = ((tiles_m * tiles_n + 7) / 8) * 8;
    hipLaunchKernelGGL((gemm_split_ab_kernel<2, 2, true, BM>), dim3(grid, sp), dim3(GT), lds, st, static_cast<const __bf16*>(Aimage),
                       static_cast<const __bf16*>(Bimage), static_cast<const float*>(nullptr), workspace, M, N, Kp, tiles_m, tiles_n, a_scales,
                       (int)EP_BIAS, static_cast<float*>(nullptr), static_cast<const float*>(nullptr), static_cast<float*>(nullptr),
                       AbImageOut{nullptr, nullptr, nullptr, 0.f, 0.f});
  }
  if (int e = check_launch("hopmi_gemm_f16x2_ab_splitk")) return e;
  hipLaunchKernelGGL(gemm_splitk_sum_kernel, dim3((unsigned)(((size_t)M * N / 4 + 255) / 256)), dim3(256), 0, st, workspace, sp, row_bias, C, M, N);
  return check_launch("hopmi_gemm_f16x2_ab_splitk(sum)");
}

// 64-row tiles for the LDS-DMA form: where 128-row tiles would leave most of the chip idle (< 160 tiles), and up to ~700 of them --
// more, smaller workgroups in flight hide the staging latency better (tools/probes/bench_ab.py, M = 4352, us, 128 / 64 rows: N = 768
// K = 768 27.6 / 23.8, N = 2304 60.4 / 56.5, N = 768 K = 2304 59.8 / 56.2, N = 768 K = 3072 75.5 / 75.6; N = 3072 (816 tiles) 78.5 /
// 83.0: the B panel is re-read twice as often).  HOPMI_GEMM_AB_BM64: 0 never, 2 always.
static bool ab_half_tiles(int M, int N) {
  const int mode = env_int("HOPMI_GEMM_AB_BM64", 1);
  if (mode == 0) return false;
  if (mode == 2) return true;
  (void)M; (void)N;
  return true;       // (in the training step 64-row tiles win at every shape: 15.01 -> 14.81 ms per step against 128-row tiles, A/B on one box)
}

// 4-wave workgroups of 32 x 64 per wave (gemm_split_ab_kernel, WV): HOPMI_GEMM_AB_WAVES = 4 / 8 forces a form
static bool ab_four_waves() { return env_int("HOPMI_GEMM_AB_WAVES", 8) == 4; }

extern "C" int hopmi_gemm_f16x2_ab_ep(const void* Aimage, const float* a_scales, const void* Bimage, const float* bias, float* C, float* C2,
                                      const float* aux, float* c_rowmax, int M, int N, int K, int epilogue, void* stream) {
  if (!Aimage || !a_scales || !Bimage || !C) { set_error("hopmi_gemm_f16x2_ab: null pointer argument"); return HOPMI_EINVAL; }
  if (M <= 0 || N <= 0 || K <= 0 || K % GK) {
    set_error("hopmi_gemm_f16x2_ab: need K %% 32 == 0 (M=%d N=%d K=%d)", M, N, K);
    return HOPMI_EINVAL;
  }
  if (epilogue < EP_BIAS || epilogue > EP_GELU_GRAD || (epilogue == EP_GELU_GRAD && !aux)) {
    set_error("hopmi_gemm_f16x2_ab: epilogue %d (0 bias, 1 gelu, 2 gelu gradient: needs aux)", epilogue);
    return HOPMI_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // tile buffers: two (64 KB of LDS: two workgroups per CU) -- with the tile-blocked images the faster form at every BERT shape
  // (tools/probes/bench_ab.py, M = 4352, us, two / three buffers: N = 2304 59 / 77, N = 3072 77 / 102, N = K = 768 26.6 / 27.0,
  // K = 3072 74 / 79, K = 2304 58 / 60); HOPMI_GEMM_NBUF=3 keeps the three-buffer form reachable
  const int forced = env_int("HOPMI_GEMM_NBUF", 0);
  const int nbuf = forced == 3 ? 3 : 2;
  // 64-row tiles when 128-row tiles would leave most of the chip idle (as the split form's mode 3)
  const bool half = ab_half_tiles(M, N);
  if (half && nbuf == 3) launch_gemm_ab<2, 3, true, 64>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, c_rowmax);
  else if (half && ab_four_waves()) launch_gemm_ab<2, 2, true, 64, 4>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, c_rowmax);
  else if (half) launch_gemm_ab<2, 2, true, 64>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, c_rowmax);
  else if (nbuf == 2) launch_gemm_ab<2, 2, true>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, c_rowmax);
  else launch_gemm_ab<2, 3, true>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, c_rowmax);
  return check_launch("hopmi_gemm_f16x2_ab");
}

extern "C" int hopmi_gemm_f16x2_ab_img(const void* Aimage, const float* a_scales, const void* Bimage, const float* bias, float* C, float* C2,
                                       const float* aux, int M, int N, int K, int epilogue, void* out_image, float* out_scales,
                                       const float* row_norm, float bound_mul, float bound_add, void* stream) {
  if (!Aimage || !a_scales || !Bimage || !out_image || !out_scales || !row_norm) { set_error("hopmi_gemm_f16x2_ab_img: null pointer argument"); return HOPMI_EINVAL; }
  if (M <= 0 || N <= 0 || K <= 0 || K % GK || N % GK) {
    set_error("hopmi_gemm_f16x2_ab_img: need K %% 32 == 0 and N %% 32 == 0 (M=%d N=%d K=%d)", M, N, K);
    return HOPMI_EINVAL;
  }
  if (epilogue < EP_BIAS || epilogue > EP_GELU_GRAD || (epilogue == EP_GELU_GRAD && !aux) || !(bound_mul >= 0.f) || !(bound_add >= 0.f)) {
    set_error("hopmi_gemm_f16x2_ab_img: epilogue %d / bound (%g, %g)", epilogue, (double)bound_mul, (double)bound_add);
    return HOPMI_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const AbImageOut io{static_cast<_Float16*>(out_image), out_scales, row_norm, bound_mul, bound_add};
  const bool half = ab_half_tiles(M, N);
  if (half && env_int("HOPMI_GEMM_NBUF", 0) == 3) launch_gemm_ab<2, 3, true, 64>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, nullptr, io);
  else if (half && ab_four_waves()) launch_gemm_ab<2, 2, true, 64, 4>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, nullptr, io);
  else if (half) launch_gemm_ab<2, 2, true, 64>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, nullptr, io);
  else if (env_int("HOPMI_GEMM_NBUF", 0) == 3) launch_gemm_ab<2, 3, true>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, nullptr, io);
  else launch_gemm_ab<2, 2, true>(Aimage, Bimage, bias, C, M, N, K, st, a_scales, epilogue, C2, aux, nullptr, io);
  return check_launch("hopmi_gemm_f16x2_ab_img");
}

extern "C" int hopmi_gemm_f16x2_ab(const void* Aimage, const float* a_scales, const void* Bimage, const float* bias, float* C, int M,
                                   int N, int K, void* stream) {
  return hopmi_gemm_f16x2_ab_ep(Aimage, a_scales, Bimage, bias, C, nullptr, nullptr, nullptr, M, N, K, EP_BIAS, stream);
}

extern "C" int hopmi_gemm_f16x2_tiles_n(int N) { return N > 0 ? (N + GN - 1) / GN : 0; }

extern "C" int hopmi_gemm_f16x2(const float* A, const float* a_scales, int a_parts, const void* Bimage, const float* bias, float* C,
                                float* C2, const float* aux, float* c_rowmax, int M, int N, int K, int epilogue, void* stream) {
  if (!A || !a_scales || !Bimage || !C || a_parts < 0 || a_parts > 1024) {
    set_error("hopmi_gemm_f16x2: null pointer argument / a_parts=%d", a_parts);
    return HOPMI_EINVAL;
  }
  if (M <= 0 || N <= 0 || K <= 0 || (K & 3) || (reinterpret_cast<uintptr_t>(A) & 15)) {
    set_error("hopmi_gemm_f16x2: need K %% 4 == 0 and a 16-byte aligned A (M=%d N=%d K=%d)", M, N, K);
    return HOPMI_EINVAL;
  }
  if (epilogue < EP_BIAS || epilogue > EP_GELU_GRAD || (epilogue == EP_GELU_GRAD && !aux)) {
    set_error("hopmi_gemm_f16x2: epilogue %d (0 bias, 1 gelu, 2 gelu gradient: needs aux)", epilogue);
    return HOPMI_EINVAL;
  }
  return launch_gemm_split<2, true>(A, Bimage, bias, C, M, N, K, static_cast<hipStream_t>(stream), epilogue, C2, aux, a_scales, a_parts,
                                    c_rowmax);
}

extern "C" int hopmi_gemm_split_prepare(const float* W, int N, int K, int parts, void* image, void* stream) {
  if (!W || !image || N <= 0 || K <= 0 || (K & 1) || (parts != 2 && parts != 3 && parts != F16_PARTS)) {
    set_error("hopmi_gemm_split_prepare: need W, image, even K and parts in {2, 3, 16} (N=%d K=%d parts=%d)", N, K, parts);
    return HOPMI_EINVAL;
  }
  const size_t n = (size_t)N * K;
  const unsigned grid = (unsigned)((n / 2 + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (parts == F16_PARTS) return hopmi_gemm_f16x2_prepare(W, N, K, image, stream);
  if (parts == 2) hipLaunchKernelGGL(gemm_split_prepare_kernel<2>, dim3(grid), dim3(256), 0, st, W, n, static_cast<unsigned*>(image));
  else hipLaunchKernelGGL(gemm_split_prepare_kernel<3>, dim3(grid), dim3(256), 0, st, W, n, static_cast<unsigned*>(image));
  return check_launch("hopmi_gemm_split_prepare");
}

extern "C" int hopmi_gemm_split_ab(const void* Aimage, const void* Bimage, const float* bias, float* C, int M, int N, int K, int parts,
                                   void* stream) {
  if (!Aimage || !Bimage || !C) { set_error("hopmi_gemm_split_ab: null pointer argument"); return HOPMI_EINVAL; }
  if (M <= 0 || N <= 0 || K <= 0 || N % GN || K % GK || (parts != 2 && parts != 3)) {
    set_error("hopmi_gemm_split_ab: need N %% 128 == 0, K %% 32 == 0, parts in {2, 3} (M=%d N=%d K=%d parts=%d)", M, N, K, parts);
    return HOPMI_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nbuf = env_int("HOPMI_GEMM_NBUF", 3);
  if (parts == 2) { if (nbuf == 2) launch_gemm_ab<2, 2>(Aimage, Bimage, bias, C, M, N, K, st); else launch_gemm_ab<2, 3>(Aimage, Bimage, bias, C, M, N, K, st); }
  else { if (nbuf == 2) launch_gemm_ab<3, 2>(Aimage, Bimage, bias, C, M, N, K, st); else launch_gemm_ab<3, 3>(Aimage, Bimage, bias, C, M, N, K, st); }
  return check_launch("hopmi_gemm_split_ab");
}

extern "C" int hopmi_gemm_split(const float* A, const void* Bimage, const float* bias, float* C, int M, int N, int K, int parts,
                                void* stream) {
  if (!A || !Bimage || !C) { set_error("hopmi_gemm_split: null pointer argument"); return HOPMI_EINVAL; }
  if (M <= 0 || N <= 0 || K <= 0 || N % GN || K % GK || (parts != 2 && parts != 3)) {
    set_error("hopmi_gemm_split: need N %% 128 == 0, K %% 32 == 0, parts in {2, 3} (M=%d N=%d K=%d parts=%d)", M, N, K, parts);
    return HOPMI_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  return parts == 2 ? launch_gemm_split<2>(A, Bimage, bias, C, M, N, K, st) : launch_gemm_split<3>(A, Bimage, bias, C, M, N, K, st);
}

extern "C" int hopmi_gemm_split_ep(const float* A, const void* Bimage, const float* bias, float* C, float* C2, const float* aux, int M,
                                   int N, int K, int parts, int epilogue, void* stream) {
  if (!A || !Bimage || !C) { set_error("hopmi_gemm_split_ep: null pointer argument"); return HOPMI_EINVAL; }
  if (M <= 0 || N <= 0 || K <= 0 || N % GN || K % GK || (parts != 2 && parts != 3)) {
    set_error("hopmi_gemm_split_ep: need N %% 128 == 0, K %% 32 == 0, parts in {2, 3} (M=%d N=%d K=%d parts=%d)", M, N, K, parts);
    return HOPMI_EINVAL;
  }
  if (epilogue < EP_BIAS || epilogue > EP_GELU_GRAD || (epilogue == EP_GELU_GRAD && !aux)) {
    set_error("hopmi_gemm_split_ep: epilogue %d (0 bias, 1 gelu, 2 gelu gradient: needs aux)", epilogue);
    return HOPMI_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  return parts == 2 ? launch_gemm_split<2>(A, Bimage, bias, C, M, N, K, st, epilogue, C2, aux)
                    : launch_gemm_split<3>(A, Bimage, bias, C, M, N, K, st, epilogue, C2, aux);
}

HOPMI_SPLIT_STATUS_SETTER(gemm)
