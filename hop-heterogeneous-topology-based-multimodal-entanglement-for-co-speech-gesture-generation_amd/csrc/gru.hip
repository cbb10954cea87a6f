// Bidirectional GRU layer recurrence of the HOP pose decoder (reference: model/HOP.py:166-167,248
// nn.GRU(…, hidden 350, 4 layers, bidirectional); the same cell serves the discriminator,
// model/multimodal_context_net.py:236-237,257), forward and back-propagation through time.
//
// Split of the work (torch.nn.GRU semantics, gate order r, z, n):
//   * the input projections gi[b][t][dir][3H] = x W_ih^T + b_ih for ALL time steps are one big GEMM
//     done by the caller (hipBLASLt) -- they have no sequential dependency;
//   * the sequential part -- gh = h_{t-1} W_hh^T + b_hh, the gates, h_t -- runs here, ONE launch per
//     time step (both directions in the same launch).  A dependent launch boundary costs ~1.5 us on
//     MI355X, cheaper and safer than an in-kernel grid barrier, and the T launches of a layer are
//     enqueued from C in one ABI call (capturable into a hipGraph).
// Decomposition of a step: workgroup (jb, bb, dir) owns hidden units [16 jb, 16 jb + 16) of batch rows
// [32 bb, 32 bb + 32): out[32 x (3 gates x 16)] = h_prev[32 x H] W_slice^T, exact-fp32 MFMA 16x16x4 with
// K = H split over the 4 waves (lane quad q of wave w owns the contiguous k range
// [(4w+q) KQ, (4w+q+1) KQ): 8-B loads straight to registers, no LDS staging), partial tiles summed
// through LDS, then the gate math as the epilogue.  h_prev is read from the layer output y itself.
#include "common.h"

namespace hopmi {

constexpr int GRU_BM = 32;     // batch rows per workgroup (2 MFMA row tiles)
constexpr int GRU_NU = 16;     // hidden units per workgroup (one MFMA column tile per gate)
constexpr int RED_LD = 49;     // LDS stride of the [32][48] partial tiles

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// acc[mt][g] += A[rows 16mt..][k] * W[unit rows of gate g][k] over this lane's k range.
// a_ptr[mt] / w_ptr[g] point at this lane's row and first k.  The trip count `nk2` (float2 steps) is
// wave-uniform -- MFMA needs every lane -- and a lane whose range ends early (the K padding) re-reads its
// last valid pair and contributes zeros.
template <int NG>
__device__ __forceinline__ void ksplit_mfma(f32x4 (&acc)[2][NG], const float* (&a_ptr)[2], const bool (&a_ok)[2],
                                            const float* (&w_ptr)[NG], bool w_ok, int nk2, int nk2_valid) {
  const int last = max(nk2_valid - 1, 0);
#pragma unroll 4
  for (int p = 0; p < nk2; ++p) {
    const bool k_ok = p < nk2_valid;
    const int pc = min(p, last);
    float2 a[2], w[NG];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      a[mt] = *reinterpret_cast<const float2*>(a_ptr[mt] + 2 * pc);
      if (!(a_ok[mt] && k_ok)) a[mt] = make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      w[g] = *reinterpret_cast<const float2*>(w_ptr[g] + 2 * pc);
      if (!(w_ok && k_ok)) w[g] = make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        acc[mt][g] = mfma16(a[mt].x, w[g].x, acc[mt][g]);
        acc[mt][g] = mfma16(a[mt].y, w[g].y, acc[mt][g]);
      }
  }
}

// one time step `s` of both directions (dir 0 walks t = s, dir 1 walks t = T-1-s)
__global__ __launch_bounds__(256) void gru_fwd_step_kernel(const float* __restrict__ gi, const float* __restrict__ whh,
                                                           const float* __restrict__ bhh, float* __restrict__ y,
                                                           float* __restrict__ gates, int B, int T, int H, int KQ, int s) {
  __shared__ float red[4 * GRU_BM * RED_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, i = lane & 15;
  const int d = blockIdx.z, j0 = blockIdx.x * GRU_NU, b0 = blockIdx.y * GRU_BM;
  const int t = d ? T - 1 - s : s, tp = d ? t + 1 : t - 1;
  const size_t ystride = (size_t)2 * H;

  if (s > 0) {
    const int k0 = (4 * w + q) * KQ;                            // this lane's k range [k0, k0 + KQ)
    const int nk2_valid = max(0, min(KQ, H - k0)) >> 1;         // H, KQ, k0 even
    const int kc = min(k0, H - 2);                              // keep the (unused) address in bounds
    f32x4 acc[2][3];
    const float* a_ptr[2];
    bool a_ok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int b = b0 + 16 * mt + i;
      a_ok[mt] = b < B;
      a_ptr[mt] = y + ((size_t)min(b, B - 1) * T + tp) * ystride + d * H + kc;
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[mt][g] = {0.f, 0.f, 0.f, 0.f};
    }
    const int ju = j0 + i;
    const bool w_ok = ju < H;
    const float* w_ptr[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) w_ptr[g] = whh + ((size_t)(d * 3 + g) * H + min(ju, H - 1)) * H + kc;
    ksplit_mfma<3>(acc, a_ptr, a_ok, w_ptr, w_ok, KQ >> 1, nk2_valid);
    // partial tile of this wave -> LDS: red[w][row][16 g + col]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * GRU_BM + 16 * mt + 4 * q + r) * RED_LD + 16 * g + i] = acc[mt][g][r];
  }
  __syncthreads();

  // gate epilogue: thread -> (row, unit), two passes of 16 rows
  const int jj = tid & 15, j = j0 + jj;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = (tid >> 4) + 16 * pass, b = b0 + row;
    if (b < B && j < H) {
      float gh[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float v = bhh[(d * 3 + g) * H + j];
        if (s > 0) {
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) v += red[(ww * GRU_BM + row) * RED_LD + 16 * g + jj];
        }
        gh[g] = v;
      }
      const float* gip = gi + (((size_t)b * T + t) * 2 + d) * 3 * H + j;
      const float r = sigmoidf_(gip[0] + gh[0]);
      const float z = sigmoidf_(gip[H] + gh[1]);
      const float n = tanhf(gip[2 * H] + r * gh[2]);
      const float hp = (s > 0) ? y[((size_t)b * T + tp) * ystride + d * H + j] : 0.f;
      y[((size_t)b * T + t) * ystride + d * H + j] = (1.f - z) * n + z * hp;
      float* gp = gates + (((size_t)b * T + t) * 2 + d) * 4 * H + j;
      gp[0] = r; gp[H] = z; gp[2 * H] = n; gp[3 * H] = gh[2];
    }
  }
}

// BPTT step `s` (s = 0 is the LAST step the forward processed).  Per direction, in forward processing
// order p (dir 0: p = t, dir 1: p = T-1-t):
//   D_p   = dy_p + Dz_{p+1} + dgh_{p+1} W_hh          (gradient w.r.t. h_p; Dz_{p+1} = D_{p+1} * z_{p+1})
//   dn    = D_p (1 - z) (1 - n^2);  dz = D_p (h_{p-1} - n) z (1 - z);  dr = dn * hn * r (1 - r)
//   dgi_p = [dr, dz, dn];  dgh_p = [dr, dz, dn * r];  Dz_p = D_p * z
// The workgroup owns units [16 jb, +16) of rows [32 bb, +32): the contraction over the 3H gates of step
// p+1 reads dgh rows (contiguous) and rows of W_hh^T (whhT[dir][unit][3H], transposed by the caller).
__global__ __launch_bounds__(256) void gru_bwd_step_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ gates, const float* __restrict__ whhT,
                                                           float* __restrict__ dgi, float* __restrict__ dgh,
                                                           float* __restrict__ dhz, int B, int T, int H, int KQ, int s) {
  __shared__ float red[4 * GRU_BM * RED_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, i = lane & 15;
  const int d = blockIdx.z, j0 = blockIdx.x * GRU_NU, b0 = blockIdx.y * GRU_BM;
  const int t = d ? s : T - 1 - s;                 // time index processed at BPTT step s
  const int tn = d ? t - 1 : t + 1;                // the step the forward processed right after t
  const int tp = d ? t + 1 : t - 1;                // ... and right before t (source of h_prev)
  const int K = 3 * H;
  const float* dhz_in = dhz + ((size_t)((s + 1) & 1) * 2 + d) * B * H;
  float* dhz_out = dhz + ((size_t)(s & 1) * 2 + d) * B * H;

  if (s > 0) {
    const int k0 = (4 * w + q) * KQ;
    const int nk2_valid = max(0, min(KQ, K - k0)) >> 1;
    const int kc = min(k0, K - 2);
    f32x4 acc[2][1];
    const float* a_ptr[2];
    bool a_ok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int b = b0 + 16 * mt + i;
      a_ok[mt] = b < B;
      a_ptr[mt] = dgh + (((size_t)min(b, B - 1) * T + tn) * 2 + d) * K + kc;
      acc[mt][0] = {0.f, 0.f, 0.f, 0.f};
    }
    const int ju = j0 + i;
    const bool w_ok = ju < H;
    const float* w_ptr[1];
    w_ptr[0] = whhT + ((size_t)d * H + min(ju, H - 1)) * K + kc;
    ksplit_mfma<1>(acc, a_ptr, a_ok, w_ptr, w_ok, KQ >> 1, nk2_valid);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(w * GRU_BM + 16 * mt + 4 * q + r) * RED_LD + i] = acc[mt][0][r];
  }
  __syncthreads();

  const int jj = tid & 15, j = j0 + jj;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = (tid >> 4) + 16 * pass, b = b0 + row;
    if (b < B && j < H) {
      float D = dy[((size_t)b * T + t) * 2 * H + d * H + j];
      if (s > 0) {
        D += dhz_in[(size_t)b * H + j];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) D += red[(ww * GRU_BM + row) * RED_LD + jj];
      }
      const float* gp = gates + (((size_t)b * T + t) * 2 + d) * 4 * H + j;
      const float r = gp[0], z = gp[H], n = gp[2 * H], hn = gp[3 * H];
      const float hp = (s < T - 1) ? y[((size_t)b * T + tp) * 2 * H + d * H + j] : 0.f;
      const float dn = D * (1.f - z) * (1.f - n * n);
      const float dz = D * (hp - n) * z * (1.f - z);
      const float dr = dn * hn * r * (1.f - r);
      const size_t o = (((size_t)b * T + t) * 2 + d) * K + j;
      dgi[o] = dr; dgi[o + H] = dz; dgi[o + 2 * H] = dn;
      dgh[o] = dr; dgh[o + H] = dz; dgh[o + 2 * H] = dn * r;
      dhz_out[(size_t)b * H + j] = D * z;
    }
  }
}

static int even_ceil_div16(int K) {
  int kq = (K + 15) / 16;
  return kq + (kq & 1);
}

static int gru_validate(const void* const* ptrs, int n, int B, int T, int H) {
  for (int i = 0; i < n; ++i)
    if (!ptrs[i]) { set_error("hopmi_gru: null pointer argument #%d", i); return HOPMI_EINVAL; }
  if (B <= 0 || T <= 0) { set_error("hopmi_gru: B=%d T=%d must be > 0", B, T); return HOPMI_EINVAL; }
  if (H < 2 || (H & 1) || H > 4096) { set_error("hopmi_gru: hidden size H=%d must be even and in [2,4096]", H); return HOPMI_EINVAL; }
  return HOPMI_OK;
}

}  // namespace hopmi

using namespace hopmi;

extern "C" int hopmi_gru_fwd(const float* gi, const float* whh, const float* bhh, float* y, float* gates,
                             int B, int T, int H, void* stream) {
  const void* ptrs[] = {gi, whh, bhh, y, gates};
  if (int e = gru_validate(ptrs, 5, B, T, H)) return e;
  const dim3 grid((H + GRU_NU - 1) / GRU_NU, (B + GRU_BM - 1) / GRU_BM, 2);
  const int KQ = even_ceil_div16(H);
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL(gru_fwd_step_kernel, grid, dim3(256), 0, st, gi, whh, bhh, y, gates, B, T, H, KQ, s);
  return check_launch("hopmi_gru_fwd");
}

extern "C" size_t hopmi_gru_bwd_ws_floats(int B, int H) {
  return (B > 0 && H > 0) ? (size_t)4 * B * H : 0;
}

extern "C" int hopmi_gru_bwd(const float* dy, const float* y, const float* gates, const float* whhT,
                             float* dgi, float* dgh, float* ws, int B, int T, int H, void* stream) {
  const void* ptrs[] = {dy, y, gates, whhT, dgi, dgh, ws};
  if (int e = gru_validate(ptrs, 7, B, T, H)) return e;
  const dim3 grid((H + GRU_NU - 1) / GRU_NU, (B + GRU_BM - 1) / GRU_BM, 2);
  const int KQ = even_ceil_div16(3 * H);
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL(gru_bwd_step_kernel, grid, dim3(256), 0, st, dy, y, gates, whhT, dgi, dgh, ws, B, T, H, KQ, s);
  return check_launch("hopmi_gru_bwd");
}
