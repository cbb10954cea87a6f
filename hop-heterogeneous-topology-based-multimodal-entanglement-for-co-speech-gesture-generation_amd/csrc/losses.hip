// The generator losses of one training step (reference: train_eval/train_llm.py:46-79) in two launches instead of the ~75
// elementwise / reduction launches the same arithmetic takes as separate tensor operations (each one pays the ~4 us launch
// floor: ~0.5 ms per step forward + backward):
//
//   huber   = smooth_l1(out / 0.1, target / 0.1) * 0.1                                   mean over all B*F elements   (:46)
//   pose_b  = sum_f smooth_l1(out[b] / beta, out_rand[b] / beta) * beta,  beta = 0.05                               (:60-62)
//   z_b     = mean_j |z_context[b] - z_rand[b]|                                                                     (:65-66)
//   div_reg = mean_b max(-pose_b / (z_b + 1e-5), -1000)                                                             (:67-69)
//   kld     = -0.5 mean(1 + logvar - mu^2 - exp(logvar))                                 (z_type == 'speaker')      (:71)
//   total   = w_reg huber + w_div div_reg + w_kld kld                                                               (:75-77)
//
// out_rand, z_context and z_rand enter detached (as in the reference), so the gradient goes to out, mu and logvar only.
// Forward: one workgroup per clip leaves four partial sums (fixed-order tree in LDS); the last workgroup to arrive turns
// them into the four scalars and the per-clip factor of the div_reg gradient -- bitwise reproducible.  Backward: one
// elementwise launch.
#include "common.h"

namespace hopmi {

__device__ __forceinline__ float smooth_l1(float d) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float smooth_l1_grad(float d) { return fminf(fmaxf(d, -1.f), 1.f); }

constexpr float LOSS_BETA_HUBER = 0.1f, LOSS_BETA_DIV = 0.05f;

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {     // fixed-order tree; every thread gets the result
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  return red[0];
}

// ws: [B][4] partial sums, [B] div_reg gradient factors, then the arrival counter (int, zero before the first call and
// left at zero by every call)
__global__ __launch_bounds__(256) void hop_losses_fwd_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                             const float* __restrict__ out_rand, const float* __restrict__ zc,
                                                             const float* __restrict__ zr, const float* __restrict__ mu,
                                                             const float* __restrict__ lv, int B, int F, int Z, float w_reg,
                                                             float w_div, float w_kld, float* __restrict__ vals, float* ws) {
  __shared__ float red[256];
  __shared__ int last;
  const int b = blockIdx.x, tid = threadIdx.x;
  float hub = 0.f, pose = 0.f, zl = 0.f, kl = 0.f;
  for (int f = tid; f < F; f += 256) {
    const float o = out[(size_t)b * F + f];
    hub += smooth_l1((o - target[(size_t)b * F + f]) / LOSS_BETA_HUBER);
    if (out_rand != nullptr) pose += smooth_l1((o - out_rand[(size_t)b * F + f]) / LOSS_BETA_DIV) * LOSS_BETA_DIV;
  }
  for (int j = tid; j < Z; j += 256) {
    if (zc != nullptr) zl += fabsf(zc[(size_t)b * Z + j] - zr[(size_t)b * Z + j]);
    if (mu != nullptr) {
      const float m = mu[(size_t)b * Z + j], l = lv[(size_t)b * Z + j];
      kl += 1.f + l - m * m - __expf(l);
    }
  }
  hub = block_sum<256>(hub, red);
  pose = block_sum<256>(pose, red);
  zl = block_sum<256>(zl, red);
  kl = block_sum<256>(kl, red);
  int* counter = reinterpret_cast<int*>(ws + (size_t)5 * B);
  if (tid == 0) {
    float* p = ws + (size_t)4 * b;
    __hip_atomic_store(p + 0, hub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, pose, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, Z > 0 ? zl / (float)Z : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 3, kl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == B - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float s_h = 0.f, s_d = 0.f, s_k = 0.f;
  for (int c = tid; c < B; c += 256) {
    const float* p = ws + (size_t)4 * c;
    s_h += __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_k += __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float coef = 0.f;
    if (out_rand != nullptr) {
      const float pose_c = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float den = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1.0e-5f;
      const float d = -(pose_c / den);
      s_d += fmaxf(d, -1000.f);
      coef = d >= -1000.f ? -w_div / ((float)B * den) : 0.f;          // d total / d pose_c (clamp: gradient 1 at the bound, as torch)
    }
    ws[(size_t)4 * B + c] = coef;
  }
  s_h = block_sum<256>(s_h, red);
  s_d = block_sum<256>(s_d, red);
  s_k = block_sum<256>(s_k, red);
  if (tid == 0) {
    const float huber = s_h / ((float)B * (float)F) * LOSS_BETA_HUBER;
    const float div_reg = out_rand != nullptr ? s_d / (float)B : 0.f;
    const float kld = mu != nullptr ? -0.5f * s_k / ((float)B * (float)Z) : 0.f;
    vals[0] = huber; vals[1] = div_reg; vals[2] = kld;
    vals[3] = w_reg * huber + (out_rand != nullptr ? w_div * div_reg : 0.f) + (mu != nullptr ? w_kld * kld : 0.f);
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// gradient of vals[3] w.r.t. out, mu, logvar, times the upstream scalar *g
__global__ __launch_bounds__(256) void hop_losses_bwd_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                             const float* __restrict__ out_rand, const float* __restrict__ mu,
                                                             const float* __restrict__ lv, const float* __restrict__ ws,
                                                             const float* __restrict__ g, int B, int F, int Z, float w_reg,
                                                             float w_kld, float* __restrict__ d_out, float* __restrict__ d_mu,
                                                             float* __restrict__ d_lv) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const float up = *g;
  if (i < (size_t)B * F) {
    const int b = (int)(i / F);
    const float o = out[i];
    float d = w_reg / ((float)B * (float)F) * smooth_l1_grad((o - target[i]) / LOSS_BETA_HUBER);
    if (out_rand != nullptr) d += ws[(size_t)4 * B + b] * smooth_l1_grad((o - out_rand[i]) / LOSS_BETA_DIV);
    d_out[i] = up * d;
  }
  if (mu != nullptr && i < (size_t)B * Z) {
    const float c = up * w_kld / ((float)B * (float)Z);
    d_mu[i] = c * mu[i];
    d_lv[i] = c * -0.5f * (1.f - __expf(lv[i]));
  }
}

__global__ void hop_losses_clear_kernel(float* ws, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ws[i] = 0.f;
}

}  // namespace hopmi

using namespace hopmi;

extern "C" size_t hopmi_hop_losses_ws_floats(int B) { return B > 0 ? (size_t)5 * B + 4 : 0; }

extern "C" int hopmi_hop_losses_fwd(const float* out, const float* target, const float* out_rand, const float* z_context,
                                    const float* z_rand, const float* mu, const float* logvar, int B, int F, int Z, float w_reg,
                                    float w_div, float w_kld, float* vals, float* ws, void* stream) {
  if (!out || !target || !vals || !ws) { set_error("hopmi_hop_losses_fwd: null pointer argument"); return HOPMI_EINVAL; }
  if (B <= 0 || F <= 0 || Z < 0) { set_error("hopmi_hop_losses_fwd: bad sizes B=%d F=%d Z=%d", B, F, Z); return HOPMI_EINVAL; }
  if ((out_rand != nullptr) != (z_context != nullptr) || (z_context != nullptr) != (z_rand != nullptr) ||
      (mu != nullptr) != (logvar != nullptr) || ((out_rand || mu) && Z <= 0)) {
    set_error("hopmi_hop_losses_fwd: out_rand / z_context / z_rand come together, mu / logvar come together, and need Z > 0");
    return HOPMI_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nws = (int)hopmi_hop_losses_ws_floats(B);
  hipLaunchKernelGGL(hop_losses_clear_kernel, dim3((nws + 255) / 256), dim3(256), 0, st, ws, nws);
  hipLaunchKernelGGL(hop_losses_fwd_kernel, dim3(B), dim3(256), 0, st, out, target, out_rand, z_context, z_rand, mu, logvar, B, F, Z,
                     w_reg, w_div, w_kld, vals, ws);
  return check_launch("hopmi_hop_losses_fwd");
}

extern "C" int hopmi_hop_losses_bwd(const float* out, const float* target, const float* out_rand, const float* mu,
                                    const float* logvar, const float* ws, const float* g, int B, int F, int Z, float w_reg,
                                    float w_kld, float* d_out, float* d_mu, float* d_logvar, void* stream) {
  if (!out || !target || !ws || !g || !d_out) { set_error("hopmi_hop_losses_bwd: null pointer argument"); return HOPMI_EINVAL; }
  if (B <= 0 || F <= 0 || Z < 0 || (mu != nullptr && (!logvar || !d_mu || !d_logvar || Z <= 0))) {
    set_error("hopmi_hop_losses_bwd: bad arguments (B=%d F=%d Z=%d)", B, F, Z);
    return HOPMI_EINVAL;
  }
  const size_t n = (size_t)B * (size_t)(F > Z ? F : Z);
  hipLaunchKernelGGL(hop_losses_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), out,
                     target, out_rand, mu, logvar, ws, g, B, F, Z, w_reg, w_kld, d_out, d_mu, d_logvar);
  return check_launch("hopmi_hop_losses_bwd");
}
