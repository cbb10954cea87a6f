// Adam over a LIST of fp32 tensors in one launch (round 6): the optimizer step of the recorded training step
// (reference: train_eval/train_llm.py:86 `model_optim.step()`, torch.optim.Adam as run_ted.py:330-333 builds it).
//
// torch's fused Adam (multi_tensor_apply) moves the generator's 65.8 M parameters -- 7 streams of 263 MB: read p, g, m, v, write
// p, m, v -- at 3.2 TB/s (576 us; 3.8 on one flat tensor of the same size): its chunking serves the 172 small tensors badly (they
// cost 210 us for 17 % of the elements) and a 4-read / 3-write copy of the same bytes runs at 4.9 TB/s on this chip
// (tools/probes/adam_floor.py).  Here: a table of (p, g, m, v, n) per tensor and a list of (tensor, chunk) work items of 8 192
// elements, one workgroup per item, 16-byte accesses wherever the four pointers allow (scalar otherwise: parameters that are views
// into packed buffers at odd offsets -- small ones), the bias corrections from the optimizer's own device-side step counter.
// Arithmetic = torch's _fused_adam_ (no weight decay, no amsgrad, no maximize; the host refuses anything else):
//   m = lerp(m, g, 1 - beta1);  v = beta2 v + (1 - beta2) g g;  p -= (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include "common.h"

namespace hopmi {

constexpr int ADAM_CHUNK = 8192;                   // elements per work item: 256 threads x 8 float4

struct AdamTensor { float* p; const float* g; float* m; float* v; long long n; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float w2, float step_size, float bc2s, float eps) {
  // torch: lerp(m, g, w) with w = 1 - beta1: m + w (g - m) below one half, g - (g - m)(1 - w) from one half on
  m = w1 < 0.5f ? m + w1 * (g - m) : g - (g - m) * (1.f - w1);
  v = v * beta2 + w2 * g * g;
  const float denom = sqrtf(v) / bc2s + eps;
  p -= step_size * m / denom;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTensor* __restrict__ T, const int2* __restrict__ items, double lr,
                                                         double beta1, double beta2, double eps, const float* __restrict__ step) {
  // the step's scalars as torch forms them: hyper-parameters are doubles, 1 - beta and 1 - beta^t are taken in double and rounded
  // to fp32 once (in fp32, 1 - 0.999f is off by 1.3e-5 relative -- and so would every second moment be)
  __shared__ float sc[6];
  if (threadIdx.x == 0) {
    const double st = (double)*step;                               // already advanced for this step (host: _foreach_add_)
    const float bc1 = (float)(1.0 - pow(beta1, st)), bc2 = (float)(1.0 - pow(beta2, st));
    sc[0] = (float)(1.0 - beta1);
    sc[1] = (float)beta2;
    sc[2] = (float)(1.0 - beta2);
    sc[3] = (float)(lr / (double)bc1);
    sc[4] = sqrtf(bc2);
    sc[5] = (float)eps;
  }
  __syncthreads();
  const float w1 = sc[0], b2 = sc[1], w2 = sc[2], step_size = sc[3], bc2s = sc[4], epsf = sc[5];
  const int2 it = items[blockIdx.x];
  const AdamTensor t = T[it.x];
  const long long base = (long long)it.y * ADAM_CHUNK;
  const long long left = t.n - base;
  const int cnt = left < ADAM_CHUNK ? (int)left : ADAM_CHUNK;
  float* p = t.p + base; const float* g = t.g + base; float* m = t.m + base; float* v = t.v + base;
  const bool vec = !((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15);
  if (vec) {
    const int n4 = cnt >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
      const float4 gg = reinterpret_cast<const float4*>(g)[i];
      adam_one(pp.x, gg.x, mm.x, vv.x, w1, b2, w2, step_size, bc2s, epsf);
      adam_one(pp.y, gg.y, mm.y, vv.y, w1, b2, w2, step_size, bc2s, epsf);
      adam_one(pp.z, gg.z, mm.z, vv.z, w1, b2, w2, step_size, bc2s, epsf);
      adam_one(pp.w, gg.w, mm.w, vv.w, w1, b2, w2, step_size, bc2s, epsf);
      reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int i = 4 * n4 + threadIdx.x; i < cnt; i += 256) adam_one(p[i], g[i], m[i], v[i], w1, b2, w2, step_size, bc2s, epsf);
  } else {
    for (int i = threadIdx.x; i < cnt; i += 256) adam_one(p[i], g[i], m[i], v[i], w1, b2, w2, step_size, bc2s, epsf);
  }
}

}  // namespace hopmi

using namespace hopmi;

extern "C" int hopmi_adam_chunk(void) { return ADAM_CHUNK; }

extern "C" int hopmi_adam_multi(const void* tensors, const void* items, int n_items, double lr, double beta1, double beta2, double eps,
                                const float* step, void* stream) {
  if (!tensors || !items || !step || n_items <= 0) { set_error("hopmi_adam_multi: null pointer argument / n_items=%d", n_items); return HOPMI_EINVAL; }
  if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) {
    set_error("hopmi_adam_multi: lr=%g beta1=%g beta2=%g eps=%g", lr, beta1, beta2, eps);
    return HOPMI_EINVAL;
  }
  hipLaunchKernelGGL(adam_multi_kernel, dim3(n_items), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const AdamTensor*>(tensors),
                     static_cast<const int2*>(items), lr, beta1, beta2, eps, step);
  return check_launch("hopmi_adam_multi");
}
