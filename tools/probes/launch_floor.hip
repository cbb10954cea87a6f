// Probe: duration of an (almost) empty kernel as a function of the launch shape (workgroups, threads, dynamic LDS),
// measured as back-to-back launches between two events and as per-dispatch start/stop events (hipExtLaunchKernelGGL).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void empty_kernel(float* p) { extern __shared__ float s[]; if (p && threadIdx.x == 9999) p[0] = s[0]; }
int main() {
  float* d; hipMalloc(&d, 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {256, 512, 1024}; const int threads[] = {64, 256, 512, 1024}; const int ldss[] = {0, 32 * 1024, 64 * 1024, 148 * 1024};
  for (int g : grids) for (int t : threads) for (int l : ldss) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(t), l, 0, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(t), l, 0, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> v;
    for (int i = 0; i < 30; ++i) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipExtLaunchKernelGGL(empty_kernel, dim3(g), dim3(t), l, 0, a, b, 0, d);
      hipEventSynchronize(b); float x; hipEventElapsedTime(&x, a, b); v.push_back(x * 1e3f);
      hipEventDestroy(a); hipEventDestroy(b);
    }
    std::sort(v.begin(), v.end());
    printf("grid %4d threads %4d lds %6d: back-to-back %.2f us/launch, dispatch events median %.2f us\n", g, t, l, ms * 1e3f / 200, v[15]);
  }
  return 0;
}
