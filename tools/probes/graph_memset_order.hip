// Probe: is a hipMemsetAsync recorded as a graph memset node ordered before the kernel node that follows it in the captured
// stream, when eager launches are queued on the same stream between replays?  Each replay: memset(sem, 0) -> check kernel
// (counts words that are not 0, then dirties them).  Build: hipcc --offload-arch=gfx950 -O2 graph_memset_order.hip -o graph_memset_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void check_and_dirty(int* sem, int n, unsigned long long* errors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    if (sem[i] != 0) atomicAdd(errors, 1ull);
    sem[i] = 7;
  }
}
__global__ void busy(float* x, int n, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float v = x[i]; for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f; x[i] = v; }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 3000, eager = argc > 2 ? atoi(argv[2]) : 20, nodes = argc > 3 ? atoi(argv[3]) : 8;
  const int n = 4096;
  int* sem; unsigned long long* err; float* x;
  CK(hipMalloc(&sem, nodes * n * sizeof(int))); CK(hipMalloc(&err, 8)); CK(hipMalloc(&x, (1 << 20) * sizeof(float)));
  CK(hipMemset(sem, 0, nodes * n * sizeof(int))); CK(hipMemset(err, 0, 8)); CK(hipMemset(x, 0, (1 << 20) * sizeof(float)));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int k = 0; k < nodes; ++k) {                       // a chain: busy kernel, memset, dependent kernel
    hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, st, x, 1 << 16, 200);
    CK(hipMemsetAsync(sem + k * n, 0, n * sizeof(int), st));
    hipLaunchKernelGGL(check_and_dirty, dim3(n / 256), dim3(256), 0, st, sem + k * n, n, err);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < reps; ++r) {
    CK(hipGraphLaunch(ge, st));
    for (int e = 0; e < eager; ++e) hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, st, x + (1 << 18), 1 << 14, 50);
  }
  CK(hipStreamSynchronize(st));
  unsigned long long h = 0; CK(hipMemcpy(&h, err, 8, hipMemcpyDeviceToHost));
  printf("replays %d, eager launches between replays %d, memset->kernel pairs per graph %d: %llu stale words seen\n", reps, eager, nodes, h);
  return 0;
}
