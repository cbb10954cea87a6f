// Probe: do 256-thread workgroups with N KB of dynamic LDS co-reside on a CU?  Each block runs a
// fixed-latency dependent chain (~5 us); if blocks co-reside, time stays flat as the grid grows to
// k x 256 blocks (k <= blocks/CU), otherwise it grows in steps of one block-time per 256 blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void body(float* out, int iters) {
  extern __shared__ float smem[];
  smem[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float v = 0.f;
  int idx = threadIdx.x;
  for (int i = 0; i < iters; ++i) {      // dependent LDS chain: latency-bound, not throughput-bound
    v += smem[idx];
    idx = (idx + 1 + (int)(v * 0.f)) & 255;
  }
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int occ = 0;
  for (int kb : {1, 16, 30, 48, 64, 65, 80}) {
    (void)hipFuncSetAttribute((const void*)body, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, body, 256, kb * 1024);
    printf("LDS %3d KB: occupancy API %d blocks/CU |", kb, occ);
    for (int blocks : {256, 512, 768, 1024, 2048}) {
      body<<<blocks, 256, kb * 1024>>>(out, 100);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      body<<<blocks, 256, kb * 1024>>>(out, 100);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("  %4d blk: %6.1f us", blocks, ms * 1e3);
    }
    printf("\n");
  }
  return 0;
}
