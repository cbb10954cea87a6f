// Probe: issue rate of the fp32 / bf16 MFMA shapes on gfx950, one wave per SIMD, independent
// accumulators, plus the shader clock (s_memtime ticks per s_memrealtime 100 MHz tick).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, long long* clk, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f32x16 d0 = {0}, d1 = d0;
  bf16x8 ba = {1, 2, 3, 4, 5, 6, 7, 8}, bb = {8, 7, 6, 5, 4, 3, 2, 1};
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {        // 4 independent 16x16x4 f32 chains
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    } else if (MODE == 1) { // 2 independent 32x32x2 f32 chains
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
    } else if (MODE == 2) { // bf16 16x16x32
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, c3, 0, 0, 0);
    } else {                // 1 dependent 16x16x4 f32 chain
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int blocks, int iters, double flop_per_mfma) {
  float* out; long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(out, clk, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double n = 4.0 * iters;
  double ghz = (double)h[0] / ((double)h[1] * 10.0) ;  // memrealtime = 100 MHz -> 10 ns per tick
  printf("%-28s blocks=%4d  %.3f ms  %.1f ns/MFMA/wave  %.1f shader-cycles/MFMA  clock %.2f GHz  %.1f TFLOP/s\n", name, blocks, ms,
         ms * 1e6 / n, (double)h[0] / n, ghz, blocks * 4 * n * flop_per_mfma / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(clk);
}

int main() {
  for (int blocks : {64, 256, 1024}) {
    run<0>("f32 16x16x4 (4 indep)", blocks, 20000, 2048);
    run<3>("f32 16x16x4 (dependent)", blocks, 20000, 2048);
    run<1>("f32 32x32x2 (2 indep)", blocks, 20000, 4096);
    run<2>("bf16 16x16x32 (4 indep)", blocks, 20000, 16384);
  }
  return 0;
}
