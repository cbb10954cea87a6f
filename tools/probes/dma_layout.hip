// Probe: L2 -> LDS staging rate of the fp16 GEMM's operand tiles by LDS-DMA, for the row-major image layout (a tile row = 64 bytes
// out of a K*2-byte row: half a 128-byte line per request) against a tile-blocked layout (the tile's 8 KiB contiguous).
//   hipcc --offload-arch=gfx950 -O3 -o dma_layout dma_layout.hip && ./dma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int GT = 512, IMG = 128 * 64;
template <bool BLOCKED, int NBUF>
__global__ __launch_bounds__(GT, 1) void k(const unsigned short* A, const unsigned short* B, int M, int N, int K, int tiles_n, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
  const int srow = 16 * wv + (lane >> 2), sslot = lane & 3;
  const size_t a_part = (size_t)M * K, b_part = (size_t)N * K;
  const int nk = K / 32;
  auto stage = [&](int kt) {
    unsigned char* dst = smem + (kt % NBUF) * 4 * IMG + wv * 1024;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const unsigned short *as, *bs;
      if (BLOCKED) {
        as = A + p * a_part + ((size_t)(tm * nk + kt) * 128 + srow) * 32 + 8 * sslot;
        bs = B + p * b_part + ((size_t)(tn * nk + kt) * 128 + srow) * 32 + 8 * sslot;
      } else {
        as = A + p * a_part + (size_t)(tm * 128 + srow) * K + kt * 32 + 8 * sslot;
        bs = B + p * b_part + (size_t)(tn * 128 + srow) * K + kt * 32 + 8 * sslot;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)as, (__attribute__((address_space(3))) void*)(dst + p * IMG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bs, (__attribute__((address_space(3))) void*)(dst + (2 + p) * IMG), 16, 0, 0);
    }
  };
  for (int pre = 0; pre < NBUF - 1; ++pre) stage(pre);
  float acc = 0.f;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NBUF - 1 < nk) stage(kt + NBUF - 1);
    if (kt + NBUF - 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NBUF - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc += reinterpret_cast<float*>(smem + (kt % NBUF) * 4 * IMG)[tid];
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 123.456f) sink[0] = acc;
}
template <bool BL, int NBUF>
float run(const unsigned short* A, const unsigned short* B, int M, int N, int K, float* sink) {
  const int tiles = (M / 128) * (N / 128);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<BL, NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<BL, NBUF>), dim3(tiles), dim3(GT), NBUF * 4 * IMG, 0, A, B, M, N, K, N / 128, sink);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<BL, NBUF>), dim3(tiles), dim3(GT), NBUF * 4 * IMG, 0, A, B, M, N, K, N / 128, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}
int main() {
  const int M = 4352;
  unsigned short *A, *B; float* sink;
  hipMalloc(&A, (size_t)2 * M * 3072 * 2); hipMalloc(&B, (size_t)2 * 3072 * 3072 * 2); hipMalloc(&sink, 16);
  hipMemset(A, 0, (size_t)2 * M * 3072 * 2); hipMemset(B, 0, (size_t)2 * 3072 * 3072 * 2);
  const int shapes[5][2] = {{2304, 768}, {768, 768}, {3072, 768}, {768, 3072}, {768, 2304}};
  for (auto& s : shapes) {
    const int N = s[0], K = s[1];
    const double bytes = (double)(M / 128) * (N / 128) * (K / 32) * 32768.0;
    const float r2 = run<false, 2>(A, B, M, N, K, sink), b2 = run<true, 2>(A, B, M, N, K, sink);
    const float r3 = run<false, 3>(A, B, M, N, K, sink), b3 = run<true, 3>(A, B, M, N, K, sink);
    printf("N=%d K=%d: staging only, row-major nbuf2 %.1f us (%.2f TB/s) blocked %.1f us (%.2f TB/s) | nbuf3 row-major %.1f blocked %.1f us\n", N, K, r2,
           bytes / r2 / 1e6, b2, bytes / b2 / 1e6, r3, b3);
  }
  return 0;
}
