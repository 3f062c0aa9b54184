// Microbenchmark: LDS atomic-add throughput vs plain LDS stores, 256-thread blocks, 32 ops/thread/iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned binmask) {
  __shared__ unsigned h[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      x = x * 1664525u + 1013904223u;
      unsigned b = (x >> 11) & binmask;
      if (MODE == 0) atomicAdd(&h[b], 1u);
      else if (MODE == 1) h[b] = x;
      else if (MODE == 2) { unsigned r = atomicAdd(&h[b], 1u); x ^= r; }
    }
    __syncthreads();
  }
  unsigned s = 0;
  for (int i = threadIdx.x; i < 2048; i += 256) s += h[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + x;
}
template <int MODE> float run(unsigned* d, int blocks, int iters, unsigned mask) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, 2, mask); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters, mask); hipEventRecord(b);
  hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 256 * 4);
  int blocks = 2048, iters = 64;
  double ops = (double)blocks * 256 * 32 * iters;
  for (unsigned mask : {2047u, 255u, 63u, 7u, 0u}) {
    float t0 = run<0>(d, blocks, iters, mask), t1 = run<1>(d, blocks, iters, mask), t2 = run<2>(d, blocks, iters, mask);
    printf("bins=%4u  atomic_noret %.3f ms (%.1f Gops/s, %.2f lane-ops/clk/CU)  store %.3f ms  atomic_ret %.3f ms\n", mask + 1, t0,
           ops / t0 / 1e6, ops / (t0 * 1e-3) / 256 / 2.4e9, t1, t2);
  }
  return 0;
}
