// cycles per MFMA (one wave per SIMD, back-to-back, independent accumulators): 16x16x32 vs 16x16x16 bf16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) float f4;
template <int MODE> __global__ void k(float* out, unsigned long long* cyc, int iters) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf8 a8, b8; s4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.01f + i); b8[i] = (__bf16)(i * 0.5f); }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(i * 3); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[j], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x + blockIdx.x * blockDim.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 10000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
    }
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %.2f cycles per MFMA (one wave per SIMD)\n", mode == 0 ? "16x16x32_bf16" : "16x16x16_bf16", (double)c / (iters * 4.0));
  }
  return 0;
}
