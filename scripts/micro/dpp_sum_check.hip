#include <hip/hip_runtime.h>
__device__ inline float x16(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float x32(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL> __device__ inline float dppf(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
__global__ void k(const float* x, float* out) {
  float v = x[threadIdx.x];
  v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); v += dppf<0x140>(v);
  v = x16(v); v = x32(v);
  out[threadIdx.x] = v;
}
int main() {
  float h[64], r[64]; float tot = 0; for (int i = 0; i < 64; ++i) { h[i] = (float)(i * i % 17) + 0.25f * i; tot += h[i]; }
  float *d, *o; (void)hipMalloc(&d, 256); (void)hipMalloc(&o, 256); (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o); (void)hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) if (r[i] != tot) ++bad;
  printf("dpp wave_sum bad=%d total=%f lane0=%f lane37=%f\n", bad, tot, r[0], r[37]);
  return bad != 0;
}
