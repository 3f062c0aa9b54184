// wave_reduce8 (sea_common.hpp) against plain sums / maxima:  hipcc --offload-arch=gfx950 -I sea-attention_amd/csrc scripts/micro/reduce8_check.hip -o scripts/micro/reduce8_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "sea_common.hpp"
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = (float)((l * 7 + i * 13) % 31) - 9.0f + 0.25f * i;
  const float xs = sea::wave_reduce8(v, [](float a, float b) { return a + b; });
  const float xm = sea::wave_reduce8(v, [](float a, float b) { return fmaxf(a, b); });
  out[l] = xs; out[64 + l] = xm;
  for (int i = 0; i < 8; ++i) { out[128 + i * 64 + l] = sea::reduce8_get(xs, i); }
}
int main() {
  float* d; hipMalloc(&d, sizeof(float) * (128 + 512));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128 + 512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 8; ++i) {
    float s = 0.f, m = -1e30f;
    for (int l = 0; l < 64; ++l) { const float v = (float)((l * 7 + i * 13) % 31) - 9.0f + 0.25f * i; s += v; m = fmaxf(m, v); }
    for (int l = 8 * i; l < 8 * i + 8; ++l) if (h[l] != s || h[64 + l] != m) { ++bad; printf("value %d lane %d: sum %g (want %g) max %g (want %g)\n", i, l, h[l], s, h[64 + l], m); }
    for (int l = 0; l < 64; ++l) if (h[128 + i * 64 + l] != s) { ++bad; if (bad < 20) printf("get %d lane %d: %g want %g\n", i, l, h[128 + i * 64 + l], s); }
  }
  printf(bad ? "FAILED %d\n" : "reduce8 ok\n", bad);
  return bad != 0;
}
