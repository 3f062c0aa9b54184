// wave64 inclusive scan by DPP (row_shr 1/2/4/8, row_bcast:15, row_bcast:31) against the serial reference
#include "../../sea-attention_amd/csrc/sea_common.hpp"
#include <cstdio>
__global__ void k(const int* x, int* out) { out[threadIdx.x] = sea::wave_incl_scan(x[threadIdx.x]); }
int main() {
  int h[64], r[64]; for (int i = 0; i < 64; ++i) h[i] = (i * 37) % 11 + (i == 17 ? 1000 : 0);
  int *d, *o; (void)hipMalloc(&d, 256); (void)hipMalloc(&o, 256); (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o); (void)hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
  int bad = 0, run = 0; for (int i = 0; i < 64; ++i) { run += h[i]; if (r[i] != run) ++bad; }
  printf("dpp wave_incl_scan bad=%d last=%d expect=%d\n", bad, r[63], run);
  return bad != 0;
}
