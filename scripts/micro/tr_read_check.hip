// Semantics check of ds_read_b64_tr_b16 (gfx950): per 16-lane group, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4x16 block of 16-bit elements; lane i receives column i (rows 0..3 in elements 0..3).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(const short* x, s4* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = x[i];
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + (8 * g + q) * 64 + 4 * p));
  out[lane] = v;
}
int main() {
  short h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (short)i;
  short* d; s4* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * sizeof(s4));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  s4 r[64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) { int g = l >> 4, i = l & 15; if (r[l][q] != (8 * g + q) * 64 + i) ++bad; }
  printf("tr_read_check bad=%d  lane5: %d %d %d %d (expect 5 69 133 197)\n", bad, r[5][0], r[5][1], r[5][2], r[5][3]);
  return bad != 0;
}
