// Unfused per-(row, head) softmax and element-wise multiply on the flat CSR, plus the library's
// error/version plumbing.  gfx950 only.
//
// Replaces (reference, src/models/perlin_attention/ops/kernels/):
//   flat_csr_softmax.py:55-125   loads the whole row BLOCK_Z wide and loops H masked passes
//   flat_csr_elmul.py:42-108     gathers from a stride-0 expanded (N,H,T,T) view
#include "sea_common.hpp"
#include <stdarg.h>

namespace sea {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__device__ inline uint32_t fkey(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ inline float fkey_inv(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  return __uint_as_float(u);
}

// One 256-thread workgroup per (n, t) row.  Per-head running max / sum live in LDS; lanes that hold
// entries of the same head are combined inside the wave before touching LDS (entries of a row are
// grouped by head, so a wave sees 1-3 distinct heads).
template <typename I>
__global__ __launch_bounds__(256) void csr_softmax_kernel(const float* in_all, float* out_all, int H, int T_dst, int T_src,
                                                         const I* crow_all, const I* col_all, int64_t col_stride_n) {
  __shared__ uint32_t s_max[1024];
  __shared__ float s_sum[1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int row = blockIdx.x;
  const int n = row / T_dst, t = row - n * T_dst;
  const I* crow = crow_all + (int64_t)n * (T_dst + 1);
  const I* col = col_all + n * col_stride_n;
  const float* in = in_all + n * col_stride_n;
  float* out = out_all + n * col_stride_n;
  const int64_t beg = crow[t], end = crow[t + 1];
  if (beg == end) return;
  for (int i = tid; i < H; i += 256) { s_max[i] = 0u; s_sum[i] = 0.f; }
  __syncthreads();
  // pass 1: per-head max
  for (int64_t e0 = beg; e0 < end; e0 += 256) {
    const int64_t e = e0 + tid;
    int h = -1;
    float x = -INFINITY;
    if (e < end) { h = (int)(col[e] / T_src); x = in[e]; }
    unsigned long long active = __ballot(h >= 0);
    while (active) {
      const int src = __ffsll((long long)active) - 1;
      const int h0 = __shfl(h, src);
      const bool mine = h == h0;
      const float mx = wave_max(mine ? x : -INFINITY);
      if (lane == src) atomicMax(&s_max[h0], fkey(mx));
      active &= ~__ballot(mine);
    }
  }
  __syncthreads();
  // pass 2: per-head sum of exp
  for (int64_t e0 = beg; e0 < end; e0 += 256) {
    const int64_t e = e0 + tid;
    int h = -1;
    float ex = 0.f;
    if (e < end) { h = (int)(col[e] / T_src); ex = __expf(in[e] - fkey_inv(s_max[h])); }
    unsigned long long active = __ballot(h >= 0);
    while (active) {
      const int src = __ffsll((long long)active) - 1;
      const int h0 = __shfl(h, src);
      const bool mine = h == h0;
      const float sm = wave_sum(mine ? ex : 0.f);
      if (lane == src) atomicAdd(&s_sum[h0], sm);
      active &= ~__ballot(mine);
    }
  }
  __syncthreads();
  // pass 3: normalise
  for (int64_t e = beg + tid; e < end; e += 256) {
    const int h = (int)(col[e] / T_src);
    out[e] = __expf(in[e] - fkey_inv(s_max[h])) / s_sum[h];
  }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void csr_elmul_kernel(const float* in_all, float* out_all, const T* other, int64_t on,
                                                       int64_t oh, int64_t ot, int64_t os, int T_dst, int T_src,
                                                       const I* crow_all, const I* col_all, int64_t col_stride_n) {
  const int row = blockIdx.x;
  const int n = row / T_dst, t = row - n * T_dst;
  const I* crow = crow_all + (int64_t)n * (T_dst + 1);
  const I* col = col_all + n * col_stride_n;
  const float* in = in_all + n * col_stride_n;
  float* out = out_all + n * col_stride_n;
  const int64_t beg = crow[t], end = crow[t + 1];
  for (int64_t e = beg + threadIdx.x; e < end; e += 256) {
    const int64_t c = col[e];
    const int64_t h = c / T_src, key = c - h * T_src;
    out[e] = in[e] * Elem<T>::to_f(other[n * on + h * oh + t * ot + key * os]);
  }
}

}  // namespace sea

using namespace sea;

extern "C" int sea_version(void) { return SEA_ABI_VERSION; }
extern "C" const char* sea_last_error(void) { return g_err; }

extern "C" int sea_csr_softmax(const float* in_values, float* out_values, int64_t N, int64_t H, int64_t T_dst,
                               int64_t T_src, const void* crow, const void* col, int idx_bytes, int64_t col_stride_n,
                               sea_stream_t stream) {
  const char* nm = "sea_csr_softmax";
  SEA_REQUIRE(in_values && out_values && crow && col, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SEA_EINVAL, "%s: idx_bytes must be 4 or 8", nm);
  SEA_REQUIRE(H > 0 && H <= 1024, SEA_EUNSUPPORTED, "%s: H must be in 1..1024", nm);
  SEA_REQUIRE(N > 0 && T_dst > 0 && T_src > 0, SEA_EINVAL, "%s: bad shape", nm);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(N * T_dst)), block(256);
  if (idx_bytes == 4)
    hipLaunchKernelGGL((csr_softmax_kernel<int32_t>), grid, block, 0, s, in_values, out_values, (int)H, (int)T_dst, (int)T_src,
                       (const int32_t*)crow, (const int32_t*)col, col_stride_n);
  else
    hipLaunchKernelGGL((csr_softmax_kernel<int64_t>), grid, block, 0, s, in_values, out_values, (int)H, (int)T_dst, (int)T_src,
                       (const int64_t*)crow, (const int64_t*)col, col_stride_n);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

template <typename I>
static void launch_elmul(const float* in, float* out, const void* other, int dtype, const int64_t* os, int64_t N,
                         int64_t T_dst, int64_t T_src, const void* crow, const void* col, int64_t cs, hipStream_t s) {
  dim3 grid((unsigned)(N * T_dst)), block(256);
  if (dtype == SEA_F32)
    hipLaunchKernelGGL((csr_elmul_kernel<float, I>), grid, block, 0, s, in, out, (const float*)other, os[0], os[1], os[2], os[3],
                       (int)T_dst, (int)T_src, (const I*)crow, (const I*)col, cs);
  else if (dtype == SEA_F16)
    hipLaunchKernelGGL((csr_elmul_kernel<__half, I>), grid, block, 0, s, in, out, (const __half*)other, os[0], os[1], os[2],
                       os[3], (int)T_dst, (int)T_src, (const I*)crow, (const I*)col, cs);
  else
    hipLaunchKernelGGL((csr_elmul_kernel<__hip_bfloat16, I>), grid, block, 0, s, in, out, (const __hip_bfloat16*)other, os[0],
                       os[1], os[2], os[3], (int)T_dst, (int)T_src, (const I*)crow, (const I*)col, cs);
}

extern "C" int sea_csr_elmul(const float* in_values, float* out_values, const void* other, int dtype,
                             const int64_t* other_strides, int64_t N, int64_t H, int64_t T_dst, int64_t T_src,
                             const void* crow, const void* col, int idx_bytes, int64_t col_stride_n, sea_stream_t stream) {
  const char* nm = "sea_csr_elmul";
  (void)H;
  SEA_REQUIRE(in_values && out_values && other && other_strides && crow && col, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SEA_EINVAL, "%s: idx_bytes must be 4 or 8", nm);
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && T_dst > 0 && T_src > 0, SEA_EINVAL, "%s: bad shape", nm);
  hipStream_t s = (hipStream_t)stream;
  if (idx_bytes == 4) launch_elmul<int32_t>(in_values, out_values, other, dtype, other_strides, N, T_dst, T_src, crow, col, col_stride_n, s);
  else launch_elmul<int64_t>(in_values, out_values, other, dtype, other_strides, N, T_dst, T_src, crow, col, col_stride_n, s);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
