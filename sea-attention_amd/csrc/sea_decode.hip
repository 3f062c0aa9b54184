// Glue of a graph-replayed decoding step (perlin_attention/decode.py; reference loop: src/main/opt_generate.py:131 ->
// the `use_cache` branches of perlin_attention/attention.py + attention_state.py:142-203).
//
// A position of the session is ~10 kernels of 4 - 25 us; the framework's own glue around them was eleven more launches
// of ~4.5 us each (three input copies, an index_copy_ into the caches, a cat + a copy for the CNN window, two counter
// adds, ...: 50 of a step's 130 us; after: profiles/r04b_decode_kernel_stats.csv).  Two kernels replace most of it:
//
//   decode_stage_kernel   copies the new q row into the static input buffer and writes the new k / v rows straight into the
//                         caches at the row the session's device-side counter names.  The only launch of a step whose
//                         arguments change (the caller's q / k / v): it runs eagerly in front of the replay.
//   c8_window_shift_kernel  the CNN window (N, rows, row) moved up by one row in place; a thread owns a 16-byte column of
//                         all rows and walks them top to bottom, so no thread reads what another one writes.  It is the
//                         LAST launch of a step and also advances the two counters (rows the state has seen, keys the next
//                         row sees): every reader of this step has finished (stream order), the next step's have not begun.
#include "sea_common.hpp"

namespace sea {

struct StageParams {
  const void *q, *k, *v;
  int64_t qs[2], ks[2], vs[2];       // element strides [n, h] of the (N, H, 1, D) inputs (feature stride 1)
  void* q_in;                        // (N, H, D) dense
  void* kv_cache;                    // (2, N, H, cap, D) dense
  const int32_t* ctr;                // [seen, tsrc] of THIS step: the new token's cache row is ctr[0]
  int N, H, D, cap;
};

template <typename T>
__global__ __launch_bounds__(256) void decode_stage_kernel(StageParams p) {
  const int pos = p.ctr[0];
  if (pos < 0 || pos >= p.cap) return;                       // (the host mirrors the length and refuses before this can happen)
  const int rows = p.N * p.H;
  const int per = p.D / 8;                                   // 16-byte chunks per row (launcher: D % 8 == 0)
  const T* srcs[3] = {reinterpret_cast<const T*>(p.q), reinterpret_cast<const T*>(p.k), reinterpret_cast<const T*>(p.v)};
  const int64_t* strs[3] = {p.qs, p.ks, p.vs};
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < 3 * rows * per; c += gridDim.x * blockDim.x) {
    const int which = c / (rows * per);
    const int r = (c - which * rows * per) / per, j = c % per;
    const int n = r / p.H, h = r - n * p.H;
    const uint4 val = *reinterpret_cast<const uint4*>(srcs[which] + n * strs[which][0] + h * strs[which][1] + j * 8);
    T* dst;
    if (which == 0) dst = reinterpret_cast<T*>(p.q_in) + (int64_t)r * p.D;
    else dst = reinterpret_cast<T*>(p.kv_cache) + (((int64_t)(which - 1) * rows + r) * p.cap + pos) * p.D;
    *reinterpret_cast<uint4*>(dst + j * 8) = val;
  }
}

__global__ __launch_bounds__(256) void c8_window_shift_kernel(uint4* xs, int rows, int64_t chunks_per_row, int64_t items_chunks,
                                                              int32_t* counters) {
  // xs (N, rows, chunks_per_row) in 16-byte chunks: xs[n, r] = xs[n, r + 1] for r < rows - 1
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid == 0 && counters != nullptr) { counters[0] += 1; counters[1] += 1; }
  if (gid >= items_chunks) return;
  const int64_t n = gid / chunks_per_row, c = gid - n * chunks_per_row;
  if (rows < 2) return;
  uint4* base = xs + n * rows * chunks_per_row + c;
  uint4 nxt = base[chunks_per_row];
  for (int r = 0; r + 1 < rows; ++r) {
    const uint4 cur = nxt;
    if (r + 2 < rows) nxt = base[(int64_t)(r + 2) * chunks_per_row];
    base[(int64_t)r * chunks_per_row] = cur;
  }
}

}  // namespace sea

using namespace sea;

extern "C" int sea_decode_stage(const void* q, const void* k, const void* v, int dtype, int64_t N, int64_t H, int64_t D,
                                const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                void* q_in, void* kv_cache, int64_t capacity, const int32_t* counters, sea_stream_t stream) {
  const char* nm = "sea_decode_stage";
  SEA_REQUIRE(q && k && v && q_strides && k_strides && v_strides && q_in && kv_cache && counters, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16, SEA_EUNSUPPORTED, "%s: 16-bit data only (dtype %d)", nm, dtype);
  SEA_REQUIRE(N > 0 && H > 0 && D > 0 && capacity > 0 && N * H * D < (1ll << 24), SEA_EINVAL, "%s: bad shape", nm);
  SEA_REQUIRE(D % 8 == 0, SEA_EUNSUPPORTED, "%s: D must be a multiple of 8 (16-byte rows)", nm);
  bool al = (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)q_in | (uintptr_t)kv_cache) & 15) == 0;
  for (int i = 0; i < 2; ++i) al = al && q_strides[i] % 8 == 0 && k_strides[i] % 8 == 0 && v_strides[i] % 8 == 0;
  SEA_REQUIRE(al, SEA_EUNSUPPORTED, "%s: rows must be 16-byte aligned", nm);
  StageParams p;
  p.q = q; p.k = k; p.v = v; p.q_in = q_in; p.kv_cache = kv_cache; p.ctr = counters;
  for (int i = 0; i < 2; ++i) { p.qs[i] = q_strides[i]; p.ks[i] = k_strides[i]; p.vs[i] = v_strides[i]; }
  p.N = (int)N; p.H = (int)H; p.D = (int)D; p.cap = (int)capacity;
  hipStream_t s = (hipStream_t)stream;
  const int64_t chunks = 3 * N * H * (D / 8);
  const unsigned blocks = (unsigned)((chunks + 255) / 256 > 1024 ? 1024 : (chunks + 255) / 256);
  if (dtype == SEA_F16) hipLaunchKernelGGL((decode_stage_kernel<__half>), dim3(blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((decode_stage_kernel<__hip_bfloat16>), dim3(blocks), dim3(256), 0, s, p);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_c8_window_shift(void* xs, int64_t N, int64_t rows, int64_t row_bytes, int32_t* counters, sea_stream_t stream) {
  const char* nm = "sea_c8_window_shift";
  SEA_REQUIRE(xs, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(N > 0 && rows > 0 && row_bytes > 0, SEA_EINVAL, "%s: bad shape", nm);
  SEA_REQUIRE(row_bytes % 16 == 0 && ((uintptr_t)xs & 15) == 0, SEA_EUNSUPPORTED, "%s: rows are whole 16-byte chunks", nm);
  const int64_t cpr = row_bytes / 16, total = N * cpr;
  SEA_REQUIRE(total < (1ll << 31), SEA_EUNSUPPORTED, "%s: window too large", nm);
  hipLaunchKernelGGL(c8_window_shift_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<uint4*>(xs), (int)rows, cpr, total, counters);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
