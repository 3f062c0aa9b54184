// Shared device/host helpers for libsea_hip.so (gfx950 only: wave64, DPP, 160 KB LDS).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sea_hip.h"

namespace sea {

constexpr int WAVE = 64;

void set_error(const char* fmt, ...);

#define SEA_REQUIRE(cond, code, ...)        \
  do {                                      \
    if (!(cond)) {                          \
      ::sea::set_error(__VA_ARGS__);        \
      return (code);                        \
    }                                       \
  } while (0)

#define SEA_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t _e = hipGetLastError();                                      \
    if (_e != hipSuccess) {                                                 \
      ::sea::set_error("%s: launch failed: %s", name, hipGetErrorString(_e)); \
      return SEA_ELAUNCH;                                                   \
    }                                                                       \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the DEVICE that is current when it is set, and the call is a slow
// driver round trip: one flag per (call site, device), not one per process.  `static DevOnce once; if (once.first()) ...`
struct DevOnce {
  unsigned long long seen = 0;                     // bit d: configured on device d (a racing first call sets it twice: harmless)
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return true;
    const unsigned long long m = 1ull << (d & 63);
    const unsigned long long old = __atomic_fetch_or(&seen, m, __ATOMIC_RELAXED);
    return (old & m) == 0;
  }
};
// a failed attribute call fails the launch that needed it (rc < 0 + sea_last_error), it is not discarded
#define SEA_MAX_LDS(fn, bytes)                                                                                       \
  do {                                                                                                               \
    const hipError_t _ea = hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
    if (_ea != hipSuccess) {                                                                                         \
      ::sea::set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d): %s", (int)(bytes), hipGetErrorString(_ea)); \
      return SEA_ELAUNCH;                                                                                            \
    }                                                                                                                \
  } while (0)

// ---- element access ------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16-byte lane load
  __device__ static inline float to_f(float x) { return x; }
};
template <> struct Elem<__half> {
  static constexpr int VEC = 8;
  __device__ static inline float to_f(__half x) { return __half2float(x); }
};
template <> struct Elem<__hip_bfloat16> {
  static constexpr int VEC = 8;
  __device__ static inline float to_f(__hip_bfloat16 x) { return __bfloat162float(x); }
};

__device__ inline float bf16_bits_to_f(uint32_t hi16) { return __uint_as_float(hi16 << 16); }

// Unpack one 16-byte lane fragment into VEC floats.
template <typename T> __device__ inline void unpack16(const uint4& r, float* f);
template <> __device__ inline void unpack16<float>(const uint4& r, float* f) {
  f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y);
  f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
template <> __device__ inline void unpack16<__hip_bfloat16>(const uint4& r, float* f) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
template <> __device__ inline void unpack16<__half>(const uint4& r, float* f) {
  const __half2* h = reinterpret_cast<const __half2*>(&r);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float2 t = __half22float2(h[i]);
    f[2 * i] = t.x; f[2 * i + 1] = t.y;
  }
}

template <typename T> __device__ inline T from_f(float x);
template <> __device__ inline float from_f<float>(float x) { return x; }
template <> __device__ inline __half from_f<__half>(float x) { return __float2half(x); }
template <> __device__ inline __hip_bfloat16 from_f<__hip_bfloat16>(float x) { return __float2bfloat16(x); }

// erf to ~2e-7 absolute (Abramowitz & Stegun 7.1.26 on hardware rcp / exp2): 12 VALU + 2 transcendental slots, branch
// free, against ~35 slots of the two-branch libm erff.  Used for GELU(x) = 0.5 x (1 + erf(x / sqrt 2)), where the
// error is absolute on a term that is added to 1 -- far below the 16-bit rounding the result goes through.
__device__ inline float erf_as(float z) {
  const float a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(a * a * -1.4426950408889634f);
  return copysignf(fmaf(-(p * t), e, 1.0f), z);
}
// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) = h + |h| E,  h = x / 2,  E = erf(|x| / sqrt 2) as in erf_as with the 1 / sqrt 2 folded
// into its constants: no copysign, no separate 1 + erf, the argument scaling gone -- 16 issue slots instead of 19, in the
// one-launch MLP that is bound by vector issue (22 k vector instructions per wave, a third of them this function).
__device__ inline float gelu_erf(float x) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));
  const float E = fmaf(-(p * t), e, 1.0f);
  const float h = 0.5f * x;
  return fmaf(fabsf(h), E, h);
}

// Two floats -> one dword of packed 16-bit values (low half = a), same rounding as from_f<T>.
template <typename T> __device__ inline uint32_t pack2(float a, float b) {
  const T x = from_f<T>(a), y = from_f<T>(b);
  return (uint32_t)__builtin_bit_cast(unsigned short, x) | ((uint32_t)__builtin_bit_cast(unsigned short, y) << 16);
}

// ---- wave-level primitives -----------------------------------------------------------------
__device__ inline int lane_id() { return threadIdx.x & 63; }

// All-lanes reductions over the 64 lanes without LDS traffic: four DPP steps inside each row of 16 lanes (quad_perm
// xor 1 / xor 2, row_half_mirror, row_mirror), then the gfx950 row-swap instructions v_permlane16_swap /
// v_permlane32_swap for the cross-row steps.  ~10 VALU instructions with a few cycles of latency each, against six
// dependent ds_bpermute round trips (~100+ cycles each) for the __shfl_xor butterfly.  Every lane ends with the
// same bits (each level combines the same two partial results, and + / max / min commute).
template <int CTRL> __device__ inline uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <typename T, typename Op> __device__ inline T wave_reduce32(T v, Op op) {
  static_assert(sizeof(T) == 4, "32-bit types");
  auto bits = [](T x) { return __builtin_bit_cast(uint32_t, x); };
  auto val = [](uint32_t x) { return __builtin_bit_cast(T, x); };
  v = op(v, val(dpp_u32<0xB1>(bits(v))));    // quad_perm [1,0,3,2]
  v = op(v, val(dpp_u32<0x4E>(bits(v))));    // quad_perm [2,3,0,1]
  v = op(v, val(dpp_u32<0x141>(bits(v))));   // row_half_mirror
  v = op(v, val(dpp_u32<0x140>(bits(v))));   // row_mirror
  {
    const auto r = __builtin_amdgcn_permlane16_swap(bits(v), bits(v), false, false);
    v = op(val(r[0]), val(r[1]));
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(bits(v), bits(v), false, false);
    v = op(val(r[0]), val(r[1]));
  }
  return v;
}
// EIGHT values per lane reduced over the 64 lanes at once: returns x with x[lane] = op over all lanes of v[lane >> 3]
// (read result k back with v_readlane from lane 8k).  A transposing butterfly -- each cross-lane step also halves the
// number of live values: v_permlane32_swap pairs (k, k+4), v_permlane16_swap pairs again, one select + row_ror:8 step,
// then three DPP steps inside the groups of 8 lanes.  18 vector instructions instead of 8 x 10 for eight wave_reduce32.
template <typename Op> __device__ inline float wave_reduce8(const float (&v)[8], Op op) {
  float r[4], q[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {          // lanes < 32: value k over (l, l+32);  lanes >= 32: value k+4
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 4]), false, false);
    r[k] = op(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {          // rows of 16 lanes: q[0] holds values 0, 2, 4, 6;  q[1] holds 1, 3, 5, 7
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[k]), __float_as_uint(r[k + 2]), false, false);
    q[k] = op(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  const bool b3 = (threadIdx.x & 8) != 0;
  const float keep = b3 ? q[1] : q[0], give = b3 ? q[0] : q[1];
  float x = op(keep, __uint_as_float(dpp_u32<0x128>(__float_as_uint(give))));   // row_ror:8 -> value index = lane >> 3
  x = op(x, __uint_as_float(dpp_u32<0xB1>(__float_as_uint(x))));                // quad_perm [1,0,3,2]
  x = op(x, __uint_as_float(dpp_u32<0x4E>(__float_as_uint(x))));                // quad_perm [2,3,0,1]
  x = op(x, __uint_as_float(dpp_u32<0x141>(__float_as_uint(x))));               // row_half_mirror
  return x;
}
// result k of wave_reduce8, wave-uniform (a scalar register)
__device__ inline float reduce8_get(float x, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 8 * k)); }

// v[l] + v[l ^ 16] and v[l] + v[l ^ 32] for every lane, through the gfx950 row-swap instructions (no LDS crossbar)
__device__ inline float xor16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <typename T> __device__ inline T wave_sum(T v) {
  if constexpr (sizeof(T) == 4) {
    return wave_reduce32(v, [](T a, T b) { return a + b; });
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  }
}
template <typename T> __device__ inline T wave_max(T v) {
  if constexpr (sizeof(T) == 4) {
    if constexpr (std::is_same<T, float>::value) return wave_reduce32(v, [](float a, float b) { return fmaxf(a, b); });   // v_max_f32_dpp
    else return wave_reduce32(v, [](T a, T b) { return a > b ? a : b; });
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
  }
}
template <typename T> __device__ inline T wave_min(T v) {
  if constexpr (sizeof(T) == 4) {
    if constexpr (std::is_same<T, float>::value) return wave_reduce32(v, [](float a, float b) { return fminf(a, b); });
    else return wave_reduce32(v, [](T a, T b) { return a < b ? a : b; });
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
  }
}
// inclusive scan over the 64 lanes.  32-bit types: the GFX9 DPP scan -- row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast:15 (lane 15 of rows 0,2 into rows 1,3) and row_bcast:31 (lane 31 into rows 2,3): six VALU instructions
// instead of six dependent ds_bpermute round trips.
template <int CTRL, int ROW_MASK> __device__ inline uint32_t dpp_shift_u32(uint32_t v) {   // lanes without a source get 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <typename T> __device__ inline T wave_incl_scan(T v) {
  if constexpr (sizeof(T) == 4 && std::is_integral<T>::value) {
    uint32_t x = (uint32_t)v;
    x += dpp_shift_u32<0x111, 0xF>(x);     // row_shr:1
    x += dpp_shift_u32<0x112, 0xF>(x);     // row_shr:2
    x += dpp_shift_u32<0x114, 0xF>(x);     // row_shr:4
    x += dpp_shift_u32<0x118, 0xF>(x);     // row_shr:8
    x += dpp_shift_u32<0x142, 0xA>(x);     // row_bcast:15 -> rows 1 and 3
    x += dpp_shift_u32<0x143, 0xC>(x);     // row_bcast:31 -> rows 2 and 3
    return (T)x;
  } else {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { T t = __shfl_up(v, o); if (l >= o) v += t; }
    return v;
  }
}

// index load at the API edge: int32 internally, int64 for torch CSR tensors
template <int IB> struct Idx;
template <> struct Idx<4> { using type = int32_t; };
template <> struct Idx<8> { using type = int64_t; };

// ---- interpolation boundaries ---------------------------------------------------------------
// Reference: scales = target_width / T_m (int64/int -> fp32 true division,
// causal_resize_m_to_t.py:642); v = round_half_away(b * scale) in fp32 (:654-655, libdevice roundf).
__device__ inline float interp_scale(int w, int T_m) { return __fdiv_rn((float)w, (float)T_m); }
__device__ inline float interp_bound(int b, float scale) { return roundf(__fmul_rn((float)b, scale)); }
__device__ inline int row_width(int t, int T_dst, int T_src, int is_causal) {
  // target_width = arange(1, T_src+1)[-T_dst:] (causal) or T_src (causal_resize_m_to_t.py:951-955)
  return is_causal ? (T_src - T_dst + t + 1) : T_src;
}

}  // namespace sea
