// Device pieces of the predictor tail (MFMA variant) shared by predictor_tail_mfma_kernel (sea_predictor.hip) and the
// fused tail + top-k selection kernel (sea_topk.hip).  See sea_predictor.hip for what the tail computes.
#pragma once
#include "sea_common.hpp"
#include <type_traits>

namespace sea {

struct TailParams {
  const void* y;     // (N, C, T, W4)
  const void* w4;    // (C, Hpad) fp32: the live row of the (H, C, 1, 1) conv weight, transposed, heads padded to 8
  const void* b4;    // (Hpad) fp32
  const void* gamma; // (T_M)
  const void* beta;  // (T_M)
  void* probs;       // (N, H, T, T_M); optional in the fused tail + selection kernels (the map stays in registers)
  void* scores;      // optional (N, H, T, T_M)
  float eps;
  int N, C, H, T, W4, UP, T_M;
  int64_t ys_n, ys_c, ys_t, ys_w;  // element strides of y (NCHW: ys_w == 1; channels-last / C8: ys_c == 1)
  int64_t ys_c8;                   // stride between blocks of 8 channels (8*ys_c for plain 4-D layouts; W*8 for C8)
  const void* w16;   // MFMA variant: (HP16, Cp) 16-bit row-major copy of the conv weight, zero padded
  int Cp;            // channels padded to a multiple of 32
  const uint32_t* tab;  // optional [3][64 E] words in global memory: tail_consts_fill's table, computed once (not per row)
  const float* z;    // optional (N, T, H, W4) fp32: the 1x1 conv's output, already computed (sea_causal_conv_c8_z's epilogue);
                     // y / w16 are then unused and the z tile is a copy into LDS
};

template <typename T, int E> __device__ inline void store_run(T* dst, const float* f, int j0, int T_M) {
  if (E == 4 && sizeof(T) == 2 && j0 + 4 <= T_M) {
    union { uint2 u; T h[4]; } pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk.h[e] = from_f<T>(f[e]);
    *reinterpret_cast<uint2*>(dst + j0) = pk.u;
  } else if (E == 4 && sizeof(T) == 4 && j0 + 4 <= T_M) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + j0) = make_float4(f[0], f[1], f[2], f[3]);
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (j0 + e < T_M) dst[j0 + e] = from_f<T>(f[e]);
  }
}

typedef __attribute__((ext_vector_type(4))) float tf4;
typedef __attribute__((ext_vector_type(8))) __bf16 tbf8;
typedef __attribute__((ext_vector_type(8))) _Float16 th8;
template <typename T> __device__ inline tf4 tail_mfma(const uint4& a, const uint4& b, tf4 c);
template <> __device__ inline tf4 tail_mfma<__hip_bfloat16>(const uint4& a, const uint4& b, tf4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tbf8, a), __builtin_bit_cast(tbf8, b), c, 0, 0, 0);
}
template <> __device__ inline tf4 tail_mfma<__half>(const uint4& a, const uint4& b, tf4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(th8, a), __builtin_bit_cast(th8, b), c, 0, 0, 0);
}


// ---- z = y W^T + b: (W4 pixels x C) @ (C x H) on v_mfma_f32_16x16x32, A fragments straight from global -----------
// z lands in LDS as [head][pixel] with [W4] = bias and [W4+1] = 0 per row (turn padded / unused taps of the area
// resize into plain reads).  256 threads; the caller barriers before reading s_z.
template <typename T>
__device__ __forceinline__ void tail_z_tile(const TailParams& p, float* s_z, int n, int t) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int HP = ((p.H + 15) / 16) * 16;
  const int LDZ = p.W4 + 3;
  const float* __restrict__ bF = reinterpret_cast<const float*>(p.b4);
  if (p.z) {                                         // block-uniform: z was produced by the convolution's epilogue
    const float* zr = p.z + ((int64_t)n * p.T + t) * ((int64_t)p.H * p.W4);
    const int qpr = p.W4 >> 2, nq = p.H * qpr;       // W4 % 4 == 0 (launchers check); 16-byte loads, one round trip
    for (int i = threadIdx.x; i < nq; i += 256) {
      const float4 v = *reinterpret_cast<const float4*>(zr + 4 * i);
      const int h = p.W4 == 64 ? (i >> 4) : i / qpr;
      float* d = s_z + h * LDZ + 4 * (i - h * qpr);
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int h = threadIdx.x; h < HP; h += 256) { s_z[h * LDZ + p.W4] = h < p.H ? bF[h] : 0.f; s_z[h * LDZ + p.W4 + 1] = 0.f; }
    return;
  }
  if constexpr (sizeof(T) == 4) {
    // fp32 data (round 5): the same product on v_mfma_f32_16x16x4_f32, exact fp32.  A = y (row = pixel li, k = channel 4 j + lg:
    // one 4-byte load per MFMA from the channels-last / C8 row), B = the transposed fp32 weights wT (C, Hpad) (k, col = head li).
    const float* __restrict__ wT = reinterpret_cast<const float*>(p.w4);
    const float* yf = reinterpret_cast<const float*>(p.y) + n * p.ys_n + t * p.ys_t;
    const int Hpad = ((p.H + 7) / 8) * 8;
    const int MTf = (p.W4 + 15) / 16, NTf = HP / 16, KS = (p.C + 3) / 4;
    for (int mt = wv; mt < MTf; mt += 4) {
      const int wpix = mt * 16 + li;
      for (int nt = 0; nt < NTf; ++nt) {
        const int h = nt * 16 + li;
        tf4 acc = tf4{0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < KS; j0 += 8) {                      // eight k-steps' operands requested together
          float a[8], b[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int c = 4 * (j0 + u) + lg;
            const bool ck = (j0 + u) < KS && c < p.C;
            a[u] = (ck && wpix < p.W4) ? yf[(int64_t)wpix * p.ys_w + (c >> 3) * p.ys_c8 + (c & 7) * p.ys_c] : 0.f;
            b[u] = (ck && h < p.H) ? wT[c * Hpad + h] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (j0 + u < KS) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);   // (block-uniform guard)
        }
        const float bias = h < p.H ? bF[h] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = mt * 16 + lg * 4 + r;
          if (px < p.W4) s_z[h * LDZ + px] = acc[r] + bias;
        }
      }
    }
    for (int h = threadIdx.x; h < HP; h += 256) { s_z[h * LDZ + p.W4] = h < p.H ? bF[h] : 0.f; s_z[h * LDZ + p.W4 + 1] = 0.f; }
    return;
  } else {
  const T* __restrict__ w16 = reinterpret_cast<const T*>(p.w16);
  const T* yb = reinterpret_cast<const T*>(p.y) + n * p.ys_n + t * p.ys_t;
  const int MT = (p.W4 + 15) / 16, NT = HP / 16, KC = p.Cp / 32;
  // All fragments of a tile are requested before the first MFMA (ONE exposed memory round trip per tile, not one per
  // MFMA: this phase is pure latency -- 8 KB of y and 4 KB of L2-resident weights per row).  Guards instead of
  // compile-time trip counts for the common shapes so that the loops unroll; other shapes loop below.
  auto tile_fixed = [&](auto ntc, auto kcc) {
    constexpr int CNT = decltype(ntc)::value, CKC = decltype(kcc)::value;
    for (int mt = wv; mt < MT; mt += 4) {
      const int wpix = mt * 16 + li;
      uint4 a[CKC], b[CNT][CKC];
#pragma unroll
      for (int kc = 0; kc < CKC; ++kc) {
        const int ci = kc * 32 + 8 * lg;
        a[kc] = make_uint4(0, 0, 0, 0);
        if (ci < p.C && wpix < p.W4) a[kc] = *reinterpret_cast<const uint4*>(yb + (int64_t)wpix * p.ys_w + (ci >> 3) * p.ys_c8);
      }
#pragma unroll
      for (int nt = 0; nt < CNT; ++nt)
#pragma unroll
        for (int kc = 0; kc < CKC; ++kc)
          b[nt][kc] = *reinterpret_cast<const uint4*>(w16 + (nt * 16 + li) * p.Cp + kc * 32 + 8 * lg);
#pragma unroll
      for (int nt = 0; nt < CNT; ++nt) {
        tf4 acc = tf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < CKC; ++kc) acc = tail_mfma<T>(a[kc], b[nt][kc], acc);
        const int h = nt * 16 + li;                // C layout: col = li -> head, row = lg*4 + r -> pixel
        const float bias = h < p.H ? bF[h] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = mt * 16 + lg * 4 + r;
          if (px < p.W4) s_z[h * LDZ + px] = acc[r] + bias;
        }
      }
    }
  };
  if (NT == 2 && KC == 2) {                        // H in 17..32, C <= 64 (OPT-1.3B ... 6.7B)
    tile_fixed(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
  } else if (NT == 1 && KC == 1) {                 // H <= 16, C <= 32 (OPT-125m / 350m)
    tile_fixed(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  } else {
  for (int mt = wv; mt < MT; mt += 4) {
    const int wpix = mt * 16 + li;
    for (int nt = 0; nt < NT; ++nt) {
      tf4 acc = tf4{0.f, 0.f, 0.f, 0.f};
      for (int kc = 0; kc < KC; ++kc) {
        const int ci = kc * 32 + 8 * lg;
        uint4 a = make_uint4(0, 0, 0, 0);
        if (ci < p.C && wpix < p.W4) a = *reinterpret_cast<const uint4*>(yb + (int64_t)wpix * p.ys_w + (ci >> 3) * p.ys_c8);
        const uint4 b = *reinterpret_cast<const uint4*>(w16 + (nt * 16 + li) * p.Cp + ci);
        acc = tail_mfma<T>(a, b, acc);
      }
      const int h = nt * 16 + li;                // C layout: col = li -> head, row = lg*4 + r -> pixel
      const float bias = h < p.H ? bF[h] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int px = mt * 16 + lg * 4 + r;
        if (px < p.W4) s_z[h * LDZ + px] = acc[r] + bias;
      }
    }
  }
  }
  for (int h = threadIdx.x; h < HP; h += 256) { s_z[h * LDZ + p.W4] = h < p.H ? bF[h] : 0.f; s_z[h * LDZ + p.W4 + 1] = 0.f; }
  }
}

// ---- per-pixel constants of the area-resize / LayerNorm stage ----------------------------------------------------
// They depend on the output pixel j only (not on the row, the head or the wave), and every wave needs all T_M of them:
// the workgroup computes each ONCE (thread <-> pixel) into an LDS table and a lane then reads its E pixels back
// (the first version recomputed them per lane in every wave: ~300 of the ~2700 vector instructions of a wave of the
// fused tail + selection kernel, which is vector-issue bound -- profiles/r02d_pmc_per_kernel.txt).
// Table: [3][TMP] words, TMP = 64 * E:  [0] taps packed 3 x 10 bits (index into the z row: pixel, W4 = bias i.e. the
// zero-padded border, W4+1 = unused tap) + the tap count (1..3, 0 beyond T_M) in bits 30..31;  [1] gamma;  [2] beta.
constexpr int TAIL_TAB_ROWS = 3;
template <typename T>
__device__ __forceinline__ void tail_consts_fill(const TailParams& p, uint32_t* s_tab, int TMP) {
  const int Wp = p.W4 * p.UP + 2;
  const T* gam = reinterpret_cast<const T*>(p.gamma);
  const T* bet = reinterpret_cast<const T*>(p.beta);
  const bool up_pow2 = (p.UP & (p.UP - 1)) == 0;            // block-uniform; x4 in every configuration the reference builds
  const int up_sh = __ffs(p.UP) - 1;
  const bool tm_pow2 = (p.T_M & (p.T_M - 1)) == 0 && (int64_t)(p.T_M + 1) * Wp < (1 << 24);   // block-uniform
  const int tm_sh = __ffs(p.T_M) - 1;
  for (int j = threadIdx.x; j < TMP; j += blockDim.x) {
    uint32_t pk = (uint32_t)(p.W4 + 1) * 0x100401u;         // three unused taps, count 0
    float g = 0.f, be = 0.f;
    if (j < p.T_M) {
      g = Elem<T>::to_f(gam[j]); be = Elem<T>::to_f(bet[j]);
      // adaptive-average-pool window of output pixel j over the Wp padded pixels (ATen's float formula; dividing an
      // integer below 2^24 by a power of two is exact in fp32, so for those T_M the floor / ceil are plain shifts)
      int xs, xe;
      if (tm_pow2) {
        xs = (j * Wp) >> tm_sh;
        xe = ((j + 1) * Wp + p.T_M - 1) >> tm_sh;
      } else {
        xs = (int)floorf((float)(j * Wp) / (float)p.T_M);
        xe = (int)ceilf((float)((j + 1) * Wp) / (float)p.T_M);
      }
      pk = (uint32_t)(xe - xs) << 30;                       // launcher: Wp <= 3 * T_M, so 1..3 taps
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int x = xs + k;
        int sidx = p.W4 + 1;
        // source pixel of tap x: (x - 1) / UP (a shift for the power-of-two factors)
        if (x < xe) sidx = (x == 0 || x == Wp - 1) ? p.W4 : (up_pow2 ? ((x - 1) >> up_sh) : (x - 1) / p.UP);
        pk |= (uint32_t)sidx << (10 * k);
      }
    }
    s_tab[j] = pk;
    s_tab[TMP + j] = __float_as_uint(g);
    s_tab[2 * TMP + j] = __float_as_uint(be);
  }
}

// ---- a lane's constants + one head's row --------------------------------------------------------------------------
template <typename T, int E>
struct TailRow {
  float g[E], be[E], rcnt[E];
  int src[E][3];   // index into the z row: pixel, W4 (bias: zero-padded border) or W4+1 (unused tap)

  // the same from a table in GLOBAL memory (computed once per weight set): three 16-byte loads per lane, no LDS, no barrier
  __device__ __forceinline__ void load_global(const uint32_t* tab, int lane) {
    constexpr int TMP = 64 * E;
    uint32_t pk[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      pk[e] = tab[lane * E + e];
      g[e] = __uint_as_float(tab[TMP + lane * E + e]);
      be[e] = __uint_as_float(tab[2 * TMP + lane * E + e]);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
      for (int k = 0; k < 3; ++k) src[e][k] = (int)__builtin_amdgcn_ubfe(pk[e], 10 * k, 10);
      const uint32_t cnt = pk[e] >> 30;
      rcnt[e] = cnt == 1 ? 1.0f : cnt == 2 ? 0.5f : cnt == 3 ? (1.0f / 3.0f) : 0.f;
    }
  }

  // after the barrier that publishes tail_consts_fill's table
  __device__ __forceinline__ void load(const uint32_t* s_tab, int lane) {
    constexpr int TMP = 64 * E;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t pk = s_tab[lane * E + e];
      g[e] = __uint_as_float(s_tab[TMP + lane * E + e]);
      be[e] = __uint_as_float(s_tab[2 * TMP + lane * E + e]);
#pragma unroll
      for (int k = 0; k < 3; ++k) src[e][k] = (int)__builtin_amdgcn_ubfe(pk, 10 * k, 10);
      const uint32_t cnt = pk >> 30;                         // 1 / cnt, as the division gives it
      rcnt[e] = cnt == 1 ? 1.0f : cnt == 2 ? 0.5f : cnt == 3 ? (1.0f / 3.0f) : 0.f;
    }
  }

  // area resize -> LayerNorm -> (scores) -> softmax -> probs of up to NBC <= 8 heads of one wave AT ONCE: the four wave
  // reductions of a head (mean, variance, max, sum of exponentials) run as four wave_reduce8 over the batch -- 4 x 26
  // vector instructions (butterfly + readlanes) instead of 8 x 4 x 10, in a kernel that is vector-issue bound.
  // Head b of the batch (b < nb, wave-uniform) reads z row zrow(b) and writes at element offset obase(b); a[b][] returns
  // its probabilities.  FULLROW: T_M == 64 * E, every lane slot is a real pixel -- no per-element predicates.
  template <bool FULLROW, int NBC, typename ZF, typename OF>
  __device__ __forceinline__ void heads_impl(const TailParams& p, int lane, int nb, ZF zrow, OF obase, float (&a)[NBC][E]) const {
    static_assert(NBC >= 1 && NBC <= 8, "one wave_reduce8 per statistic");
    const float invT = 1.0f / (float)p.T_M;
    float red[8], st[NBC];
#pragma unroll
    for (int b = 0; b < 8; ++b) red[b] = 0.f;
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
      float s1 = 0.f;
      if (b < nb) {
        const float* zr = zrow(b);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          a[b][e] = (zr[src[e][0]] + zr[src[e][1]] + zr[src[e][2]]) * rcnt[e];
          s1 += a[b][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) a[b][e] = 0.f;
      }
      red[b] = s1;
    }
    float x = wave_reduce8(red, [](float u, float v) { return u + v; }) * invT;
#pragma unroll
    for (int b = 0; b < NBC; ++b) st[b] = reduce8_get(x, b);                       // mean
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
      float s2 = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (FULLROW || lane * E + e < p.T_M) { const float d = a[b][e] - st[b]; s2 += d * d; }
      red[b] = s2;
    }
    x = rsqrtf(wave_reduce8(red, [](float u, float v) { return u + v; }) * invT + p.eps);
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
      const float rstd = reduce8_get(x, b);
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        a[b][e] = (a[b][e] - st[b]) * rstd * g[e] + be[e];
        if (FULLROW || lane * E + e < p.T_M) mx = fmaxf(mx, a[b][e]);
      }
      red[b] = mx;
    }
#pragma unroll
    for (int b = NBC; b < 8; ++b) red[b] = -INFINITY;
    x = wave_reduce8(red, [](float u, float v) { return fmaxf(u, v); });
#pragma unroll
    for (int b = NBC; b < 8; ++b) red[b] = 0.f;
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
      const float mx = reduce8_get(x, b);
      if (p.scores && b < nb) store_run<T, E>(reinterpret_cast<T*>(p.scores) + obase(b), a[b], lane * E, FULLROW ? 64 * E : p.T_M);
      float se = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        a[b][e] = (FULLROW || lane * E + e < p.T_M) ? __expf(a[b][e] - mx) : 0.f;
        se += a[b][e];
      }
      red[b] = se;
    }
    x = 1.0f / wave_reduce8(red, [](float u, float v) { return u + v; });          // (slots >= NBC: 1/0 = inf, never read)
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
      const float inv = reduce8_get(x, b);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        a[b][e] *= inv;
        // the fp32 product exists in a register before anything rounds it to 16 bits: without this the compiler folds the
        // multiply into a mixed-precision convert (v_fma_mixlo_f16: ONE rounding) in some kernels and not in others, and
        // the same row then differs by an fp16 ulp between the stand-alone tail and the fused tail + selection kernels
        // (6e-5 of the elements) -- the selection's keys must be the very numbers a later sea_predictor_tail call writes
        asm volatile("" : "+v"(a[b][e]));
      }
      if (p.probs && b < nb) store_run<T, E>(reinterpret_cast<T*>(p.probs) + obase(b), a[b], lane * E, FULLROW ? 64 * E : p.T_M);
    }
  }
  template <int NBC, typename ZF, typename OF>
  __device__ __forceinline__ void heads(const TailParams& p, int lane, int nb, ZF zrow, OF obase, float (&a)[NBC][E]) const {
    if (p.T_M == 64 * E) heads_impl<true>(p, lane, nb, zrow, obase, a);      // wave-uniform
    else heads_impl<false>(p, lane, nb, zrow, obase, a);
  }
};

}  // namespace sea
