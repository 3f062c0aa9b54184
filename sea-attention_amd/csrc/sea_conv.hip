// SEA predictor CNN in channels-last (NHWC) form, hand-written for gfx950 (16-bit data, bf16/f16 MFMA).
//
// Replaces, for 16-bit tensors (reference: src/models/perlin_attention/attention.py:266-281,
// modules.py:96-192):
//   ChannelSplit + cnn.lnorm1                         -> split_layernorm_nhwc_kernel   (writes NHWC)
//   cnn.keepres.conv1 / conv2 (+ the ReLU after each) -> causal_conv_nhwc_kernel       (NHWC -> NHWC)
// MIOpen's implicit-GEMM needs NHWC too and therefore brackets every NCHW conv with two layout transposes,
// a padded copy of the input and separate bias / ReLU passes; here the tensors simply stay NHWC between the
// LayerNorm and the predictor tail, padding is done by predication and bias + ReLU live in the epilogue.
//
// causal_conv_nhwc_kernel: implicit GEMM, M = pixels, N = C_out, K = taps x C_in, v_mfma_f32_16x16x32.
//   one wave = 64 consecutive pixels of one (n, t) row  x  all C_out   (4 M-tiles x NT N-tiles)
//   A fragments (8 channels of one tap of one pixel = 16 B) come straight from global memory (L1/L2 absorb the
//   9x tap reuse); B fragments (weights, [co][tap][ci] with padded rows) are staged once per workgroup in LDS.
#include "sea_common.hpp"

namespace sea {

typedef __attribute__((ext_vector_type(4))) float cf4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;

template <typename T> struct Mfma16;
template <> struct Mfma16<__hip_bfloat16> {
  __device__ static inline cf4 run(const uint4& a, const uint4& b, cf4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma16<__half> {
  __device__ static inline cf4 run(const uint4& a, const uint4& b, cf4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
  }
};

struct ConvParams {
  const void* x;    // (N, T, W, Cin)   NHWC
  const void* w;    // (Cout, taps*CinP) packed [co][tap][ci], ci padded to CinP (multiple of 32), 16-bit
  const float* b;   // (Cout) fp32
  void* y;          // (N, T, W, Cout)  NHWC
  int N, T, W, Cin, Cout, CinP;
  int KS, dil, pad_w, relu;
};

// NT = number of 16-wide output-channel tiles (Cout <= 16*NT)
template <typename T, int NT>
__global__ __launch_bounds__(256) void causal_conv_nhwc_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int taps = p.KS * p.KS;
  const int KP = taps * p.CinP;                 // packed K extent (elements)
  const int ldw = KP + 8;                       // LDS row stride in elements (+16 B: conflict-free b128 reads)
  T* sW = reinterpret_cast<T*>(smem);           // (16*NT) x ldw
  // ---- stage the weights once per workgroup ----------------------------------------------------------
  {
    const T* wg = reinterpret_cast<const T*>(p.w);
    const int chunks_per_row = KP / 8;
    for (int ch = threadIdx.x; ch < 16 * NT * chunks_per_row; ch += 256) {
      const int co = ch / chunks_per_row, kc = (ch - co * chunks_per_row) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (co < p.Cout) v = *reinterpret_cast<const uint4*>(wg + (int64_t)co * KP + kc);
      *reinterpret_cast<uint4*>(sW + co * ldw + kc) = v;
    }
  }
  __syncthreads();

  const int segs = (p.W + 63) / 64;                          // 64-pixel segments per row
  const int64_t nwork = (int64_t)p.N * p.T * segs;
  for (int64_t work = (int64_t)blockIdx.x * 4 + wv; work < nwork; work += (int64_t)gridDim.x * 4) {
    const int seg = (int)(work % segs);
    const int64_t nt_ = work / segs;
    const int t = (int)(nt_ % p.T), n = (int)(nt_ / p.T);
    const int w0 = seg * 64;
    const T* xn = reinterpret_cast<const T*>(p.x) + (int64_t)n * p.T * p.W * p.Cin;

    cf4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = cf4{0.f, 0.f, 0.f, 0.f};

    for (int ti = 0; ti < p.KS; ++ti) {
      const int tr = t + p.dil * (ti - (p.KS - 1));          // causal: rows t-(KS-1)*dil .. t
      if (tr < 0) continue;                                  // wave-uniform: zero padding on top
      for (int tj = 0; tj < p.KS; ++tj) {
        const int tap = ti * p.KS + tj;
        for (int cc = 0; cc < p.CinP; cc += 32) {
          const int ci = cc + 8 * lg;                        // this lane's 8 input channels
          uint4 a[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int wpix = w0 + mt * 16 + li;
            const int wc = wpix + p.dil * tj - p.pad_w;
            a[mt] = make_uint4(0, 0, 0, 0);
            if (ci < p.Cin && wc >= 0 && wc < p.W && wpix < p.W)
              a[mt] = *reinterpret_cast<const uint4*>(xn + ((int64_t)tr * p.W + wc) * p.Cin + ci);
          }
          const T* wrow = sW + tap * p.CinP + ci;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const uint4 b = *reinterpret_cast<const uint4*>(wrow + (nt * 16 + li) * ldw);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = Mfma16<T>::run(a[mt], b, acc[mt][nt]);
          }
        }
      }
    }
    // ---- epilogue: bias (+ ReLU), NHWC store.  C layout: col = li (channel), row = lg*4 + r (pixel) -------
    T* yn = reinterpret_cast<T*>(p.y) + ((int64_t)n * p.T + t) * p.W * p.Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = nt * 16 + li;
      const float bias = co < p.Cout ? p.b[co] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wpix = w0 + mt * 16 + lg * 4 + r;
          float v = acc[mt][nt][r] + bias;
          if (p.relu) v = fmaxf(v, 0.f);
          if (co < p.Cout && wpix < p.W) yn[(int64_t)wpix * p.Cout + co] = from_f<T>(v);
        }
      }
    }
  }
}

// ChannelSplit + LayerNorm with a channels-last result:
//   x (N, C, T, S*W) -> out[n, t, w, c*S+i] = LN(x[n, c, t, i*W:(i+1)*W])[w] * gamma[w] + beta[w]
// One workgroup per (n, t): the C*S rows are normalised by 8-lane groups, transposed through LDS and written
// as one contiguous W x (C*S) block.
template <typename T>
__global__ __launch_bounds__(256) void split_layernorm_nhwc_kernel(const T* x, T* out, const T* gamma, const T* beta, float eps,
                                                                  int C, int Tn, int S, int W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC = 8;
  T* tile = reinterpret_cast<T*>(smem);        // W x (CS + 8)
  const int CS = C * S;
  const int ldt = CS + 8;
  const int nt = blockIdx.x;
  const int n = nt / Tn, t = nt - n * Tn;
  const int lpr = W / VEC;                      // lanes per row (W % 8 == 0, W <= 512)
  const int rows_per_pass = 256 / lpr;
  const int sub = threadIdx.x % lpr, rloc = threadIdx.x / lpr;
  float g[VEC], b[VEC];
  unpack16<T>(*reinterpret_cast<const uint4*>(gamma + sub * VEC), g);
  unpack16<T>(*reinterpret_cast<const uint4*>(beta + sub * VEC), b);
  const float invW = 1.0f / (float)W;
  for (int r0 = 0; r0 < CS; r0 += rows_per_pass) {
    const int r = r0 + rloc;                    // output channel c*S + i
    const bool ok = r < CS && rloc < rows_per_pass;
    float f[VEC];
    {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        const int c = r / S, i = r - c * S;
        v = *reinterpret_cast<const uint4*>(x + (((int64_t)n * C + c) * Tn + t) * ((int64_t)S * W) + i * W + sub * VEC);
      }
      unpack16<T>(v, f);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += f[j];
    for (int o = 1; o < lpr; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * invW;
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { const float d = f[j] - mean; q2 += d * d; }
    for (int o = 1; o < lpr; o <<= 1) q2 += __shfl_xor(q2, o);
    const float rstd = rsqrtf(q2 * invW + eps);
    if (ok) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) tile[(sub * VEC + j) * ldt + r] = from_f<T>((f[j] - mean) * rstd * g[j] + b[j]);
    }
  }
  __syncthreads();
  T* on = out + (int64_t)nt * W * CS;
  const int cpr = CS / VEC;                     // 16-byte chunks per pixel
  for (int ch = threadIdx.x; ch < W * cpr; ch += 256) {
    const int w = ch / cpr, c0 = (ch - w * cpr) * VEC;
    *reinterpret_cast<uint4*>(on + (int64_t)w * CS + c0) = *reinterpret_cast<const uint4*>(tile + w * ldt + c0);
  }
}

}  // namespace sea

using namespace sea;

template <typename T>
static int launch_conv(const ConvParams& p, hipStream_t s) {
  const int nt = (p.Cout + 15) / 16;
  const size_t lds = (size_t)(16 * nt) * (size_t)(p.KS * p.KS * p.CinP + 8) * sizeof(T);
  if (lds > 160 * 1024) return SEA_EUNSUPPORTED;
  const int64_t nwork = (int64_t)p.N * p.T * ((p.W + 63) / 64);
  int64_t blocks = (nwork + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;      // persistent-ish: weights are staged once per workgroup
  dim3 grid((unsigned)blocks), block(256);
#define SEA_CONV(NTV)                                                                                               \
  do {                                                                                                              \
    static bool configured = false;                                                                                 \
    if (lds > 64 * 1024 && !configured) {                                                                           \
      (void)hipFuncSetAttribute((const void*)causal_conv_nhwc_kernel<T, NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      configured = true;                                                                                            \
    }                                                                                                               \
    hipLaunchKernelGGL((causal_conv_nhwc_kernel<T, NTV>), grid, block, lds, s, p);                                  \
  } while (0)
  switch (nt) {
    case 1: SEA_CONV(1); break; case 2: SEA_CONV(2); break; case 3: SEA_CONV(3); break; case 4: SEA_CONV(4); break;
    case 5: SEA_CONV(5); break; case 6: SEA_CONV(6); break; case 8: SEA_CONV(8); break;
    default: return SEA_EUNSUPPORTED;
  }
#undef SEA_CONV
  return SEA_OK;
}

extern "C" int sea_causal_conv_nhwc(const void* x, int dtype, int64_t N, int64_t T, int64_t W, int64_t Cin, int64_t Cout,
                                    const void* w_packed, int64_t CinP, const float* bias, int ksize, int dilation,
                                    int pad_w, int relu, void* y, sea_stream_t stream) {
  const char* nm = "sea_causal_conv_nhwc";
  SEA_REQUIRE(x && w_packed && bias && y, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16, SEA_EUNSUPPORTED, "%s: 16-bit data only (dtype %d)", nm, dtype);
  SEA_REQUIRE(N > 0 && T > 0 && W > 0 && Cin > 0 && Cout > 0 && ksize > 0 && dilation > 0 && pad_w >= 0, SEA_EINVAL,
              "%s: bad shape", nm);
  SEA_REQUIRE(Cin % 8 == 0 && CinP % 32 == 0 && CinP >= Cin && Cout <= 128, SEA_EUNSUPPORTED,
              "%s: needs Cin %% 8 == 0, CinP %% 32 == 0, Cout <= 128", nm);
  SEA_REQUIRE(W + dilation * (ksize - 1) - 2 * pad_w == W, SEA_EUNSUPPORTED, "%s: width-preserving padding only", nm);
  SEA_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed) & 15) == 0, SEA_EUNSUPPORTED, "%s: 16-byte alignment", nm);
  ConvParams p;
  p.x = x; p.w = w_packed; p.b = bias; p.y = y;
  p.N = (int)N; p.T = (int)T; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout; p.CinP = (int)CinP;
  p.KS = ksize; p.dil = dilation; p.pad_w = pad_w; p.relu = relu;
  hipStream_t s = (hipStream_t)stream;
  const int rc = dtype == SEA_BF16 ? launch_conv<__hip_bfloat16>(p, s) : launch_conv<__half>(p, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported channel count / kernel size for the LDS weight tile", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_split_layernorm_nhwc(const void* x, int dtype, int64_t N, int64_t C, int64_t T, int64_t S, int64_t W,
                                        const void* gamma, const void* beta, float eps, void* out, sea_stream_t stream) {
  const char* nm = "sea_split_layernorm_nhwc";
  SEA_REQUIRE(x && gamma && beta && out, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16, SEA_EUNSUPPORTED, "%s: 16-bit data only (dtype %d)", nm, dtype);
  SEA_REQUIRE(N > 0 && C > 0 && T > 0 && S > 0 && W > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int64_t lpr = W / 8;
  SEA_REQUIRE(W % 8 == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && (C * S) % 8 == 0, SEA_EUNSUPPORTED,
              "%s: needs W a power-of-two multiple of 8 (<= 512) and C*S %% 8 == 0", nm);
  const size_t lds = (size_t)W * (size_t)(C * S + 8) * 2;
  SEA_REQUIRE(lds <= 64 * 1024, SEA_EUNSUPPORTED, "%s: tile needs %zu B of LDS", nm, lds);
  SEA_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, SEA_EUNSUPPORTED,
              "%s: 16-byte alignment", nm);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(N * T)), block(256);
  if (dtype == SEA_BF16)
    hipLaunchKernelGGL((split_layernorm_nhwc_kernel<__hip_bfloat16>), grid, block, lds, s, (const __hip_bfloat16*)x,
                       (__hip_bfloat16*)out, (const __hip_bfloat16*)gamma, (const __hip_bfloat16*)beta, eps, (int)C, (int)T, (int)S, (int)W);
  else
    hipLaunchKernelGGL((split_layernorm_nhwc_kernel<__half>), grid, block, lds, s, (const __half*)x, (__half*)out,
                       (const __half*)gamma, (const __half*)beta, eps, (int)C, (int)T, (int)S, (int)W);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
