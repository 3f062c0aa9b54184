// MFMA fragment helpers shared by the C8 convolution kernel (sea_conv.hip) and the fused decode-step kernel (sea_topk.hip).
#pragma once
#include "sea_common.hpp"

namespace sea {

typedef __attribute__((ext_vector_type(4))) float cf4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) unsigned int cu4;

template <typename T> struct Mfma16;
template <> struct Mfma16<__hip_bfloat16> {
  __device__ static inline cf4 run(const uint4& a, const cu4& b, cf4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma16<__half> {
  __device__ static inline cf4 run(const uint4& a, const cu4& b, cf4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
  }
};

// MFMA row (nt, 4 g + r) of the convolution's weight operand <-> output channel conv_chan<NT>(nt, g) + r (see sea_conv.hip)
template <int NT> __device__ __forceinline__ constexpr int conv_chan(int nt, int g) {
  return nt < (NT & ~1) ? (nt >> 1) * 32 + g * 8 + (nt & 1) * 4 : (NT - 1) * 16 + g * 4;
}

// ONE output row of causal_conv_c8_kernel<T, NT, 3, ONESEG> (W <= 64) computed by a 4-wave workgroup from its three input rows
// (tap rows t - 2 dil, t - dil, t, each a C8 row (C/8, W, 8)): wave w takes pixel tile w x all NT channel tiles.  Same operand
// placement, same k order (tap row, 32-channel chunk, tap column), same epilogue arithmetic as the convolution kernel: the row
// is bit for bit the one that kernel writes.  KCH = CinP / 32.
// Memory plan (one row has no work to hide latency behind, so every load is issued as early as it can be): the pixel
// fragments of the rows that exist already (`pre`) are requested first, then the whole weight image -- (16 NT) x 9 KCH x 64
// bytes, 74 KB at 64 -> 64 channels -- goes through registers into LDS in the convolution kernel's fragment order (all of a
// thread's chunks in flight at once: one round trip, not one per k-step), barrier, then 9 KCH steps of ds_read + MFMA.
// 256 threads; sW: 16 NT x 9 KCH x 64 bytes of LDS nobody else touches between the two barriers inside.
template <typename T, int NT, int KCH>
struct ConvRowC8 {
  static constexpr int ROWS = 16 * NT, NSTEPS = 9 * KCH, CinP = KCH * 32;
  static constexpr int NCH = NSTEPS * 4 * ROWS;                 // 16-byte chunks of the LDS image
  static constexpr int PER = (NCH + 255) / 256;                 // chunks per thread
  uint4 bf[3][KCH][3];                                          // this lane's pixel fragments, [tap row][chunk][tap column]

  __device__ __forceinline__ void fetch_row(int ti, const T* __restrict__ r, int C, int W, int dil, int pad_w) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int cci = 0; cci < KCH; ++cci)
#pragma unroll
      for (int tj = 0; tj < 3; ++tj) {
        const int px = wv * 16 + li + dil * tj - pad_w;
        const int blk = cci * 4 + lg;
        bf[ti][cci][tj] = make_uint4(0, 0, 0, 0);
        if ((unsigned)px < (unsigned)W && blk < (C >> 3)) bf[ti][cci][tj] = *reinterpret_cast<const uint4*>(r + ((int64_t)blk * W + px) * 8);
      }
  }

  // weights -> registers -> LDS (the convolution kernel's image: chunk ch = (st * 4 + g) * ROWS + row).  Two halves, so that
  // the NEXT layer's weights can be in flight while this layer's row is computed; store_weights ends with a barrier.
  uint4 wv_[PER];
  __device__ __forceinline__ void load_weights(const T* __restrict__ wp, int C) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int ch = i * 256 + (int)threadIdx.x;
      wv_[i] = make_uint4(0, 0, 0, 0);
      if (ch < NCH) {
        const int row = ch % ROWS, sl = ch / ROWS;
        const int g = sl & 3, st = sl >> 2;
        const int tj = st % 3, tc = st / 3;
        const int cci = tc % KCH, ti = tc / KCH;
        const int nt = row >> 4, rr = row & 15;
        const int co = conv_chan<NT>(nt, rr >> 2) + (rr & 3);
        if (co < C) wv_[i] = *reinterpret_cast<const uint4*>(wp + ((int64_t)co * 9 + ti * 3 + tj) * CinP + cci * 32 + g * 8);
      }
    }
  }
  __device__ __forceinline__ void store_weights(T* sW) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int ch = i * 256 + (int)threadIdx.x;
      if (ch < NCH) *reinterpret_cast<uint4*>(sW + (int64_t)ch * 8) = wv_[i];
    }
    __syncthreads();
  }

  // this lane's biases (channels conv_chan<NT>(nt, lg) .. +3 per tile), requested with the first loads of the kernel: read in the
  // epilogue they were one more exposed round trip per row (a row has nothing to hide it behind)
  float bs[NT][4];
  __device__ __forceinline__ void load_bias(const float* __restrict__ bias, int C) {
    const int lg = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c0 = conv_chan<NT>(nt, lg);
#pragma unroll
      for (int r = 0; r < 4; ++r) bs[nt][r] = c0 + r < C ? bias[c0 + r] : 0.f;
    }
  }

  // the row: MFMAs in the convolution kernel's order, its epilogue, the C8 store.  Ends with a barrier (the stored row is
  // visible to the workgroup, and sW may be overwritten).  load_bias() first.
  __device__ __forceinline__ void run(const T* sW, T* __restrict__ out, int C, int W, int relu) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const T* wl = sW + (lg * ROWS + li) * 8;
    cf4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = cf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ti = 0; ti < 3; ++ti)
#pragma unroll
      for (int cci = 0; cci < KCH; ++cci)
#pragma unroll
        for (int tj = 0; tj < 3; ++tj) {
          const int st = (ti * KCH + cci) * 3 + tj;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const uint4 a = *reinterpret_cast<const uint4*>(wl + (int64_t)st * (4 * ROWS * 8) + nt * 128);
            acc[nt] = Mfma16<T>::run(a, __builtin_bit_cast(cu4, bf[ti][cci][tj]), acc[nt]);
          }
        }
    const int wpix = wv * 16 + li, C8 = C >> 3;
    unsigned pk[2 * NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v0 = acc[nt][0] + bs[nt][0], v1 = acc[nt][1] + bs[nt][1];
      float v2 = acc[nt][2] + bs[nt][2], v3 = acc[nt][3] + bs[nt][3];
      if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
      pk[2 * nt] = pack2<T>(v0, v1);
      pk[2 * nt + 1] = pack2<T>(v2, v3);
    }
    if (wpix < W) {
#pragma unroll
      for (int q = 0; q < NT / 2; ++q) {
        const int blk = q * 4 + lg;
        if (blk < C8) *reinterpret_cast<uint4*>(out + ((int64_t)blk * W + wpix) * 8) = make_uint4(pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]);
      }
      if constexpr (NT & 1) {
        const int c0 = conv_chan<NT>(NT - 1, lg);
        if (c0 < C) *reinterpret_cast<uint2*>(out + ((int64_t)(c0 >> 3) * W + wpix) * 8 + (c0 & 7)) = make_uint2(pk[2 * NT - 2], pk[2 * NT - 1]);
      }
    }
    __syncthreads();
  }
};

}  // namespace sea
