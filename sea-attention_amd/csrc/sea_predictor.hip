// Bandwidth-bound pieces of SEA's estimator and epilogue, hand-written for gfx950.
//
// Replaces (reference, src/models/perlin_attention/):
//   attention.py:123-131,623 + 266 (cnn.lnorm1)    ChannelSplit + LayerNorm(T_M/4)      -> split_layernorm_kernel
//   attention.py:271-281, modules.py:42-55,77-92     UpsampleFP32(1,4) + CausalConv2d(2H->H, 1x1, pad 1)
//                                                    + KeepRes area-resize to T_M + LayerNorm(T_M)
//   attention.py:670-673                             + softmax over T_M                   -> predictor_tail_kernel
//   attention.py:1208-1244                           cumsum(v)/arange(1..T)               -> cumavg_kernel
//
// In the reference each of these is a chain of small framework kernels over (N,H,T,T_M)-sized tensors
// (upsample writes a 4x larger tensor which the 1x1 conv, the area pooling, the LayerNorm and the softmax
// each read and write again).  Algebra used here: a 1x1 convolution commutes with nearest-neighbour
// upsampling, so the channel GEMM runs on the T_M/4-wide tensor, and everything after it happens on-chip.
#include "sea_common.hpp"
#include "sea_tail.hpp"

#ifdef SEA_STAMP
__device__ unsigned long long sea_dbg_tail[8];
#define TSTAMP(i) do { if (threadIdx.x == 0) { unsigned long long _t = __builtin_amdgcn_s_memtime(); atomicAdd(&sea_dbg_tail[i], _t - _tprev); _tprev = _t; } } while (0)
#else
#define TSTAMP(i) do {} while (0)
#endif

namespace sea {

// ------------------------------------------------------------------------------------------------------
// ChannelSplit + LayerNorm:  x (N,C,T,S*W) -> out (N,C*S,T,W),  out[n,c*S+i,t,:] = LN(x[n,c,t,i*W:(i+1)*W])
// LPR lanes own one output row (16 B each); a wave-instruction covers 64/LPR rows.
// ------------------------------------------------------------------------------------------------------
template <int LPR> __device__ inline float lpr_sum(float x) {
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) x += __shfl_xor(x, o);
  return x;
}

template <typename T, int LPR>
__global__ __launch_bounds__(256) void split_layernorm_kernel(const T* x, T* out, const T* gamma, const T* beta, float eps,
                                                             int64_t rows, int C, int Tn, int S, int W, int act_gelu) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int RPW = 64 / LPR;  // rows per wave-instruction
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, sub = lane - grp * LPR;
  const bool act = sub * VEC < W;
  float g[VEC], b[VEC];
  {
    uint4 gr = make_uint4(0, 0, 0, 0), br = make_uint4(0, 0, 0, 0);
    if (act) { gr = *reinterpret_cast<const uint4*>(gamma + sub * VEC); br = *reinterpret_cast<const uint4*>(beta + sub * VEC); }
    unpack16<T>(gr, g); unpack16<T>(br, b);
  }
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float invW = 1.0f / (float)W;
  for (int64_t r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t r = r0 + grp;  // output row id = ((n*C + c)*S + i)*T + t
    const bool ok = r < rows && act;
    float f[VEC];
    int64_t o_off = 0;
    {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        const int64_t t = r % Tn;
        const int64_t q = r / Tn;          // (n*C + c)*S + i
        const int64_t i = q % S;
        const int64_t nc = q / S;          // n*C + c
        v = *reinterpret_cast<const uint4*>(x + (nc * Tn + t) * ((int64_t)S * W) + i * W + sub * VEC);
        o_off = r * W + sub * VEC;
      }
      unpack16<T>(v, f);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += f[j];
    const float mean = lpr_sum<LPR>(s) * invW;
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { const float d = act ? f[j] - mean : 0.f; q2 += d * d; }
    const float rstd = rsqrtf(lpr_sum<LPR>(q2) * invW + eps);
    if (ok) {
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = (f[j] - mean) * rstd * g[j] + b[j];
      if (act_gelu) {   // nn.GELU() default: exact erf form
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = gelu_erf(o[j]);
      }
      if (VEC == 4) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o_off) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        union { uint4 u; T h[8]; } pk;
#pragma unroll
        for (int j = 0; j < VEC; ++j) pk.h[j] = from_f<T>(o[j]);
        *reinterpret_cast<uint4*>(out + o_off) = pk.u;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Predictor tail.  One wave per (n, t) row, four rows per workgroup.
//   z[h][w]  = b4[h] + sum_c w4[h][c] * y[n,c,t,w]                       (1x1 conv, before upsampling)
//   P[h][x]  = b4[h] for x in {0, 4*W4+1},  z[h][(x-1)/UP] otherwise     (upsample x UP, zero pad 1 -> bias)
//   a[h][j]  = mean(P[h][floor(j*Wp/T_M) : ceil((j+1)*Wp/T_M)])         (area resize Wp -> T_M, adaptive avg pool)
//   s[h][j]  = LayerNorm_j(a[h][:]) * gamma[j] + beta[j]                  (estimated_attention_score)
//   p[h][j]  = softmax_j(s[h][:])                                         (estimated_attention_probs)
// LDS per wave: the y tile (C x W4, fp32) which is then overwritten by z (H x W4); weights shared.
// ------------------------------------------------------------------------------------------------------

// E = output pixels per lane (lane owns j in [lane*E, lane*E+E)).
// One workgroup (4 waves) per (n, t) row: the y tile is staged once in LDS (fp32), every wave takes blocks of
// HB = 8 heads.  The conv weights are read through the SCALAR cache (wave-uniform addresses, transposed
// fp32 copy wT[c][Hpad]) and enter the FMAs as SGPR operands, so LDS carries only y and the z block.
template <typename T, int E>
__global__ __launch_bounds__(256) void predictor_tail_kernel(TailParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HB = 8;
  constexpr int VEC = Elem<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int CW = p.C * p.W4;
  const int LDY = p.W4 + 1;                                            // padded row stride of the y tile
  float* s_y = reinterpret_cast<float*>(smem);                        // C x LDY
  float* s_z = s_y + p.C * LDY + wv * (HB * p.W4);                    // per-wave HB x W4
  const float* __restrict__ wT = reinterpret_cast<const float*>(p.w4); // (C, Hpad) fp32, transposed
  const float* __restrict__ bF = reinterpret_cast<const float*>(p.b4); // (Hpad) fp32
  const int Hpad = ((p.H + HB - 1) / HB) * HB;

  const int row = blockIdx.x;
  const int n = row / p.T, t = row - n * p.T;
#ifdef SEA_STAMP
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
#endif

  // ---- stage y[n, :, t, :] (C x W4) as fp32, 16-byte loads -----------------------------------------------
  {
    const T* yb = reinterpret_cast<const T*>(p.y) + n * p.ys_n + t * p.ys_t;
    if (p.ys_w == 1) {                            // NCHW: vectors run along the width
      const int cpr = p.W4 / VEC;
      for (int ch = threadIdx.x; ch < CW / VEC; ch += 256) {
        const int c = ch / cpr, w = (ch - c * cpr) * VEC;
        float f[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(yb + c * p.ys_c + w), f);
        float* dst = s_y + c * LDY + w;
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = f[j];
      }
    } else {                                      // channels-last: vectors run along the channels
      const int cpp = p.C / VEC;
      for (int ch = threadIdx.x; ch < CW / VEC; ch += 256) {
        const int w = ch / cpp, c = (ch - w * cpp) * VEC;
        float f[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(yb + (int64_t)w * p.ys_w + (c >> 3) * p.ys_c8 + (c & 7)), f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s_y[(c + j) * LDY + w] = f[j];
      }
    }
  }

  // ---- per-lane constants of the area-resize / LayerNorm stage --------------------------------------------
  // output pixel j averages P[xs..xe) (torch adaptive_avg_pool index arithmetic), P = [bias, z upsampled, bias]
  const int Wp = p.W4 * p.UP + 2;
  const T* gam = reinterpret_cast<const T*>(p.gamma);
  const T* bet = reinterpret_cast<const T*>(p.beta);
  float g[E], be[E], rcnt[E];
  int src[E][3];   // source column in z (or -1 = padding -> bias, -2 = unused tap)
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = lane * E + e;
    g[e] = 0.f; be[e] = 0.f; rcnt[e] = 0.f;
    src[e][0] = src[e][1] = src[e][2] = -2;
    if (j < p.T_M) {
      g[e] = Elem<T>::to_f(gam[j]); be[e] = Elem<T>::to_f(bet[j]);
      const int xs = (int)floorf((float)(j * Wp) / (float)p.T_M);
      const int xe = (int)ceilf((float)((j + 1) * Wp) / (float)p.T_M);
      rcnt[e] = 1.0f / (float)(xe - xs);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int x = xs + k;
        if (x < xe) src[e][k] = (x == 0 || x == Wp - 1) ? -1 : (x - 1) / p.UP;
      }
    }
  }
  const float invT = 1.0f / (float)p.T_M;
  __syncthreads();
  TSTAMP(0);   // staging + per-lane constants

  const int hblocks = Hpad / HB;
  for (int it = 0; it * 4 < hblocks; ++it) {
    const int hb = it * 4 + wv;                 // wave-uniform
    const bool active = hb < hblocks;
    const int h0 = hb * HB;
    // ---- z[h0..h0+HB) = W y + b: lane owns columns w = lane (+64 ...) ---------------------------------------
    if (active) {
      for (int w = lane; w < p.W4; w += 64) {
        float acc[HB];
#pragma unroll
        for (int jh = 0; jh < HB; ++jh) acc[jh] = 0.f;
        const float* wp = wT + h0;
#pragma unroll 4
        for (int c = 0; c < p.C; ++c) {
          const float yv = s_y[c * LDY + w];
          const float4 wa = *reinterpret_cast<const float4*>(wp + (int64_t)c * Hpad);
          const float4 wb = *reinterpret_cast<const float4*>(wp + (int64_t)c * Hpad + 4);
          acc[0] = fmaf(wa.x, yv, acc[0]); acc[1] = fmaf(wa.y, yv, acc[1]);
          acc[2] = fmaf(wa.z, yv, acc[2]); acc[3] = fmaf(wa.w, yv, acc[3]);
          acc[4] = fmaf(wb.x, yv, acc[4]); acc[5] = fmaf(wb.y, yv, acc[5]);
          acc[6] = fmaf(wb.z, yv, acc[6]); acc[7] = fmaf(wb.w, yv, acc[7]);
        }
#pragma unroll
        for (int jh = 0; jh < HB; ++jh) s_z[jh * p.W4 + w] = acc[jh] + bF[h0 + jh];
      }
    }
    __syncthreads();
    TSTAMP(1);   // channel GEMM
    // ---- per head: area resize -> LayerNorm -> (scores) -> softmax -> probs -----------------------------------
    if (active) {
      for (int jh = 0; jh < HB && h0 + jh < p.H; ++jh) {
        const int h = h0 + jh;
        const float* zr = s_z + jh * p.W4;
        const float bias = bF[h];
        float a[E];
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int sc = src[e][k];
            acc += sc >= 0 ? zr[sc] : (sc == -1 ? bias : 0.f);
          }
          a[e] = acc * rcnt[e];
          s1 += a[e];
        }
        const float mean = wave_sum(s1) * invT;
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (lane * E + e < p.T_M) { const float d = a[e] - mean; s2 += d * d; }
        const float rstd = rsqrtf(wave_sum(s2) * invT + p.eps);
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          a[e] = (a[e] - mean) * rstd * g[e] + be[e];
          if (lane * E + e < p.T_M) mx = fmaxf(mx, a[e]);
        }
        mx = wave_max(mx);
        const int64_t obase = (((int64_t)n * p.H + h) * p.T + t) * p.T_M;
        if (p.scores) store_run<T, E>(reinterpret_cast<T*>(p.scores) + obase, a, lane * E, p.T_M);
        float se = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          a[e] = (lane * E + e < p.T_M) ? __expf(a[e] - mx) : 0.f;
          se += a[e];
        }
        const float inv = 1.0f / wave_sum(se);
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] *= inv;
        store_run<T, E>(reinterpret_cast<T*>(p.probs) + obase, a, lane * E, p.T_M);
      }
    }
    __syncthreads();
    TSTAMP(2);   // resize + LN + softmax + stores
  }
}

// MFMA variant of the predictor tail for channels-last 16-bit input: the 1x1 conv z = y W^T is
// (W4 pixels x C) @ (C x H) on v_mfma_f32_16x16x32; A fragments (8 channels of a pixel = 16 B) come straight from
// global memory, so there is no staging pass at all.  z lands in LDS as [head][pixel] (+ a bias slot and a zero
// slot per row, which turn the padded / unused taps of the area resize into plain reads).
template <typename T, int E>
__global__ __launch_bounds__(256) void predictor_tail_mfma_kernel(TailParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int LDZ = p.W4 + 3;                     // z row: W4 pixels, [W4] = bias, [W4+1] = 0 (odd stride: no bank aliasing)
  float* s_z = reinterpret_cast<float*>(smem);  // HP x LDZ
  uint32_t* s_tab = reinterpret_cast<uint32_t*>(s_z + ((p.H + 15) / 16) * 16 * LDZ);   // per-pixel constants [3][64 E]
  const int row = blockIdx.x;
  const int n = row / p.T, t = row - n * p.T;
  tail_z_tile<T>(p, s_z, n, t);
  tail_consts_fill<T>(p, s_tab, 64 * E);
  __syncthreads();
  TailRow<T, E> tr;
  tr.load(s_tab, lane);
  const int mine = (p.H - wv + 3) / 4;                     // heads wv, wv + 4, ... of this wave, in batches of 8
  for (int k0 = 0; k0 < mine; k0 += 8) {
    float a[8][E];
    tr.heads(p, lane, min(8, mine - k0),
             [&](int b) { return s_z + (wv + 4 * (k0 + b)) * LDZ; },
             [&](int b) { return (((int64_t)n * p.H + (wv + 4 * (k0 + b))) * p.T + t) * p.T_M; }, a);
  }
}

// ------------------------------------------------------------------------------------------------------
// Causal cumulative average:  out[n,h,t,:] = (sum_{s<=t} v[n,h,s,:]) / (t+1), fp32 accumulation.
// One workgroup per (n, h, 64-wide feature slab): 16 waves split T into segments; lanes own features.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void cumavg_kernel(const T* v, T* out, int Tn, int D, int64_t vs_n, int64_t vs_h,
                                                     int64_t vs_t, int H, int dslabs) {
  __shared__ float s_seg[16][64];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int blk = blockIdx.x;
  const int slab = blk % dslabs;
  const int nh = blk / dslabs;
  const int n = nh / H, h = nh - n * H;
  const int dcol = slab * 64 + lane;
  const bool act = dcol < D;
  const int per = (Tn + 15) / 16;
  const int t0 = seg * per, t1 = min(Tn, t0 + per);
  const T* vb = v + n * vs_n + h * vs_h + dcol;
  T* ob = out + ((int64_t)nh * Tn) * D + dcol;
  float s = 0.f;
  if (act) {
    int t = t0;
    for (; t + 4 <= t1; t += 4) {
      const float a0 = Elem<T>::to_f(vb[(int64_t)t * vs_t]), a1 = Elem<T>::to_f(vb[(int64_t)(t + 1) * vs_t]);
      const float a2 = Elem<T>::to_f(vb[(int64_t)(t + 2) * vs_t]), a3 = Elem<T>::to_f(vb[(int64_t)(t + 3) * vs_t]);
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; t < t1; ++t) s += Elem<T>::to_f(vb[(int64_t)t * vs_t]);
  }
  s_seg[seg][lane] = s;
  __syncthreads();
  float run = 0.f;
  for (int i = 0; i < seg; ++i) run += s_seg[i][lane];
  if (act) {
    for (int t = t0; t < t1; ++t) {
      run += Elem<T>::to_f(vb[(int64_t)t * vs_t]);
      ob[(int64_t)t * D] = from_f<T>(run / (float)(t + 1));
    }
  }
}

// 16-bit fast path (D = 8*FG, FG | 64): a thread owns 8 adjacent features (one 16-byte vector) and a run of
// consecutive rows; a wave is RL = 64/FG row-lanes x FG feature groups, 16 waves split T.  Pass 1 sums the runs,
// the run prefixes come from RL-1 shuffles inside the wave plus the 16 wave totals in LDS, pass 2 re-reads the
// rows (L2 / Infinity Cache) and writes the averages.  Every load and store moves whole 128-byte lines.
// Few (n,h) pairs: the rows are cut into `gridDim.y` slices of slice_len rows.  SUMS_ONLY leaves every slice's column
// totals in carry[(nh * nslices + slice) * D], the second launch starts a slice from the totals before it.
template <typename T, int FG, bool SUMS_ONLY>
__global__ __launch_bounds__(1024) void cumavg_vec_kernel(const T* v, T* out, int T_all, int64_t vs_n, int64_t vs_h, int64_t vs_t,
                                                         int H, int slice_len, float* carry) {
  constexpr int D = FG * 8, RL = 64 / FG;
  const int slice = blockIdx.y, nslices = gridDim.y;
  const int ts0 = slice * slice_len, Tn = min(T_all, ts0 + slice_len);     // this block's rows: [ts0, Tn)
  __shared__ float s_tot[16][D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int fg = lane % FG, r = lane / FG;
  const int nh = blockIdx.x;
  const int n = nh / H, h = nh - n * H;
  const int seg_len = (Tn - ts0 + 16 * RL - 1) / (16 * RL);
  // FG = 10 (d = 80): RL = 6 row-lanes use 60 lanes, the last 4 lanes of a wave own no rows
  const int t0 = r < RL ? min(Tn, ts0 + (wv * RL + r) * seg_len) : Tn, t1 = min(Tn, t0 + seg_len);
  const T* vb = v + n * vs_n + h * vs_h + fg * 8;
  T* ob = out + (int64_t)nh * T_all * D + fg * 8;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  {
    int t = t0;
    for (; t + 4 <= t1; t += 4) {                       // 4 independent 16-byte loads in flight
      uint4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const uint4*>(vb + (int64_t)(t + u) * vs_t);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack16<T>(x[u], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += f[j];
      }
    }
    for (; t < t1; ++t) {
      float f[8];
      unpack16<T>(*reinterpret_cast<const uint4*>(vb + (int64_t)t * vs_t), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
  }
  float run[8];                                          // exclusive prefix over the row-lanes of this wave
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float inc = s[j];
#pragma unroll
    for (int k = 1; k < RL; k <<= 1) {
      const float up = __shfl_up(inc, k * FG);
      if (r >= k) inc += up;
    }
    run[j] = inc - s[j];
    if (r == RL - 1) s_tot[wv][fg * 8 + j] = inc;
  }
  __syncthreads();
  if constexpr (SUMS_ONLY) {
    if (threadIdx.x < D) {                                 // column totals of the slice, waves added in order
      float tot = 0.f;
      for (int w = 0; w < 16; ++w) tot += s_tot[w][threadIdx.x];
      carry[((int64_t)nh * nslices + slice) * D + threadIdx.x] = tot;
    }
    return;
  }
  if (slice > 0 && r < RL) {
    for (int s2 = 0; s2 < slice; ++s2) {                   // fixed order
      const float* c = carry + ((int64_t)nh * nslices + s2) * D + fg * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) run[j] += c[j];
    }
  }
  for (int w = 0; w < wv; ++w) {
    const float4 a = *reinterpret_cast<const float4*>(&s_tot[w][fg * 8]);
    const float4 b = *reinterpret_cast<const float4*>(&s_tot[w][fg * 8 + 4]);
    run[0] += a.x; run[1] += a.y; run[2] += a.z; run[3] += a.w;
    run[4] += b.x; run[5] += b.y; run[6] += b.z; run[7] += b.w;
  }
  for (int t = t0; t < t1; ++t) {
    float f[8];
    unpack16<T>(*reinterpret_cast<const uint4*>(vb + (int64_t)t * vs_t), f);
    const float den = (float)(t + 1);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { run[j] += f[j]; o[j] = run[j] / den; }
    *reinterpret_cast<uint4*>(ob + (int64_t)t * D) =
        make_uint4(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7]));
  }
}

// the per-pixel constants of the tail (tail_consts_fill: taps of the area resize, gamma, beta) into GLOBAL memory, once per
// weight set -- the fused tail + selection kernels then read a lane's 12 words with three 16-byte loads instead of computing
// the table for every row (round 5)
template <typename T> __global__ __launch_bounds__(256) void tail_consts_kernel(TailParams p, uint32_t* tab, int TMP) {
  tail_consts_fill<T>(p, tab, TMP);
}

}  // namespace sea

using namespace sea;

template <typename T>
static int launch_split_ln(const void* x, void* out, const void* g, const void* b, float eps, int64_t N, int64_t C,
                           int64_t Tn, int64_t S, int64_t W, int act, hipStream_t s) {
  constexpr int VEC = Elem<T>::VEC;
  int lpr = 1;
  while (lpr * VEC < W) lpr *= 2;
  const int64_t rows = N * C * S * Tn;
  const int rpw = 64 / lpr;
  int64_t blocks = (rows + 4 * rpw - 1) / (4 * rpw);
  if (blocks > 16384) blocks = 16384;
  dim3 grid((unsigned)blocks), block(256);
#define SEA_SLN(L) hipLaunchKernelGGL((split_layernorm_kernel<T, L>), grid, block, 0, s, (const T*)x, (T*)out, (const T*)g, \
                                      (const T*)b, eps, rows, (int)C, (int)Tn, (int)S, (int)W, act)
  switch (lpr) {
    case 1: SEA_SLN(1); break; case 2: SEA_SLN(2); break; case 4: SEA_SLN(4); break; case 8: SEA_SLN(8); break;
    case 16: SEA_SLN(16); break; case 32: SEA_SLN(32); break; case 64: SEA_SLN(64); break;
    default: return SEA_EUNSUPPORTED;
  }
#undef SEA_SLN
  return SEA_OK;
}

extern "C" int sea_split_layernorm(const void* x, int dtype, int64_t N, int64_t C, int64_t T, int64_t S, int64_t W,
                                   const void* gamma, const void* beta, float eps, int activation, void* out,
                                   sea_stream_t stream) {
  const char* nm = "sea_split_layernorm";
  SEA_REQUIRE(x && gamma && beta && out, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && C > 0 && T > 0 && S > 0 && W > 0, SEA_EINVAL, "%s: bad shape", nm);
  SEA_REQUIRE(activation == 0 || activation == 1, SEA_EINVAL, "%s: activation must be 0 (none) or 1 (GELU)", nm);
  const int vec = dtype == SEA_F32 ? 4 : 8;
  SEA_REQUIRE(W % vec == 0 && W <= 64 * vec, SEA_EUNSUPPORTED, "%s: W=%lld must be a multiple of %d and <= %d", nm,
              (long long)W, vec, 64 * vec);
  SEA_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, SEA_EUNSUPPORTED,
              "%s: tensors must be 16-byte aligned", nm);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (dtype == SEA_F32) rc = launch_split_ln<float>(x, out, gamma, beta, eps, N, C, T, S, W, activation, s);
  else if (dtype == SEA_F16) rc = launch_split_ln<__half>(x, out, gamma, beta, eps, N, C, T, S, W, activation, s);
  else rc = launch_split_ln<__hip_bfloat16>(x, out, gamma, beta, eps, N, C, T, S, W, activation, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported width", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

template <typename T>
static int launch_tail_mfma(const TailParams& p, dim3 grid, hipStream_t s) {
  const int E = (p.T_M + 63) / 64;
  const int HP = ((p.H + 15) / 16) * 16;
  const size_t lds = (size_t)HP * (p.W4 + 3) * sizeof(float) + (size_t)TAIL_TAB_ROWS * 64 * (E == 5 ? 6 : E == 7 ? 8 : E) * sizeof(uint32_t);
  if (lds > 64 * 1024 || p.W4 + 1 >= 1024 || p.W4 * p.UP + 2 > 2 * p.T_M) return SEA_EUNSUPPORTED;   // 10-bit taps, windows of <= 3 of them
#define SEA_TAILM(EE) hipLaunchKernelGGL((predictor_tail_mfma_kernel<T, EE>), grid, dim3(256), lds, s, p)
  switch (E) {
    case 1: SEA_TAILM(1); break; case 2: SEA_TAILM(2); break; case 3: SEA_TAILM(3); break; case 4: SEA_TAILM(4); break;
    case 5: case 6: SEA_TAILM(6); break; case 7: case 8: SEA_TAILM(8); break;
    default: return SEA_EUNSUPPORTED;
  }
#undef SEA_TAILM
  return SEA_OK;
}

template <typename T>
static int launch_tail(const TailParams& p, size_t lds, dim3 grid, hipStream_t s) {
  const int E = (p.T_M + 63) / 64;
#define SEA_TAIL(EE)                                                                                         \
  do {                                                                                                       \
    if (lds > 64 * 1024) SEA_MAX_LDS((predictor_tail_kernel<T, EE>), lds);   /* per launch: the size varies with the shape */ \
    hipLaunchKernelGGL((predictor_tail_kernel<T, EE>), grid, dim3(256), lds, s, p);                          \
  } while (0)
  switch (E) {
    case 1: SEA_TAIL(1); break; case 2: SEA_TAIL(2); break; case 3: SEA_TAIL(3); break; case 4: SEA_TAIL(4); break;
    case 5: case 6: SEA_TAIL(6); break; case 7: case 8: SEA_TAIL(8); break;
    default: return SEA_EUNSUPPORTED;
  }
#undef SEA_TAIL
  return SEA_OK;
}

#ifdef SEA_STAMP
extern "C" int sea_debug_tail_stamps(unsigned long long* host8) {
  (void)hipMemcpyFromSymbol(host8, HIP_SYMBOL(sea_dbg_tail), sizeof(unsigned long long) * 8);
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(sea_dbg_tail), z, sizeof(z));
  return 0;
}
#endif

extern "C" int sea_predictor_tail(const void* y, int dtype, int64_t N, int64_t C, int64_t H, int64_t T, int64_t W4,
                                  int64_t up, int64_t T_m, const int64_t* y_strides, const void* conv_w,
                                  const void* conv_b, const void* conv_w16, int64_t Cp, const void* gamma,
                                  const void* beta, float eps, void* probs, void* scores, sea_stream_t stream) {
  const char* nm = "sea_predictor_tail";
  SEA_REQUIRE(y && y_strides && conv_w && conv_b && gamma && beta && probs, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && C > 0 && H > 0 && T > 0 && W4 > 0 && up > 0 && T_m > 0, SEA_EINVAL, "%s: bad shape", nm);
  SEA_REQUIRE(T_m <= 512, SEA_EUNSUPPORTED, "%s: T_m=%lld > 512", nm, (long long)T_m);
  SEA_REQUIRE(W4 * up == T_m, SEA_EUNSUPPORTED, "%s: needs W4*up == T_m (area resize of T_m+2 -> T_m)", nm);
  SEA_REQUIRE((((uintptr_t)probs | (uintptr_t)scores) & 15) == 0 && T_m % 4 == 0, SEA_EUNSUPPORTED,
              "%s: outputs must be 16-byte aligned", nm);
  const int vec = dtype == SEA_F32 ? 4 : 8;
  const bool nchw = y_strides[3] == 1, nhwc = y_strides[1] == 1;
  SEA_REQUIRE(nchw || nhwc, SEA_EUNSUPPORTED, "%s: y must have unit stride along the width (NCHW) or the channels (NHWC)", nm);
  SEA_REQUIRE(nchw ? y_strides[4] == 8 * y_strides[1] : (C % 8 == 0 || y_strides[4] == 8), SEA_EUNSUPPORTED,
              "%s: the block-of-8 stride must be 8x the channel stride unless y is channel-blocked with C %% 8 == 0", nm);
  SEA_REQUIRE((nchw ? (W4 % vec == 0 && y_strides[1] % vec == 0) : (C % vec == 0 && y_strides[3] % vec == 0 && y_strides[4] % vec == 0)) &&
                  y_strides[0] % vec == 0 && y_strides[2] % vec == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)conv_w & 15) == 0,
              SEA_EUNSUPPORTED, "%s: y vectors must be 16-byte aligned", nm);
  const size_t lds = (size_t)(C * (W4 + 1) + 4 * 8 * W4) * sizeof(float);
  SEA_REQUIRE(lds <= 160 * 1024, SEA_EUNSUPPORTED, "%s: needs %zu B of LDS", nm, lds);
  TailParams p;
  p.y = y; p.w4 = conv_w; p.b4 = conv_b; p.gamma = gamma; p.beta = beta; p.probs = probs; p.scores = scores; p.eps = eps;
  p.N = (int)N; p.C = (int)C; p.H = (int)H; p.T = (int)T; p.W4 = (int)W4; p.UP = (int)up; p.T_M = (int)T_m;
  p.ys_n = y_strides[0]; p.ys_c = y_strides[1]; p.ys_t = y_strides[2]; p.ys_w = y_strides[3]; p.ys_c8 = y_strides[4];
  p.w16 = conv_w16; p.Cp = (int)Cp; p.z = nullptr; p.tab = nullptr;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(N * T));
  int rc;
  const bool mfma = conv_w16 != nullptr && dtype != SEA_F32 && nhwc && !nchw && Cp % 32 == 0 && Cp >= C &&
                    (((uintptr_t)conv_w16) & 15) == 0;
  // fp32 channels-last / C8 input (round 5): the MFMA-structured kernel on the fp32 MFMA -- the device code the fused fp32 tail +
  // selection launch runs, so that the two agree bit for bit (dense mode reads this map, sparse mode selects from that one)
  const bool mfma32 = dtype == SEA_F32 && nhwc && !nchw && C % 4 == 0 && T_m % 4 == 0;
  if (mfma) rc = dtype == SEA_F16 ? launch_tail_mfma<__half>(p, grid, s) : launch_tail_mfma<__hip_bfloat16>(p, grid, s);
  else if (mfma32) rc = launch_tail_mfma<float>(p, grid, s);
  else if (dtype == SEA_F32) rc = launch_tail<float>(p, lds, grid, s);
  else if (dtype == SEA_F16) rc = launch_tail<__half>(p, lds, grid, s);
  else rc = launch_tail<__hip_bfloat16>(p, lds, grid, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported T_m", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_predictor_tail_consts(int dtype, int64_t W4, int64_t up, int64_t T_m, const void* gamma, const void* beta,
                                         uint32_t* tab, sea_stream_t stream) {
  const char* nm = "sea_predictor_tail_consts";
  SEA_REQUIRE(gamma && beta && tab, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16 || dtype == SEA_F32, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(W4 > 0 && up > 0 && T_m == 256 && W4 * up == T_m && W4 + 1 < 1024, SEA_EUNSUPPORTED,
              "%s: the table serves the T_m = 256 kernels (3 x 256 words)", nm);
  TailParams p;
  p.y = nullptr; p.w4 = nullptr; p.b4 = nullptr; p.gamma = gamma; p.beta = beta; p.probs = nullptr; p.scores = nullptr; p.eps = 0.f;
  p.N = 1; p.C = 0; p.H = 1; p.T = 1; p.W4 = (int)W4; p.UP = (int)up; p.T_M = (int)T_m;
  p.ys_n = p.ys_c = p.ys_t = p.ys_w = p.ys_c8 = 0; p.w16 = nullptr; p.Cp = 0; p.z = nullptr; p.tab = nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SEA_F16) hipLaunchKernelGGL((tail_consts_kernel<__half>), dim3(1), dim3(256), 0, s, p, tab, 256);
  else if (dtype == SEA_F32) hipLaunchKernelGGL((tail_consts_kernel<float>), dim3(1), dim3(256), 0, s, p, tab, 256);
  else hipLaunchKernelGGL((tail_consts_kernel<__hip_bfloat16>), dim3(1), dim3(256), 0, s, p, tab, 256);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

// The tail from z = the 1x1 convolution's output, as sea_causal_conv_c8_z's epilogue writes it: (N, T, H, W4) fp32.
extern "C" int sea_predictor_tail_z(const float* z, int dtype, int64_t N, int64_t H, int64_t T, int64_t W4, int64_t up, int64_t T_m,
                                    const float* conv_b, const void* gamma, const void* beta, float eps, void* probs, void* scores,
                                    sea_stream_t stream) {
  const char* nm = "sea_predictor_tail_z";
  SEA_REQUIRE(z && conv_b && gamma && beta && probs, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16, SEA_EUNSUPPORTED, "%s: 16-bit maps only (dtype %d)", nm, dtype);
  SEA_REQUIRE(N > 0 && H > 0 && T > 0 && W4 > 0 && up > 0 && T_m > 0, SEA_EINVAL, "%s: bad shape", nm);
  SEA_REQUIRE(T_m <= 512 && W4 * up == T_m && T_m % 4 == 0 && W4 % 4 == 0, SEA_EUNSUPPORTED,
              "%s: needs W4 * up == T_m <= 512, T_m %% 4 == 0, W4 %% 4 == 0", nm);
  SEA_REQUIRE((((uintptr_t)probs | (uintptr_t)scores | (uintptr_t)z) & 15) == 0, SEA_EUNSUPPORTED, "%s: 16-byte alignment", nm);
  TailParams p;
  p.y = nullptr; p.w4 = nullptr; p.b4 = conv_b; p.gamma = gamma; p.beta = beta; p.probs = probs; p.scores = scores; p.eps = eps;
  p.N = (int)N; p.C = 0; p.H = (int)H; p.T = (int)T; p.W4 = (int)W4; p.UP = (int)up; p.T_M = (int)T_m;
  p.ys_n = p.ys_c = p.ys_t = p.ys_w = p.ys_c8 = 0; p.w16 = nullptr; p.Cp = 0; p.z = z; p.tab = nullptr;
  hipStream_t s = (hipStream_t)stream;
  const int rc = dtype == SEA_F16 ? launch_tail_mfma<__half>(p, dim3((unsigned)(N * T)), s) : launch_tail_mfma<__hip_bfloat16>(p, dim3((unsigned)(N * T)), s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported T_m / H for the tail's LDS plan", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

static int cumavg_entry(const char* nm, const void* v, int dtype, int64_t N, int64_t H, int64_t T, int64_t D,
                        const int64_t* v_strides, void* out, int64_t n_slices, void* workspace, int64_t workspace_bytes,
                        sea_stream_t stream);

extern "C" int sea_cumavg(const void* v, int dtype, int64_t N, int64_t H, int64_t T, int64_t D, const int64_t* v_strides,
                          void* out, sea_stream_t stream) {
  return cumavg_entry("sea_cumavg", v, dtype, N, H, T, D, v_strides, out, 1, nullptr, 0, stream);
}

extern "C" int sea_cumavg_sliced(const void* v, int dtype, int64_t N, int64_t H, int64_t T, int64_t D, const int64_t* v_strides,
                                 void* out, int64_t n_slices, void* workspace, int64_t workspace_bytes, sea_stream_t stream) {
  return cumavg_entry("sea_cumavg_sliced", v, dtype, N, H, T, D, v_strides, out, n_slices, workspace, workspace_bytes, stream);
}

static int cumavg_entry(const char* nm, const void* v, int dtype, int64_t N, int64_t H, int64_t T, int64_t D,
                        const int64_t* v_strides, void* out, int64_t n_slices, void* workspace, int64_t workspace_bytes,
                        sea_stream_t stream) {
  SEA_REQUIRE(v && v_strides && out, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && H > 0 && T > 0 && D > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int dslabs = (int)((D + 63) / 64);
  SEA_REQUIRE(N * H * dslabs < (1ll << 31), SEA_EUNSUPPORTED, "%s: grid too large", nm);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(N * H * dslabs)), block(1024);
  // 16-byte vector path: 16-bit data, D in {32, 64, 80, 128}, everything 16-byte aligned
  const bool vec = dtype != SEA_F32 && (D == 32 || D == 64 || D == 80 || D == 128) && v_strides[0] % 8 == 0 && v_strides[1] % 8 == 0 &&
                   v_strides[2] % 8 == 0 && (((uintptr_t)v | (uintptr_t)out) & 15) == 0;
  SEA_REQUIRE(n_slices >= 1 && n_slices <= 64 && n_slices <= T, SEA_EINVAL, "%s: n_slices %lld outside 1..min(64, T)", nm, (long long)n_slices);
  SEA_REQUIRE(n_slices == 1 || vec, SEA_EUNSUPPORTED, "%s: slices need the vector kernel (16-bit data, D in {32,64,80,128}, aligned)", nm);
  SEA_REQUIRE(n_slices == 1 || (workspace && workspace_bytes >= N * H * n_slices * D * (int64_t)sizeof(float)), SEA_EINVAL,
              "%s: workspace of %lld bytes needed", nm, (long long)(N * H * n_slices * D * (int64_t)sizeof(float)));
  if (vec) {
    const int slice_len = (int)((T + n_slices - 1) / n_slices);
    const int ns = (int)((T + slice_len - 1) / slice_len);       // no empty slice
    float* carry = reinterpret_cast<float*>(workspace);
    dim3 g2((unsigned)(N * H), (unsigned)ns);
#define SEA_CUMAVG1(TT, FGV, SO) hipLaunchKernelGGL((cumavg_vec_kernel<TT, FGV, SO>), g2, block, 0, s, (const TT*)v, (TT*)out, (int)T, \
                                                    v_strides[0], v_strides[1], v_strides[2], (int)H, slice_len, carry)
#define SEA_CUMAVG(TT, FGV) do { if (ns > 1) SEA_CUMAVG1(TT, FGV, true); SEA_CUMAVG1(TT, FGV, false); } while (0)
    if (dtype == SEA_F16) { if (D == 32) SEA_CUMAVG(__half, 4); else if (D == 64) SEA_CUMAVG(__half, 8); else if (D == 80) SEA_CUMAVG(__half, 10); else SEA_CUMAVG(__half, 16); }
    else { if (D == 32) SEA_CUMAVG(__hip_bfloat16, 4); else if (D == 64) SEA_CUMAVG(__hip_bfloat16, 8); else if (D == 80) SEA_CUMAVG(__hip_bfloat16, 10); else SEA_CUMAVG(__hip_bfloat16, 16); }
#undef SEA_CUMAVG
#undef SEA_CUMAVG1
    SEA_CHECK_LAUNCH(nm);
    return SEA_OK;
  }
  if (dtype == SEA_F32)
    hipLaunchKernelGGL((cumavg_kernel<float>), grid, block, 0, s, (const float*)v, (float*)out, (int)T, (int)D, v_strides[0],
                       v_strides[1], v_strides[2], (int)H, dslabs);
  else if (dtype == SEA_F16)
    hipLaunchKernelGGL((cumavg_kernel<__half>), grid, block, 0, s, (const __half*)v, (__half*)out, (int)T, (int)D, v_strides[0],
                       v_strides[1], v_strides[2], (int)H, dslabs);
  else
    hipLaunchKernelGGL((cumavg_kernel<__hip_bfloat16>), grid, block, 0, s, (const __hip_bfloat16*)v, (__hip_bfloat16*)out, (int)T,
                       (int)D, v_strides[0], v_strides[1], v_strides[2], (int)H, dslabs);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
