// SEA predictor MLP in one launch, hand-written for gfx950 (16-bit data, bf16/f16 MFMA).
//
// Replaces, for 16-bit tensors (reference: src/models/perlin_attention/attention.py):
//   attention_predictor_enc      Linear(3d -> 2d) + LayerNorm(2d) + GELU              (:190-196, :605-617)
//   attention_predictor_dec_row  Linear(2d -> S*W) + ChannelSplit(S)                  (:123-131, :256-262, :623)
//   cnn.lnorm1                   LayerNorm(W) on every split                          (:266)
//   attention_predictor_dec_scaler  Linear(2d -> 2) (+ the sigmoid of :1158-1166)
// i.e. three library GEMMs, two LayerNorm kernels and their five activation round trips through HBM.
//
// One wave owns 16 rows x = performer_value[n, h0..h0+15, t, :] (16 heads of one token) and keeps the whole
// chain in registers.  All products are computed TRANSPOSED, D^T = W . X^T (A operand = weights, B = activations):
// in the v_mfma_f32_16x16x32 accumulator a lane then holds 4 consecutive features (row 4*(lane/16)+r of each
// 16-feature tile) of ONE activation row (column lane%16), so
//   * a LayerNorm reduction is an in-lane sum plus two cross-lane adds (lanes l, l^16, l^32, l^48 share a row);
//   * the encoder output, rounded to 16 bit, IS the B operand of the next product -- the 8 values a lane holds of
//     tiles (2k, 2k+1) are taken as the k-th 8-element K chunk of that lane, and the decoder weights are packed
//     with the same K permutation.  No LDS transpose, no shuffle.
// The weights (A fragments, pre-packed in fragment order by the host) live in LDS for the lifetime of the
// persistent workgroup.  Outputs: the decoder rows LayerNorm'ed per split, written straight in the C8 layout of
// the conv kernels (channel = head*S + split); optionally the encoder output; the two sigmoid gates.
#include "sea_common.hpp"


namespace sea {

typedef __attribute__((ext_vector_type(4))) float mf4;
typedef __attribute__((ext_vector_type(8))) __bf16 mbf8;
typedef __attribute__((ext_vector_type(8))) _Float16 mh8;

template <typename T> __device__ inline mf4 mlp_mfma(const uint4& a, const uint4& b, mf4 c);
template <> __device__ inline mf4 mlp_mfma<__hip_bfloat16>(const uint4& a, const uint4& b, mf4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mbf8, a), __builtin_bit_cast(mbf8, b), c, 0, 0, 0);
}
template <> __device__ inline mf4 mlp_mfma<__half>(const uint4& a, const uint4& b, mf4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mh8, a), __builtin_bit_cast(mh8, b), c, 0, 0, 0);
}

// value of from_f<T>(x) as a float (the rounding a 16-bit Linear / LayerNorm output goes through)
template <typename T> __device__ inline float round16(float x) { return Elem<T>::to_f(from_f<T>(x)); }

struct MlpParams {
  const void* x;            // (N, H, T, Din), Din contiguous
  int64_t xs_n, xs_h, xs_t; // element strides
  const void* w1p;          // [KS1][NT1][64 lanes][8]: enc weight A fragments
  const void* w2p;          // [NT1/2][NT2+1][64][8]:   dec weight A fragments, K permuted; tile NT2 = the 2 scaler rows
  const float* vec;         // b1[D1] g1[D1] be1[D1] b2[D2] g2[Wd] be2[Wd] bsc[2]  (fp32)
  void* x_c8;               // (N, T, H*2/8, Wd, 8)
  void* tpred;              // optional (N, H, T, D1)
  float* row_scale;         // optional (N, H, T)
  float* avg_scale;         // optional (N, H, T)
  int N, H, T, Din, KS1;
  float eps1, eps2;
  int Wd;                   // width of one decoder split (= T_M / 4); <= 16 * (NT2 / 2), smaller only in the PADW instantiations
  int64_t xc8_n;            // element stride between batch items of x_c8 (dense: T * (H*2/8) * Wd * 8; a decode session writes the
                            // one new row of every item straight behind its CNN window: (rows + 1) * row)
  int spread;               // launches too small to fill 256 persistent workgroups (a decoding step: 2 - 16 items): item i goes to
                            // workgroup i % grid, wave (i / grid) % waves -- one CU per item -- instead of 16 items to one CU
};

// MLP_WAVES waves per (persistent) workgroup: 16 where the accumulators leave room under 128 VGPRs, else 8.
// STAGE: the (16 heads x 2 splits x Wd) result tile of an item is transposed through a wave-private LDS tile and
// leaves as 16-byte vectors (whole C8 blocks); without it every lane stores its 4-byte (split0, split1) pairs.
// W1S: the encoder weights do not fit in LDS beside the decoder's (d = 128: 12 k-steps x 16 tiles = 192 KB): they are
// STREAMED -- the workgroup's waves walk the k-steps of their items in lockstep, one k-step's 16 fragments (16 KB) at a
// time through a two-slot LDS ring, each thread carrying two 16-byte chunks of the next k-step in registers; one barrier
// per k-step (a slot is rewritten two steps later, after the barrier every reader has passed).  The 192 KB stay in L2.
// PADW: the decoder's split width is not a multiple of 16 (T_M = 96: 24; any T_M % 32 == 0 the reference's grid may ask for,
// src/main/benchmark_opt_ablation.py:160-186): each split owns HT = ceil(Wd / 16) whole tiles, its rows past Wd are zero
// weights / zero bias (host packing), they stay out of the LayerNorm statistics and are never stored.
// PACK (round 5): a wave's 16 rows are 16 CONSECUTIVE (token, head) rows of the flattened (n, t, h) order instead of heads
// 16 j .. 16 j + 15 of one token.  With H % 16 == 0 the two are the same thing (PACK off: that code is untouched); with H = 40
// a token's third tile was half empty (2.5 of 3 tiles filled), with H = 12 a quarter of the only tile -- packed, every tile
// but the launch's last is full.  H % 4 == 0 keeps every group of 4 rows inside one token, so the C8 stores (4 heads x 2
// splits = one 8-channel block) stay whole blocks; each group just has its own (n, t, block).
template <typename T, int NT1, int NT2, int MLP_WAVES, bool STAGE, bool W1S = false, bool PADW = false, bool PACK = false>
__global__ __launch_bounds__(MLP_WAVES * 64) __attribute__((amdgpu_waves_per_eu(MLP_WAVES / 4, MLP_WAVES / 4)))
void predictor_mlp_kernel(MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D1 = NT1 * 16, D2 = NT2 * 16, WdP = D2 / 2, KS2 = NT1 / 2, HT = NT2 / 2;
  const int Wd = PADW ? p.Wd : WdP;
  static_assert(NT1 % 2 == 0 && NT2 % 2 == 0, "tile counts must be even");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  // ---- LDS image: weights in fragment order + the fp32 vectors ----------------------------------------
  T* sW1 = reinterpret_cast<T*>(smem);                                   // KS1*NT1 fragments of 512 elements (W1S: 2*NT1)
  T* sW2 = sW1 + (size_t)(W1S ? 2 : p.KS1) * NT1 * 512;                  // KS2*(NT2+1) fragments
  float* sV = reinterpret_cast<float*>(sW2 + (size_t)KS2 * (NT2 + 1) * 512);
  constexpr int NVEC = 3 * D1 + D2 + 2 * WdP + 2;
  constexpr int QSTR = WdP * 16 + 16;                                    // bytes per 4-head block row of the tile (+16: banks)
  char* sTile = reinterpret_cast<char*>(sV + ((NVEC + 3) & ~3)) + (size_t)wv * (4 * QSTR);
  {
    const int n1 = p.KS1 * NT1 * 64, n2 = KS2 * (NT2 + 1) * 64;          // 16-byte chunks
    const uint4* g1 = reinterpret_cast<const uint4*>(p.w1p);
    const uint4* g2 = reinterpret_cast<const uint4*>(p.w2p);
    if (!W1S)
      for (int i = threadIdx.x; i < n1; i += MLP_WAVES * 64) reinterpret_cast<uint4*>(sW1)[i] = g1[i];
    for (int i = threadIdx.x; i < n2; i += MLP_WAVES * 64) reinterpret_cast<uint4*>(sW2)[i] = g2[i];
    for (int i = threadIdx.x; i < NVEC; i += MLP_WAVES * 64) sV[i] = p.vec[i];
  }
  __syncthreads();
  const float* sB1 = sV, *sG1 = sV + D1, *sE1 = sV + 2 * D1;
  const float* sB2 = sV + 3 * D1, *sG2 = sB2 + D2, *sE2 = sG2 + WdP, *sBsc = sE2 + WdP;
  const T* w1l = sW1 + lane * 8;                                         // + (ks*NT1 + tile)*512
  const T* w2l = sW2 + lane * 8;                                         // + (ks*(NT2+1) + tile)*512

  const int htiles = (p.H + 15) / 16;
  const int total_rows = p.N * p.T * p.H;                                // (launcher: < 2^31)
  const int nitems = PACK ? (total_rows + 15) / 16 : p.N * p.T * htiles;
  const int C8 = p.H >> 2;                                               // 8-channel blocks: (H*2)/8
  const int wstride = p.spread ? (int)gridDim.x : 1;                     // items between neighbouring waves of a workgroup
  for (int base = blockIdx.x * (p.spread ? 1 : MLP_WAVES); base < nitems; base += gridDim.x * MLP_WAVES) {
    // (workgroup-uniform trip count: with W1S every wave joins the barriers of the weight ring, item or not)
    const bool active = base + wv * wstride < nitems;
    if (!W1S && !active) continue;
    const int item = active ? base + wv * wstride : nitems - 1;
    int ht = 0, t, n, h;
    bool hok;
    if constexpr (PACK) {                                                // this lane's row of the flattened (n, t, h) order
      const int R = item * 16 + li;
      hok = active && R < total_rows;
      const int Rc = hok ? R : 0;
      const int nt_ = Rc / p.H;
      h = Rc - nt_ * p.H;
      n = nt_ / p.T;
      t = nt_ - n * p.T;
    } else {
      ht = item % htiles;
      const int nt_ = item / htiles;
      t = nt_ % p.T; n = nt_ / p.T;
      h = ht * 16 + li;
      hok = active && h < p.H;
    }
    const T* xr = reinterpret_cast<const T*>(p.x) + n * p.xs_n + (hok ? h : 0) * p.xs_h + t * p.xs_t + 8 * lg;

    // ---- product 1: enc^T = W1 . x^T ------------------------------------------------------------------
    mf4 acc1[NT1];
#pragma unroll
    for (int i = 0; i < NT1; ++i) acc1[i] = mf4{0.f, 0.f, 0.f, 0.f};
    {
      // all K fragments of the 16 rows are requested up front (one exposed memory round trip per item; the
      // registers are free here -- the accumulators of product 2 do not exist yet)
      constexpr int MAXKS = W1S ? 12 : 8;
      uint4 xf[MAXKS];
#pragma unroll
      for (int ks = 0; ks < MAXKS; ++ks) {
        xf[ks] = make_uint4(0, 0, 0, 0);
        if (ks < p.KS1 && hok && 32 * ks + 8 * lg < p.Din) xf[ks] = *reinterpret_cast<const uint4*>(xr + 32 * ks);
      }
      if constexpr (W1S) {
        constexpr int WCH = NT1 / MLP_WAVES;                              // 16-byte chunks of a k-step per thread
        static_assert(NT1 % MLP_WAVES == 0, "a k-step's fragments divide over the workgroup");
        const uint4* g1 = reinterpret_cast<const uint4*>(p.w1p);
        // a thread's chunks of the next PF k-steps ride in registers: the L2 round trip of a k-step's weights (~1 us under
        // load) is three MFMA blocks long, one step of look-ahead left every barrier waiting for it
        constexpr int PF = 3;
        uint4 wreg[PF][WCH];
#pragma unroll
        for (int f = 0; f < PF; ++f)
#pragma unroll
          for (int u = 0; u < WCH; ++u)
            wreg[f][u] = (f < p.KS1) ? g1[(size_t)f * NT1 * 64 + threadIdx.x + u * MLP_WAVES * 64] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
          if (ks < p.KS1) {                                                 // workgroup-uniform
            uint4* slot = reinterpret_cast<uint4*>(sW1) + (ks & 1) * NT1 * 64;
#pragma unroll
            for (int u = 0; u < WCH; ++u) slot[threadIdx.x + u * MLP_WAVES * 64] = wreg[ks % PF][u];
            if (ks + PF < p.KS1) {
#pragma unroll
              for (int u = 0; u < WCH; ++u)
                wreg[ks % PF][u] = g1[(size_t)(ks + PF) * NT1 * 64 + threadIdx.x + u * MLP_WAVES * 64];
            }
            __syncthreads();
            const T* wk = w1l + (size_t)(ks & 1) * NT1 * 512;
#pragma unroll
            for (int i = 0; i < NT1; ++i)
              acc1[i] = mlp_mfma<T>(*reinterpret_cast<const uint4*>(wk + i * 512), xf[ks], acc1[i]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
#pragma unroll
      for (int ks = 0; ks < MAXKS; ++ks) {
        if (ks < p.KS1) {
          const T* wk = w1l + (size_t)ks * NT1 * 512;
#pragma unroll
          for (int i = 0; i < NT1; ++i)
            acc1[i] = mlp_mfma<T>(*reinterpret_cast<const uint4*>(wk + i * 512), xf[ks], acc1[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      }
    }
    // ---- bias, round (the Linear's 16-bit output), LayerNorm(D1) + exact GELU, round -> B operand of product 2 --
    uint4 tb[KS2];                                                        // lane's 8 values of tiles (2k, 2k+1), packed
    {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NT1; ++i) {
        const float4 b = *reinterpret_cast<const float4*>(sB1 + i * 16 + lg * 4);
        acc1[i][0] = round16<T>(acc1[i][0] + b.x); acc1[i][1] = round16<T>(acc1[i][1] + b.y);
        acc1[i][2] = round16<T>(acc1[i][2] + b.z); acc1[i][3] = round16<T>(acc1[i][3] + b.w);
        s += (acc1[i][0] + acc1[i][1]) + (acc1[i][2] + acc1[i][3]);
      }
      s = xor32_sum(xor16_sum(s));
      const float mean = s * (1.0f / (float)D1);
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < NT1; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = acc1[i][r] - mean; q2 += d * d; }
      q2 = xor32_sum(xor16_sum(q2));
      const float rstd = rsqrtf(q2 * (1.0f / (float)D1) + p.eps1);
#pragma unroll
      for (int i = 0; i < NT1; ++i) {
        const float4 g = *reinterpret_cast<const float4*>(sG1 + i * 16 + lg * 4);
        const float4 e = *reinterpret_cast<const float4*>(sE1 + i * 16 + lg * 4);
        float o[4] = {(acc1[i][0] - mean) * rstd * g.x + e.x, (acc1[i][1] - mean) * rstd * g.y + e.y,
                      (acc1[i][2] - mean) * rstd * g.z + e.z, (acc1[i][3] - mean) * rstd * g.w + e.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = gelu_erf(o[r]);                     // nn.GELU(): erf form
        const uint32_t p0 = pack2<T>(o[0], o[1]), p1 = pack2<T>(o[2], o[3]);
        if (i & 1) { tb[i >> 1].z = p0; tb[i >> 1].w = p1; } else { tb[i >> 1].x = p0; tb[i >> 1].y = p1; }
        __builtin_amdgcn_sched_barrier(0);   // one tile's erf chains at a time: interleaving all 32 blows the register file
      }
      if (p.tpred && hok) {                                               // encoder output (N,H,T,D1): 8-byte pieces
        T* tp = reinterpret_cast<T*>(p.tpred) + (((int64_t)n * p.H + h) * p.T + t) * D1 + lg * 4;
#pragma unroll
        for (int i = 0; i < NT1; ++i)
          *reinterpret_cast<uint2*>(tp + i * 16) = (i & 1) ? make_uint2(tb[i >> 1].z, tb[i >> 1].w) : make_uint2(tb[i >> 1].x, tb[i >> 1].y);
      }
    }
    // ---- product 2: dec^T = W2 . enc^T (tile NT2: the two gate rows) -----------------------------------
    mf4 acc2[NT2 + 1];
#pragma unroll
    for (int i = 0; i <= NT2; ++i) acc2[i] = mf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      const T* wk = w2l + (size_t)ks * (NT2 + 1) * 512;
#pragma unroll
      for (int i = 0; i <= NT2; ++i)
        acc2[i] = mlp_mfma<T>(*reinterpret_cast<const uint4*>(wk + i * 512), tb[ks], acc2[i]);
      __builtin_amdgcn_sched_barrier(0);     // keep the fragment reads of later k-steps out of this one's registers
    }
    // ---- gates: bias, round, sigmoid (fp32) -----------------------------------------------------------
    if (lg == 0 && hok) {
      const int64_t o = ((int64_t)n * p.H + h) * p.T + t;
      const float s0 = round16<T>(acc2[NT2][0] + sBsc[0]), s1 = round16<T>(acc2[NT2][1] + sBsc[1]);
      if (p.row_scale) p.row_scale[o] = 1.0f / (1.0f + __expf(-s0));
      if (p.avg_scale) p.avg_scale[o] = 1.0f / (1.0f + __expf(-s1));
    }
    // ---- bias, round, LayerNorm(Wd) per split, round, C8 store ---------------------------------------------
    uint32_t outp[HT][4];                                                 // (split 0, split 1) pairs of column w
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < HT; ++i) {
        mf4& a = acc2[hf * HT + i];
        const float4 b = *reinterpret_cast<const float4*>(sB2 + (hf * HT + i) * 16 + lg * 4);
        a[0] = round16<T>(a[0] + b.x); a[1] = round16<T>(a[1] + b.y); a[2] = round16<T>(a[2] + b.z); a[3] = round16<T>(a[3] + b.w);
        s += (a[0] + a[1]) + (a[2] + a[3]);
      }
      s = xor32_sum(xor16_sum(s));                                        // (padded features are exact zeros here)
      const float invWd = 1.0f / (float)Wd;
      const float mean = s * invWd;
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < HT; ++i) {
        if (!PADW || i * 16 + lg * 4 < Wd) {                              // a lane's 4 features are real or padding together
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = acc2[hf * HT + i][r] - mean; q2 += d * d; }
        }
      }
      q2 = xor32_sum(xor16_sum(q2));
      const float rstd = rsqrtf(q2 * invWd + p.eps2);
#pragma unroll
      for (int i = 0; i < HT; ++i) {
        const mf4& a = acc2[hf * HT + i];
        const float4 g = *reinterpret_cast<const float4*>(sG2 + i * 16 + lg * 4);
        const float4 e = *reinterpret_cast<const float4*>(sE2 + i * 16 + lg * 4);
        const float o[4] = {(a[0] - mean) * rstd * g.x + e.x, (a[1] - mean) * rstd * g.y + e.y,
                            (a[2] - mean) * rstd * g.z + e.z, (a[3] - mean) * rstd * g.w + e.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t hv = (uint32_t)__builtin_bit_cast(unsigned short, from_f<T>(o[r]));
          outp[i][r] = hf == 0 ? hv : (outp[i][r] | (hv << 16));
        }
      }
    }
    if constexpr (STAGE) {
      // lane -> LDS: its pair of column w sits at [h/4][w][(h%4)*4 bytes]; then 16-byte rows out, coalesced
      char* tl = sTile + (li >> 2) * QSTR + (li & 3) * 4;
#pragma unroll
      for (int i = 0; i < HT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<uint32_t*>(tl + (i * 16 + lg * 4 + r) * 16) = outp[i][r];
      if constexpr (PACK) {
#pragma unroll
        for (int c0 = 0; c0 < 4 * WdP; c0 += 64) {
          const int c = c0 + lane, q = c / Wd, w = c - q * Wd;
          const bool cin = !PADW || c < 4 * Wd;
          const uint4 v = *reinterpret_cast<const uint4*>(sTile + (cin ? q * QSTR + w * 16 : 0));
          const int Rq = item * 16 + 4 * q;                              // first row of group q: its token and 4-head block
          const bool qv = active && cin && Rq < total_rows;
          const int Rc = qv ? Rq : 0;
          const int ntq = Rc / p.H, hq = Rc - ntq * p.H;
          const int nq = ntq / p.T, tq = ntq - nq * p.T;
          T* ybq = reinterpret_cast<T*>(p.x_c8) + nq * p.xc8_n + ((int64_t)tq * C8 + (hq >> 2)) * (Wd * 8);
          if (qv) *reinterpret_cast<uint4*>(ybq + (int64_t)w * 8) = v;
        }
      } else {
      T* yb = reinterpret_cast<T*>(p.x_c8) + n * p.xc8_n + ((int64_t)t * C8 + ht * 4) * (Wd * 8);
#pragma unroll
      for (int c0 = 0; c0 < 4 * WdP; c0 += 64) {
        const int c = c0 + lane, q = c / Wd, w = c - q * Wd;
        const bool cin = !PADW || c < 4 * Wd;
        const uint4 v = *reinterpret_cast<const uint4*>(sTile + (cin ? q * QSTR + w * 16 : 0));
        if (active && cin && ht * 4 + q < C8) *reinterpret_cast<uint4*>(yb + (int64_t)c * 8) = v;
      }
      }
    } else if (hok) {   // channel = 2h + split: this lane's pair is bytes [4*(h%4), +4) of block h/4, pixel w
      T* yb = reinterpret_cast<T*>(p.x_c8) + n * p.xc8_n + ((int64_t)t * C8 + (h >> 2)) * (Wd * 8) + (h & 3) * 2;
#pragma unroll
      for (int i = 0; i < HT; ++i) {
        if (!PADW || i * 16 + lg * 4 < Wd) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *reinterpret_cast<uint32_t*>(yb + (i * 16 + lg * 4 + r) * 8) = outp[i][r];
        }
      }
    }
  }
}

}  // namespace sea

using namespace sea;

template <typename T>
static int launch_mlp(const MlpParams& p_in, int nt1, int nt2, hipStream_t s) {
  MlpParams p = p_in;
  const bool pack = p.H % 16 != 0;                 // rows packed across tokens (the kernel's PACK): every tile full
  const int64_t nitems = pack ? ((int64_t)p.N * p.T * p.H + 15) / 16 : (int64_t)p.N * p.T * ((p.H + 15) / 16);
  int rc = SEA_EUNSUPPORTED;
  const bool padw = p.Wd != nt2 * 8;
#define SEA_MLP(A, B) do { if (padw) SEA_MLP_(A, B, true); else SEA_MLP_(A, B, false); } while (0)
#define SEA_MLP_(A, B, PW) do { if (pack) SEA_MLP__(A, B, PW, true); else SEA_MLP__(A, B, PW, false); } while (0)
#define SEA_MLP__(A, B, PW, PK)                                                                                          \
  do {                                                                                                              \
    constexpr int NW = (A + B <= 16) ? 16 : 8;                                                                      \
    const size_t wbytes = ((size_t)p.KS1 * A + (size_t)(A / 2) * (B + 1)) * 1024 + (size_t)((3 * A * 16 + 2 * B * 16 + 2 + 3) & ~3) * sizeof(float); \
    const size_t tile = (size_t)NW * 4 * (B * 8 * 16 + 16);                                                         \
    const bool stage = wbytes + tile <= 160 * 1024;                                                                 \
    const size_t lds = wbytes + (stage ? tile : 0);                                                                 \
    if (lds > 160 * 1024) break;                                                                                    \
    int64_t blocks = (nitems + NW - 1) / NW;                                                                        \
    if (blocks > 256) blocks = 256;               /* persistent: one workgroup per CU keeps the weights in LDS */   \
    p.spread = nitems < 256 * NW;                 /* fewer: one item per CU first (MlpParams::spread) */           \
    if (p.spread) blocks = nitems < 256 ? nitems : 256;                                                             \
    static DevOnce once;                                                                                            \
    if (once.first()) {                                                                                             \
      SEA_MAX_LDS((predictor_mlp_kernel<T, A, B, NW, true, false, PW, PK>), 160 * 1024);                            \
      SEA_MAX_LDS((predictor_mlp_kernel<T, A, B, NW, false, false, PW, PK>), 160 * 1024);                           \
    }                                                                                                               \
    if (stage) hipLaunchKernelGGL((predictor_mlp_kernel<T, A, B, NW, true, false, PW, PK>), dim3((unsigned)blocks), dim3(NW * 64), lds, s, p);  \
    else hipLaunchKernelGGL((predictor_mlp_kernel<T, A, B, NW, false, false, PW, PK>), dim3((unsigned)blocks), dim3(NW * 64), lds, s, p);       \
    rc = SEA_OK;                                                                                                    \
  } while (0)
  if (nt1 == 16 && nt2 == 8 && !padw) {                               // d = 128: encoder weights streamed through a two-slot LDS ring
    constexpr int A = 16, B = 8, NW = 8;
    const size_t wbytes = ((size_t)2 * A + (size_t)(A / 2) * (B + 1)) * 1024 + (size_t)((3 * A * 16 + 2 * B * 16 + 2 + 3) & ~3) * sizeof(float);
    const size_t lds = wbytes + (size_t)NW * 4 * (B * 8 * 16 + 16);
    if (lds <= 160 * 1024 && p.KS1 <= 12) {
      int64_t blocks = (nitems + NW - 1) / NW;
      if (blocks > 256) blocks = 256;
      p.spread = nitems < 256 * NW;
      if (p.spread) blocks = nitems < 256 ? nitems : 256;
      static DevOnce once;
      if (once.first()) {
        SEA_MAX_LDS((predictor_mlp_kernel<T, A, B, NW, true, true, false, false>), 160 * 1024);
        SEA_MAX_LDS((predictor_mlp_kernel<T, A, B, NW, true, true, false, true>), 160 * 1024);
      }
      if (pack) hipLaunchKernelGGL((predictor_mlp_kernel<T, A, B, NW, true, true, false, true>), dim3((unsigned)blocks), dim3(NW * 64), lds, s, p);
      else hipLaunchKernelGGL((predictor_mlp_kernel<T, A, B, NW, true, true, false, false>), dim3((unsigned)blocks), dim3(NW * 64), lds, s, p);
      rc = SEA_OK;
    }
  } else if (nt1 == 8 && nt2 == 8) SEA_MLP(8, 8);
  else if (nt1 == 8 && nt2 == 2) SEA_MLP(8, 2);       // d = 64 at every predictor length T_M % 32 == 0 up to 512 (T_M / 4 = 8 .. 128
  else if (nt1 == 8 && nt2 == 4) SEA_MLP(8, 4);       //  pixels per split: 1 .. 8 tiles, the last one partly padding)
  else if (nt1 == 8 && nt2 == 6) SEA_MLP(8, 6);
  else if (nt1 == 8 && nt2 == 10) SEA_MLP(8, 10);
  else if (nt1 == 8 && nt2 == 12) SEA_MLP(8, 12);
  else if (nt1 == 8 && nt2 == 14) SEA_MLP(8, 14);
  else if (nt1 == 8 && nt2 == 16) SEA_MLP(8, 16);
  else if (nt1 == 10 && nt2 == 8 && !padw) SEA_MLP_(10, 8, false);
#undef SEA_MLP__
#undef SEA_MLP_
#undef SEA_MLP
  return rc;
}

extern "C" int sea_predictor_mlp(const void* x, int dtype, int64_t N, int64_t H, int64_t T, int64_t Din, const int64_t* x_strides,
                                 int64_t D1, int64_t D2, const void* w1_packed, const void* w2_packed, const float* vectors,
                                 float eps1, float eps2, void* x_c8, int64_t x_c8_stride_n, void* tpred, float* row_scale,
                                 float* avg_scale, sea_stream_t stream) {
  const char* nm = "sea_predictor_mlp";
  SEA_REQUIRE(x && x_strides && w1_packed && w2_packed && vectors && x_c8, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16, SEA_EUNSUPPORTED, "%s: 16-bit data only (dtype %d)", nm, dtype);
  SEA_REQUIRE(N > 0 && H > 0 && T > 0 && Din > 0 && D1 > 0 && D2 > 0, SEA_EINVAL, "%s: bad shape", nm);
  SEA_REQUIRE(Din % 8 == 0 && Din <= (D1 == 256 ? 384 : 256) && D1 % 32 == 0 && D2 % 16 == 0 && H % 4 == 0, SEA_EUNSUPPORTED,
              "%s: needs Din %% 8 == 0, Din <= 256 (384 with D1 = 256), D1 %% 32 == 0, D2 %% 16 == 0, H %% 4 == 0", nm);
  SEA_REQUIRE(x_strides[0] % 8 == 0 && x_strides[1] % 8 == 0 && x_strides[2] % 8 == 0 &&
                  (((uintptr_t)x | (uintptr_t)w1_packed | (uintptr_t)w2_packed | (uintptr_t)x_c8 | (uintptr_t)tpred) & 15) == 0,
              SEA_EUNSUPPORTED, "%s: 16-byte alignment", nm);
  SEA_REQUIRE(N * T * ((H + 15) / 16) * 16 < (1ll << 31), SEA_EUNSUPPORTED, "%s: too many rows", nm);
  MlpParams p;
  p.x = x; p.xs_n = x_strides[0]; p.xs_h = x_strides[1]; p.xs_t = x_strides[2];
  p.w1p = w1_packed; p.w2p = w2_packed; p.vec = vectors;
  p.x_c8 = x_c8; p.tpred = tpred; p.row_scale = row_scale; p.avg_scale = avg_scale;
  p.N = (int)N; p.H = (int)H; p.T = (int)T; p.Din = (int)Din; p.KS1 = (int)((Din + 31) / 32);
  p.eps1 = eps1; p.eps2 = eps2;
  p.Wd = (int)(D2 / 2);
  const int64_t dense_n = T * (H * 2 / 8) * (D2 / 2) * 8;
  SEA_REQUIRE(x_c8_stride_n == 0 || (x_c8_stride_n >= dense_n && x_c8_stride_n % 8 == 0), SEA_EINVAL,
              "%s: x_c8_stride_n must be 0 (dense) or >= %lld elements in whole 16-byte blocks", nm, (long long)dense_n);
  p.xc8_n = x_c8_stride_n ? x_c8_stride_n : dense_n;
  const int nt2 = 2 * (int)((D2 / 2 + 15) / 16);               // tiles of the padded decoder: each split owns ceil(Wd / 16)
  hipStream_t s = (hipStream_t)stream;
  const int rc = dtype == SEA_BF16 ? launch_mlp<__hip_bfloat16>(p, (int)(D1 / 16), nt2, s)
                                   : launch_mlp<__half>(p, (int)(D1 / 16), nt2, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported (D1=%lld, D2=%lld) combination", nm, (long long)D1, (long long)D2);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
