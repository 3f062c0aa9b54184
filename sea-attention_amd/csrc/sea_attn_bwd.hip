// Backward of the row-indexed sparse attention over the flat CSR, hand-written for gfx950.
//
// Reference shape: src/models/perlin_attention/masked_mm.py:169-267 (MatMultWithMask.backward: gradients of a masked
// product flow only through the kept entries) and the dense branch's autograd through
// softmax(q k^T + mask) v (attention.py:1061-1133).  Forward (sea_attn.hip), per (n, h, t) with entries j:
//     s_j = q_t . k_j        p_j = softmax_j(s)        o_t = sum_j p_j v_j
// (row scale and the average-pool mix are applied by the caller in torch, so autograd owns their gradients).
// Given dO:   dp_j = dO_t . v_j     delta = dO_t . o_t = sum_j p_j dp_j     ds_j = p_j (dp_j - delta)
//             dQ_t = sum_j ds_j k_j          dK_j += ds_j q_t          dV_j += p_j dO_t
//
// Mapping = the forward gather kernel's: one LPR-lane group per query row walks the row's entries, K and V rows are
// fetched as whole 16-byte lane fragments; p_j comes from the forward's per-entry output (`probs_out` of
// sea_sparse_attention_ex), so no score is recomputed.  dQ is owned by the row (plain store); dK and dV rows are shared
// by every query that keeps the key: fp32 global atomics (global_atomic_add_f32 executes at the memory side, ~1.3 TB/s of
// added bytes chip-wide -- MI355X_MICROARCH.md -- which bounds this first backward; a CSC pass that turns the scatter into
// a gather is the known next step).  All gradients are fp32; the caller casts.
#include "sea_attn.hpp"

namespace sea {

struct AttnBwdParams {
  const void *q, *k, *v;
  int64_t qs[3], ks[3], vs[3];
  int N, H, T_dst, T_src, D;
  const int32_t* crow;
  const int32_t* col;
  int64_t col_stride_n;
  const int32_t* head_off;
  const float* probs;      // (N, probs_stride_n) p_j of the forward (softmax only: row_scale = NULL there)
  int64_t probs_stride_n;
  const float* out;        // (N, H, T_dst, D) fp32 contiguous: o of the forward
  const float* dout;       // same layout: dL/do
  float* dq;               // (N, H, T_dst, D) fp32 contiguous
  float* dk;               // (N, H, T_src, D) fp32 contiguous, zero-initialised by the caller
  float* dv;
  int TB;
};

template <int CTRL> __device__ inline float bdpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int LPR> __device__ inline float bgroup_sum(float x) {
  if (LPR >= 2) x += bdpp<0xB1>(x);
  if (LPR >= 4) x += bdpp<0x4E>(x);
  if (LPR >= 8) x += bdpp<0x141>(x);
  if (LPR >= 16) x += bdpp<0x140>(x);
  if (LPR >= 32) x += __shfl_xor(x, 16);
  if (LPR >= 64) x += __shfl_xor(x, 32);
  return x;
}

// one LPR-lane group per query row, 64 / LPR rows per wave, 4 waves per workgroup (the forward's geometry)
template <typename T, int LPR>
__global__ __launch_bounds__(256) void sparse_attn_bwd_kernel(AttnBwdParams p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int RPW = 64 / LPR, RPB = 4 * RPW;
  int pair, tb;
  if (!map_block(p.N * p.H, p.TB, &pair, &tb)) return;
  const int n = pair / p.H, h = pair - n * p.H;
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, sub = lane - grp * LPR;
  const int t = tb * RPB + (threadIdx.x >> 6) * RPW + grp;
  const bool rowok = t < p.T_dst;
  const bool dact = sub * VEC < p.D;
  const int tt = rowok ? t : p.T_dst - 1;
  const int d0 = dact ? sub * VEC : 0;

  float qf[VEC], go[VEC], of[VEC];
  {
    uint4 qraw = make_uint4(0, 0, 0, 0);
    if (dact) qraw = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] + (int64_t)tt * p.qs[2] + d0);
    unpack16<T>(qraw, qf);
    const int64_t ro = (((int64_t)n * p.H + h) * p.T_dst + tt) * p.D + d0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      go[j] = (dact && rowok) ? p.dout[ro + j] : 0.f;
      of[j] = (dact && rowok) ? p.out[ro + j] : 0.f;
    }
  }
  float dl = 0.f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) dl = fmaf(go[j], of[j], dl);
  const float delta = bgroup_sum<LPR>(dl);                 // dO . o = sum_j p_j dp_j

  const int row_beg = p.crow[(int64_t)n * (p.T_dst + 1) + tt];
  const int32_t* ho = p.head_off + ((int64_t)n * p.T_dst + tt) * (p.H + 1);
  const int beg = row_beg + ho[h];
  const int end = rowok ? row_beg + ho[h + 1] : beg;
  const int32_t* col = p.col + n * p.col_stride_n;
  const float* pr = p.probs + n * p.probs_stride_n;
  const int hcol = h * p.T_src;
  const T* kb = reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1] + d0;
  const T* vb = reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1] + d0;
  float* dkb = p.dk + ((int64_t)n * p.H + h) * p.T_src * p.D + d0;
  float* dvb = p.dv + ((int64_t)n * p.H + h) * p.T_src * p.D + d0;

  float dqa[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) dqa[j] = 0.f;
  int zmax = end - beg;                                    // the wave walks as long as its longest row
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) zmax = max(zmax, __shfl_xor(zmax, o));
  for (int i = 0; i < zmax; ++i) {
    const int e = beg + i;
    const bool ok = e < end;
    const int key = ok ? col[e] - hcol : 0;
    const float pj = ok ? pr[e] : 0.f;
    uint4 kr = make_uint4(0, 0, 0, 0), vr = make_uint4(0, 0, 0, 0);
    if (dact && ok) {
      kr = *reinterpret_cast<const uint4*>(kb + (int64_t)key * p.ks[2]);
      vr = *reinterpret_cast<const uint4*>(vb + (int64_t)key * p.vs[2]);
    }
    float kf[VEC], vf[VEC];
    unpack16<T>(kr, kf);
    unpack16<T>(vr, vf);
    float dp = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) dp = fmaf(go[j], vf[j], dp);
    dp = bgroup_sum<LPR>(dp);
    const float ds = pj * (dp - delta);
    if (dact && ok) {
      float* dkr = dkb + (int64_t)key * p.D;
      float* dvr = dvb + (int64_t)key * p.D;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        dqa[j] = fmaf(ds, kf[j], dqa[j]);
        atomicAdd(dkr + j, ds * qf[j]);
        atomicAdd(dvr + j, pj * go[j]);
      }
    }
  }
  if (rowok && dact) {
    float* dqr = p.dq + (((int64_t)n * p.H + h) * p.T_dst + t) * p.D + d0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) dqr[j] = dqa[j];
  }
}

static int blanes_per_row(int D, int vec) {
  const int need = (D + vec - 1) / vec;
  int l = 1;
  while (l < need) l *= 2;
  return l < 4 ? 4 : l;
}

template <typename T>
static int launch_bwd(AttnBwdParams p, hipStream_t s) {
  const int lpr = blanes_per_row(p.D, Elem<T>::VEC);
  const int rpb = 4 * (64 / lpr);
  p.TB = (p.T_dst + rpb - 1) / rpb;
  const int64_t blocks = (int64_t)8 * ((p.N * p.H + 7) / 8) * p.TB;
  if (blocks >= (1ll << 31)) return SEA_EUNSUPPORTED;
  dim3 grid((unsigned)blocks), block(256);
  switch (lpr) {
    case 4: hipLaunchKernelGGL((sparse_attn_bwd_kernel<T, 4>), grid, block, 0, s, p); break;
    case 8: hipLaunchKernelGGL((sparse_attn_bwd_kernel<T, 8>), grid, block, 0, s, p); break;
    case 16: hipLaunchKernelGGL((sparse_attn_bwd_kernel<T, 16>), grid, block, 0, s, p); break;
    case 32: hipLaunchKernelGGL((sparse_attn_bwd_kernel<T, 32>), grid, block, 0, s, p); break;
    case 64: hipLaunchKernelGGL((sparse_attn_bwd_kernel<T, 64>), grid, block, 0, s, p); break;
    default: return SEA_EUNSUPPORTED;
  }
  return SEA_OK;
}

}  // namespace sea

using namespace sea;

extern "C" int sea_sparse_attention_bwd(const void* q, const void* k, const void* v, int dtype, int64_t N, int64_t H,
                                        int64_t T_dst, int64_t T_src, int64_t D, const int64_t* q_strides,
                                        const int64_t* k_strides, const int64_t* v_strides, const int32_t* crow,
                                        const int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                                        const float* probs, int64_t probs_stride_n, const float* out, const float* dout,
                                        float* dq, float* dk, float* dv, sea_stream_t stream) {
  const char* nm = "sea_sparse_attention_bwd";
  SEA_REQUIRE(q && k && v && crow && col && head_off && probs && out && dout && dq && dk && dv && q_strides && k_strides &&
                  v_strides, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && H > 0 && T_dst > 0 && T_src > 0 && D > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int vec = dtype == SEA_F32 ? 4 : 8;
  SEA_REQUIRE(D % vec == 0 && D <= 64 * vec, SEA_EUNSUPPORTED, "%s: D=%lld must be a multiple of %d and <= %d", nm,
              (long long)D, vec, 64 * vec);
  for (int i = 0; i < 3; ++i)
    SEA_REQUIRE(q_strides[i] % vec == 0 && k_strides[i] % vec == 0 && v_strides[i] % vec == 0, SEA_EUNSUPPORTED,
                "%s: row strides must be multiples of %d elements", nm, vec);
  SEA_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, SEA_EUNSUPPORTED, "%s: q/k/v must be 16-byte aligned", nm);
  AttnBwdParams p;
  p.q = q; p.k = k; p.v = v;
  for (int i = 0; i < 3; ++i) { p.qs[i] = q_strides[i]; p.ks[i] = k_strides[i]; p.vs[i] = v_strides[i]; }
  p.N = (int)N; p.H = (int)H; p.T_dst = (int)T_dst; p.T_src = (int)T_src; p.D = (int)D;
  p.crow = crow; p.col = col; p.col_stride_n = col_stride_n; p.head_off = head_off;
  p.probs = probs; p.probs_stride_n = probs_stride_n; p.out = out; p.dout = dout; p.dq = dq; p.dk = dk; p.dv = dv;
  p.TB = 0;
  int rc;
  if (dtype == SEA_F32) rc = launch_bwd<float>(p, (hipStream_t)stream);
  else if (dtype == SEA_F16) rc = launch_bwd<__half>(p, (hipStream_t)stream);
  else rc = launch_bwd<__hip_bfloat16>(p, (hipStream_t)stream);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported head size %lld", nm, (long long)D);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
