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

// ---- gather form (round 3): dK / dV without float atomics -----------------------------------------------------------------
// The scatter of the row pass above (every (row, key) entry adds D floats to dK_key and dV_key with memory-side atomics,
// ~1.3 TB/s chip-wide) becomes a gather over the TRANSPOSED pattern:
//   1. csc_count_kernel    entries per (n, head * T_src + key): one int32 atomic per entry (L2 integer atomics);
//   2. csc_scan_local / _finish_kernel  exclusive scan -> start of every key's list (`cptr`) + a working copy (`cursor`);
//   3. bwd_rows_kernel     one lane group per query row (the forward's geometry): dp_j = dO . v_j, ds_j = p_j (dp_j - delta),
//                          dQ = sum_j ds_j k_j (plain store), and a 16-byte record {t, p_j, ds_j} dropped into the key's
//                          list at a slot taken from `cursor` (one returning int32 atomic per entry);
//   4. bwd_cols_kernel     one lane group per (n, h, key): walks the key's records, gathers q_t (input dtype) and dO_t
//                          (fp32) rows, dK = sum ds q, dV = sum p dO -- plain stores, every row written (no zero-fill).
// The order of a key's records is the order the row pass reached them: sums differ between runs in fp32 rounding only,
// like the atomic form.
struct BwdRec { int32_t t; float p; float ds; int32_t pad; };

struct BwdGatherParams {
  AttnBwdParams a;
  int32_t* cptr;       // (N, C + 1), C = H * T_src
  int32_t* cursor;     // (N, C + 1)
  BwdRec* recs;        // (N, rec_stride_n)
  int64_t rec_stride_n;
  int64_t C;
};

__global__ __launch_bounds__(256) void csc_count_kernel(const int32_t* __restrict__ crow, const int32_t* __restrict__ col,
                                                        int64_t col_stride_n, int T_dst, int64_t C, int32_t* __restrict__ cnt) {
  const int n = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int z = crow[(int64_t)n * (T_dst + 1) + T_dst];
  if (e >= z) return;
  atomicAdd(cnt + (int64_t)n * (C + 1) + col[n * col_stride_n + e], 1);
}

// cnt[n][0..C) -> exclusive scan in place (cnt[n][C] = total), copied to cursor: two launches over (chunk, batch item)
// workgroups.  Launch 1 scans its chunk locally and leaves the chunk total in `tot`; launch 2 adds the totals of the chunks
// before it and writes both copies.  (The first version walked all C = H * T_src = 131072 counters of an item in ONE
// workgroup, 128 passes of three barriers: 170 us alone on the critical path of the backward at one sequence per GPU --
// ADVICE r3; a thread-owns-16-consecutive-counters form of that single workgroup measured the same 170 us: uncoalesced.)
constexpr int CSC_CHUNKS = 32;
__global__ __launch_bounds__(1024) void csc_scan_local_kernel(int32_t* __restrict__ cptr, int64_t C, int64_t L, int32_t* __restrict__ tot) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const int ch = blockIdx.x, n = blockIdx.y;
  int32_t* c = cptr + (int64_t)n * (C + 1);
  const int64_t lo = (int64_t)ch * L, hi = min(C, lo + L);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int64_t i0 = lo; i0 < hi; i0 += 1024) {
    const int64_t i = i0 + threadIdx.x;
    const int x = i < hi ? c[i] : 0;
    const int inc = wave_incl_scan(x);
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) woff += (w < wv) ? s_wave[w] : 0;
    const int carry = s_carry;
    if (i < hi) c[i] = carry + woff + inc - x;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) tot[n * CSC_CHUNKS + ch] = s_carry;
}

__global__ __launch_bounds__(1024) void csc_scan_finish_kernel(int32_t* __restrict__ cptr, int32_t* __restrict__ cursor, int64_t C, int64_t L,
                                                              const int32_t* __restrict__ tot) {
  const int ch = blockIdx.x, n = blockIdx.y;
  int32_t* c = cptr + (int64_t)n * (C + 1);
  int32_t* u = cursor + (int64_t)n * (C + 1);
  int base = 0;
  for (int j = 0; j < ch; ++j) base += tot[n * CSC_CHUNKS + j];
  const int64_t lo = (int64_t)ch * L, hi = min(C, lo + L);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) {
    const int v = c[i] + base;
    c[i] = v; u[i] = v;
  }
  if (ch == CSC_CHUNKS - 1 && threadIdx.x == 0) {
    const int all = base + tot[n * CSC_CHUNKS + ch];
    c[C] = all; u[C] = all;
  }
}

template <typename T, int LPR, int U, int NWB>
__global__ __launch_bounds__(NWB * 64) void sparse_attn_bwd_rows_kernel(BwdGatherParams g) {
  const AttnBwdParams& p = g.a;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int RPW = 64 / LPR, RPB = NWB * RPW;
  int pair, tb;
  if (!map_block(p.N * p.H, p.TB, &pair, &tb)) return;
  const int n = pair / p.H;
  const int h = (pair - n * p.H + n) % p.H;                // heads rotated over the XCDs per item, as in the forward
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, sub = lane - grp * LPR;
  // the block's rows dealt to the lane groups by length (sea_attn.hpp: rows_by_length), as in the forward
  const int gi = (int)(threadIdx.x >> 6) * RPW + grp;
  int len = -1;
  {
    const int tn = tb * RPB + gi;
    if (tn < p.T_dst) {
      const int32_t* hon = p.head_off + ((int64_t)n * p.T_dst + tn) * (p.H + 1);
      len = hon[h + 1] - hon[h];
    }
  }
  bool rowok;
  const int t = tb * RPB + rows_by_length<LPR, RPB>(len, gi, sub, &rowok);
  const bool dact = sub * VEC < p.D;
  const int tt = t < p.T_dst ? t : p.T_dst - 1;
  const int d0 = dact ? sub * VEC : 0;

  float go[VEC];
  float dl = 0.f;
  {
    const int64_t ro = (((int64_t)n * p.H + h) * p.T_dst + tt) * p.D + d0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      go[j] = (dact && rowok) ? p.dout[ro + j] : 0.f;
      const float of = (dact && rowok) ? p.out[ro + j] : 0.f;
      dl = fmaf(go[j], of, dl);
    }
  }
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
  int32_t* cursor = g.cursor + (int64_t)n * (g.C + 1);
  BwdRec* recs = g.recs + n * g.rec_stride_n;

  float dqa[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) dqa[j] = 0.f;
  int zmax = end - beg;                                    // the wave walks as long as its longest row
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) zmax = max(zmax, __shfl_xor(zmax, o));
  const int grp_lane0 = (lane - sub) << 2;
  for (int i0 = 0; i0 < zmax; i0 += LPR) {
    // lane `sub` owns entry i0 + sub of the row: its column, its probability, and -- after the walk -- its ds
    const int e_mine = beg + i0 + sub;
    const bool mine = e_mine < end;
    // lanes past the row's end re-read the row's own last entry with p = 0 (a kept key: finite rows), empty rows key 0
    const int c_mine = mine ? col[e_mine] : (end > beg ? col[end - 1] : hcol);   // head * T_src + key
    const float p_mine = mine ? pr[e_mine] : 0.f;
    float ds_mine = 0.f;
#pragma unroll
    for (int u0 = 0; u0 < LPR; u0 += U) {
      if (i0 + u0 < zmax) {                                // wave-uniform
        uint4 kr[U], vr[U];
        float pu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int key = __builtin_amdgcn_ds_bpermute(grp_lane0 + ((u0 + u) << 2), c_mine) - hcol;
          pu[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(grp_lane0 + ((u0 + u) << 2), __float_as_int(p_mine)));
          kr[u] = make_uint4(0, 0, 0, 0);
          vr[u] = make_uint4(0, 0, 0, 0);
          if (dact) {
            kr[u] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * p.ks[2]);
            vr[u] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * p.vs[2]);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float kf[VEC], vf[VEC];
          unpack16<T>(kr[u], kf);
          unpack16<T>(vr[u], vf);
          float dp = 0.f;
#pragma unroll
          for (int j = 0; j < VEC; ++j) dp = fmaf(go[j], vf[j], dp);
          dp = bgroup_sum<LPR>(dp);
          const float ds = (pu[u] != 0.f) ? pu[u] * (dp - delta) : 0.f;   // p = 0: padding, or an entry that underflowed (a non-finite V row there must not reach dQ)
#pragma unroll
          for (int j = 0; j < VEC; ++j) dqa[j] = fmaf(ds, kf[j], dqa[j]);
          if (sub == u0 + u) ds_mine = ds;
        }
      }
    }
    if (mine) {
      const int pos = atomicAdd(cursor + c_mine, 1);
      BwdRec r;
      r.t = t; r.p = p_mine; r.ds = ds_mine; r.pad = 0;
      *reinterpret_cast<uint4*>(recs + pos) = __builtin_bit_cast(uint4, r);
    }
  }
  if (rowok && dact) {
    float* dqr = p.dq + (((int64_t)n * p.H + h) * p.T_dst + t) * p.D + d0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) dqr[j] = dqa[j];
  }
}

// one LPR-lane group per (n, h, key): dK_key = sum over the key's records of ds * q_t, dV_key = sum of p * dO_t
template <typename T, int LPR, int U, int NWB>
__global__ __launch_bounds__(NWB * 64) void sparse_attn_bwd_cols_kernel(BwdGatherParams g) {
  const AttnBwdParams& p = g.a;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int RPW = 64 / LPR, RPB = NWB * RPW;
  int pair, tb;
  if (!map_block(p.N * p.H, p.TB, &pair, &tb)) return;
  const int n = pair / p.H;
  const int h = (pair - n * p.H + n) % p.H;
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, sub = lane - grp * LPR;
  // the block's keys dealt to the lane groups by the length of their record lists (same lockstep argument as for rows)
  const int32_t* cptr = g.cptr + (int64_t)n * (g.C + 1) + (int64_t)h * p.T_src;
  const int gi = (int)(threadIdx.x >> 6) * RPW + grp;
  int len = -1;
  {
    const int kn = tb * RPB + gi;
    if (kn < p.T_src) len = cptr[kn + 1] - cptr[kn];
  }
  bool keyok;
  const int key = tb * RPB + rows_by_length<LPR, RPB>(len, gi, sub, &keyok);
  const bool dact = sub * VEC < p.D;
  const int d0 = dact ? sub * VEC : 0;
  const int beg = keyok ? cptr[key] : 0;
  const int end = keyok ? cptr[key + 1] : 0;
  const BwdRec* recs = g.recs + n * g.rec_stride_n;
  const T* qb = reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] + d0;
  const float* gob = p.dout + ((int64_t)n * p.H + h) * p.T_dst * p.D + d0;

  float dk[VEC], dv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { dk[j] = 0.f; dv[j] = 0.f; }
  int zmax = end - beg;
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) zmax = max(zmax, __shfl_xor(zmax, o));
  const int grp_lane0 = (lane - sub) << 2;
  for (int i0 = 0; i0 < zmax; i0 += LPR) {
    const int e_mine = beg + i0 + sub;
    // padding slots (lists shorter than the wave's longest) carry zero weights and point at the key's OWN last record: the
    // q / dO rows they gather are rows that contribute to this key anyway -- never an unrelated row 0 whose Inf / NaN
    // (an fp16 loss-scale overflow step) would reach this key as 0 * Inf; keys without records: zeroed below
    uint4 rm = make_uint4(0, 0, 0, 0);
    if (end > beg) rm = *reinterpret_cast<const uint4*>(recs + (e_mine < end ? e_mine : end - 1));
    if (e_mine >= end) { rm.y = 0u; rm.z = 0u; }
#pragma unroll
    for (int u0 = 0; u0 < LPR; u0 += U) {
      if (i0 + u0 < zmax) {                                // wave-uniform
        uint4 qr[U];
        float gf[U][VEC], pu[U], dsu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int src = grp_lane0 + ((u0 + u) << 2);
          const int t = __builtin_amdgcn_ds_bpermute(src, (int)rm.x);
          pu[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, (int)rm.y));
          dsu[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, (int)rm.z));
          qr[u] = make_uint4(0, 0, 0, 0);
          if (dact) qr[u] = *reinterpret_cast<const uint4*>(qb + (int64_t)t * p.qs[2]);
          const float* gr = gob + (int64_t)t * p.D;
#pragma unroll
          for (int j4 = 0; j4 < VEC; j4 += 4) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dact) x = *reinterpret_cast<const float4*>(gr + j4);
            gf[u][j4] = x.x; gf[u][j4 + 1] = x.y; gf[u][j4 + 2] = x.z; gf[u][j4 + 3] = x.w;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float qf[VEC];
          unpack16<T>(qr[u], qf);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            dk[j] = fmaf(dsu[u], qf[j], dk[j]);
            dv[j] = fmaf(pu[u], gf[u][j], dv[j]);
          }
        }
      }
    }
  }
  if (end == beg) {                                        // no query keeps this key: exact zeros, whatever the padding gathered
#pragma unroll
    for (int j = 0; j < VEC; ++j) { dk[j] = 0.f; dv[j] = 0.f; }
  }
  if (keyok && dact) {
    const int64_t ro = (((int64_t)n * p.H + h) * p.T_src + key) * p.D + d0;
#pragma unroll
    for (int j4 = 0; j4 < VEC; j4 += 4) {
      *reinterpret_cast<float4*>(p.dk + ro + j4) = make_float4(dk[j4], dk[j4 + 1], dk[j4 + 2], dk[j4 + 3]);
      *reinterpret_cast<float4*>(p.dv + ro + j4) = make_float4(dv[j4], dv[j4 + 1], dv[j4 + 2], dv[j4 + 3]);
    }
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

template <typename T>
static int launch_bwd_gather(BwdGatherParams g, hipStream_t s) {
  AttnBwdParams& p = g.a;
  const int lpr = blanes_per_row(p.D, Elem<T>::VEC);
  if (lpr > 16) return SEA_EUNSUPPORTED;                    // fp32 d >= 128: the atomic form serves those
  const int nwb = lpr == 4 ? 4 : 8;                          // rows (keys) per block, sorted by length inside the kernels
  const int rpb = nwb * (64 / lpr);
  const int NH8 = 8 * ((p.N * p.H + 7) / 8);
  if (hipMemsetAsync(g.cptr, 0, (size_t)p.N * (g.C + 1) * sizeof(int32_t), s) != hipSuccess) return SEA_ELAUNCH;
  const int64_t zb = (p.col_stride_n + 255) / 256;
  if (zb >= (1ll << 31) || p.N > 65535) return SEA_EUNSUPPORTED;
  hipLaunchKernelGGL(csc_count_kernel, dim3((unsigned)zb, (unsigned)p.N), dim3(256), 0, s, p.crow, p.col, p.col_stride_n, p.T_dst,
                     g.C, g.cptr);
  {
    // chunk totals live at the start of the record area, which the row pass fills only after the scan
    int32_t* tot = reinterpret_cast<int32_t*>(g.recs);
    const int64_t L = (((g.C + CSC_CHUNKS - 1) / CSC_CHUNKS) + 1023) / 1024 * 1024;
    hipLaunchKernelGGL(csc_scan_local_kernel, dim3(CSC_CHUNKS, (unsigned)p.N), dim3(1024), 0, s, g.cptr, g.C, L, tot);
    hipLaunchKernelGGL(csc_scan_finish_kernel, dim3(CSC_CHUNKS, (unsigned)p.N), dim3(1024), 0, s, g.cptr, g.cursor, g.C, L, tot);
  }
  p.TB = (p.T_dst + rpb - 1) / rpb;
  int64_t blocks = (int64_t)NH8 * p.TB;
  if (blocks >= (1ll << 31)) return SEA_EUNSUPPORTED;
  switch (lpr) {
    case 4: hipLaunchKernelGGL((sparse_attn_bwd_rows_kernel<T, 4, 4, 4>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    case 8: hipLaunchKernelGGL((sparse_attn_bwd_rows_kernel<T, 8, 4, 8>), dim3((unsigned)blocks), dim3(512), 0, s, g); break;
    default: hipLaunchKernelGGL((sparse_attn_bwd_rows_kernel<T, 16, 4, 8>), dim3((unsigned)blocks), dim3(512), 0, s, g); break;
  }
  p.TB = (p.T_src + rpb - 1) / rpb;
  blocks = (int64_t)NH8 * p.TB;
  if (blocks >= (1ll << 31)) return SEA_EUNSUPPORTED;
  switch (lpr) {
    case 4: hipLaunchKernelGGL((sparse_attn_bwd_cols_kernel<T, 4, 4, 4>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    // (8-lane rows: eight records in flight per lane group -- this pass lives on loads in flight, not on occupancy: U = 1 / 2 /
    //  4 / 8 at 8 / 7 / 4 / 3 waves per SIMD: train step 2.14 / 1.96 / 1.70 / 1.66 ms, one OPT-1.3B sequence; same sums in the same order)
    case 8: hipLaunchKernelGGL((sparse_attn_bwd_cols_kernel<T, 8, 8, 8>), dim3((unsigned)blocks), dim3(512), 0, s, g); break;
    default: hipLaunchKernelGGL((sparse_attn_bwd_cols_kernel<T, 16, 4, 8>), dim3((unsigned)blocks), dim3(512), 0, s, g); break;
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

static int64_t bwd_ws_ints(int64_t N, int64_t H, int64_t T_src) { return ((N * (H * T_src + 1) + 3) / 4) * 4; }   // 16-byte multiples

extern "C" int64_t sea_sparse_attention_bwd_workspace_bytes(int64_t N, int64_t H, int64_t T_src, int64_t col_stride_n) {
  if (N <= 0 || H <= 0 || T_src <= 0 || col_stride_n <= 0) return 0;
  const int64_t recs = N * col_stride_n * (int64_t)sizeof(BwdRec);
  const int64_t tot = N * 32 * 4;                              // the scan's chunk totals share the record area
  return 2 * bwd_ws_ints(N, H, T_src) * 4 + (recs > tot ? recs : tot);
}

extern "C" int sea_sparse_attention_bwd_gather(const void* q, const void* k, const void* v, int dtype, int64_t N, int64_t H,
                                               int64_t T_dst, int64_t T_src, int64_t D, const int64_t* q_strides,
                                               const int64_t* k_strides, const int64_t* v_strides, const int32_t* crow,
                                               const int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                                               const float* probs, int64_t probs_stride_n, const float* out,
                                               const float* dout, float* dq, float* dk, float* dv, void* workspace,
                                               int64_t workspace_bytes, sea_stream_t stream) {
  const char* nm = "sea_sparse_attention_bwd_gather";
  SEA_REQUIRE(q && k && v && crow && col && head_off && probs && out && dout && dq && dk && dv && q_strides && k_strides &&
                  v_strides && workspace, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && H > 0 && T_dst > 0 && T_src > 0 && D > 0 && col_stride_n > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int vec = dtype == SEA_F32 ? 4 : 8;
  SEA_REQUIRE(D % vec == 0 && D <= 16 * vec, SEA_EUNSUPPORTED, "%s: D=%lld must be a multiple of %d and <= %d", nm,
              (long long)D, vec, 16 * vec);
  for (int i = 0; i < 3; ++i)
    SEA_REQUIRE(q_strides[i] % vec == 0 && k_strides[i] % vec == 0 && v_strides[i] % vec == 0, SEA_EUNSUPPORTED,
                "%s: row strides must be multiples of %d elements", nm, vec);
  SEA_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)workspace | (uintptr_t)dout | (uintptr_t)dk |
                (uintptr_t)dv) & 15) == 0, SEA_EUNSUPPORTED, "%s: q/k/v/dout/dk/dv/workspace must be 16-byte aligned", nm);
  SEA_REQUIRE(workspace_bytes >= sea_sparse_attention_bwd_workspace_bytes(N, H, T_src, col_stride_n), SEA_EINVAL,
              "%s: workspace of %lld bytes, need %lld", nm, (long long)workspace_bytes,
              (long long)sea_sparse_attention_bwd_workspace_bytes(N, H, T_src, col_stride_n));
  SEA_REQUIRE(H * T_src < (1ll << 31) && N * (H * T_src + 1) < (1ll << 40), SEA_EUNSUPPORTED, "%s: column space too large", nm);
  BwdGatherParams g;
  AttnBwdParams& p = g.a;
  p.q = q; p.k = k; p.v = v;
  for (int i = 0; i < 3; ++i) { p.qs[i] = q_strides[i]; p.ks[i] = k_strides[i]; p.vs[i] = v_strides[i]; }
  p.N = (int)N; p.H = (int)H; p.T_dst = (int)T_dst; p.T_src = (int)T_src; p.D = (int)D;
  p.crow = crow; p.col = col; p.col_stride_n = col_stride_n; p.head_off = head_off;
  p.probs = probs; p.probs_stride_n = probs_stride_n; p.out = out; p.dout = dout; p.dq = dq; p.dk = dk; p.dv = dv;
  p.TB = 0;
  g.C = H * T_src;
  g.cptr = reinterpret_cast<int32_t*>(workspace);
  g.cursor = g.cptr + bwd_ws_ints(N, H, T_src);
  g.recs = reinterpret_cast<BwdRec*>(g.cursor + bwd_ws_ints(N, H, T_src));
  g.rec_stride_n = col_stride_n;
  int rc;
  if (dtype == SEA_F32) rc = launch_bwd_gather<float>(g, (hipStream_t)stream);
  else if (dtype == SEA_F16) rc = launch_bwd_gather<__half>(g, (hipStream_t)stream);
  else rc = launch_bwd_gather<__hip_bfloat16>(g, (hipStream_t)stream);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported shape (rows wider than 16 lanes take sea_sparse_attention_bwd)", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
