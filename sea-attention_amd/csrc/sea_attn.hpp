// Shared declarations of the fused sparse-attention kernels (sea_attn.hip: gather kernels; sea_attn_tile.hip: MFMA tile kernel).
#pragma once
#include "sea_common.hpp"

namespace sea {

struct AttnParams {
  const void *q, *k, *v;
  int64_t qs[3], ks[3], vs[3];
  int N, H, T_dst, T_src, D;
  const int32_t* crow;
  const int32_t* col;
  int64_t col_stride_n;
  const int32_t* head_off;
  const float* row_scale;
  const void* avg;
  int64_t as[3];
  const float* mix;
  void* out;
  int64_t os[3];
  float* probs;            // optional (N, probs_stride_n): rs * softmax per entry (gather kernels only)
  int64_t probs_stride_n;
  // kernel choice by plan (sea_attention_plan): `sel` = one byte per (n, h, 16-row block), 1 where the block's entries per
  // staged 16-key tile favour the MFMA tile kernel, followed by the int32 count of such blocks.  BOTH kernels are launched
  // over the same rows and read the count: when the tile blocks' share exceeds sel_num / sel_den the tile kernel runs the
  // whole launch and the gather kernel's workgroups exit at once, otherwise the other way round (round 3: since the gather
  // kernels deal rows by length, a launch split per block between the two kernels is slower than the better of them alone)
  const uint8_t* sel;
  int sel_want, TB16;
  const int32_t* sel_count;
  int sel_total, sel_num, sel_den;
  int TB;  // row blocks per (n, h): ceil(T_dst / 4)
  // fused interpolation (sea_sparse_attention_fused): the gather kernel expands the kept pixels itself -- `col` holds no
  // columns yet, the kernel WRITES them (col_w) while it walks them.  bits (N, T_dst, W) from the selection launch.
  const uint32_t* bits;
  int32_t* col_w;
  int T_m, W, max_k, is_causal, fuse_cap;
  // fused interpolation: 0 = the expanded columns stay in LDS (blocks whose lists do not fit still go through `col`): the CSR's
  // column array is NOT an output of this launch -- the caller's handle keeps its columns pending and sea_csr_emit writes
  // them if anybody ever reads them (round 4: the copy-out was 266 MB and ~50 us of the headline launch for no reader)
  int write_cols;
  // decode form of the fused interpolation (sea_sparse_attention_fused_at): the row widths follow *t_src_dev (the sequence
  // length, in device memory: a graph-replayed step has static arguments) while T_src above stays the FIXED capacity the
  // column ids are encoded with (head * T_src + key) and the K / V caches are laid out for.  NULL: widths follow T_src.
  const int32_t* t_src_dev;
};

template <typename TO, int VEC> __device__ inline void store_frag(TO* dst, const float* f);
template <> __device__ inline void store_frag<float, 4>(float* dst, const float* f) {
  *reinterpret_cast<float4*>(dst) = make_float4(f[0], f[1], f[2], f[3]);
}
template <> __device__ inline void store_frag<float, 8>(float* dst, const float* f) {
  *reinterpret_cast<float4*>(dst) = make_float4(f[0], f[1], f[2], f[3]);
  *reinterpret_cast<float4*>(dst + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
template <> __device__ inline void store_frag<__hip_bfloat16, 8>(__hip_bfloat16* dst, const float* f) {
  union { uint4 u; __hip_bfloat16 h[8]; } r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.h[i] = __float2bfloat16(f[i]);
  *reinterpret_cast<uint4*>(dst) = r.u;
}
template <> __device__ inline void store_frag<__half, 8>(__half* dst, const float* f) {
  union { uint4 u; __half h[8]; } r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.h[i] = __float2half(f[i]);
  *reinterpret_cast<uint4*>(dst) = r.u;
}
template <> __device__ inline void store_frag<__hip_bfloat16, 4>(__hip_bfloat16* dst, const float* f) {
  union { uint2 u; __hip_bfloat16 h[4]; } r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.h[i] = __float2bfloat16(f[i]);
  *reinterpret_cast<uint2*>(dst) = r.u;
}
template <> __device__ inline void store_frag<__half, 4>(__half* dst, const float* f) {
  union { uint2 u; __half h[4]; } r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.h[i] = __float2half(f[i]);
  *reinterpret_cast<uint2*>(dst) = r.u;
}

// (n, h, row-block) from blockIdx, XCD-aware: blocks b and b+8 share an XCD (observed dispatch
// order; only affects L2 locality).  Returns false if this block has no work.
__device__ inline bool map_block(int NH, int TB, int* pair, int* tb) {
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int pl = slot / TB;
  const int full = NH >> 3;                      // groups of 8 pairs: pair 8 pl + x lives on XCD x (its K / V stay in that L2)
  if (pl < full) {
    *tb = slot - pl * TB;
    *pair = pl * 8 + xcd;
    return true;
  }
  // the last, partial group (NH % 8 = r pairs; round 5): its r TB row blocks are dealt round-robin to ALL eight XCDs instead
  // of TB blocks to each of r XCDs while 8 - r sit idle -- 12 pairs (OPT-125m, one sequence: the long-context and grid legs)
  // were 2 + 2 + 2 + 2 + 1 + 1 + 1 + 1 pairs per XCD, i.e. the launch lasted as long as 16 pairs
  const int r = NH - full * 8;
  const int item = (slot - pl * TB) * 8 + xcd;
  if (pl > full || item >= r * TB) return false;
  const int pr = item / TB;
  *tb = item - pr * TB;
  *pair = full * 8 + pr;
  return true;
}

// true when the plan hands this launch to the OTHER kernel (uniform over the whole grid: every workgroup returns at once)
__device__ inline bool kernel_is_idle(const AttnParams& p) {
  if (p.sel == nullptr) return false;
  const bool tile_runs = (int64_t)p.sel_count[0] * p.sel_den > (int64_t)p.sel_total * p.sel_num;
  return tile_runs ? (p.sel_want != 1) : (p.sel_want != 0);
}

// ---- rows of a block dealt to the lane groups BY LENGTH ---------------------------------------------------------------
// The grouped top-k pools all heads of a query row, so the entries one head gets vary a lot from row to row (at the
// headline shape: mean 64, standard deviation ~32 -- four pixels of 16 keys each, Poisson-like).  A wave of the gather
// kernels walks its rows in lockstep for as long as its longest row lasts: with the rows taken in natural order only 0.64
// of the issued lane-steps carried an entry (0.73 on softmax(randn), 0.59 on the structured map).  Sorting the block's rows
// by length first (ranks 8w .. 8w+7 to wave w) puts rows of similar length into one wave: 0.83 at 32 rows per block, 0.89
// at 64, 0.93 at 128 (scripts/time_attn_variants.py).  Measured at OPT-1.3B x 8, the layer's own selection:
// 1.15 ms (natural order) -> 0.99 (32 rows) -> 0.96 (64 rows per block) -> 1.00 (128).
// `len` = entries of the row that sits at block slot `gi` in natural order (-1: no row here / not this kernel's); returns
// the slot this lane group walks; *rowok = that slot holds a row.  Two workgroup barriers: EVERY thread of the block calls.
template <int LPR> __device__ inline int lane_group_sum_i(int x) {
  static_assert(LPR == 4 || LPR == 8 || LPR == 16, "lane groups inside one DPP row");
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);                   // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);                   // quad_perm [2,3,0,1]
  if (LPR >= 8) x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);    // row_half_mirror
  if (LPR >= 16) x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);   // row_mirror
  return x;
}

template <int LPR, int RPB>
__device__ inline int rows_by_length(int len, int gi, int sub, bool* rowok) {
  __shared__ int s_len[RPB], s_row[RPB];
  if (sub == 0) s_len[gi] = len;
  __syncthreads();
  {
    // rank by (length descending, slot ascending): lane `sub` compares against slots sub, sub + LPR, ...
    const int mine = s_len[gi];
    int before = 0;
#pragma unroll
    for (int r = sub; r < RPB; r += LPR) {
      const int o = s_len[r];
      before += (o > mine || (o == mine && r < gi)) ? 1 : 0;
    }
    const int rank = lane_group_sum_i<LPR>(before);
    if (sub == 0) s_row[rank] = gi;
  }
  __syncthreads();
  const int slot = s_row[gi];                              // this lane group walks the row of rank gi
  *rowok = s_len[slot] >= 0;
  return slot;
}

// the tile-block counter sits behind the plan's bytes, 4-byte aligned
__host__ __device__ inline int64_t plan_count_offset(int64_t N, int64_t H, int64_t TB16) { return (N * H * TB16 + 3) & ~(int64_t)3; }

// launchers of the MFMA tile kernel (sea_attn_tile.hip); `flags`: see SEA_ATTN_* in sea_hip.h
bool attn_tile_supported(int dtype, int D, int T_src, const AttnParams& p);
int launch_attn_tile(const AttnParams& p, int dtype, int out_dtype, int flags, hipStream_t s);
int launch_attn_plan(const uint32_t* bits, int N, int H, int T_dst, int T_src, int T_m, int causal, float entries_per_tile,
                     uint8_t* sel, hipStream_t s);

}  // namespace sea
