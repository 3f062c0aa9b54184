// Row-indexed sparse attention on the flat CSR, hand-written for gfx950 (wave64).
//
// Replaces (reference, src/models/perlin_attention/ops/kernels/):
//   flat_csr_masked_bmm.py:39-125   SDDMM  (1-warp programs, scalar entry loop)
//   flat_csr_softmax.py:55-125      per-(row, head) softmax (H masked passes over the whole row)
//   flat_csr_elmul.py:42-108        row-scale multiply through a stride-0 (N,H,T,T) view
//   flat_csr_sdbmm.py:48-127,141-313  head-count pass + SpMM
// and the mix of attention.py:1236-1237.
//
// Mapping: ONE WAVEFRONT PER (n, h, t) QUERY ROW.  The 64 lanes form 64/LPR groups of LPR lanes; a
// group reads one whole K (or V) row per instruction as LPR x 16 B, so every gathered row is a
// run of full, contiguous 16-byte lane loads (a 128 B bf16 d=64 row = 8 lanes).  q stays in
// registers; the dot product is reduced inside the group with DPP; every group keeps its own
// online-softmax state (m, l, acc[VEC]) so the key loop has no cross-group traffic, and the
// groups are merged once at the end (flash-decoding style).  No LDS, no barriers.
//
// HBM / L2: K and V of one (n, h) are T_src*D*2*s bytes (1 MiB at OPT-1.3B T=4096 bf16), so the
// gathers are served by L2 when the workgroups of a head run on one XCD: blockIdx is remapped so
// that the (n, h) pair index is congruent to the XCD label blockIdx % 8 (speed only).
#include "sea_attn.hpp"

namespace sea {

template <int CTRL> __device__ inline float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

// sum over the LPR lanes of an aligned group; result in every lane of the group
template <int LPR> __device__ inline float group_sum(float x) {
  if (LPR >= 2) x += dpp_f<0xB1>(x);    // quad_perm [1,0,3,2]
  if (LPR >= 4) x += dpp_f<0x4E>(x);    // quad_perm [2,3,0,1]
  if (LPR >= 8) x += dpp_f<0x141>(x);   // row_half_mirror (values are quad-uniform here)
  if (LPR >= 16) x += dpp_f<0x140>(x);  // row_mirror      (values are 8-uniform here)
  if (LPR >= 32) x += __shfl_xor(x, 16);
  if (LPR >= 64) x += __shfl_xor(x, 32);
  return x;
}

// q . k over one 16-byte lane fragment.  16-bit inputs use the packed dot-product instructions
// (v_dot2c_f32_bf16 / v_dot2_f32_f16: exact products, fp32 accumulation) on the raw registers -- no unpacking.
typedef __attribute__((ext_vector_type(2))) __bf16 sea_bf2;
typedef __attribute__((ext_vector_type(2))) _Float16 sea_h2;
template <typename T> __device__ inline float frag_dot(const uint4& q, const uint4& k);
template <> __device__ inline float frag_dot<float>(const uint4& q, const uint4& k) {
  float d = __uint_as_float(q.x) * __uint_as_float(k.x);
  d = fmaf(__uint_as_float(q.y), __uint_as_float(k.y), d);
  d = fmaf(__uint_as_float(q.z), __uint_as_float(k.z), d);
  return fmaf(__uint_as_float(q.w), __uint_as_float(k.w), d);
}
template <> __device__ inline float frag_dot<__hip_bfloat16>(const uint4& q, const uint4& k) {
  float d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sea_bf2, q.x), __builtin_bit_cast(sea_bf2, k.x), 0.f, false);
  d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sea_bf2, q.y), __builtin_bit_cast(sea_bf2, k.y), d, false);
  d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sea_bf2, q.z), __builtin_bit_cast(sea_bf2, k.z), d, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sea_bf2, q.w), __builtin_bit_cast(sea_bf2, k.w), d, false);
}
template <> __device__ inline float frag_dot<__half>(const uint4& q, const uint4& k) {
  float d = __builtin_amdgcn_fdot2(__builtin_bit_cast(sea_h2, q.x), __builtin_bit_cast(sea_h2, k.x), 0.f, false);
  d = __builtin_amdgcn_fdot2(__builtin_bit_cast(sea_h2, q.y), __builtin_bit_cast(sea_h2, k.y), d, false);
  d = __builtin_amdgcn_fdot2(__builtin_bit_cast(sea_h2, q.z), __builtin_bit_cast(sea_h2, k.z), d, false);
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(sea_h2, q.w), __builtin_bit_cast(sea_h2, k.w), d, false);
}

// WP: also write the per-entry values rs * softmax to p.probs (`partial_attention_probs`, attention.py:1162-1171).  Pass 1
// leaves the raw score at the entry's slot, a second walk turns it into the probability once (m, l) are final; every slot
// is re-read by the lane that wrote it (a thread sees its own stores), so no fence is needed.
template <typename T, typename TO, int LPR, int U, bool WP>
__global__ __launch_bounds__(256) void sparse_attn_kernel(AttnParams p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KPI = 64 / LPR;  // keys per wave-instruction
  int pair, tb;
  if (!map_block(p.N * p.H, p.TB, &pair, &tb)) return;
  const int n = pair / p.H, h = pair - n * p.H;
  const int t = tb * 4 + (threadIdx.x >> 6);
  if (t >= p.T_dst) return;
  if (kernel_is_idle(p)) return;
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, sub = lane - grp * LPR;
  const bool dact = sub * VEC < p.D;

  const T* qp = reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] + t * p.qs[2] + sub * VEC;
  const T* kb = reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1] + sub * VEC;
  const T* vb = reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1] + sub * VEC;

  uint4 qraw = make_uint4(0, 0, 0, 0);
  if (dact) qraw = *reinterpret_cast<const uint4*>(qp);
  const int row_beg = p.crow[(int64_t)n * (p.T_dst + 1) + t];
  const int32_t* ho = p.head_off + ((int64_t)n * p.T_dst + t) * (p.H + 1);
  const int beg = row_beg + ho[h], end = row_beg + ho[h + 1];
  const int32_t* col = p.col + n * p.col_stride_n;
  const int hcol = h * p.T_src;

  float m = -INFINITY, l = 0.f;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;

  for (int e0 = beg; e0 < end; e0 += KPI * U) {
    bool ok[U];
    uint4 kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * KPI + grp;
      ok[u] = e < end;
      const int key = col[ok[u] ? e : end - 1] - hcol;      // masked groups re-read the row's last entry (a finite V row)
      kr[u] = make_uint4(0, 0, 0, 0);
      vr[u] = make_uint4(0, 0, 0, 0);
      if (dact) {
        kr[u] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * p.ks[2]);
        vr[u] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * p.vs[2]);
      }
    }
    float s[U];
    float mnew = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d = frag_dot<T>(qraw, kr[u]);
      d = group_sum<LPR>(d);
      s[u] = ok[u] ? d : -INFINITY;
      mnew = fmaxf(mnew, s[u]);
      if (WP && ok[u] && sub == 0) p.probs[n * p.probs_stride_n + e0 + u * KPI + grp] = d;
    }
    if (mnew == -INFINITY) continue;  // this group has seen no entry yet
    const float alpha = __expf(m - mnew);
    l *= alpha;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] *= alpha;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float pu = __expf(s[u] - mnew);
      float vf[VEC];
      unpack16<T>(vr[u], vf);
      l += pu;
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = fmaf(pu, vf[j], acc[j]);
    }
    m = mnew;
  }

  // merge the KPI groups (lanes with equal `sub` hold the same feature slice)
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
    const float mo = __shfl_xor(m, o);
    const float lo = __shfl_xor(l, o);
    const float M = fmaxf(m, mo);
    const float a = (m == -INFINITY) ? 0.f : __expf(m - M);
    const float b = (mo == -INFINITY) ? 0.f : __expf(mo - M);
    l = l * a + lo * b;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float ao = __shfl_xor(acc[j], o);
      acc[j] = acc[j] * a + ao * b;
    }
    m = M;
  }

  if (WP && sub == 0) {                                    // every group: the entries it scored in pass 1
    const int64_t ridx = ((int64_t)n * p.H + h) * p.T_dst + t;
    float c = (l > 0.f) ? (1.0f / l) : 0.f;
    if (p.row_scale) c *= p.row_scale[ridx];
    float* pr = p.probs + n * p.probs_stride_n;
    for (int e = beg + grp; e < end; e += KPI) pr[e] = __expf(pr[e] - m) * c;
  }
  if (grp == 0 && dact) {
    const int64_t ridx = ((int64_t)n * p.H + h) * p.T_dst + t;
    float scale = (l > 0.f) ? (1.0f / l) : 0.f;
    if (p.row_scale) scale *= p.row_scale[ridx];
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = (l > 0.f) ? acc[j] * scale : 0.f;   // empty row: 0 even if the clamped key's V is not finite
    if (p.mix) {
      const float a = p.mix[ridx];
      const T* ap = reinterpret_cast<const T*>(p.avg) + n * p.as[0] + h * p.as[1] + t * p.as[2] + sub * VEC;
      float af[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(ap), af);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = o[j] * a + (1.0f - a) * af[j];
    }
    TO* op = reinterpret_cast<TO*>(p.out) + n * p.os[0] + h * p.os[1] + t * p.os[2] + sub * VEC;
    store_frag<TO, VEC>(op, o);
  }
}

// natural-order length of slot gi's row for rows_by_length (sea_attn.hpp): -1 for slots past T_dst
__device__ inline int slot_length(const AttnParams& p, int n, int h, int tn) {
  if (tn >= p.T_dst) return -1;
  const int32_t* hon = p.head_off + ((int64_t)n * p.T_dst + tn) * (p.H + 1);
  return hon[h + 1] - hon[h];
}

// ---- steps I + J in one launch: the interpolation of ONE (row, head) by the lane group that walks it -----------------------
// The lane group expands its row's kept pixels of head h to key columns -- the emit kernel's arithmetic, bit for bit
// (interp_scale / interp_bound, keys descending inside a pixel, the reference's fp32 stepping for a thinned pixel) -- into
// a group-private list in LDS, copies the list to `col` (the CSR's column array is an output of the layer) and then walks
// it from LDS: no separate emit launch, no column re-read from memory, and the expansion's scalar-heavy loops run in the
// shadow of a kernel that is bound by its gathers.  Lane `sub` owns mask word `sub` of the head (T_m / 32 words; more:
// rounds of LPR); a group scan orders the runs.  The list is written and read by the same wave, whose LDS operations
// execute in order: no workgroup barrier after the one that sizes the lists.  Blocks whose lists would not fit the LDS
// budget expand straight into `col` and read it back past the L1 (rare: a head that takes most of a row).
// Every thread of the block calls (one workgroup barrier inside).
template <int LPR, int RPB>
__device__ inline void fused_expand(const AttnParams& p, int n, int h, int tt, bool rowok, int gi, int sub, int beg, int end,
                                    int hcol, int t_src, int* s_keys, int* lbase_out, bool* fits_out, int* total_out) {
  __shared__ int s_glen[RPB];
  const int mylen = end - beg;
  if (sub == 0) s_glen[gi] = mylen;
  __syncthreads();
  int bef = 0, tot = 0;
#pragma unroll
  for (int r = sub; r < RPB; r += LPR) {
    const int v_ = s_glen[r];
    tot += v_;
    bef += r < gi ? v_ : 0;
  }
  const int lbase = lane_group_sum_i<LPR>(bef);
  const int total = lane_group_sum_i<LPR>(tot);
  const bool fits = total <= p.fuse_cap;                           // block-uniform
  int32_t* gcol = p.col_w + n * p.col_stride_n;
  const int WPH = p.T_m >> 5;                              // mask words per head (launcher: T_m % 32 == 0)
  const int w_t = row_width(tt, p.T_dst, t_src, p.is_causal);      // (t_src: p.T_src, or the decode form's device counter)
  const float scale = interp_scale(w_t, p.T_m);
  const uint32_t* brow = p.bits + ((int64_t)n * p.T_dst + tt) * p.W + h * WPH;
  int carry = 0;
  for (int w0 = 0; w0 < WPH; w0 += LPR) {                  // uniform
    const int wi = w0 + sub;
    const uint32_t word = (rowok && wi < WPH) ? brow[wi] : 0u;
    int nent = 0;
    for (uint32_t mm = word; mm;) {
      const int b = wi * 32 + __ffs(mm) - 1;
      mm &= mm - 1;
      const int wd = (int)interp_bound(b + 1, scale) - (int)interp_bound(b, scale);
      nent += wd < p.max_k ? wd : p.max_k;
    }
    int incl = nent;                                       // inclusive scan over the group's LPR lanes (word order)
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
      const int up = __shfl_up(incl, o, LPR);
      if (sub >= o) incl += up;
    }
    int off = carry + incl - nent;
    carry += __shfl(incl, LPR - 1, LPR);
    for (uint32_t mm = word; mm;) {
      const int b = wi * 32 + __ffs(mm) - 1;
      mm &= mm - 1;
      const int lo = (int)interp_bound(b, scale), hi = (int)interp_bound(b + 1, scale);
      const int wd = hi - lo;
      if (wd <= p.max_k) {
        const int c0 = hcol + hi - 1;
        if (fits) { for (int j = 0; j < wd; ++j) s_keys[lbase + off + j] = c0 - j; }
        else { for (int j = 0; j < wd; ++j) gcol[beg + off + j] = c0 - j; }
        off += wd;
      } else {                                             // thinned pixel: the reference's fp32 stepping (csr_emit_kernel)
        const float rs = (float)lo + (float)hcol, re = (float)hi + (float)hcol;
        const float step = __fdiv_rn(re - rs, (float)p.max_k);
        for (int j = 0; j < p.max_k; ++j) {
          const int c = (int)((re - (float)(int)__fmul_rn((float)j, step)) - 1.0f);
          if (fits) s_keys[lbase + off + j] = c; else gcol[beg + off + j] = c;
        }
        off += p.max_k;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (fits) {                                              // the CSR's columns: the list, copied out by its own lane group
    if (p.write_cols)                                      // (grid-uniform; 0: nobody reads them, the handle stays pending)
      for (int i = sub; i < mylen; i += LPR) gcol[beg + i] = s_keys[lbase + i];
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                         // this wave's column stores have left before it reads them back
  }
  *lbase_out = lbase;
  *fits_out = fits;
  *total_out = total;
}

// Decode form (DEC kernels, T_dst of a few rows): the block's key lists are complete in LDS (or, not fitting, in `col`); before
// the one lane group per row starts its dependent walk, EVERY lane group of the block touches the K / V rows of the lists --
// what the unfused kernel's warming pass does from `col` (see there).  Every thread of the block calls (two barriers).
template <int LPR, int NG>
__device__ inline void fused_warm(const int* s_keys, int total, bool fits, int hcol, const char* kbase, const char* vbase,
                                  uint32_t kst, uint32_t vst, uint32_t lane_off) {
  __syncthreads();                                         // the lists of the other waves
  if (fits) {                                              // (block-uniform)
    const int g = (int)threadIdx.x / LPR;
    uint32_t sink = 0;
    for (int e = g; e < total; e += NG) {
      const uint32_t key_c = (uint32_t)(s_keys[e] - hcol);
      const uint4 a = *reinterpret_cast<const uint4*>(kbase + (__umul24(key_c, kst) + lane_off));
      const uint4 b = *reinterpret_cast<const uint4*>(vbase + (__umul24(key_c, vst) + lane_off));
      sink ^= a.x ^ b.x;
    }
    asm volatile("" ::"v"(sink));                          // the loads must complete; their values are not used
  }
  __syncthreads();
}

// lane `sub`'s column of entry `ec` of its row: from the group's own list (fused, fits), from `col` past the L1 (fused, the
// block's lists did not fit: written just above by this wave), or from `col` as the emit launch left it
template <bool FUSE>
__device__ inline int entry_column(const AttnParams& p, const int32_t* col, int n, int ec, int beg, const int* s_keys, int lbase, bool fits) {
  if (FUSE && fits) return s_keys[lbase + ec - beg];
  if (FUSE) return __hip_atomic_load(p.col_w + n * p.col_stride_n + ec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return col[ec];
}

constexpr int SEA_ATTN_WARM_ROWS = 8;            // T_dst up to which the idle lane groups pre-touch the rows' K / V lines

// ---- variant B: one LPR-lane GROUP per query row (64/LPR rows per wave) -----------------------------------
// Each group walks its own row's entries, one key per instruction step, U keys in flight, with its own online
// softmax; nothing is merged across groups.  A wave therefore has 64/LPR independent
// (offsets -> col -> K/V) load chains in flight instead of one, which is what the wave-per-row mapping lacks.
// Workgroup = 4 waves = 256/LPR consecutive query rows of one (n, h).
// DEC (with FUSE): the decode form -- row widths from *p.t_src_dev, the key lists warmed by the whole block (fused_warm).
template <typename T, typename TO, int LPR, int U, bool WP, int NWB = 4, bool FUSE = false, bool DEC = false>
// the fused 16-bit inference forms are held to 64 registers (8 waves per SIMD; 4 - 8 spilled registers): measured 0.307 ms
// against 0.328 ms at 88 registers for the 16-lane form (LLaMA-13B d = 128), and the same direction for the 8-lane form
__global__ __launch_bounds__(NWB * 64, (FUSE && !DEC && sizeof(T) == 2 && !WP) ? 8 : 1) void sparse_attn_rows_kernel(AttnParams p) {
  static_assert(!DEC || FUSE, "the decode form is a fused form");
  constexpr int VEC = Elem<T>::VEC;
  constexpr int RPW = 64 / LPR;       // rows per wave
  constexpr int RPB = NWB * RPW;      // rows per workgroup
  int pair, tb;
  if (kernel_is_idle(p) || !map_block(p.N * p.H, p.TB, &pair, &tb)) return;      // (grid / workgroup uniform: before any barrier)
  const int n = pair / p.H;
  // heads rotated over the XCDs per batch item: the pair index is congruent to the XCD label (map_block), so without the
  // rotation an XCD would serve the same residue class of heads for every item and a heavy head would load one XCD only
  const int h = (pair - n * p.H + n) % p.H;
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, sub = lane - grp * LPR;
  const int gi = (int)(threadIdx.x >> 6) * RPW + grp;      // lane group index inside the block
  bool rowok;
  const int t = tb * RPB + rows_by_length<LPR, RPB>(slot_length(p, n, h, tb * RPB + gi), gi, sub, &rowok);
  const bool dact = sub * VEC < p.D;
  const int tt = t < p.T_dst ? t : p.T_dst - 1;
  const int sube = dact ? sub : 0;    // lanes beyond D re-read fragment 0 (their q fragment is zero)

  // wave-uniform 64-bit bases + per-lane 32-bit byte offsets (launcher guarantees T_src*stride*sizeof(T) < 2^31)
  const char* kbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1]);
  const char* vbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1]);
  const uint32_t kst = (uint32_t)p.ks[2] * (uint32_t)sizeof(T), vst = (uint32_t)p.vs[2] * (uint32_t)sizeof(T);
  const uint32_t lane_off = (uint32_t)(sube * VEC) * (uint32_t)sizeof(T);

  uint4 qraw = make_uint4(0, 0, 0, 0);
  if (dact) qraw = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] +
                                                  (int64_t)tt * p.qs[2] + sub * VEC);
  const int row_beg = p.crow[(int64_t)n * (p.T_dst + 1) + tt];
  const int32_t* ho = p.head_off + ((int64_t)n * p.T_dst + tt) * (p.H + 1);
  const int beg = row_beg + ho[h];
  const int end = rowok ? row_beg + ho[h + 1] : beg;
  const int32_t* col = p.col + n * p.col_stride_n;
  const int hcol = h * p.T_src;

  // ---- FUSE: the interpolation of THIS (row, head) inside the attention launch (fused_expand below the helpers) ----------
  extern __shared__ int s_keys[];
  int lbase = 0;
  bool fits = false;
  int ltotal = 0;
  if constexpr (FUSE) fused_expand<LPR, RPB>(p, n, h, tt, rowok, gi, sub, beg, end, hcol, DEC ? *p.t_src_dev : p.T_src, s_keys, &lbase, &fits, &ltotal);
  if constexpr (DEC) fused_warm<LPR, NWB * RPW>(s_keys, ltotal, fits, hcol, kbase, vbase, kst, vst, lane_off);

  // ---- a decoding step: T_dst of one or a few rows, so all but a few of the block's lane groups have no row -- and the one
  // that has walks its ~k entries down ONE dependent chain (column indices -> four K / V rows -> the next four ...: three
  // memory round trips per 8 entries, 40 us for an OPT-1.3B x 8 step whose K / V lines come cold from HBM).  The idle groups
  // first touch every K / V row the block's rows will gather, all at once: the walk below then finds them in this CU's L1 /
  // the L2.  Arithmetic untouched: the step stays bitwise the stateless forward.
  if constexpr (!FUSE) {
    if (p.T_dst <= SEA_ATTN_WARM_ROWS) {                   // (grid-uniform)
      constexpr int NG = NWB * RPW;
      const int g = (int)threadIdx.x / LPR;
      uint32_t sink = 0;
      for (int r = 0; r < p.T_dst; ++r) {
        const int rb2 = p.crow[(int64_t)n * (p.T_dst + 1) + r];
        const int32_t* ho2 = p.head_off + ((int64_t)n * p.T_dst + r) * (p.H + 1);
        const int b2 = rb2 + ho2[h], e2 = rb2 + ho2[h + 1];
        for (int e = b2 + g; e < e2; e += NG) {
          const uint32_t key_c = (uint32_t)(col[e] - hcol);
          const uint4 a = *reinterpret_cast<const uint4*>(kbase + (__umul24(key_c, kst) + lane_off));
          const uint4 b = *reinterpret_cast<const uint4*>(vbase + (__umul24(key_c, vst) + lane_off));
          sink ^= a.x ^ b.x;
        }
      }
      asm volatile("" ::"v"(sink));                        // the loads must complete; their values are not used
      __syncthreads();
    }
  }

  float m = -INFINITY, l = 0.f;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;

  // longest row of the wave bounds the loop (rows of a wave are neighbours: similar lengths)
  int zmax = end - beg;
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) zmax = max(zmax, __shfl_xor(zmax, o));
  const int last = end - 1;

  // Column indices: lane `sub` of a group fetches entry (i0 + sub) -- one coalesced load per LPR entries -- and the
  // group reads them back one by one through the LDS crossbar (ds_bpermute), which this kernel otherwise leaves idle.
  const int grp_lane0 = (lane - sub) << 2;                 // byte address of the group's first lane for bpermute
  for (int i0 = 0; i0 < zmax; i0 += LPR) {
    int cidx;
    {
      const int e = beg + i0 + sub;
      const int ec = e < end ? e : (last >= beg ? last : 0);
      cidx = (end > beg) ? (entry_column<FUSE>(p, col, n, ec, beg, s_keys, lbase, fits) - hcol) : 0;   // rows without entries read key 0 (masked below)
    }
#pragma unroll
    for (int u0 = 0; u0 < LPR; u0 += U) {
      if (i0 + u0 < zmax) {                                // wave-uniform
        bool ok[U];
        uint4 kr[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ok[u] = beg + i0 + u0 + u < end;
          const uint32_t key_c = (uint32_t)__builtin_amdgcn_ds_bpermute(grp_lane0 + ((u0 + u) << 2), cidx);
          kr[u] = *reinterpret_cast<const uint4*>(kbase + (__umul24(key_c, kst) + lane_off));
          vr[u] = *reinterpret_cast<const uint4*>(vbase + (__umul24(key_c, vst) + lane_off));
        }
        float s[U];
        float mnew = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float d = frag_dot<T>(qraw, kr[u]);
          d = group_sum<LPR>(d);
          s[u] = ok[u] ? d : -INFINITY;
          mnew = fmaxf(mnew, s[u]);
          if (WP && ok[u] && sub == u0 + u) p.probs[n * p.probs_stride_n + beg + i0 + u0 + u] = d;   // entry j by lane j % LPR
        }
        const float msafe = (mnew == -INFINITY) ? 0.f : mnew;     // rows that have seen nothing yet: exp(-inf - 0) = 0
        const float alpha = __expf(m - msafe);
        l *= alpha;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] *= alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float pu = __expf(s[u] - msafe);
          float vf[VEC];
          unpack16<T>(vr[u], vf);
          l += pu;
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] = fmaf(pu, vf[j], acc[j]);
        }
        m = mnew;
      }
    }
  }

  if (WP && rowok) {
    const int64_t ridx = ((int64_t)n * p.H + h) * p.T_dst + t;
    float c = (l > 0.f) ? (1.0f / l) : 0.f;
    if (p.row_scale) c *= p.row_scale[ridx];
    float* pr = p.probs + n * p.probs_stride_n;
    for (int e = beg + sub; e < end; e += LPR) pr[e] = __expf(pr[e] - m) * c;
  }
  if (rowok && dact) {
    const int64_t ridx = ((int64_t)n * p.H + h) * p.T_dst + t;
    float scale = (l > 0.f) ? (1.0f / l) : 0.f;
    if (p.row_scale) scale *= p.row_scale[ridx];
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = (l > 0.f) ? acc[j] * scale : 0.f;   // empty row: 0 even if the clamped key's V is not finite
    if (p.mix) {
      const float a = p.mix[ridx];
      const T* ap = reinterpret_cast<const T*>(p.avg) + n * p.as[0] + h * p.as[1] + (int64_t)t * p.as[2] + sub * VEC;
      float af[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(ap), af);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = o[j] * a + (1.0f - a) * af[j];
    }
    TO* op = reinterpret_cast<TO*>(p.out) + n * p.os[0] + h * p.os[1] + (int64_t)t * p.os[2] + sub * VEC;
    store_frag<TO, VEC>(op, o);
  }
}

// ---- a decoding step of ONE new row per sequence (T_dst = 1): the whole workgroup serves the row (round 5) ----------------
// The lane-group kernels above give the row one lane group: its ~k entries go down one dependent chain -- column -> K, V ->
// score -> online softmax -> accumulate, four entries per memory round trip, 8 - 10 us -- while 63 of the block's 64 lane
// groups watch.  Here one 4-wave workgroup per (sequence, head):
//   B  expands the head's kept pixels with one THREAD per pixel (block scan of the widths; fused_expand's arithmetic: scale,
//      bounds, keys descending inside a pixel, the reference's fp32 stepping for a thinned pixel),
//   C  gathers K and V of all entries at once, one lane group per entry: the score q . k (the same frag_dot + group_sum, so the
//      same bits whoever computes it) goes to LDS, the V row too,
//   D  one wave turns the scores into the online softmax's per-step quantities -- the running maximum is a prefix MAX (exact in
//      any order), alpha = exp(m_before - m_after) and p = exp(s - m_after) follow per step of four entries,
//   E  one lane group runs the accumulation chain l = l * alpha + p0 + p1 + p2 + p3, acc = acc * alpha (+)= p_u * v_u from LDS in
//      the walk's order: the SAME floating-point operations in the SAME order as sparse_attn_rows_kernel (U = 4), entries past
//      the row's end included (probability exp(-inf) = 0 times the last entry's V row) -- the step stays bitwise the stateless
//      forward (tests/test_decode_session.py, test_kv_cache.py).
// Rows longer than CH entries pass through in chunks (state carried).  16-bit data, rows of 8 or 16 lanes, T_m <= 256.
// ---- d = 80 (OPT-2.7B heads), 16-bit data: 8 lanes per row, each with 8 + 2 elements -------------------------
// The power-of-two mapping above gives a d = 80 row 16 lanes of which 10 carry data (4 rows per wave).  Here a row
// takes 8 lanes: lane j holds elements 8j..8j+7 (one 16-byte load) plus the pair 64+2j, 65+2j (one 4-byte load), so a
// wave advances 8 rows with every lane busy.  Same walk, same online softmax, same epilogue as sparse_attn_rows_kernel.
template <typename T> __device__ inline float tail_dot(uint32_t q, uint32_t k, float d);
template <> __device__ inline float tail_dot<__hip_bfloat16>(uint32_t q, uint32_t k, float d) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sea_bf2, q), __builtin_bit_cast(sea_bf2, k), d, false);
}
template <> __device__ inline float tail_dot<__half>(uint32_t q, uint32_t k, float d) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(sea_h2, q), __builtin_bit_cast(sea_h2, k), d, false);
}
template <typename T> __device__ inline void unpack2(uint32_t r, float* f);
template <> __device__ inline void unpack2<__hip_bfloat16>(uint32_t r, float* f) {
  f[0] = __uint_as_float(r << 16); f[1] = __uint_as_float(r & 0xffff0000u);
}
template <> __device__ inline void unpack2<__half>(uint32_t r, float* f) {
  const float2 t = __half22float2(__builtin_bit_cast(__half2, r));
  f[0] = t.x; f[1] = t.y;
}
template <typename TO> __device__ inline void store2(TO* dst, float a, float b) { *reinterpret_cast<uint32_t*>(dst) = pack2<TO>(a, b); }
template <> __device__ inline void store2<float>(float* dst, float a, float b) { *reinterpret_cast<float2*>(dst) = make_float2(a, b); }

template <typename T, typename TO, int U, bool WP, int NWB = 4, bool FUSE = false, bool DEC = false>
// (no wave floor here: capping this kernel at 80 / 64 registers for 6 / 8 waves per SIMD measured 0.427 / 0.489 ms against
// 0.418 ms as compiled, OPT-2.7B T = 8192 -- DESIGN.md section 9)
__global__ __launch_bounds__(NWB * 64) void sparse_attn_rows80_kernel(AttnParams p) {
  static_assert(!DEC || FUSE, "the decode form is a fused form");
  constexpr int LPR = 8, VEC = 8, XT = 2, DM = LPR * VEC;     // DM = 64 elements in the 16-byte fragments
  constexpr int RPW = 64 / LPR, RPB = NWB * RPW;
  int pair, tb;
  if (kernel_is_idle(p) || !map_block(p.N * p.H, p.TB, &pair, &tb)) return;
  const int n = pair / p.H;
  const int h = (pair - n * p.H + n) % p.H;                // heads rotated over the XCDs per item (see sparse_attn_rows_kernel)
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, sub = lane - grp * LPR;
  const int gi = (int)(threadIdx.x >> 6) * RPW + grp;
  bool rowok;
  const int t = tb * RPB + rows_by_length<LPR, RPB>(slot_length(p, n, h, tb * RPB + gi), gi, sub, &rowok);
  const int tt = t < p.T_dst ? t : p.T_dst - 1;

  const char* kbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1]);
  const char* vbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1]);
  const uint32_t kst = (uint32_t)p.ks[2] * (uint32_t)sizeof(T), vst = (uint32_t)p.vs[2] * (uint32_t)sizeof(T);
  const uint32_t off_m = (uint32_t)(sub * VEC) * (uint32_t)sizeof(T);              // 16-byte fragment
  const uint32_t off_x = (uint32_t)(DM + sub * XT) * (uint32_t)sizeof(T);          // 4-byte pair

  const T* qrow = reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] + (int64_t)tt * p.qs[2];
  const uint4 qraw = *reinterpret_cast<const uint4*>(qrow + sub * VEC);
  const uint32_t qx = *reinterpret_cast<const uint32_t*>(qrow + DM + sub * XT);
  const int row_beg = p.crow[(int64_t)n * (p.T_dst + 1) + tt];
  const int32_t* ho = p.head_off + ((int64_t)n * p.T_dst + tt) * (p.H + 1);
  const int beg = row_beg + ho[h];
  const int end = rowok ? row_beg + ho[h + 1] : beg;
  const int32_t* col = p.col + n * p.col_stride_n;
  const int hcol = h * p.T_src;
  extern __shared__ int s_keys[];                          // fused interpolation: see fused_expand
  int lbase = 0;
  bool fits = false;
  int ltotal = 0;
  if constexpr (FUSE) fused_expand<LPR, RPB>(p, n, h, tt, rowok, gi, sub, beg, end, hcol, DEC ? *p.t_src_dev : p.T_src, s_keys, &lbase, &fits, &ltotal);
  // (the 16-byte fragments of the lists' K / V rows: 128 of a row's 160 bytes, i.e. both of its cache lines)
  if constexpr (DEC) fused_warm<LPR, NWB * RPW>(s_keys, ltotal, fits, hcol, kbase, vbase, kst, vst, off_m);

  float m = -INFINITY, l = 0.f;
  float acc[VEC + XT];
#pragma unroll
  for (int j = 0; j < VEC + XT; ++j) acc[j] = 0.f;

  int zmax = end - beg;
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) zmax = max(zmax, __shfl_xor(zmax, o));
  const int last = end - 1;
  const int grp_lane0 = (lane - sub) << 2;
  for (int i0 = 0; i0 < zmax; i0 += LPR) {
    int cidx;
    {
      const int e = beg + i0 + sub;
      const int ec = e < end ? e : (last >= beg ? last : 0);
      cidx = (end > beg) ? (entry_column<FUSE>(p, col, n, ec, beg, s_keys, lbase, fits) - hcol) : 0;
    }
#pragma unroll
    for (int u0 = 0; u0 < LPR; u0 += U) {
      if (i0 + u0 < zmax) {                                // wave-uniform
        bool ok[U];
        uint4 kr[U], vr[U];
        uint32_t kx[U], vx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ok[u] = beg + i0 + u0 + u < end;
          const uint32_t key_c = (uint32_t)__builtin_amdgcn_ds_bpermute(grp_lane0 + ((u0 + u) << 2), cidx);
          const uint32_t ko = __umul24(key_c, kst), vo = __umul24(key_c, vst);
          kr[u] = *reinterpret_cast<const uint4*>(kbase + (ko + off_m));
          kx[u] = *reinterpret_cast<const uint32_t*>(kbase + (ko + off_x));
          vr[u] = *reinterpret_cast<const uint4*>(vbase + (vo + off_m));
          vx[u] = *reinterpret_cast<const uint32_t*>(vbase + (vo + off_x));
        }
        float s[U];
        float mnew = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float d = tail_dot<T>(qx, kx[u], frag_dot<T>(qraw, kr[u]));
          d = group_sum<LPR>(d);
          s[u] = ok[u] ? d : -INFINITY;
          mnew = fmaxf(mnew, s[u]);
          if (WP && ok[u] && sub == u0 + u) p.probs[n * p.probs_stride_n + beg + i0 + u0 + u] = d;
        }
        const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = __expf(m - msafe);
        l *= alpha;
#pragma unroll
        for (int j = 0; j < VEC + XT; ++j) acc[j] *= alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float pu = __expf(s[u] - msafe);
          float vf[VEC + XT];
          unpack16<T>(vr[u], vf);
          unpack2<T>(vx[u], vf + VEC);
          l += pu;
#pragma unroll
          for (int j = 0; j < VEC + XT; ++j) acc[j] = fmaf(pu, vf[j], acc[j]);
        }
        m = mnew;
      }
    }
  }

  if (WP && rowok) {
    const int64_t ridx = ((int64_t)n * p.H + h) * p.T_dst + t;
    float c = (l > 0.f) ? (1.0f / l) : 0.f;
    if (p.row_scale) c *= p.row_scale[ridx];
    float* pr = p.probs + n * p.probs_stride_n;
    for (int e = beg + sub; e < end; e += LPR) pr[e] = __expf(pr[e] - m) * c;
  }
  if (rowok) {
    const int64_t ridx = ((int64_t)n * p.H + h) * p.T_dst + t;
    float scale = (l > 0.f) ? (1.0f / l) : 0.f;
    if (p.row_scale) scale *= p.row_scale[ridx];
    float o[VEC + XT];
#pragma unroll
    for (int j = 0; j < VEC + XT; ++j) o[j] = (l > 0.f) ? acc[j] * scale : 0.f;
    if (p.mix) {
      const float a = p.mix[ridx];
      const T* ap = reinterpret_cast<const T*>(p.avg) + n * p.as[0] + h * p.as[1] + (int64_t)t * p.as[2];
      float af[VEC + XT];
      unpack16<T>(*reinterpret_cast<const uint4*>(ap + sub * VEC), af);
      unpack2<T>(*reinterpret_cast<const uint32_t*>(ap + DM + sub * XT), af + VEC);
#pragma unroll
      for (int j = 0; j < VEC + XT; ++j) o[j] = o[j] * a + (1.0f - a) * af[j];
    }
    TO* op = reinterpret_cast<TO*>(p.out) + n * p.os[0] + h * p.os[1] + (int64_t)t * p.os[2];
    store_frag<TO, VEC>(op + sub * VEC, o);
    store2<TO>(op + DM + sub * XT, o[VEC], o[VEC + 1]);
  }
}

// X80: d = 80 on 8 lanes per row -- lane j holds elements 8j .. 8j+7 plus the pair 64+2j, 65+2j (sparse_attn_rows80_kernel's map)
template <typename T, typename TO, int LPR, bool X80 = false>
__global__ __launch_bounds__(256) void sparse_attn_decode1_kernel(AttnParams p) {
  constexpr int VEC = Elem<T>::VEC, NT = 256, NG = NT / LPR, XT = X80 ? 2 : 0, DL = LPR * (VEC + XT);
  static_assert(!X80 || LPR == 8, "d = 80: eight lanes per row");
  constexpr int CH = LPR == 8 ? 256 : 128;                 // entries per chunk: CH / 4 steps fit one wave, CH / NG = 8 per lane group
  constexpr int EPG = CH / NG;
  static_assert(VEC == 8 && CH / 4 <= 64 && CH % NG == 0, "16-bit rows; one lane per step in phase D");
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  int* s_keys = reinterpret_cast<int*>(dsm);               // [CH]     key index (without the head offset)
  float* s_sc = reinterpret_cast<float*>(s_keys + CH);     // [CH]     scores, then probabilities
  float* s_al = s_sc + CH;                                 // [CH / 4] alpha per step
  T* s_v = reinterpret_cast<T*>(s_al + CH / 4);            // [CH][DL] V rows
  __shared__ int s_wsum[4];
  __shared__ float s_mcarry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int grp = tid / LPR, sub = tid - grp * LPR;
  const int n = (int)blockIdx.x / p.H, h = (int)blockIdx.x - n * p.H;
  const bool dact = X80 || sub * VEC < p.D;
  const int sube = dact ? sub : 0;
  const char* kbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1]);
  const char* vbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1]);
  const uint32_t kst = (uint32_t)p.ks[2] * (uint32_t)sizeof(T), vst = (uint32_t)p.vs[2] * (uint32_t)sizeof(T);
  const uint32_t lane_off = (uint32_t)(sube * VEC) * (uint32_t)sizeof(T);
  uint4 qraw = make_uint4(0, 0, 0, 0);
  if (dact) qraw = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] + sub * VEC);
  constexpr int DM = LPR * VEC;                            // X80: elements in the 16-byte fragments; the pairs sit behind them
  const uint32_t off_x = (uint32_t)(DM + sub * 2) * (uint32_t)sizeof(T);
  uint32_t qx = 0;
  if constexpr (X80) qx = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] + DM + sub * 2);
  const int beg = p.crow[(int64_t)n * 2] + p.head_off[(int64_t)n * (p.H + 1) + h];
  const int hcol = h * p.T_src;

  // ---- B: one thread per pixel of the head --------------------------------------------------------------------------
  const int w_t = row_width(0, 1, *p.t_src_dev, p.is_causal);
  const float scale = interp_scale(w_t, p.T_m);
  const uint32_t* brow = p.bits + (int64_t)n * p.W + h * (p.T_m >> 5);
  const bool kept = tid < p.T_m && ((brow[tid >> 5] >> (tid & 31)) & 1u);
  const int lo = (int)interp_bound(tid, scale), hi = (int)interp_bound(tid + 1, scale);
  const int wd = hi - lo;
  const int cnt = kept ? (wd < p.max_k ? wd : p.max_k) : 0;
  const int incl = wave_incl_scan(cnt);
  if (lane == 63) s_wsum[wv] = incl;
  if (tid == 0) s_mcarry = -INFINITY;
  __syncthreads();
  int off = incl - cnt, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { off += w < wv ? s_wsum[w] : 0; total += s_wsum[w]; }
  const int padded = (total + 3) & ~3;                     // the walk's steps of four: entries past the end re-read the last one
  const bool owns_last = cnt > 0 && off + cnt == total;
  int32_t* gcol = p.col_w + n * p.col_stride_n;

  float l = 0.f;
  float acc[VEC + XT];
#pragma unroll
  for (int j = 0; j < VEC + XT; ++j) acc[j] = 0.f;

  for (int c0 = 0; c0 < padded; c0 += CH) {                // (block-uniform)
    // this thread's pixel: its entries that fall into the chunk
    if (cnt > 0 && off < c0 + CH && off + cnt > c0) {
      const int j0 = max(0, c0 - off), j1 = min(cnt, c0 + CH - off);
      if (wd <= p.max_k) {
        for (int j = j0; j < j1; ++j) s_keys[off + j - c0] = hi - 1 - j;
      } else {                                             // thinned pixel: the reference's fp32 stepping (csr_emit_kernel)
        const float rs = (float)lo + (float)hcol, re = (float)hi + (float)hcol;
        const float step = __fdiv_rn(re - rs, (float)p.max_k);
        for (int j = j0; j < j1; ++j) s_keys[off + j - c0] = (int)((re - (float)(int)__fmul_rn((float)j, step)) - 1.0f) - hcol;
      }
      if (owns_last && j1 == cnt) {                        // the last entry's key again in the slots up to the step boundary
        const int lastkey = s_keys[total - 1 - c0];
        for (int e = total; e < padded; ++e) s_keys[e - c0] = lastkey;
      }
    }
    __syncthreads();
    const int nent = min(CH, padded - c0);                 // entries of the chunk, padding included (a multiple of 4)
    if (p.write_cols)
      for (int e = tid; e < min(nent, total - c0); e += NT) gcol[beg + c0 + e] = hcol + s_keys[e];

    // ---- C: one lane group per entry, EPG entries each, every load in flight at once ------------------------------------
    {
      uint4 kr[EPG], vr[EPG];
      uint32_t kx[EPG], vx[EPG];
#pragma unroll
      for (int i = 0; i < EPG; ++i) {
        const int e = grp + i * NG;
        const uint32_t key_c = (uint32_t)s_keys[e < nent ? e : 0];
        const uint32_t ko = __umul24(key_c, kst), vo = __umul24(key_c, vst);
        kr[i] = *reinterpret_cast<const uint4*>(kbase + (ko + lane_off));
        vr[i] = *reinterpret_cast<const uint4*>(vbase + (vo + lane_off));
        if constexpr (X80) {
          kx[i] = *reinterpret_cast<const uint32_t*>(kbase + (ko + off_x));
          vx[i] = *reinterpret_cast<const uint32_t*>(vbase + (vo + off_x));
        }
      }
#pragma unroll
      for (int i = 0; i < EPG; ++i) {
        const int e = grp + i * NG;
        float d = frag_dot<T>(qraw, kr[i]);
        if constexpr (X80) d = tail_dot<T>(qx, kx[i], d);
        d = group_sum<LPR>(d);
        if (e < nent) {
          if (sub == 0) s_sc[e] = c0 + e < total ? d : -INFINITY;
          *reinterpret_cast<uint4*>(s_v + (size_t)e * DL + sub * VEC) = vr[i];
          if constexpr (X80) *reinterpret_cast<uint32_t*>(s_v + (size_t)e * DL + DM + sub * 2) = vx[i];
        }
      }
    }
    __syncthreads();

    // ---- D: per step of four entries (lane = step): running maximum, alpha, probabilities --------------------------------
    if (wv == 0) {
      const int nsteps = nent >> 2;
      const bool on = lane < nsteps;
      float4 sc = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (on) sc = *reinterpret_cast<const float4*>(s_sc + 4 * lane);
      float mx = fmaxf(fmaxf(sc.x, sc.y), fmaxf(sc.z, sc.w));
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {                   // inclusive prefix maximum over the steps
        const float t = __shfl_up(mx, o);
        if (lane >= o) mx = fmaxf(mx, t);
      }
      const float m_in = s_mcarry;
      const float m_after = fmaxf(m_in, mx);
      float m_before = __shfl_up(m_after, 1);
      if (lane == 0) m_before = m_in;
      const float msafe = (m_after == -INFINITY) ? 0.f : m_after;     // rows that have seen nothing yet: exp(-inf - 0) = 0
      if (on) {
        s_al[lane] = __expf(m_before - msafe);
        *reinterpret_cast<float4*>(s_sc + 4 * lane) = make_float4(__expf(sc.x - msafe), __expf(sc.y - msafe), __expf(sc.z - msafe), __expf(sc.w - msafe));
      }
      if (lane == nsteps - 1) s_mcarry = m_after;
    }
    __syncthreads();

    // ---- E: the accumulation chain, in the walk's order ------------------------------------------------------------------
    if (grp == 0) {
      const int nsteps = nent >> 2;
      for (int st = 0; st < nsteps; ++st) {
        const float alpha = s_al[st];
        const float4 pu4 = *reinterpret_cast<const float4*>(s_sc + 4 * st);
        const float pu[4] = {pu4.x, pu4.y, pu4.z, pu4.w};
        uint4 vr[4];
        uint32_t vx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          vr[u] = *reinterpret_cast<const uint4*>(s_v + (size_t)(4 * st + u) * DL + sub * VEC);
          if constexpr (X80) vx[u] = *reinterpret_cast<const uint32_t*>(s_v + (size_t)(4 * st + u) * DL + DM + sub * 2);
        }
        l *= alpha;
#pragma unroll
        for (int j = 0; j < VEC + XT; ++j) acc[j] *= alpha;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float vf[VEC + 2];
          unpack16<T>(vr[u], vf);
          if constexpr (X80) unpack2<T>(vx[u], vf + VEC);
          l += pu[u];
#pragma unroll
          for (int j = 0; j < VEC + XT; ++j) acc[j] = fmaf(pu[u], vf[j], acc[j]);
        }
      }
    }
    __syncthreads();                                       // the chunk's LDS is free again
  }

  if (grp == 0 && dact) {                                  // sparse_attn_rows_kernel's epilogue (t = 0, T_dst = 1)
    const int64_t ridx = (int64_t)n * p.H + h;
    float sc_ = (l > 0.f) ? (1.0f / l) : 0.f;
    if (p.row_scale) sc_ *= p.row_scale[ridx];
    float o[VEC + 2];
#pragma unroll
    for (int j = 0; j < VEC + XT; ++j) o[j] = (l > 0.f) ? acc[j] * sc_ : 0.f;
    if (p.mix) {
      const float a = p.mix[ridx];
      const T* ap = reinterpret_cast<const T*>(p.avg) + n * p.as[0] + h * p.as[1];
      float af[VEC + 2];
      unpack16<T>(*reinterpret_cast<const uint4*>(ap + sub * VEC), af);
      if constexpr (X80) unpack2<T>(*reinterpret_cast<const uint32_t*>(ap + DM + sub * 2), af + VEC);
#pragma unroll
      for (int j = 0; j < VEC + XT; ++j) o[j] = o[j] * a + (1.0f - a) * af[j];
    }
    TO* op = reinterpret_cast<TO*>(p.out) + n * p.os[0] + h * p.os[1];
    store_frag<TO, VEC>(op + sub * VEC, o);
    if constexpr (X80) store2<TO>(op + DM + sub * 2, o[VEC], o[VEC + 1]);
  }
}

// ---- unfused SDDMM: one wave per (n, t) row, all heads -----------------------------------------------
struct SddmmParams {
  const void *q, *k;
  int64_t qs[3], ks[3];
  int N, H, T_dst, T_src, D;
  const void* crow;
  const void* col;
  int64_t col_stride_n;
  float* values;
};

template <typename T, typename I, int LPR>
__global__ __launch_bounds__(256) void csr_sddmm_kernel(SddmmParams p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KPI = 64 / LPR;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)p.N * p.T_dst) return;
  const int n = (int)(row / p.T_dst), t = (int)(row - (int64_t)n * p.T_dst);
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, sub = lane - grp * LPR;
  const bool dact = sub * VEC < p.D;
  const I* crow = reinterpret_cast<const I*>(p.crow) + (int64_t)n * (p.T_dst + 1);
  const I* col = reinterpret_cast<const I*>(p.col) + n * p.col_stride_n;
  float* vals = p.values + n * p.col_stride_n;
  const int64_t beg = crow[t], end = crow[t + 1];
  const T* qb = reinterpret_cast<const T*>(p.q) + n * p.qs[0] + t * p.qs[2] + sub * VEC;
  const T* kb = reinterpret_cast<const T*>(p.k) + n * p.ks[0] + sub * VEC;
  for (int64_t e0 = beg; e0 < end; e0 += KPI) {
    const int64_t e = e0 + grp;
    const bool ok = e < end;
    int64_t c = ok ? (int64_t)col[e] : 0;
    const int h = (int)(c / p.T_src);
    const int key = (int)(c - (int64_t)h * p.T_src);
    uint4 qr = make_uint4(0, 0, 0, 0), kr = make_uint4(0, 0, 0, 0);
    if (dact) {
      qr = *reinterpret_cast<const uint4*>(qb + h * p.qs[1]);
      kr = *reinterpret_cast<const uint4*>(kb + h * p.ks[1] + (int64_t)key * p.ks[2]);
    }
    float qf[VEC], kf[VEC];
    unpack16<T>(qr, qf);
    unpack16<T>(kr, kf);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) d = fmaf(qf[j], kf[j], d);
    d = group_sum<LPR>(d);
    if (ok && sub == 0) vals[e] = d;
  }
}

// ---- unfused SpMM: one wave per (n, h, t), needs head_off --------------------------------------------
struct SpmmParams {
  const float* values;
  const void* v;
  int64_t vs[3];
  int N, H, T_dst, T_src, D;
  const void* crow;
  const void* col;
  int64_t col_stride_n;
  const int32_t* head_off;
  float* out;
  int TB;
};

template <typename T, typename I, int LPR>
__global__ __launch_bounds__(256) void csr_spmm_kernel(SpmmParams p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KPI = 64 / LPR;
  int pair, tb;
  if (!map_block(p.N * p.H, p.TB, &pair, &tb)) return;
  const int n = pair / p.H, h = pair - n * p.H;
  const int t = tb * 4 + (threadIdx.x >> 6);
  if (t >= p.T_dst) return;
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR, sub = lane - grp * LPR;
  const bool dact = sub * VEC < p.D;
  const I* crow = reinterpret_cast<const I*>(p.crow) + (int64_t)n * (p.T_dst + 1);
  const I* col = reinterpret_cast<const I*>(p.col) + n * p.col_stride_n;
  const float* vals = p.values + n * p.col_stride_n;
  const int32_t* ho = p.head_off + ((int64_t)n * p.T_dst + t) * (p.H + 1);
  const int64_t beg = (int64_t)crow[t] + ho[h], end = (int64_t)crow[t] + ho[h + 1];
  const T* vb = reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1] + sub * VEC;
  const int64_t hcol = (int64_t)h * p.T_src;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  for (int64_t e0 = beg; e0 < end; e0 += KPI) {
    const int64_t e = e0 + grp;
    const bool ok = e < end;
    const int64_t key = ok ? ((int64_t)col[e] - hcol) : 0;
    const float pv = ok ? vals[e] : 0.f;
    uint4 vr = make_uint4(0, 0, 0, 0);
    if (dact) vr = *reinterpret_cast<const uint4*>(vb + key * p.vs[2]);
    float vf[VEC];
    unpack16<T>(vr, vf);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(pv, vf[j], acc[j]);
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] += __shfl_xor(acc[j], o);
  }
  if (grp == 0 && dact) {
    float* op = p.out + (((int64_t)n * p.H + h) * p.T_dst + t) * p.D + sub * VEC;
    store_frag<float, VEC>(op, acc);
  }
}

// ---- host side ---------------------------------------------------------------------------------------
static int lanes_per_row(int D, int vec) {
  const int need = (D + vec - 1) / vec;
  int l = 1;
  while (l < need) l *= 2;
  return l < 4 ? 4 : l;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static bool strides_ok(const int64_t* s, int vec) { return s[0] % vec == 0 && s[1] % vec == 0 && s[2] % vec == 0; }

// A decoding step (T_dst of a few rows per (n, h)): the fused form has nothing to win there (one row per workgroup), and the
// plain lane-group kernel warms the row's K / V lines with its idle lane groups first (below).  sea_attention_few_rows().
constexpr int64_t SEA_ATTN_FEW_ROWS = 2048;
template <typename T, typename TO, bool WP>
static int launch_attn_wp(AttnParams p, hipStream_t s) {
  constexpr int VEC = Elem<T>::VEC;
  const int lpr = lanes_per_row(p.D, VEC);
  const int NH = p.N * p.H;
  const int esz = (int)sizeof(T);
  const bool small = p.T_src < (1 << 24) && p.ks[2] * esz < (1 << 24) && p.vs[2] * esz < (1 << 24) &&
                     (int64_t)p.T_src * p.ks[2] * esz < (1ll << 31) && (int64_t)p.T_src * p.vs[2] * esz < (1ll << 31);
  if constexpr (sizeof(T) == 2 && !WP) {
    // one new row per sequence (a DecodeSession position): the whole workgroup serves the row (sparse_attn_decode1_kernel)
    if (p.bits && p.t_src_dev && p.T_dst == 1 && p.T_m <= 256 && small && (lpr == 8 || lpr == 16 || p.D == 80)) {
      if (p.D == 80) {
        constexpr int CH = 256, DL = 80;
        hipLaunchKernelGGL((sparse_attn_decode1_kernel<T, TO, 8, true>), dim3((unsigned)NH), dim3(256), CH * 8 + CH + CH * DL * 2, s, p);
      } else if (lpr == 8) {
        constexpr int CH = 256, DL = 64;
        hipLaunchKernelGGL((sparse_attn_decode1_kernel<T, TO, 8>), dim3((unsigned)NH), dim3(256), CH * 8 + CH + CH * DL * 2, s, p);
      } else {
        constexpr int CH = 128, DL = 128;
        hipLaunchKernelGGL((sparse_attn_decode1_kernel<T, TO, 16>), dim3((unsigned)NH), dim3(256), CH * 8 + CH + CH * DL * 2, s, p);
      }
      return SEA_OK;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (p.D == 80 && small) {                          // d = 80: 8 lanes x (8 + 2) elements per row
      const int rpb = 8 * 8;                               // 8 waves x 8 rows, sorted by length
      p.TB = (p.T_dst + rpb - 1) / rpb;
      const int64_t blocks = (int64_t)8 * ((NH + 7) / 8) * p.TB;
      if (p.bits && p.t_src_dev) {
        if constexpr (WP) return SEA_EUNSUPPORTED;
        else hipLaunchKernelGGL((sparse_attn_rows80_kernel<T, TO, 4, false, 8, true, true>), dim3((unsigned)blocks), dim3(512), p.fuse_cap * (int)sizeof(int), s, p);
      } else if (p.bits) hipLaunchKernelGGL((sparse_attn_rows80_kernel<T, TO, 4, WP, 8, true>), dim3((unsigned)blocks), dim3(512), p.fuse_cap * (int)sizeof(int), s, p);
      else hipLaunchKernelGGL((sparse_attn_rows80_kernel<T, TO, 4, WP, 8>), dim3((unsigned)blocks), dim3(512), 0, s, p);
      return SEA_OK;
    }
  }
  if (lpr <= 16 && small) {
    // rows per workgroup (sorted by length inside the kernel): 64 for rows of 4 or 8 lanes, 32 for rows of 16 lanes.
    // Measured at the headline shape (layer's own selection, bf16 d = 64): 32 / 64 / 128 rows per block 0.99 / 0.96 / 1.00 ms.
    const int nwb = lpr == 4 ? 4 : 8;
    const int rpb = nwb * (64 / lpr);
    p.TB = (p.T_dst + rpb - 1) / rpb;
    const int64_t blocks = (int64_t)8 * ((NH + 7) / 8) * p.TB;
    dim3 grid((unsigned)blocks), block(nwb * 64);
    if (p.bits) {                                          // fused interpolation: group-private key lists in LDS
      if (lpr == 4) return SEA_EUNSUPPORTED;
      const int lds = p.fuse_cap * (int)sizeof(int);
      if (p.t_src_dev) {                                   // decode form
        if constexpr (WP) return SEA_EUNSUPPORTED;
        else {
          if (lpr == 8) hipLaunchKernelGGL((sparse_attn_rows_kernel<T, TO, 8, 4, false, 8, true, true>), grid, block, lds, s, p);
          else hipLaunchKernelGGL((sparse_attn_rows_kernel<T, TO, 16, 4, false, 8, true, true>), grid, block, lds, s, p);
          return SEA_OK;
        }
      }
      if (lpr == 8) hipLaunchKernelGGL((sparse_attn_rows_kernel<T, TO, 8, 4, WP, 8, true>), grid, block, lds, s, p);
      else hipLaunchKernelGGL((sparse_attn_rows_kernel<T, TO, 16, 4, WP, 8, true>), grid, block, lds, s, p);
      return SEA_OK;
    }
    switch (lpr) {
      case 4: hipLaunchKernelGGL((sparse_attn_rows_kernel<T, TO, 4, 4, WP, 4>), grid, block, 0, s, p); break;
      case 8: hipLaunchKernelGGL((sparse_attn_rows_kernel<T, TO, 8, 4, WP, 8>), grid, block, 0, s, p); break;
      default: hipLaunchKernelGGL((sparse_attn_rows_kernel<T, TO, 16, 4, WP, 8>), grid, block, 0, s, p); break;
    }
    return SEA_OK;
  }
  if (p.bits) return SEA_EUNSUPPORTED;                      // (the wave-per-row kernel below has no fused form)
  const int64_t blocks = (int64_t)8 * ((NH + 7) / 8) * p.TB;
  dim3 grid((unsigned)blocks), block(256);
  switch (lpr) {
    case 4: hipLaunchKernelGGL((sparse_attn_kernel<T, TO, 4, 2, WP>), grid, block, 0, s, p); break;
    case 8: hipLaunchKernelGGL((sparse_attn_kernel<T, TO, 8, 4, WP>), grid, block, 0, s, p); break;
    case 16: hipLaunchKernelGGL((sparse_attn_kernel<T, TO, 16, 4, WP>), grid, block, 0, s, p); break;
    case 32: hipLaunchKernelGGL((sparse_attn_kernel<T, TO, 32, 4, WP>), grid, block, 0, s, p); break;
    case 64: hipLaunchKernelGGL((sparse_attn_kernel<T, TO, 64, 4, WP>), grid, block, 0, s, p); break;
    default: return SEA_EUNSUPPORTED;
  }
  return SEA_OK;
}

// Kernel choice (fixed, no run-time switches): 16-bit d = 80 -> rows80; rows of <= 16 lanes with 32-bit byte offsets ->
// lane-group-per-row kernel; anything else (fp32 d >= 128, huge strides) -> wave-per-row kernel.
template <typename T, typename TO>
static int launch_attn(const AttnParams& p, hipStream_t s) {
  return p.probs ? launch_attn_wp<T, TO, true>(p, s) : launch_attn_wp<T, TO, false>(p, s);
}

template <typename T, typename I>
static int launch_sddmm(const SddmmParams& p, hipStream_t s) {
  const int lpr = lanes_per_row(p.D, Elem<T>::VEC);
  const int64_t rows = (int64_t)p.N * p.T_dst;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  switch (lpr) {
    case 4: hipLaunchKernelGGL((csr_sddmm_kernel<T, I, 4>), grid, block, 0, s, p); break;
    case 8: hipLaunchKernelGGL((csr_sddmm_kernel<T, I, 8>), grid, block, 0, s, p); break;
    case 16: hipLaunchKernelGGL((csr_sddmm_kernel<T, I, 16>), grid, block, 0, s, p); break;
    case 32: hipLaunchKernelGGL((csr_sddmm_kernel<T, I, 32>), grid, block, 0, s, p); break;
    case 64: hipLaunchKernelGGL((csr_sddmm_kernel<T, I, 64>), grid, block, 0, s, p); break;
    default: return SEA_EUNSUPPORTED;
  }
  return SEA_OK;
}

template <typename T, typename I>
static int launch_spmm(const SpmmParams& p, hipStream_t s) {
  const int lpr = lanes_per_row(p.D, Elem<T>::VEC);
  const int NH = p.N * p.H;
  const int64_t blocks = (int64_t)8 * ((NH + 7) / 8) * p.TB;
  dim3 grid((unsigned)blocks), block(256);
  switch (lpr) {
    case 4: hipLaunchKernelGGL((csr_spmm_kernel<T, I, 4>), grid, block, 0, s, p); break;
    case 8: hipLaunchKernelGGL((csr_spmm_kernel<T, I, 8>), grid, block, 0, s, p); break;
    case 16: hipLaunchKernelGGL((csr_spmm_kernel<T, I, 16>), grid, block, 0, s, p); break;
    case 32: hipLaunchKernelGGL((csr_spmm_kernel<T, I, 32>), grid, block, 0, s, p); break;
    case 64: hipLaunchKernelGGL((csr_spmm_kernel<T, I, 64>), grid, block, 0, s, p); break;
    default: return SEA_EUNSUPPORTED;
  }
  return SEA_OK;
}

}  // namespace sea

using namespace sea;

static int check_dtype(const char* name, int dtype) {
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", name, dtype);
  return SEA_OK;
}

// bits != NULL: the fused form (sea_sparse_attention_fused) -- `col` is written by the launch, not read
static int attention_entry(const char* nm, const void* q, const void* k, const void* v, int dtype, int64_t N, int64_t H,
                           int64_t T_dst, int64_t T_src, int64_t D, const int64_t* q_strides,
                           const int64_t* k_strides, const int64_t* v_strides, const int32_t* crow,
                           const int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                           const float* row_scale, const void* avg, const int64_t* avg_strides,
                           const float* mix, void* out, int out_dtype, const int64_t* out_strides,
                           float* probs_out, int64_t probs_stride_n, const uint8_t* block_path, int flags,
                           const uint32_t* bits, int64_t T_m, int is_causal, int max_k, int write_cols, const int32_t* t_src_dev,
                           sea_stream_t stream) {
  SEA_REQUIRE(q && k && v && crow && col && head_off && out && q_strides && k_strides && v_strides && out_strides,
              SEA_EINVAL, "%s: null pointer", nm);
  if (int e = check_dtype(nm, dtype)) return e;
  SEA_REQUIRE(out_dtype == SEA_F32 || out_dtype == dtype, SEA_EUNSUPPORTED, "%s: out dtype must be fp32 or the input dtype", nm);
  SEA_REQUIRE((mix == nullptr) == (avg == nullptr), SEA_EINVAL, "%s: avg and mix go together", nm);
  SEA_REQUIRE(!avg || avg_strides, SEA_EINVAL, "%s: avg_strides is null", nm);
  SEA_REQUIRE(N > 0 && H > 0 && T_dst > 0 && T_src > 0 && D > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int path = flags & 0xff;
  SEA_REQUIRE(path == SEA_ATTN_AUTO || path == SEA_ATTN_GATHER || path == SEA_ATTN_TILE, SEA_EINVAL, "%s: bad path %d", nm, path);
  const int vec = dtype == SEA_F32 ? 4 : 8;
  SEA_REQUIRE(D % vec == 0 && D <= 64 * vec, SEA_EUNSUPPORTED, "%s: D=%lld must be a multiple of %d and <= %d", nm,
              (long long)D, vec, 64 * vec);
  SEA_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && (!avg || aligned16(avg)) && aligned16(out), SEA_EUNSUPPORTED,
              "%s: tensors must be 16-byte aligned", nm);
  SEA_REQUIRE(strides_ok(q_strides, vec) && strides_ok(k_strides, vec) && strides_ok(v_strides, vec) &&
                  strides_ok(out_strides, vec) && (!avg || strides_ok(avg_strides, vec)),
              SEA_EUNSUPPORTED, "%s: row strides must be multiples of %d elements", nm, vec);
  SEA_REQUIRE(N * H * ((T_dst + 3) / 4) < (1ll << 28), SEA_EUNSUPPORTED, "%s: grid too large", nm);
  AttnParams p;
  p.q = q; p.k = k; p.v = v;
  for (int i = 0; i < 3; ++i) {
    p.qs[i] = q_strides[i]; p.ks[i] = k_strides[i]; p.vs[i] = v_strides[i]; p.os[i] = out_strides[i];
    p.as[i] = avg ? avg_strides[i] : 0;
  }
  p.N = (int)N; p.H = (int)H; p.T_dst = (int)T_dst; p.T_src = (int)T_src; p.D = (int)D;
  p.crow = crow; p.col = col; p.col_stride_n = col_stride_n; p.head_off = head_off;
  p.row_scale = row_scale; p.avg = avg; p.mix = mix; p.out = out;
  p.probs = probs_out; p.probs_stride_n = probs_stride_n;
  p.sel = nullptr; p.sel_want = 0; p.TB16 = (int)((T_dst + 15) / 16);
  p.sel_count = nullptr; p.sel_total = (int)(N * H * p.TB16);
  // share of tile-favouring blocks above which the tile kernel takes the launch (scripts/sweep_plan_threshold.py, round 3:
  // with the plan's cut at 30 entries per staged tile the structured map gives 0.53 / 0.71 / 0.51 at d = 64 / 80 / 128 and
  // the tile kernel wins by 4 % / 13 % / loses by 13 %; the layer's own and independent rows give <= 0.33 and lose always)
  p.sel_num = D >= 128 ? 13 : 1; p.sel_den = D >= 128 ? 20 : 2;
  p.bits = bits; p.col_w = const_cast<int32_t*>(col); p.T_m = (int)T_m; p.W = (int)((H * T_m + 31) / 32);
  p.max_k = max_k; p.is_causal = is_causal;
  p.fuse_cap = 8192;                                        // entries of a block's key lists held in LDS (32 KB)
  p.write_cols = write_cols;
  p.t_src_dev = t_src_dev;
  if (t_src_dev) {
    SEA_REQUIRE(bits != nullptr && probs_out == nullptr && T_dst <= SEA_ATTN_WARM_ROWS, SEA_EUNSUPPORTED,
                "%s: the decode form takes T_dst <= %d rows per sequence, no probs_out", nm, SEA_ATTN_WARM_ROWS);
  }
  if (bits) {
    SEA_REQUIRE(path != SEA_ATTN_TILE && block_path == nullptr, SEA_EUNSUPPORTED, "%s: the fused interpolation runs on the gather kernels", nm);
    SEA_REQUIRE(T_m > 0 && T_m % 32 == 0 && max_k > 0, SEA_EUNSUPPORTED, "%s: fused interpolation needs T_m %% 32 == 0", nm);
  }
  p.TB = (int)((T_dst + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  const bool tile_ok = attn_tile_supported(dtype, (int)D, (int)T_src, p) && probs_out == nullptr;
  if (path == SEA_ATTN_TILE) {
    SEA_REQUIRE(tile_ok, SEA_EUNSUPPORTED, "%s: the tile kernel takes 16-bit data with D in {64, 80, 128} and no probs_out", nm);
  }
  int rc;
  if (path == SEA_ATTN_TILE) {
    rc = launch_attn_tile(p, dtype, out_dtype, flags, s);
    SEA_REQUIRE(rc == SEA_OK, rc, "%s: tile kernel launch parameters rejected (flags 0x%x)", nm, flags);
    SEA_CHECK_LAUNCH(nm);
    return SEA_OK;
  }
  if (path == SEA_ATTN_AUTO && block_path && tile_ok) {      // both kernels over the same rows; the plan's tile-block count
    p.sel = block_path;                                     // decides ON THE DEVICE which of them runs this launch
    p.sel_count = reinterpret_cast<const int32_t*>(block_path + plan_count_offset(N, H, p.TB16));
    p.sel_want = 1;
    rc = launch_attn_tile(p, dtype, out_dtype, flags, s);
    SEA_REQUIRE(rc == SEA_OK, rc, "%s: tile kernel launch parameters rejected (flags 0x%x)", nm, flags);
    SEA_CHECK_LAUNCH(nm);
    p.sel_want = 0;
  }
  if (dtype == SEA_F32) rc = launch_attn<float, float>(p, s);
  else if (dtype == SEA_F16) rc = out_dtype == SEA_F32 ? launch_attn<__half, float>(p, s) : launch_attn<__half, __half>(p, s);
  else rc = out_dtype == SEA_F32 ? launch_attn<__hip_bfloat16, float>(p, s) : launch_attn<__hip_bfloat16, __hip_bfloat16>(p, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported head size %lld", nm, (long long)D);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_sparse_attention_ex(const void* q, const void* k, const void* v, int dtype, int64_t N, int64_t H,
                                       int64_t T_dst, int64_t T_src, int64_t D, const int64_t* q_strides,
                                       const int64_t* k_strides, const int64_t* v_strides, const int32_t* crow,
                                       const int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                                       const float* row_scale, const void* avg, const int64_t* avg_strides,
                                       const float* mix, void* out, int out_dtype, const int64_t* out_strides,
                                       float* probs_out, int64_t probs_stride_n, const uint8_t* block_path, int flags,
                                       sea_stream_t stream) {
  return attention_entry("sea_sparse_attention", q, k, v, dtype, N, H, T_dst, T_src, D, q_strides, k_strides, v_strides, crow, col,
                         col_stride_n, head_off, row_scale, avg, avg_strides, mix, out, out_dtype, out_strides, probs_out,
                         probs_stride_n, block_path, flags, nullptr, 0, 0, 0, 1, nullptr, stream);
}

// Steps I + J in ONE launch: the gather kernel expands the selection's kept pixels to key columns itself (the emit's
// arithmetic), writes them to `col` and walks them from LDS.  crow / head_off as for sea_sparse_attention (from
// sea_csr_row_scan and the selection launch); `col` (N, col_stride_n) is OUTPUT here.
extern "C" int64_t sea_attention_few_rows(void) { return SEA_ATTN_FEW_ROWS; }

extern "C" int sea_sparse_attention_fused(const void* q, const void* k, const void* v, int dtype, int64_t N, int64_t H,
                                          int64_t T_dst, int64_t T_src, int64_t D, const int64_t* q_strides,
                                          const int64_t* k_strides, const int64_t* v_strides, const int32_t* crow,
                                          int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                                          const float* row_scale, const void* avg, const int64_t* avg_strides,
                                          const float* mix, void* out, int out_dtype, const int64_t* out_strides,
                                          float* probs_out, int64_t probs_stride_n, const uint32_t* bits, int64_t T_m,
                                          int is_causal, int max_k, int write_columns, sea_stream_t stream) {
  SEA_REQUIRE(bits != nullptr, SEA_EINVAL, "sea_sparse_attention_fused: null pointer");
  return attention_entry("sea_sparse_attention_fused", q, k, v, dtype, N, H, T_dst, T_src, D, q_strides, k_strides, v_strides, crow,
                         col, col_stride_n, head_off, row_scale, avg, avg_strides, mix, out, out_dtype, out_strides, probs_out,
                         probs_stride_n, nullptr, SEA_ATTN_GATHER, bits, T_m, is_causal, max_k, write_columns != 0, nullptr, stream);
}

// The decode form of the fused launch: a graph-replayed step (opt_generate.py:131) has static arguments, so the
// sequence length the row widths follow is read from device memory and T_src = T_cap is the fixed capacity of the K / V
// caches, with which the column ids are encoded (sea_csr_emit_at's convention).  T_dst <= 8 rows per sequence; the lane groups
// that have no row touch the K / V rows of the block's expanded lists before the walk starts.  Same arithmetic, same order
// as sea_csr_emit_at + sea_sparse_attention: the step stays bitwise the stateless forward.
extern "C" int sea_sparse_attention_fused_at(const void* q, const void* k, const void* v, int dtype, int64_t N, int64_t H,
                                             int64_t T_dst, int64_t T_cap, int64_t D, const int64_t* q_strides,
                                             const int64_t* k_strides, const int64_t* v_strides, const int32_t* crow,
                                             int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                                             const float* row_scale, const void* avg, const int64_t* avg_strides,
                                             const float* mix, void* out, int out_dtype, const int64_t* out_strides,
                                             const uint32_t* bits, int64_t T_m, const int32_t* t_src_dev, int is_causal,
                                             int max_k, int write_columns, sea_stream_t stream) {
  SEA_REQUIRE(bits != nullptr && t_src_dev != nullptr, SEA_EINVAL, "sea_sparse_attention_fused_at: null pointer");
  return attention_entry("sea_sparse_attention_fused_at", q, k, v, dtype, N, H, T_dst, T_cap, D, q_strides, k_strides, v_strides,
                         crow, col, col_stride_n, head_off, row_scale, avg, avg_strides, mix, out, out_dtype, out_strides, nullptr,
                         0, nullptr, SEA_ATTN_GATHER, bits, T_m, is_causal, max_k, write_columns != 0, t_src_dev, stream);
}

extern "C" int sea_sparse_attention(const void* q, const void* k, const void* v, int dtype, int64_t N, int64_t H,
                                    int64_t T_dst, int64_t T_src, int64_t D, const int64_t* q_strides,
                                    const int64_t* k_strides, const int64_t* v_strides, const int32_t* crow,
                                    const int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                                    const float* row_scale, const void* avg, const int64_t* avg_strides,
                                    const float* mix, void* out, int out_dtype, const int64_t* out_strides,
                                    sea_stream_t stream) {
  return sea_sparse_attention_ex(q, k, v, dtype, N, H, T_dst, T_src, D, q_strides, k_strides, v_strides, crow, col,
                                 col_stride_n, head_off, row_scale, avg, avg_strides, mix, out, out_dtype, out_strides,
                                 nullptr, 0, nullptr, SEA_ATTN_AUTO, stream);
}

extern "C" int sea_attention_plan(const uint32_t* bits, int64_t N, int64_t H, int64_t T_dst, int64_t T_src, int64_t T_m,
                                  int is_causal, float entries_per_tile, uint8_t* block_path, sea_stream_t stream) {
  const char* nm = "sea_attention_plan";
  SEA_REQUIRE(bits && block_path, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(N > 0 && H > 0 && T_dst > 0 && T_src > 0 && T_m > 0, SEA_EINVAL, "%s: bad shape", nm);
  SEA_REQUIRE(hipMemsetAsync(block_path + plan_count_offset(N, H, (T_dst + 15) / 16), 0, 4, (hipStream_t)stream) == hipSuccess,
              SEA_ELAUNCH, "%s: memset failed", nm);
  const int rc = launch_attn_plan(bits, (int)N, (int)H, (int)T_dst, (int)T_src, (int)T_m, is_causal,
                                  entries_per_tile > 0.f ? entries_per_tile : 30.0f, block_path, (hipStream_t)stream);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: needs T_m %% 32 == 0, H <= 64 and H * T_m <= 32768", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int64_t sea_sparse_attention_bytes(int64_t Z, int64_t N, int64_t H, int64_t T_dst, int64_t D, int elem_bytes) {
  return Z * (2 * D * elem_bytes + 4) + N * H * T_dst * (2 * D * elem_bytes + 4);
}

extern "C" int sea_csr_sddmm(const void* q, const void* k, int dtype, int64_t N, int64_t H, int64_t T_dst, int64_t T_src,
                             int64_t D, const int64_t* q_strides, const int64_t* k_strides, const void* crow,
                             const void* col, int idx_bytes, int64_t col_stride_n, float* values, sea_stream_t stream) {
  const char* nm = "sea_csr_sddmm";
  SEA_REQUIRE(q && k && crow && col && values && q_strides && k_strides, SEA_EINVAL, "%s: null pointer", nm);
  if (int e = check_dtype(nm, dtype)) return e;
  SEA_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SEA_EINVAL, "%s: idx_bytes must be 4 or 8", nm);
  const int vec = dtype == SEA_F32 ? 4 : 8;
  SEA_REQUIRE(D % vec == 0 && D <= 64 * vec, SEA_EUNSUPPORTED, "%s: D=%lld must be a multiple of %d", nm, (long long)D, vec);
  SEA_REQUIRE(aligned16(q) && aligned16(k) && strides_ok(q_strides, vec) && strides_ok(k_strides, vec), SEA_EUNSUPPORTED,
              "%s: q/k rows must be 16-byte aligned", nm);
  SddmmParams p;
  p.q = q; p.k = k;
  for (int i = 0; i < 3; ++i) { p.qs[i] = q_strides[i]; p.ks[i] = k_strides[i]; }
  p.N = (int)N; p.H = (int)H; p.T_dst = (int)T_dst; p.T_src = (int)T_src; p.D = (int)D;
  p.crow = crow; p.col = col; p.col_stride_n = col_stride_n; p.values = values;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (idx_bytes == 4) {
    if (dtype == SEA_F32) rc = launch_sddmm<float, int32_t>(p, s);
    else if (dtype == SEA_F16) rc = launch_sddmm<__half, int32_t>(p, s);
    else rc = launch_sddmm<__hip_bfloat16, int32_t>(p, s);
  } else {
    if (dtype == SEA_F32) rc = launch_sddmm<float, int64_t>(p, s);
    else if (dtype == SEA_F16) rc = launch_sddmm<__half, int64_t>(p, s);
    else rc = launch_sddmm<__hip_bfloat16, int64_t>(p, s);
  }
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported head size", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_csr_spmm(const float* values, const void* v, int dtype, int64_t N, int64_t H, int64_t T_dst,
                            int64_t T_src, int64_t D, const int64_t* v_strides, const void* crow, const void* col,
                            int idx_bytes, int64_t col_stride_n, const int32_t* head_off, float* out,
                            sea_stream_t stream) {
  const char* nm = "sea_csr_spmm";
  SEA_REQUIRE(values && v && crow && col && head_off && out && v_strides, SEA_EINVAL, "%s: null pointer", nm);
  if (int e = check_dtype(nm, dtype)) return e;
  SEA_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SEA_EINVAL, "%s: idx_bytes must be 4 or 8", nm);
  const int vec = dtype == SEA_F32 ? 4 : 8;
  SEA_REQUIRE(D % vec == 0 && D <= 64 * vec, SEA_EUNSUPPORTED, "%s: D=%lld must be a multiple of %d", nm, (long long)D, vec);
  SEA_REQUIRE(aligned16(v) && aligned16(out) && strides_ok(v_strides, vec), SEA_EUNSUPPORTED,
              "%s: v rows must be 16-byte aligned", nm);
  SpmmParams p;
  p.values = values; p.v = v;
  for (int i = 0; i < 3; ++i) p.vs[i] = v_strides[i];
  p.N = (int)N; p.H = (int)H; p.T_dst = (int)T_dst; p.T_src = (int)T_src; p.D = (int)D;
  p.crow = crow; p.col = col; p.col_stride_n = col_stride_n; p.head_off = head_off; p.out = out;
  p.TB = (int)((T_dst + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (idx_bytes == 4) {
    if (dtype == SEA_F32) rc = launch_spmm<float, int32_t>(p, s);
    else if (dtype == SEA_F16) rc = launch_spmm<__half, int32_t>(p, s);
    else rc = launch_spmm<__hip_bfloat16, int32_t>(p, s);
  } else {
    if (dtype == SEA_F32) rc = launch_spmm<float, int64_t>(p, s);
    else if (dtype == SEA_F16) rc = launch_spmm<__half, int64_t>(p, s);
    else rc = launch_spmm<__hip_bfloat16, int64_t>(p, s);
  }
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported head size", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
