// Tiled sparse attention on the matrix cores (gfx950, wave64): shared key rows are fetched ONCE per block of query
// rows instead of once per (row, key) entry.
//
// Why: in the flat CSR of SEA an entry is a (query row, key) pair, but entries come in runs -- an interpolated pixel
// is a run of up to ceil(w_t / T_M) consecutive keys (causal_resize_m_to_t.py:565-569) -- and neighbouring query rows
// keep largely the same pixels.  The reference's own SpMM is tile shaped for that reason
// (flat_csr_sdbmm.py:141-313, MAX_ROW_T sizing :382-388).  The gather kernels of sea_attn.hip pull 2 x D x s bytes from
// L2 for every entry; here a wave owns 16*RT consecutive query rows of one (n, h) and works on 16-key tiles:
//
//   1. the rows' CSR entries of head h are turned into a key bitmap in LDS (one bit per (row, key), ds_or), window by
//      window of KW keys; OR-ing the rows' words gives the list of 16-key tiles that hold any kept key;
//   2. per pair of listed tiles (32 key slots): S^T = K_tile . Q^T on v_mfma_f32_16x16x32 (K fragments straight from
//      global: lane = key, 16 contiguous bytes; Q^T fragments live in registers for the whole kernel), the bitmap
//      masks the scores, online softmax with the row on the lane (4 lanes x 4 keys per row and tile),
//      P^T split into two 16-bit terms (bf16 data: 16 significand bits; fp16 data: one term), and
//      O^T += V^T . P^T on the MFMA with V staged row-major in LDS (coalesced 16-byte global loads) and read
//      k-major by ds_read_b64_tr_b16;
//   3. epilogue as in the gather kernels: 1/l, row scale, mix with the cumulative average, strided store.
//
// Every wave is independent (own LDS region, no workgroup barrier): LDS operations of one wave execute in order, so
// the zero -> ds_or -> read sequence and the single V staging buffer need no synchronisation beyond program order.
//
// Contract differences from the gather kernels: a (row, column) pair must not occur twice in the CSR (a bitmap cannot
// count) -- true of every CSR the interpolation emits; 16-bit data, D in {64, 80, 128}.
#include "sea_attn.hpp"
#include <type_traits>

namespace sea {

typedef __attribute__((ext_vector_type(8))) __bf16 at_bf8;
typedef __attribute__((ext_vector_type(8))) _Float16 at_h8;
typedef __attribute__((ext_vector_type(2))) __bf16 at_bf2;
typedef __attribute__((ext_vector_type(2))) _Float16 at_h2;
typedef __attribute__((ext_vector_type(2))) float at_f2;
typedef __attribute__((ext_vector_type(4))) float at_f4;
typedef __attribute__((ext_vector_type(4))) short at_s4;

template <typename T> struct TileT;
template <> struct TileT<__hip_bfloat16> {
  static constexpr bool SPLIT = true;     // P enters the matrix cores as hi + lo (2 x 8 significand bits)
  __device__ static inline at_f4 mfma(const uint4& a, const uint4& b, at_f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, a), __builtin_bit_cast(at_bf8, b), c, 0, 0, 0);
  }
  __device__ static inline uint32_t pack(float a, float b) {          // RNE, low half = a
    const at_f2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, at_bf2));
  }
  __device__ static inline float lo_val(uint32_t pk) { return __uint_as_float(pk << 16); }
  __device__ static inline float hi_val(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }
};
template <> struct TileT<__half> {
  static constexpr bool SPLIT = false;    // 11 significand bits: 2.4e-4 rms on p <= 1, inside the 1e-3 bar
  __device__ static inline at_f4 mfma(const uint4& a, const uint4& b, at_f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(at_h8, a), __builtin_bit_cast(at_h8, b), c, 0, 0, 0);
  }
  __device__ static inline uint32_t pack(float a, float b) {
    const at_f2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, at_h2));
  }
  __device__ static inline float lo_val(uint32_t) { return 0.f; }
  __device__ static inline float hi_val(uint32_t) { return 0.f; }
};

__device__ inline float xor16_max(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int D> struct TileGeom {
  static constexpr int KK = (D + 31) / 32;                 // 32-wide k-steps of K . Q^T
  static constexpr int MT = D / 16;                        // 16-wide feature tiles of O^T
  static constexpr int CH = D / 8;                         // 16-byte chunks per K / V row
  // V staging rows: a stride of 40 (mod 64) words keeps the transposing reads of 8 consecutive rows on disjoint banks
  static constexpr int VST = (D == 80) ? 160 : D * 2 + 32; // bytes
  static constexpr int VLD = (32 * CH) / 64;               // 16-byte V loads per lane and tile pair
  static_assert((32 * CH) % 64 == 0 && D % 16 == 0, "head size");
};

// bytes of LDS one wave needs: key bitmap [ROWS][KW/32 + 1] + tile list [KW/16] + V stage [32][VST]
__host__ __device__ inline int tile_wave_lds(int D, int RT, int KW) {
  const int vst = (D == 80) ? 160 : D * 2 + 32;
  int b = 16 * RT * (KW / 32 + 1) * 4 + (KW / 16) * 2;
  b = (b + 15) & ~15;
  return b + 32 * vst;
}

template <typename T, typename TO> __device__ inline void store4(TO* dst, const float* o);
template <> __device__ inline void store4<__hip_bfloat16, float>(float* dst, const float* o) { *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]); }
template <> __device__ inline void store4<__half, float>(float* dst, const float* o) { *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]); }
template <> __device__ inline void store4<__hip_bfloat16, __hip_bfloat16>(__hip_bfloat16* dst, const float* o) {
  *reinterpret_cast<uint2*>(dst) = make_uint2(TileT<__hip_bfloat16>::pack(o[0], o[1]), TileT<__hip_bfloat16>::pack(o[2], o[3]));
}
template <> __device__ inline void store4<__half, __half>(__half* dst, const float* o) {
  *reinterpret_cast<uint2*>(dst) = make_uint2(TileT<__half>::pack(o[0], o[1]), TileT<__half>::pack(o[2], o[3]));
}
template <typename T> __device__ inline void unpack4(const uint2& r, float* f);
template <> __device__ inline void unpack4<__hip_bfloat16>(const uint2& r, float* f) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <> __device__ inline void unpack4<__half>(const uint2& r, float* f) {
  const float2 a = __half22float2(__builtin_bit_cast(__half2, r.x)), b = __half22float2(__builtin_bit_cast(__half2, r.y));
  f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y;
}

// T: 16-bit data type; TO: output type; D: head size; RT: 16-row tiles per wave; NW: waves per workgroup
template <typename T, typename TO, int D, int RT, int NW>
__global__ __launch_bounds__(NW * 64) void sparse_attn_tile_kernel(AttnParams p, int KW, int wave_lds) {
  using G = TileGeom<D>;
  using X = TileT<T>;
  constexpr int ROWS = 16 * RT, KK = G::KK, MT = G::MT, CH = G::CH, VST = G::VST, VLD = G::VLD;
  constexpr bool SPLIT = X::SPLIT;   // (one bf16 term for 16-bit outputs was tried: no faster -- the kernel is not VALU bound -- and one ulp worse)
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  int pair, tb;
  if (kernel_is_idle(p) || !map_block(p.N * p.H, p.TB, &pair, &tb)) return;
  const int n = pair / p.H, h = pair - n * p.H;
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int t0 = (tb * NW + wave) * ROWS;
  if (t0 >= p.T_dst) return;                               // wave-uniform; the kernel has no workgroup barrier

  const int WPR = KW >> 5, BST = WPR + 1;                  // bitmap words per row, padded row stride (banks)
  char* wbase = at_smem + wave * wave_lds;
  uint32_t* bm = reinterpret_cast<uint32_t*>(wbase);                       // [ROWS][BST]
  unsigned short* tl = reinterpret_cast<unsigned short*>(bm + ROWS * BST); // [KW / 16]
  char* vs = wbase + ((ROWS * BST * 4 + (KW / 16) * 2 + 15) & ~15);        // [32][VST]

  const char* kbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1]);
  const char* vbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1]);
  const uint32_t kst = (uint32_t)p.ks[2] * 2u, vst = (uint32_t)p.vs[2] * 2u;   // bytes (launcher: T_src * stride < 2^31)
  const int32_t* col = p.col + n * p.col_stride_n;
  const int hcol = h * p.T_src;
  const int klast = p.T_src - 1;

  // Q^T fragments (B operand: lane = query row li, k = feature 32 kk + 8 g + j), kept for the whole kernel
  uint4 qf[RT][KK];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int t = min(t0 + 16 * rt + li, p.T_dst - 1);
    const T* qrow = reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] + (int64_t)t * p.qs[2];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int d0 = 32 * kk + 8 * g;
      qf[rt][kk] = (d0 < D) ? *reinterpret_cast<const uint4*>(qrow + d0) : make_uint4(0, 0, 0, 0);
    }
  }
  // entry range of row `lane` (lanes >= ROWS and rows past T_dst: empty)
  int rbeg = 0, rend = 0;
  if (lane < ROWS && t0 + lane < p.T_dst) {
    const int t = t0 + lane;
    const int row_beg = p.crow[(int64_t)n * (p.T_dst + 1) + t];
    const int32_t* ho = p.head_off + ((int64_t)n * p.T_dst + t) * (p.H + 1);
    rbeg = row_beg + ho[h];
    rend = row_beg + ho[h + 1];
  }

  float m[RT], l[RT];
  at_f4 acc[RT][MT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    m[rt] = -INFINITY; l[rt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[rt][mt] = at_f4{0.f, 0.f, 0.f, 0.f};
  }

  int kmax = -1;                                           // largest key of the block (found by the first scan)
  for (int W0 = 0; W0 == 0 || W0 <= kmax; W0 += KW) {
    // ---- 1. key bitmap of the window ----------------------------------------------------------------------------
    for (int i = lane; i < ROWS * BST; i += 64) bm[i] = 0u;
    int kmx = -1;
#pragma unroll
    for (int r0 = 0; r0 < ROWS; r0 += 8) {
      int cv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                        // eight rows' first 64 entries in flight together
        const int b = __builtin_amdgcn_readlane(rbeg, r0 + u), e = __builtin_amdgcn_readlane(rend, r0 + u);
        const int i = b + lane;
        cv[u] = i < e ? col[i] - hcol : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = __builtin_amdgcn_readlane(rbeg, r0 + u), e = __builtin_amdgcn_readlane(rend, r0 + u);
        uint32_t* row = bm + (r0 + u) * BST;
        int c = cv[u];
        for (int i = b + lane;;) {
          if (c >= 0) {
            kmx = max(kmx, c);
            const uint32_t kw = (uint32_t)(c - W0);
            if (kw < (uint32_t)KW) atomicOr(row + (kw >> 5), 1u << (kw & 31));
          }
          i += 64;
          if (!__builtin_amdgcn_readfirstlane((int)(__ballot(i < e) != 0ull))) break;   // rows longer than 64 entries: rare
          c = i < e ? col[i] - hcol : -1;
        }
      }
    }
    if (W0 == 0) kmax = wave_max(kmx);
    __builtin_amdgcn_wave_barrier();

    // ---- 2. list of the 16-key tiles that hold a kept key (ascending), with the row tiles that use them -----------
    int NT = 0;
    for (int w0 = 0; w0 < WPR; w0 += 64) {
      const int w = w0 + lane;
      uint32_t u[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        u[rt] = 0u;
        if (w < WPR) {
#pragma unroll
          for (int r = 0; r < 16; ++r) u[rt] |= bm[(16 * rt + r) * BST + w];
        }
      }
      uint32_t ua = u[0];
      if (RT > 1) ua |= u[RT - 1];
      const bool lo = (ua & 0xffffu) != 0u, hi = (ua >> 16) != 0u;
      const uint64_t blo = __ballot(lo), bhi = __ballot(hi);
      const uint64_t lt = (1ull << lane) - 1ull;
      const int pos = NT + __popcll(blo & lt) + __popcll(bhi & lt);
      int rml = 0, rmh = 0;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        rml |= ((u[rt] & 0xffffu) != 0u) << rt;
        rmh |= ((u[rt] >> 16) != 0u) << rt;
      }
      if (lo) tl[pos] = (unsigned short)((2 * w) | (rml << 9));
      if (hi) tl[pos + (lo ? 1 : 0)] = (unsigned short)((2 * w + 1) | (rmh << 9));
      NT += __popcll(blo) + __popcll(bhi);
    }
    __builtin_amdgcn_wave_barrier();
    if (NT == 0) continue;

    // ---- 3. tile pairs ------------------------------------------------------------------------------------------------
    struct PairRegs { uint4 kf[2][KK], vr[VLD]; };
    auto load_pair = [&](int i) -> PairRegs {              // global loads of pair i: K fragments + the pair's V rows
      PairRegs r;
      const int ea = __builtin_amdgcn_readfirstlane((int)tl[i]);
      const int eb = __builtin_amdgcn_readfirstlane((int)tl[min(i + 1, NT - 1)]);
      const int ka0 = W0 + 16 * (ea & 0x1ff), kb0 = W0 + 16 * (eb & 0x1ff);
      const uint32_t oa = (uint32_t)min(ka0 + li, klast) * kst, ob = (uint32_t)min(kb0 + li, klast) * kst;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int d0 = 32 * kk + 8 * g;
        if (d0 < D) {
          r.kf[0][kk] = *reinterpret_cast<const uint4*>(kbase + (oa + (uint32_t)d0 * 2u));
          r.kf[1][kk] = *reinterpret_cast<const uint4*>(kbase + (ob + (uint32_t)d0 * 2u));
        } else {
          r.kf[0][kk] = make_uint4(0, 0, 0, 0);
          r.kf[1][kk] = make_uint4(0, 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < VLD; ++u) {
        const int idx = lane + 64 * u, ks = idx / CH, ch = idx - ks * CH;
        const int key = min((ks < 16 ? ka0 : kb0 - 16) + ks, klast);
        r.vr[u] = *reinterpret_cast<const uint4*>(vbase + ((uint32_t)key * vst + (uint32_t)ch * 16u));
      }
      return r;
    };
    PairRegs cur = load_pair(0);
    for (int i = 0; i < NT; i += 2) {
      const int ea = __builtin_amdgcn_readfirstlane((int)tl[i]);
      const bool has_b = i + 1 < NT;
      const int eb = __builtin_amdgcn_readfirstlane((int)tl[has_b ? i + 1 : i]);
      const int ca = ea & 0x1ff, cb = eb & 0x1ff;

      // S^T = K_tile . Q^T : lane (li, g) gets the scores of query row li against keys 4 g .. 4 g + 3 of the tile
      at_f4 s[RT][2];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          at_f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) a = X::mfma(cur.kf[tt][kk], qf[rt][kk], a);
          s[rt][tt] = a;
        }
      // this pair's V rows -> LDS (row = key slot: tile a keys 0..15, then tile b keys 0..15)
#pragma unroll
      for (int u = 0; u < VLD; ++u) {
        const int idx = lane + 64 * u, ks = idx / CH, ch = idx - ks * CH;
        *reinterpret_cast<uint4*>(vs + ks * VST + ch * 16) = cur.vr[u];
      }
      if (i + 2 < NT) cur = load_pair(i + 2);                       // next pair's loads fly under the softmax and P.V below

      // mask + online softmax, one query row per lane (its 4 + 4 keys of the pair; the row's other keys sit in g' != g)
      uint4 ph[RT], pl[RT];
      float pmax[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const uint32_t* row = bm + (16 * rt + li) * BST;
        const uint32_t na = (row[ca >> 1] >> (16 * (ca & 1) + 4 * g)) & 0xfu;
        const uint32_t nb = has_b ? ((row[cb >> 1] >> (16 * (cb & 1) + 4 * g)) & 0xfu) : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[rt][0][j] = (na >> j) & 1u ? s[rt][0][j] : -INFINITY;
          s[rt][1][j] = (nb >> j) & 1u ? s[rt][1][j] : -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(s[rt][0][0], s[rt][0][1]), fmaxf(s[rt][0][2], s[rt][0][3])),
                         fmaxf(fmaxf(s[rt][1][0], s[rt][1][1]), fmaxf(s[rt][1][2], s[rt][1][3])));
        mx = xor16_max(mx);
        mx = xor32_max(mx);
        pmax[rt] = mx;
      }
      bool grow = false;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) grow = grow || (pmax[rt] > m[rt]);
      if (__ballot(grow) != 0ull) {                        // some row's maximum grew: rescale (wave-uniform branch)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float mn = fmaxf(m[rt], pmax[rt]);
          const float alpha = (mn == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m[rt] - mn) * LOG2E);
          l[rt] *= alpha;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc[rt][mt][0] *= alpha; acc[rt][mt][1] *= alpha; acc[rt][mt][2] *= alpha; acc[rt][mt][3] *= alpha;
          }
          m[rt] = mn;
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float ms2 = (m[rt] == -INFINITY) ? 0.f : -m[rt] * LOG2E;    // exp(s - m) = exp2(s * log2e - m * log2e): fma + exp
        float pv[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pv[j] = __builtin_amdgcn_exp2f(fmaf(s[rt][0][j], LOG2E, ms2));
          pv[4 + j] = __builtin_amdgcn_exp2f(fmaf(s[rt][1][j], LOG2E, ms2));
        }
        l[rt] += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
        // P^T fragment (B operand): k = 8 g + j <-> j < 4: tile a key 4 g + j, j >= 4: tile b key 4 g + j - 4
        ph[rt] = make_uint4(X::pack(pv[0], pv[1]), X::pack(pv[2], pv[3]), X::pack(pv[4], pv[5]), X::pack(pv[6], pv[7]));
        if (SPLIT) {
          pl[rt] = make_uint4(X::pack(pv[0] - X::lo_val(ph[rt].x), pv[1] - X::hi_val(ph[rt].x)),
                              X::pack(pv[2] - X::lo_val(ph[rt].y), pv[3] - X::hi_val(ph[rt].y)),
                              X::pack(pv[4] - X::lo_val(ph[rt].z), pv[5] - X::hi_val(ph[rt].z)),
                              X::pack(pv[6] - X::lo_val(ph[rt].w), pv[7] - X::hi_val(ph[rt].w)));
        }
      }
      // O^T += V^T . P^T : V^T fragments (A operand: lane = feature 16 mt + li, k as above) by transposing LDS reads
      const char* va = vs + (4 * g + (li >> 2)) * VST + (li & 3) * 8;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const at_s4 xa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_s4 __attribute__((address_space(3)))*)(va + mt * 32));
        const at_s4 xb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_s4 __attribute__((address_space(3)))*)(va + 16 * VST + mt * 32));
        const uint2 ua2 = __builtin_bit_cast(uint2, xa), ub2 = __builtin_bit_cast(uint2, xb);
        const uint4 vf = make_uint4(ua2.x, ua2.y, ub2.x, ub2.y);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc[rt][mt] = X::mfma(vf, ph[rt], acc[rt][mt]);
          if (SPLIT) acc[rt][mt] = X::mfma(vf, pl[rt], acc[rt][mt]);
        }
      }
    }
  }

  // ---- epilogue: 1 / l, row scale, mix with the cumulative average, store (lane: row li, features 16 mt + 4 g ..) ----
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const float ls = xor32_sum(xor16_sum(l[rt]));
    const int t = t0 + 16 * rt + li;
    if (t < p.T_dst) {
      const int64_t ridx = ((int64_t)n * p.H + h) * p.T_dst + t;
      float scale = (ls > 0.f) ? (1.0f / ls) : 0.f;
      if (p.row_scale) scale *= p.row_scale[ridx];
      const float a = p.mix ? p.mix[ridx] : 1.f;
      const T* ap = reinterpret_cast<const T*>(p.avg) + n * p.as[0] + h * p.as[1] + (int64_t)t * p.as[2];
      TO* op = reinterpret_cast<TO*>(p.out) + n * p.os[0] + h * p.os[1] + (int64_t)t * p.os[2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int d0 = 16 * mt + 4 * g;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (ls > 0.f) ? acc[rt][mt][j] * scale : 0.f;
        if (p.mix) {
          float af[4];
          unpack4<T>(*reinterpret_cast<const uint2*>(ap + d0), af);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = o[j] * a + (1.0f - a) * af[j];
        }
        store4<T, TO>(op + d0, o);
      }
    }
  }
}

// ---- per-block dispatch plan -----------------------------------------------------------------------------------------
// Which kernel should own a 16-row block of one head?  The tile kernel's time goes with the 16-key tiles it stages, the
// gather kernels' with the entries they walk; measured on MI355X (OPT-1.3B / 2.7B / LLaMA-13B shapes, three kinds of map,
// scripts/time_attn_paths.py, scripts/sweep_plan_cut.py) the two cross at about 30 entries per staged tile.  Both figures are estimated from the
// kept-PIXEL bit masks of the selection kernel (no CSR walk): entries = sum over rows of kept pixels x pixel width,
// tiles = union pixels x width / 16 plus about two tiles of slack per run of adjacent union pixels (run ends are not
// tile aligned and pixel boundaries drift by up to 15 keys over 16 rows), capped by the tiles a row can see.
struct PlanParams {
  const uint32_t* bits;    // (N, T_dst, W)
  uint8_t* sel;            // (N, H, TB16)
  int32_t* count;          // blocks given to the tile kernel (zeroed by the caller)
  int N, H, T_dst, T_src, T_m, W, TB16, causal;
  float entries_per_tile;
};

__global__ __launch_bounds__(256) void attn_plan_kernel(PlanParams p) {
  __shared__ uint32_t s_or[1024];
  __shared__ int s_cnt[3 * 64];                            // per head: union pixels, kept pixels over the rows, runs
  const int blk = blockIdx.x;
  const int n = blk / p.TB16, t16 = blk - n * p.TB16;
  const int t0 = t16 * 16, rows = min(16, p.T_dst - t0);
  const int WPH = p.T_m >> 5;                              // words per head
  for (int i = threadIdx.x; i < 3 * p.H; i += 256) s_cnt[i] = 0;
  const uint32_t* b0 = p.bits + ((int64_t)n * p.T_dst + t0) * p.W;
  uint32_t orw[4];
  int cnt[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int w = threadIdx.x + 256 * u;
    orw[u] = 0u; cnt[u] = 0;
    if (w < p.W) {
      for (int r = 0; r < rows; ++r) {
        const uint32_t x = b0[(int64_t)r * p.W + w];
        orw[u] |= x;
        cnt[u] += __popc(x);
      }
      s_or[w] = orw[u];
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int w = threadIdx.x + 256 * u;
    if (w < p.W && orw[u] != 0u) {
      const int hh = w / WPH, wi = w - hh * WPH;
      const uint32_t carry = wi ? (s_or[w - 1] >> 31) : 0u;
      const int runs = __popc(orw[u] & ~((orw[u] << 1) | carry));
      atomicAdd(&s_cnt[3 * hh], __popc(orw[u]));
      atomicAdd(&s_cnt[3 * hh + 1], cnt[u]);
      atomicAdd(&s_cnt[3 * hh + 2], runs);
    }
  }
  __syncthreads();
  for (int hh = threadIdx.x; hh < p.H; hh += 256) {
    const int w_last = p.causal ? (p.T_src - p.T_dst + t0 + rows) : p.T_src;      // keys the block's last row sees
    const float pw = (float)w_last / (float)p.T_m;                                // keys per pixel
    const float entries = (float)s_cnt[3 * hh + 1] * fminf(pw, 64.f);
    float tiles = (float)s_cnt[3 * hh] * pw * (1.0f / 16.0f) + 2.0f * (float)s_cnt[3 * hh + 2];
    tiles = fminf(tiles, (float)((w_last + 15) / 16 + 1));
    const bool tile = entries >= p.entries_per_tile * tiles && tiles > 0.f;
    p.sel[((int64_t)n * p.H + hh) * p.TB16 + t16] = tile ? 1 : 0;
    if (tile) atomicAdd(p.count, 1);
  }
}

int launch_attn_plan(const uint32_t* bits, int N, int H, int T_dst, int T_src, int T_m, int causal, float entries_per_tile,
                     uint8_t* sel, hipStream_t s) {
  PlanParams p;
  p.bits = bits; p.sel = sel; p.count = reinterpret_cast<int32_t*>(sel + plan_count_offset(N, H, (T_dst + 15) / 16)); p.N = N; p.H = H; p.T_dst = T_dst; p.T_src = T_src; p.T_m = T_m;
  p.W = (H * T_m + 31) / 32; p.TB16 = (T_dst + 15) / 16; p.causal = causal;
  p.entries_per_tile = entries_per_tile;
  if (T_m % 32 != 0 || p.W > 1024 || H > 64) return SEA_EUNSUPPORTED;
  hipLaunchKernelGGL(attn_plan_kernel, dim3((unsigned)(N * p.TB16)), dim3(256), 0, s, p);
  return SEA_OK;
}

// ---- host side ---------------------------------------------------------------------------------------------------
bool attn_tile_supported(int dtype, int D, int T_src, const AttnParams& p) {
  if (dtype != SEA_F16 && dtype != SEA_BF16) return false;
  if (D != 64 && D != 80 && D != 128) return false;
  return (int64_t)T_src * p.ks[2] * 2 < (1ll << 31) && (int64_t)T_src * p.vs[2] * 2 < (1ll << 31);
}

template <typename T, typename TO, int D, int RT, int NW>
static int launch_tile_inst(AttnParams p, int KW, hipStream_t s) {
  const int wl = tile_wave_lds(D, RT, KW);
  const int lds = wl * NW;
  static int lds_set[64] = {0};                             // per instantiation and device; grows monotonically (benign if two threads race)
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  if (lds > lds_set[dev_ & 63]) {
    SEA_MAX_LDS((sparse_attn_tile_kernel<T, TO, D, RT, NW>), lds);
    lds_set[dev_ & 63] = lds;
  }
  const int rpb = NW * 16 * RT;
  p.TB = (p.T_dst + rpb - 1) / rpb;
  const int64_t blocks = (int64_t)8 * ((p.N * p.H + 7) / 8) * p.TB;
  if (blocks >= (1ll << 31)) return SEA_EUNSUPPORTED;
  hipLaunchKernelGGL((sparse_attn_tile_kernel<T, TO, D, RT, NW>), dim3((unsigned)blocks), dim3(NW * 64), lds, s, p, KW, wl);
  return SEA_OK;
}

template <typename T, typename TO, int D>
static int launch_tile_d(const AttnParams& p, int rt, int KW, hipStream_t s) {
  if (rt == 2) return launch_tile_inst<T, TO, D, 2, 2>(p, KW, s);
  return launch_tile_inst<T, TO, D, 1, 2>(p, KW, s);
}

template <typename T, typename TO>
static int launch_tile_t(const AttnParams& p, int rt, int KW, hipStream_t s) {
  switch (p.D) {
    case 64: return launch_tile_d<T, TO, 64>(p, rt, KW, s);
    case 80: return launch_tile_d<T, TO, 80>(p, rt, KW, s);
    case 128: return launch_tile_d<T, TO, 128>(p, rt, KW, s);
    default: return SEA_EUNSUPPORTED;
  }
}

int launch_attn_tile(const AttnParams& p, int dtype, int out_dtype, int flags, hipStream_t s) {
  // flags (SEA_ATTN_* in sea_hip.h): bits 8..11 = row tiles per wave (0: default), bits 12..15 = log2 of the key window
  int rt = (flags >> 8) & 0xf;
  if (rt == 0 || p.sel) rt = 1;                            // a dispatch plan speaks of 16-row blocks
  if (rt != 1 && rt != 2) return SEA_EINVAL;
  const int kwl = (flags >> 12) & 0xf;
  int KW = kwl ? (1 << kwl) : 2048;
  if (KW < 64 || KW > 4096) return SEA_EINVAL;
  while (KW / 2 >= 64 && KW / 2 >= p.T_src) KW /= 2;       // short sequences: no wider than needed
  if (dtype == SEA_BF16)
    return out_dtype == SEA_F32 ? launch_tile_t<__hip_bfloat16, float>(p, rt, KW, s)
                                : launch_tile_t<__hip_bfloat16, __hip_bfloat16>(p, rt, KW, s);
  return out_dtype == SEA_F32 ? launch_tile_t<__half, float>(p, rt, KW, s) : launch_tile_t<__half, __half>(p, rt, KW, s);
}

}  // namespace sea
