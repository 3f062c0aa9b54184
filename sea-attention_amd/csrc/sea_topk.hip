// Grouped top-k selection + mask interpolation to flat CSR, hand-written for gfx950.
//
// Replaces (reference, src/models/perlin_attention/):
//   attention.py:774-947                     full torch.sort + int64 rank scatter + compare
//   ops/kernels/causal_resize_m_to_t.py:631-762   n_pixels / cumsum / nonzero / __scan_col_4_compute
//
// Pipeline (three launches, no host synchronisation):
//   topk_select_kernel   one 256-thread workgroup per (n, t) row: the H*T_m pooled probabilities of
//                        the row live in registers (EPT per thread); an adaptive MSB radix select with
//                        an LDS histogram finds the K-th largest key; survivors are written as a bit
//                        mask, and the per-head / per-row entry counts of the interpolated row follow
//                        from the closed-form pixel widths.
//   row_scan_kernel      crow = exclusive scan of the row counts (one workgroup per batch item).
//   csr_emit_kernel      one workgroup per row: expand every kept pixel to its key columns.
//
// Everything here is HBM-bound integer/compare work; LDS holds the score histogram only.
#include "sea_common.hpp"
#include "sea_tail.hpp"
#include "sea_convfrag.hpp"


#ifdef SEA_STAMP
__device__ unsigned long long sea_dbg[16];
#define STAMP(i) do { if (threadIdx.x == 0) { unsigned long long _t = __builtin_amdgcn_s_memtime(); \
  atomicAdd(&sea_dbg[i], _t - _tprev); _tprev = _t; } } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

namespace sea {

constexpr int TK_THREADS = 256;
constexpr int TK_WAVES = TK_THREADS / WAVE;
constexpr int TK_MAX_BINS = 2048;  // 11-bit digits

// float -> order-preserving uint32 (larger float <=> larger key); -0.0 is folded onto +0.0
__device__ inline uint32_t f2key(float f) {
  f = f + 0.0f;
  uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

struct TopkParams {
  const void* src;  // probs or mask
  int64_t sn, sh, st;
  int H, T_dst, T_m, T_src;
  int is_causal, max_k;
  int M;          // H*T_m
  int nchunks;    // M/4
  int W;          // ceil(M/32)
  int G;          // lanes per aligned group that share one head
  const int32_t* keep;
  int64_t keep_stride_n;
  uint32_t* bits;
  float* mask_out;
  int32_t* row_nnz;
  int32_t* head_off;
  // decode step replayed as a HIP graph: T_src (the rows' absolute widths) read from device memory; `keep` is then a
  // table over ABSOLUTE row indices (entry i = K of the row with i+1 visible keys), not over the T_dst rows of the call
  const int32_t* t_src_dev;
  // decode step, ONE row per batch item: the row scan is trivial -- crow = [0, row total] -- and is written here (N, 2),
  // which saves the step a launch; nullptr everywhere else
  int32_t* crow1;
};

template <typename T> __device__ inline void load4(const T* p, float* f);
template <> __device__ inline void load4<float>(const float* p, float* f) {
  float4 v = *reinterpret_cast<const float4*>(p);
  f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ inline void load4<__hip_bfloat16>(const __hip_bfloat16* p, float* f) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <> __device__ inline void load4<__half>(const __half* p, float* f) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  const __half2* h = reinterpret_cast<const __half2*>(&v);
  float2 a = __half22float2(h[0]), b = __half22float2(h[1]);
  f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y;
}

// Block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix, total in *total.
__device__ inline int block_excl_scan(int v, int* s_wave /*[TK_WAVES]*/, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = wave_incl_scan(v);
  if (lane == 63) s_wave[w] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < TK_WAVES; ++i) {
    int x = s_wave[i];
    if (i < w) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

constexpr int TK_CAND_CAP = 1024;  // threshold-bin keys resolved by direct ranking (more -> multi-pass fallback)

// Slow path, rarely taken (massive ties / all-equal rows / overfull threshold bin): classic MSB radix passes
// followed by an ordered tie scan.  It re-reads the row from memory in every pass instead of using the caller's
// register-resident keys, so it adds nothing to the register budget of the hot path.
// (scalars are passed by value: a reference to the kernel's parameter struct would make the compiler spill the
// whole struct to per-lane scratch at kernel entry)
template <typename T>
__device__ __noinline__ unsigned long long select_multipass(const T* base, int64_t sh, int T_m, int nchunks, int M, int K,
                                                            uint32_t umin, uint32_t umax, int rounds, int* s_hist,
                                                            int* s_wave, int* s_bcast) {
  const int tid = threadIdx.x;
  auto load_keys = [&](int j, uint32_t* k4) -> bool {
    const int c = j * TK_THREADS + tid;
    if (c >= nchunks) { k4[0] = k4[1] = k4[2] = k4[3] = 0u; return false; }
    const int f0 = c * 4;
    const int h = f0 / T_m, b0 = f0 - h * T_m;
    float f[4];
    load4<T>(base + h * sh + b0, f);
    for (int e = 0; e < 4; ++e) k4[e] = f2key(f[e]);
    return true;
  };
  unsigned long long sel = 0;
  uint32_t prefix = umax;
  int bits_left = (umax == umin) ? 0 : (32 - __clz(umax ^ umin));
  int kth = K, n_eq = M;
  while (bits_left > 0) {
    const int d = bits_left < 11 ? bits_left : 11;
    const int shift = bits_left - d;
    const int nb = 1 << d;
    const int hi = shift + d;
    for (int i = tid; i < nb; i += TK_THREADS) s_hist[i] = 0;
    __syncthreads();
    for (int j = 0; j < rounds; ++j) {
      uint32_t k4[4];
      if (!load_keys(j, k4)) continue;
      for (int e = 0; e < 4; ++e) {
        const uint32_t u = k4[e];
        const bool cand = (hi >= 32) ? true : ((u >> hi) == (prefix >> hi));
        if (cand) atomicAdd(&s_hist[nb - 1 - (int)((u >> shift) & (uint32_t)(nb - 1))], 1);
      }
    }
    __syncthreads();
    const int per = (nb + TK_THREADS - 1) / TK_THREADS;
    int mine = 0;
    for (int i = 0; i < per; ++i) {
      const int bi = tid * per + i;
      if (bi < nb) mine += s_hist[bi];
    }
    int total;
    const int excl = block_excl_scan(mine, s_wave, &total);
    if (excl < kth && kth <= excl + mine) {
      int run = excl;
      for (int i = 0; i < per; ++i) {
        const int bi = tid * per + i;
        const int c = s_hist[bi];
        if (kth <= run + c) { s_bcast[0] = bi; s_bcast[1] = run; s_bcast[2] = c; break; }
        run += c;
      }
    }
    __syncthreads();
    const int rb = s_bcast[0];
    kth -= s_bcast[1];
    n_eq = s_bcast[2];
    const uint32_t dmask = (uint32_t)(nb - 1) << shift;
    prefix = (prefix & ~dmask) | ((uint32_t)(nb - 1 - rb) << shift);
    bits_left = shift;
    __syncthreads();
  }
  const uint32_t tau = prefix;
  const int r = kth;   // how many of the keys == tau are kept: lowest flat index first
  int seen = 0;        // ties in earlier rounds (flat order = round-major, then thread, then element)
  for (int j = 0; j < rounds; ++j) {
    uint32_t k4[4];
    const bool valid = load_keys(j, k4);
    int cnt = 0;
    for (int e = 0; e < 4; ++e) cnt += (valid && k4[e] == tau) ? 1 : 0;
    int total = 0, rank = 0;
    if (n_eq != r) rank = seen + block_excl_scan(cnt, s_wave, &total);   // block-uniform condition
    for (int e = 0; e < 4; ++e) {
      if (!valid) continue;
      if (k4[e] > tau) sel |= 1ull << (4 * j + e);
      if (k4[e] == tau) {
        if (n_eq == r || rank < r) sel |= 1ull << (4 * j + e);
        ++rank;
      }
    }
    seen += total;
  }
  return sel;
}

// FULL: H*T_m == 256*EPT, i.e. every register slot holds a real pixel (no validity tests on the hot path).
// 64-bit variant (two packed counters)
__device__ inline long long block_excl_scan64(long long v, long long* s_wave /*[TK_WAVES]*/, long long* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long long incl = wave_incl_scan(v);
  if (lane == 63) s_wave[w] = incl;
  __syncthreads();
  long long base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < TK_WAVES; ++i) {
    const long long x = s_wave[i];
    if (i < w) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// Everything after the keys of the row are in registers: threshold search, bit mask, per-head entry counts.
// Shared by topk_select_kernel (keys loaded from the probability map) and predictor_tail_select_kernel (keys
// produced in registers by the predictor tail).  `base` = the row's probabilities in memory (slow path only).
// HMAX: bound on p.H (sizes the per-head counters; the fused kernel's 64 keeps its LDS small).
// cand: LDS for the candidate list, 2 * TK_CAND_CAP words that nobody else touches from the first barrier in here on; nullptr
// -> a static array of this function (the fused tail + selection kernel lends the z tile it no longer needs: 8 KB less LDS per
// workgroup, which is what lets a sixth workgroup onto a CU).
// K16: the keys are non-negative 16-bit values (a 16-bit map's probabilities: their own bit patterns order like the
// numbers), TWO per register -- element i is half (i & 1) of key[i >> 1].  Half the key registers, packed 16-bit min / max, and
// the packed pair is the very word the tail stores to the map.  Same selection, bit for bit: the order of the keys and the
// tie rule are what the 32-bit keys give.
// HCW: the per-head entry counts by a transposing wave reduction -- the T_m = 256 fused kernel only (round j of wave wv IS
// head 4j + wv there); any other caller counts over the kept pixels.
template <typename T, int EPT, bool FROM_MASK, bool FULL, int HMAX = 1024, bool EXT_CAND = false, bool K16 = false, bool HCW = K16>
__device__ __forceinline__ void select_body(const TopkParams& p, uint32_t (&key)[K16 ? EPT / 2 : EPT], unsigned long long sel, int n,
                                            int t, int row, const T* base, uint32_t* cand = nullptr) {
  constexpr int R = EPT / 4;  // chunk rounds
  static_assert(!K16 || (sizeof(T) == 2 && !FROM_MASK), "packed keys: 16-bit maps");
  auto key_at = [&](int i) -> uint32_t {                   // (i is a compile-time constant after unrolling)
    if constexpr (K16) return (i & 1) ? (key[i >> 1] >> 16) : (key[i >> 1] & 0xffffu);
    else return key[i];
  };
  auto digit_at = [&](int i, int shift, int d) -> int {    // digit of the DESCENDING bin order
    if constexpr (K16) return (int)__builtin_amdgcn_ubfe(~key[i >> 1], shift + 16 * (i & 1), d);
    else return (int)__builtin_amdgcn_ubfe(~key[i], shift, d);
  };
  __shared__ int s_hist[TK_MAX_BINS + 1];          // +1: dump bin for unused register slots
  __shared__ int s_head[HMAX];
  __shared__ int s_wave[TK_WAVES];
  __shared__ uint32_t s_red[2 * TK_WAVES];
  __shared__ int s_bcast[4];
  uint32_t* s_ckey;                                // threshold-bin candidates: key ...
  uint32_t* s_cidx;                                // ... and flat pixel index
  if constexpr (EXT_CAND) {
    s_ckey = cand; s_cidx = cand + TK_CAND_CAP;
  } else {
    __shared__ uint32_t s_cand[2 * TK_CAND_CAP];
    s_ckey = s_cand; s_cidx = s_cand + TK_CAND_CAP;
  }
  __shared__ uint32_t s_selbits[512];              // resolved candidates, one bit per flat pixel (M <= 16384)
  __shared__ int s_ncand;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
#ifdef SEA_STAMP
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
#endif
  // LDS scratch is cleared while the loads are in flight
  for (int i = tid; i <= TK_MAX_BINS; i += TK_THREADS) s_hist[i] = 0;
  for (int i = tid; i < 512; i += TK_THREADS) s_selbits[i] = 0;
  for (int i = tid; i < p.H; i += TK_THREADS) s_head[i] = 0;
  if (tid == 0) s_ncand = 0;

  if (!FROM_MASK) {
    const int K = p.keep[n * p.keep_stride_n + t + (p.t_src_dev ? *p.t_src_dev - p.T_dst : 0)];
    if (K >= p.M) {
#pragma unroll
      for (int j = 0; j < R; ++j)
        if (FULL || j * TK_THREADS + tid < p.nchunks) sel |= 0xFull << (4 * j);
      __syncthreads();
    } else if (K <= 0) {
      __syncthreads();
    } else {
      // ---- common leading bits of all keys: skipped, so the histogram digit starts where keys differ --------
      uint32_t umin = 0xFFFFFFFFu, umax = 0u;
      if constexpr (K16) {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        us2 mn = us2{0xFFFF, 0xFFFF}, mx = us2{0, 0};
#pragma unroll
        for (int j = 0; j < R; ++j) {
          if (FULL || j * TK_THREADS + tid < p.nchunks) {
            const us2 a = __builtin_bit_cast(us2, key[2 * j]), b = __builtin_bit_cast(us2, key[2 * j + 1]);
            mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(a, b));       // v_pk_min_u16
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(a, b));
          }
        }
        umin = min((uint32_t)mn[0], (uint32_t)mn[1]);
        umax = max((uint32_t)mx[0], (uint32_t)mx[1]);
      } else {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (FULL || j * TK_THREADS + tid < p.nchunks) {
          umin = min(umin, min(min(key[4 * j], key[4 * j + 1]), min(key[4 * j + 2], key[4 * j + 3])));
          umax = max(umax, max(max(key[4 * j], key[4 * j + 1]), max(key[4 * j + 2], key[4 * j + 3])));
        }
      }
      }
      umin = wave_min(umin);
      umax = wave_max(umax);
      if (lane == 0) { s_red[wv] = umin; s_red[TK_WAVES + wv] = umax; }
      __syncthreads();   // also publishes the cleared scratch
#pragma unroll
      for (int i = 0; i < TK_WAVES; ++i) { umin = min(umin, s_red[i]); umax = max(umax, s_red[TK_WAVES + i]); }
      STAMP(1);   // min/max prologue
      const int bits_left = (umax == umin) ? 0 : (32 - __clz(umax ^ umin));
      bool fast = bits_left > 0;
      int rb_sel = 0, kth = K, c_in = 0, shift = 0, d = 0;
      if (fast) {
        // ---- ONE histogram pass over the top `d` differing bits (descending bins) ------------------------------
        d = bits_left < 11 ? bits_left : 11;
        shift = bits_left - d;
        const int nb = 1 << d;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          int rb = digit_at(i, shift, d);
          if (!FULL) rb = (((i >> 2) * TK_THREADS + tid) < p.nchunks) ? rb : TK_MAX_BINS;
          atomicAdd(&s_hist[rb], 1);
        }
        __syncthreads();
        const int per = (nb + TK_THREADS - 1) / TK_THREADS;
        int mine = 0;
        for (int i = 0; i < per; ++i) {
          const int bi = tid * per + i;
          if (bi < nb) mine += s_hist[bi];
        }
        int total;
        const int excl = block_excl_scan(mine, s_wave, &total);
        if (excl < K && K <= excl + mine) {
          int run = excl;
          for (int i = 0; i < per; ++i) {
            const int bi = tid * per + i;
            const int c = s_hist[bi];
            if (K <= run + c) { s_bcast[0] = bi; s_bcast[1] = run; s_bcast[2] = c; break; }
            run += c;
          }
        }
        __syncthreads();
        rb_sel = s_bcast[0];
        kth = K - s_bcast[1];       // rank still to find inside the threshold bin
        c_in = s_bcast[2];          // keys in the threshold bin
        // the bin is resolved by direct ranking unless it is overfull
        fast = c_in <= TK_CAND_CAP;
      }
      STAMP(2);   // histogram + bin search
      if (fast) {
        // ---- sweep: above the bin -> kept; inside the bin -> candidate list (key, flat index) -----------------
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          const int rb = digit_at(i, shift, d);
          const bool valid = FULL || (((i >> 2) * TK_THREADS + tid) < p.nchunks);
          if (valid && rb < rb_sel) sel |= 1ull << i;
          if (valid && rb == rb_sel) {
            const int slot = atomicAdd(&s_ncand, 1);
            s_ckey[slot] = key_at(i);
            s_cidx[slot] = (uint32_t)(((i >> 2) * TK_THREADS + tid) * 4 + (i & 3));
          }
        }
        __syncthreads();
        // ---- rank inside the bin: (key desc, flat index asc); the first `kth` are kept (ties resolved here) ---
        for (int ci = tid; ci < c_in; ci += TK_THREADS) {
          const uint32_t ku = s_ckey[ci], kf = s_cidx[ci];
          int rank = 0;
          for (int cj = 0; cj < c_in; ++cj) {
            const uint32_t ou = s_ckey[cj], of = s_cidx[cj];
            rank += (ou > ku || (ou == ku && of < kf)) ? 1 : 0;
          }
          if (rank < kth) atomicOr(&s_selbits[kf >> 5], 1u << (kf & 31));
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int c = j * TK_THREADS + tid;
          const unsigned long long nibx = (s_selbits[(c >> 3) & 511] >> (4 * (c & 7))) & 0xFu;
          sel |= nibx << (4 * j);
        }
      } else if constexpr (K16) {
        // Overfull threshold bin or an all-equal row, packed keys: resolved from the REGISTERS (the fused tail + selection
        // kernels need not write the map at all, so there may be nothing to re-read).  The histogram above fixed the top d of
        // the differing bits; 16-bit keys leave at most 5 more, one second histogram over the bin's keys; what is left are
        // exact ties, kept in flat order (round-major, then thread, then element) by block scans.
        const bool hist = bits_left > 0;                       // block-uniform: the first pass ran and found bin rb_sel
        int n_eq = hist ? c_in : p.M;
        uint32_t tau_low = 0u;
        const uint32_t lowmask = hist ? ((1u << shift) - 1u) : 0u;
        auto in_bin = [&](int i) -> bool {
          const bool valid = FULL || (((i >> 2) * TK_THREADS + tid) < p.nchunks);
          return valid && (!hist || digit_at(i, shift, d) == rb_sel);
        };
        if (hist) {
#pragma unroll
          for (int i = 0; i < EPT; ++i) {
            const bool valid = FULL || (((i >> 2) * TK_THREADS + tid) < p.nchunks);
            if (valid && digit_at(i, shift, d) < rb_sel) sel |= 1ull << i;
          }
          if (shift > 0) {                                     // (block-uniform) <= 5 bits: <= 32 bins
            const int nb2 = 1 << shift;
            __syncthreads();                                   // every reader of the first histogram has passed
            if (tid < nb2) s_hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < EPT; ++i)
              if (in_bin(i)) atomicAdd(&s_hist[nb2 - 1 - (int)(key_at(i) & lowmask)], 1);
            __syncthreads();
            const int mine = tid < nb2 ? s_hist[tid] : 0;
            int total;
            const int excl = block_excl_scan(mine, s_wave, &total);
            if (tid < nb2 && excl < kth && kth <= excl + mine) { s_bcast[0] = tid; s_bcast[1] = excl; s_bcast[2] = mine; }
            __syncthreads();
            tau_low = (uint32_t)(nb2 - 1 - s_bcast[0]);
            kth -= s_bcast[1];
            n_eq = s_bcast[2];
#pragma unroll
            for (int i = 0; i < EPT; ++i)
              if (in_bin(i) && (key_at(i) & lowmask) > tau_low) sel |= 1ull << i;
          }
        }
        int seen = 0;                                          // ties in earlier rounds
#pragma unroll
        for (int j = 0; j < R; ++j) {
          bool tie[4];
          int cnt = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            tie[e] = in_bin(4 * j + e) && (key_at(4 * j + e) & lowmask) == tau_low;
            cnt += tie[e] ? 1 : 0;
          }
          int total = 0, rank = 0;
          if (n_eq != kth) rank = seen + block_excl_scan(cnt, s_wave, &total);   // block-uniform condition
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (tie[e]) {
              if (n_eq == kth || rank < kth) sel |= 1ull << (4 * j + e);
              ++rank;
            }
          }
          seen += total;
        }
        __syncthreads();
      } else {
        sel = select_multipass<T>(base, p.sh, p.T_m, p.nchunks, p.M, K, umin, umax, R, s_hist, s_wave, s_bcast);
        for (int i = tid; i < p.H; i += TK_THREADS) s_head[i] = 0;
        __syncthreads();
      }
    }
  } else {
    __syncthreads();
  }

  STAMP(3);   // selection flags
  // ---- outputs ----------------------------------------------------------------------------------
  const int w_t = row_width(t, p.T_dst, p.t_src_dev ? *p.t_src_dev : p.T_src, p.is_causal);
  const float scale = interp_scale(w_t, p.T_m);
  const bool tm_pow2 = (p.T_m & (p.T_m - 1)) == 0;      // block-uniform
  const int tm_sh = __ffs(p.T_m) - 1;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int c = j * TK_THREADS + tid;
    const bool valid = FULL || c < p.nchunks;
    const uint32_t nib = (uint32_t)(sel >> (4 * j)) & 0xFu;
    // bit mask: 8 consecutive lanes own one 32-bit word
    uint32_t word = nib << (4 * (lane & 7));
    word |= dpp_u32<0xB1>(word);     // OR over the 8 lanes by DPP: xor 1, xor 2, half-row mirror
    word |= dpp_u32<0x4E>(word);
    word |= dpp_u32<0x141>(word);
    if ((lane & 7) == 0 && (c >> 3) < p.W) p.bits[(int64_t)row * p.W + (c >> 3)] = word;
    if (p.mask_out != nullptr && valid) {
      const int f0 = c * 4;
      const int h = f0 / p.T_m, b0 = f0 - h * p.T_m;
      float4 m;
      m.x = (nib & 1u) ? 1.f : 0.f; m.y = (nib & 2u) ? 1.f : 0.f;
      m.z = (nib & 4u) ? 1.f : 0.f; m.w = (nib & 8u) ? 1.f : 0.f;
      *reinterpret_cast<float4*>(p.mask_out + (((int64_t)n * p.H + h) * p.T_dst + t) * p.T_m + b0) = m;
    }
  }
  // entries each kept pixel will emit: min(v_end - v_start, max_k), accumulated per head.  Only kept pixels
  // are visited (a row keeps ~K_t << H*T_m of them once t is large).
  if constexpr (K16 && HCW) {
    // The fused tail + selection (T_m = 256: round j of wave wv IS head 4j + wv, lane l its pixels 4l .. 4l+3): a lane's four
    // pixel widths are the same for every head, so a head's entries are the widths summed over the kept nibble's bits and
    // over the wave's lanes -- eight heads per transposing reduction, no loop over kept pixels, no LDS atomics (ablation at
    // 7 waves per SIMD: the loop below was 50 of the kernel's 406 us).
    int wpx[4];
    {
      float bprev = interp_bound(4 * lane, scale);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float bnext = interp_bound(4 * lane + e + 1, scale);
        const int w = (int)(bnext - bprev);
        wpx[e] = w < p.max_k ? w : p.max_k;
        bprev = bnext;
      }
    }
#pragma unroll
    for (int j0 = 0; j0 < R; j0 += 8) {
      float red[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        int sum = 0;
        if (j0 + b < R) {
          const uint32_t nib = (uint32_t)(sel >> (4 * (j0 + b))) & 0xFu;
#pragma unroll
          for (int e = 0; e < 4; ++e) sum += (nib >> e) & 1u ? wpx[e] : 0;
        }
        red[b] = (float)sum;                                 // <= 64 lanes x 4 pixels x max_k: exact
      }
      const float x = wave_reduce8(red, [](float u, float v) { return u + v; });
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int h = 4 * (j0 + b) + wv;
        if (j0 + b < R && h < p.H && lane == 0) s_head[h] = (int)reduce8_get(x, b);
      }
    }
  } else {
    unsigned long long m = sel;
    while (m) {
      const int i = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int f = ((i >> 2) * TK_THREADS + tid) * 4 + (i & 3);
      const int h = tm_pow2 ? (f >> tm_sh) : f / p.T_m, b = f - h * p.T_m;   // T_m is a power of two in practice: shift
      int w = (int)(interp_bound(b + 1, scale) - interp_bound(b, scale));
      w = w < p.max_k ? w : p.max_k;
      if (w > 0) atomicAdd(&s_head[h], w);
    }
  }
  __syncthreads();
  STAMP(4);   // bit mask + widths + head counts
  // exclusive scan over heads (first wave, 64 heads per step)
  if (wv == 0) {
    int carry = 0;
    int32_t* ho = p.head_off + (int64_t)row * (p.H + 1);
    for (int h0 = 0; h0 < p.H; h0 += 64) {
      const int h = h0 + lane;
      const int v = h < p.H ? s_head[h] : 0;
      const int incl = wave_incl_scan(v);
      if (h < p.H) ho[h] = carry + incl - v;
      carry += __shfl(incl, 63);
    }
    if (lane == 0) {
      ho[p.H] = carry; p.row_nnz[row] = carry;
      if (p.crow1 != nullptr) { p.crow1[2 * row] = 0; p.crow1[2 * row + 1] = carry; }
    }
  }
}

template <typename T, int EPT, bool FROM_MASK, bool FULL>
// (no register cap here: 80 registers for a sixth wave spill 12-14 of this kernel's and cost 280 -> 378 us at OPT-1.3B x 8)
__global__ __launch_bounds__(TK_THREADS) void topk_select_kernel(TopkParams p) {
  constexpr int R = EPT / 4;  // chunk rounds
  const int tid = threadIdx.x;
  const int row = blockIdx.x;
  const int n = row / p.T_dst, t = row - n * p.T_dst;
  const T* base = reinterpret_cast<const T*>(p.src) + n * p.sn + t * p.st;
#ifdef SEA_STAMP
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
#endif

  // ---- load the row: chunk c = j*256 + tid covers flat pixels 4c..4c+3 (head-major) ------------
  uint32_t key[EPT];
  unsigned long long sel = 0;  // bit (4*j + e): element e of round j is kept
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int c = j * TK_THREADS + tid;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    const bool valid = FULL || c < p.nchunks;
    if (valid) {
      const int f0 = c * 4;
      const int h = f0 / p.T_m, b0 = f0 - h * p.T_m;
      load4<T>(base + h * p.sh + b0, f);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (FROM_MASK) {
        if (valid && f[e] != 0.f) sel |= 1ull << (4 * j + e);
        key[4 * j + e] = 0;
      } else {
        key[4 * j + e] = valid ? f2key(f[e]) : 0u;
      }
    }
  }
  STAMP(0);   // row loaded, keys built
  select_body<T, EPT, FROM_MASK, FULL>(p, key, sel, n, t, row, base);
}

// ---- predictor tail + grouped top-k in one launch ("emit probs in the top-k kernel's layout") ------------
// With T_m = 256 the two kernels already agree on who holds what: wave w of the 256-thread workgroup of row (n, t)
// produces the probabilities of heads 4j + w (j = 0, 1, ...), lane l those of pixels 4l .. 4l+3 -- exactly flat chunk
// c = 256j + tid of the selection kernel.  So the tail's softmax output, rounded to the map's dtype, is stored for the
// caller (the module returns the map) AND becomes the selection key in registers: the selection never re-reads the
// (N,H,T,T_m) map from memory.  Results are bit-identical to the two-launch path (same arithmetic, same rounding).
// Register caps that buy waves of occupancy per SIMD (4 waves per workgroup, so more resident workgroups per CU), measured at
// OPT-1.3B x 8 (H = 32, EPT 32): as first compiled 87-90 registers, 5 waves, 499 us; capped at 80 (1 spill; the candidate list
// moved into the dead z tile so that six workgroups' LDS fits) 412 us; with two 16-bit keys per register and the ragged form
// (78 registers uncapped) capped at 72 (4 spills), 7 waves: 399 us.  64 registers: 40 spills, 574 us.  H <= 16: 76 -> 72,
// 6 -> 7 waves, -5 %.  H = 40: 97-108 -> 90-96, 4 -> 5 waves, -7 %.
// (the body of predictor_tail_select_kernel below as a device function, for the fused decode kernel.  The prefill kernel keeps
// its own copy on purpose: routed through this function hipcc allocates it differently -- 9 spilled vector registers
// instead of 3 at its 72-register cap, H = 32 -- and that kernel is 17 % of the headline step.)
template <typename T, int EPT, bool FULL>
__device__ __forceinline__ void tail_select_row(const TailParams& tp, const TopkParams& p, float* s_z, int row) {
  constexpr int R = EPT / 4, E = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = row / tp.T, t = row - n * tp.T;
  constexpr int LDZ = 64 + 3;      // = W4 + 3 (sea_predictor_tail_select checks W4 == 64): z-row offsets become immediates
#ifdef SEA_STAMP
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
#endif
  uint32_t* s_tab = reinterpret_cast<uint32_t*>(s_z + ((tp.H + 15) / 16) * 16 * LDZ);   // per-pixel constants [3][64 E]
  TailRow<T, E> tr;
  if (tp.tab) tr.load_global(tp.tab, lane);                        // (block-uniform) the table was computed once per weight set
  tail_z_tile<T>(tp, s_z, n, t);
  if (!tp.tab) tail_consts_fill<T>(tp, s_tab, 64 * E);
  __syncthreads();
  if (!tp.tab) tr.load(s_tab, lane);
  STAMP(8);   // z tile (MFMA) + per-pixel constants
  uint32_t key[EPT / 2];                                           // two 16-bit keys per register (select_body, K16)
  const int mine = FULL ? R : max(0, (tp.H - wv + 3) / 4);         // heads wv, wv + 4, ... of this wave (wave-uniform)
  auto batch = [&](auto j0c, auto nbc) {                           // heads 4 (J0 + b) + wv, b < NBC, through one batch
    constexpr int J0 = decltype(j0c)::value, NBC = decltype(nbc)::value;
    float a[NBC][E];
    const int nb = min(NBC, mine - J0);
    if (nb > 0) {
      // T_M == 256 == 64 E here (sea_predictor_tail_select checks): the full-row form, without its ragged twin in the kernel
      tr.template heads_impl<true>(tp, lane, nb, [&](int b) { return s_z + (4 * (J0 + b) + wv) * LDZ; },
                                   [&](int b) { return (((int64_t)n * tp.H + (4 * (J0 + b) + wv)) * tp.T + t) * (64 * E); }, a);
    }
#pragma unroll
    for (int b = 0; b < NBC; ++b) {  // probabilities are >= +0: the 16-bit pattern the map stores orders like the number
      key[2 * (J0 + b)] = (b < nb) ? pack2<T>(a[b][0], a[b][1]) : 0u;
      key[2 * (J0 + b) + 1] = (b < nb) ? pack2<T>(a[b][2], a[b][3]) : 0u;
    }
  };
  static_assert(R <= 16, "two batches of eight heads per wave");
  if constexpr (R <= 8) {
    batch(std::integral_constant<int, 0>{}, std::integral_constant<int, R>{});
  } else {
    batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
    batch(std::integral_constant<int, 8>{}, std::integral_constant<int, R - 8>{});
  }
  STAMP(9);   // 8 heads per wave: resize + LayerNorm + softmax + store
  // H <= 64: sea_predictor_tail_select checks.  The z tile and the constants table are dead once every wave has left the head
  // loop, i.e. from select_body's first barrier on: the candidate list lives there (the launcher sizes the dynamic LDS for both).
  // The packed-key selection never re-reads the map (its slow path works on the registers too): tp.probs may be null.
  select_body<T, EPT, false, FULL, 64, true, true>(p, key, 0ull, n, t, row, (const T*)nullptr, reinterpret_cast<uint32_t*>(s_z));
}

template <typename T, int EPT, bool FULL>
#ifndef SEA_TSEL_OCC32
#define SEA_TSEL_OCC32 7
#endif
#ifndef SEA_TSEL_OCC16
#define SEA_TSEL_OCC16 7
#endif
__global__ __launch_bounds__(TK_THREADS, EPT <= 16 ? SEA_TSEL_OCC16 : EPT == 32 ? SEA_TSEL_OCC32 : EPT == 40 ? 5 : 1) void predictor_tail_select_kernel(TailParams tp, TopkParams p) {
  constexpr int R = EPT / 4, E = 4;
  extern __shared__ __attribute__((aligned(16))) float s_z[];     // HP x (W4 + 3)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.x;
  const int n = row / tp.T, t = row - n * tp.T;
  constexpr int LDZ = 64 + 3;      // = W4 + 3 (sea_predictor_tail_select checks W4 == 64): z-row offsets become immediates
#ifdef SEA_STAMP
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
#endif
  uint32_t* s_tab = reinterpret_cast<uint32_t*>(s_z + ((tp.H + 15) / 16) * 16 * LDZ);   // per-pixel constants [3][64 E]
  TailRow<T, E> tr;
  if (tp.tab) tr.load_global(tp.tab, lane);                        // (block-uniform) the table was computed once per weight set
  tail_z_tile<T>(tp, s_z, n, t);
  if (!tp.tab) tail_consts_fill<T>(tp, s_tab, 64 * E);
  __syncthreads();
  if (!tp.tab) tr.load(s_tab, lane);
  STAMP(8);   // z tile (MFMA) + per-pixel constants
  uint32_t key[EPT / 2];                                           // two 16-bit keys per register (select_body, K16)
  const int mine = FULL ? R : max(0, (tp.H - wv + 3) / 4);         // heads wv, wv + 4, ... of this wave (wave-uniform)
  auto batch = [&](auto j0c, auto nbc) {                           // heads 4 (J0 + b) + wv, b < NBC, through one batch
    constexpr int J0 = decltype(j0c)::value, NBC = decltype(nbc)::value;
    float a[NBC][E];
    const int nb = min(NBC, mine - J0);
    if (nb > 0) {
      // T_M == 256 == 64 E here (sea_predictor_tail_select checks): the full-row form, without its ragged twin in the kernel
      tr.template heads_impl<true>(tp, lane, nb, [&](int b) { return s_z + (4 * (J0 + b) + wv) * LDZ; },
                                   [&](int b) { return (((int64_t)n * tp.H + (4 * (J0 + b) + wv)) * tp.T + t) * (64 * E); }, a);
    }
#pragma unroll
    for (int b = 0; b < NBC; ++b) {  // probabilities are >= +0: the 16-bit pattern the map stores orders like the number
      key[2 * (J0 + b)] = (b < nb) ? pack2<T>(a[b][0], a[b][1]) : 0u;
      key[2 * (J0 + b) + 1] = (b < nb) ? pack2<T>(a[b][2], a[b][3]) : 0u;
    }
  };
  static_assert(R <= 16, "two batches of eight heads per wave");
  if constexpr (R <= 8) {
    batch(std::integral_constant<int, 0>{}, std::integral_constant<int, R>{});
  } else {
    batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
    batch(std::integral_constant<int, 8>{}, std::integral_constant<int, R - 8>{});
  }
  STAMP(9);   // 8 heads per wave: resize + LayerNorm + softmax + store
  // H <= 64: sea_predictor_tail_select checks.  The z tile and the constants table are dead once every wave has left the head
  // loop, i.e. from select_body's first barrier on: the candidate list lives there (the launcher sizes the dynamic LDS for both).
  // The packed-key selection never re-reads the map (its slow path works on the registers too): tp.probs may be null.
  select_body<T, EPT, false, FULL, 64, true, true>(p, key, 0ull, n, t, row, (const T*)nullptr, reinterpret_cast<uint32_t*>(s_z));
}

// ---- fp32 DATA (round 5): the same fusion for the reference's fp32 measurement protocol (benchmark_bert.py:196-239) ----------
// Same structure and layouts; the z tile runs on the fp32 MFMA (tail_z_tile<float>), the map is fp32 and IS written (nothing
// lazy about the fp32 path), the keys are the 32-bit patterns of the probabilities (select_body's unpacked form, whose slow
// path re-reads the row of the map this launch has just stored).  Bit-identical to predictor_tail_mfma_kernel<float> followed
// by topk_select_kernel<float>: same device functions, same key layout (chunk 256 j + tid = head 4 j + wave, pixels 4 lane ..).
template <int EPT>
__global__ __launch_bounds__(TK_THREADS) void predictor_tail_select_f32_kernel(TailParams tp, TopkParams p) {
  using T = float;
  constexpr int R = EPT / 4, E = 4;
  extern __shared__ __attribute__((aligned(16))) float s_z[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.x;
  const int n = row / tp.T, t = row - n * tp.T;
  constexpr int LDZ = 64 + 3;
  uint32_t* s_tab = reinterpret_cast<uint32_t*>(s_z + ((tp.H + 15) / 16) * 16 * LDZ);
  TailRow<T, E> tr;
  if (tp.tab) tr.load_global(tp.tab, lane);
  tail_z_tile<T>(tp, s_z, n, t);
  if (!tp.tab) tail_consts_fill<T>(tp, s_tab, 64 * E);
  __syncthreads();
  if (!tp.tab) tr.load(s_tab, lane);
  uint32_t key[EPT];
  const int mine = max(0, (tp.H - wv + 3) / 4);
  auto batch = [&](auto j0c, auto nbc) {
    constexpr int J0 = decltype(j0c)::value, NBC = decltype(nbc)::value;
    float a[NBC][E];
    const int nb = min(NBC, mine - J0);
    if (nb > 0) {
      tr.template heads_impl<true>(tp, lane, nb, [&](int b) { return s_z + (4 * (J0 + b) + wv) * LDZ; },
                                   [&](int b) { return (((int64_t)n * tp.H + (4 * (J0 + b) + wv)) * tp.T + t) * (64 * E); }, a);
    }
#pragma unroll
    for (int b = 0; b < NBC; ++b)
#pragma unroll
      for (int e = 0; e < E; ++e) key[4 * (J0 + b) + e] = (b < nb) ? f2key(a[b][e]) : 0u;
  };
  static_assert(R <= 8, "one batch of eight heads per wave (H <= 32)");
  batch(std::integral_constant<int, 0>{}, std::integral_constant<int, R>{});
  const T* base = reinterpret_cast<const T*>(tp.probs) + (int64_t)n * p.sn + (int64_t)t * p.st;
  select_body<T, EPT, false, false, 64, true, false, false>(p, key, 0ull, n, t, row, base, reinterpret_cast<uint32_t*>(s_z));
}

// ---- the same fusion for ANY predictor length (T_m % 4 == 0, T_m <= 512; the reference's own grid runs 64 / 96 / 128 / 384:
// src/main/benchmark_opt_ablation.py:160-186, exp_long_context.py:152).  The tail keeps its natural layout (wave <-> heads
// wv, wv + 4, ..., lane <-> E consecutive pixels); the rounded probabilities go through a flat 16-bit image of the row in
// LDS ([head][pixel], 2 H T_m bytes) from which every thread takes the selection's layout (chunk c = 256 j + tid <-> flat
// pixels 4c .. 4c+3) as packed keys.  Same arithmetic as predictor_tail_mfma_kernel + topk_select_kernel: bit-identical.
template <typename T, int E, int EPT>
__global__ __launch_bounds__(TK_THREADS) void predictor_tail_select_gen_kernel(TailParams tp, TopkParams p) {
  constexpr int R = EPT / 4;
  constexpr int NBC = E >= 6 ? 4 : 8;                             // heads per batch of the tail stage (register budget)
  extern __shared__ __attribute__((aligned(16))) float s_z[];     // HP x (W4 + 3) | constants [3][64 E] | flat map (H T_m 16-bit)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.x;
  const int n = row / tp.T, t = row - n * tp.T;
  const int LDZ = tp.W4 + 3;
  uint32_t* s_tab = reinterpret_cast<uint32_t*>(s_z + ((tp.H + 15) / 16) * 16 * LDZ);
  // the flat image sits behind z + table, or behind the 8 KB the candidate list takes over from them if they are smaller
  const int zt_words = ((tp.H + 15) / 16) * 16 * LDZ + TAIL_TAB_ROWS * 64 * E;
  unsigned short* s_flat = reinterpret_cast<unsigned short*>(s_z + max(zt_words, 2 * TK_CAND_CAP));
  tail_z_tile<T>(tp, s_z, n, t);
  tail_consts_fill<T>(tp, s_tab, 64 * E);
  __syncthreads();
  TailRow<T, E> tr;
  tr.load(s_tab, lane);
  const int mine = max(0, (tp.H - wv + 3) / 4);                   // heads wv, wv + 4, ... of this wave
  for (int k0 = 0; k0 < mine; k0 += NBC) {
    float a[NBC][E];
    const int nb = min(NBC, mine - k0);
    tr.heads(tp, lane, nb, [&](int b) { return s_z + (wv + 4 * (k0 + b)) * LDZ; },
             [&](int b) { return (((int64_t)n * tp.H + (wv + 4 * (k0 + b))) * tp.T + t) * tp.T_M; }, a);
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
      if (b < nb) {
        unsigned short* fr = s_flat + (wv + 4 * (k0 + b)) * tp.T_M + lane * E;
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (lane * E + e < tp.T_M) fr[e] = __builtin_bit_cast(unsigned short, from_f<T>(a[b][e]));
      }
    }
  }
  __syncthreads();
  uint32_t key[EPT / 2];                                           // two 16-bit keys per register (select_body, K16)
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int c = j * TK_THREADS + tid;
    uint2 v = make_uint2(0u, 0u);
    if (c < p.nchunks) v = *reinterpret_cast<const uint2*>(s_flat + 4 * c);
    key[2 * j] = v.x;
    key[2 * j + 1] = v.y;
  }
  select_body<T, EPT, false, false, 64, true, true, false>(p, key, 0ull, n, t, row, (const T*)nullptr, reinterpret_cast<uint32_t*>(s_z));
}

// ---- crow = exclusive scan of row_nnz ------------------------------------------------------------
template <typename I>
__global__ __launch_bounds__(1024) void row_scan_kernel(const int32_t* row_nnz, int T_dst, I* crow) {
  // 1024 rows per pass (coalesced loads and stores).  Round 5: the row totals of the next PER passes are requested together,
  // so a pass is a wave scan + two barriers instead of a dependent memory round trip + the same (one 32768-token sequence:
  // 32 round trips, 27 us of a 1.38 ms step at N = 1; a per-thread run of consecutive rows measured worse, 33 us: its
  // loads and stores touch 64 lines per instruction).
  constexpr int PER = 16;
  __shared__ int s_w[2][16];                                   // two sets: a pass needs ONE barrier (pass j + 2 rewrites set j & 1
  const int n = blockIdx.x;                                    // only after every thread has passed barrier j + 1, i.e. read it)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t* src = row_nnz + (int64_t)n * T_dst;
  I* dst = crow + (int64_t)n * (T_dst + 1);
  long long carry = 0;
  for (int t0 = 0; t0 < T_dst; t0 += 1024 * PER) {
    int vv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int t = t0 + j * 1024 + tid;
      vv[j] = t < T_dst ? src[t] : 0;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (t0 + j * 1024 < T_dst) {                               // block-uniform
        const int t = t0 + j * 1024 + tid;
        const int v = vv[j];
        const int incl = wave_incl_scan(v);
        if (lane == 63) s_w[j & 1][w] = incl;
        __syncthreads();
        int base = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int x = s_w[j & 1][i]; if (i < w) base += x; tot += x; }
        if (t < T_dst) dst[t] = (I)(carry + base + incl - v);
        carry += tot;
      }
    }
    __syncthreads();                                             // (PER is even: the next span starts on set 0 again)
  }
  if (tid == 0) dst[T_dst] = (I)carry;
}

// ---- emit column indices ----------------------------------------------------------------------------
struct EmitParams {
  const uint32_t* bits;
  const void* crow;
  int H, T_dst, T_m, T_src, is_causal, max_k, W;
  void* col;
  int64_t col_stride_n, z_cap;
  float* values_out;
  int T_enc;                       // column = head * T_enc + key (T_src, or a fixed cache capacity in the decode form)
  const int32_t* t_src_dev;        // decode step replayed as a HIP graph: T_src (row widths) read from device memory
};

// One workgroup per row.
//   1. the T_m+1 pixel bounds of this row (round_half_away(m * w_t / T_m)) go to LDS once -- every head shares them;
//   2. each thread owns 32-pixel mask words (word w, w+256, ...): it counts the entries its kept pixels expand to;
//   3. one block scan turns the counts into output offsets (flat order = head-major, pixels ascending);
//   4. the threads write their runs (keys descending inside a pixel, causal_resize_m_to_t.py:569) into an LDS
//      window of the row, which is flushed with fully coalesced stores.
// A pixel whose width exceeds max_k is thinned with the reference's fp32 step arithmetic; every other pixel is
// plain integer work (ids stay below 2^24, so the reference's fp32 ids are exact integers there).
constexpr int EM_TABLE = 1024;                        // pixel-bound table (T_m <= 1024; else bounds on the fly)
constexpr int EM_WIN = 4096;                          // entries staged per flush

template <typename I>
__device__ __forceinline__ void csr_emit_row(const EmitParams& p, const int row) {
  __shared__ int s_wave[TK_WAVES];
  __shared__ int s_bound[EM_TABLE + 1];
  __shared__ int s_out[EM_WIN];
  const int tid = threadIdx.x;
  const int n = row / p.T_dst, t = row - n * p.T_dst;
  const I* crow = reinterpret_cast<const I*>(p.crow) + (int64_t)n * (p.T_dst + 1);
  const int64_t row_beg = (int64_t)crow[t];
  const int64_t row_end = (int64_t)crow[t + 1];
  if (row_end == row_beg) return;
  I* col = reinterpret_cast<I*>(p.col) + n * p.col_stride_n;
  float* vals = p.values_out ? p.values_out + n * p.col_stride_n : nullptr;
  const uint32_t* bits = p.bits + (int64_t)row * p.W;
  const int w_t = row_width(t, p.T_dst, p.t_src_dev ? *p.t_src_dev : p.T_src, p.is_causal);
  const float scale = interp_scale(w_t, p.T_m);
  const bool table = p.T_m <= EM_TABLE;
  if (table)
    for (int m = tid; m <= p.T_m; m += TK_THREADS) s_bound[m] = (int)interp_bound(m, scale);
  __syncthreads();
  auto bnd = [&](int m) { return table ? s_bound[m] : (int)interp_bound(m, scale); };
  const bool aligned = (p.T_m & 31) == 0;             // a mask word never straddles two heads
  // (row-uniform) every pixel at least 4 keys wide and none wider than max_k, int32 columns, no values output
  const bool direct = sizeof(I) == 4 && vals == nullptr && aligned && w_t >= 4 * p.T_m && (w_t + p.T_m - 1) / p.T_m <= p.max_k;

  // the row is walked in passes of 256 mask words (flat order); carry = entries of the earlier passes
  int carry = 0;
  for (int w0 = 0; w0 < p.W; w0 += TK_THREADS) {
    const int wi = w0 + tid;
    uint32_t word = wi < p.W ? bits[wi] : 0u;
    const int f0 = wi * 32;
    const int h0 = f0 / p.T_m, b0 = f0 - h0 * p.T_m;
    int nent = 0;
    if (direct) {                                                   // wave-uniform trip count, no thinning (see below)
      const int npmax = __builtin_amdgcn_readfirstlane(wave_max(__popc(word)));
      uint32_t m = word;
      for (int it = 0; it < npmax; ++it) {
        const bool act = m != 0u;
        const int b = b0 + (act ? __ffs(m) - 1 : 0);
        m &= m - 1;
        const int wd = bnd(b + 1) - bnd(b);
        nent += act ? wd : 0;
      }
    } else {
      for (uint32_t m = word; m;) {
        const int bit = __ffs(m) - 1;
        m &= m - 1;
        int b = b0 + bit;
        if (!aligned) b %= p.T_m;
        const int wd = bnd(b + 1) - bnd(b);
        nent += wd < p.max_k ? wd : p.max_k;
      }
    }
    int total;
    const int excl = block_excl_scan(nent, s_wave, &total);
    if (direct) {
      // Wide pixels (every pixel of the row spans >= 4 keys, none is thinned): each thread writes its runs STRAIGHT to
      // memory, four entries per 16-byte store -- no LDS window, no flush, no barrier -- and with WAVE-UNIFORM loops:
      // this kernel is bound by SCALAR instruction issue (one scalar unit per CU: 8.6e7 scalar against 8.2e7 vector
      // instructions per launch, profiles/r02d_pmc_per_kernel.txt = 334 k cycles per CU of its 190 us), and what issues
      // them is the exec-mask bookkeeping of per-lane trip counts (a run loop of 16 iterations costs 3 scalar instructions
      // each).  Every pixel of the row is wlo or wlo + 1 keys wide (wlo = floor(w_t / T_m)), so a run is written as
      // ceil((wlo + 1) / 4) four-entry stores for every lane alike, the last one pulled back to end at the run's end
      // (it overlaps its predecessor with the same values); the walk over a word's kept pixels runs to the wave's maximum.
      const int wlo = w_t / p.T_m;
      const int nst = (wlo + 1 + 3) >> 2;                           // stores per run (row-uniform)
      const int npmax = __builtin_amdgcn_readfirstlane(wave_max(__popc(word)));                     // kept pixels of the busiest lane (wave-uniform)
      int64_t off = row_beg + carry + excl;
      uint32_t m = word;
      const int hb = h0 * p.T_enc;
      for (int it = 0; it < npmax; ++it) {
        const bool act = m != 0u;
        const int bit = act ? __ffs(m) - 1 : 0;
        m &= m - 1;
        const int b = b0 + bit;                                     // (direct requires T_m % 32 == 0: no straddling)
        const int lo = bnd(b), hi = bnd(b + 1);
        const int wd = hi - lo;
        if (act && off + wd <= p.z_cap) {
          int32_t* dst = reinterpret_cast<int32_t*>(col) + off;
          const int c0 = hb + hi - 1;
          typedef int ei4 __attribute__((ext_vector_type(4), aligned(4)));   // a run starts on any 4-byte boundary
          for (int i = 0; i < nst; ++i) {
            const int j = min(4 * i, wd - 4);
            const int c = c0 - j;
            const ei4 v = {c, c - 1, c - 2, c - 3};
            *reinterpret_cast<ei4*>(dst + j) = v;
          }
        }
        off += act ? wd : 0;
      }
      carry += total;
      continue;
    }
    // windows of the row covered by this pass: [carry, carry + total)
    for (int win = (carry / EM_WIN) * EM_WIN; win < carry + total; win += EM_WIN) {
      int off = carry + excl;                                       // row-relative offset of this thread's first entry
      if (nent > 0 && off < win + EM_WIN && off + nent > win) {
        const bool inside = off >= win && off + nent <= win + EM_WIN;
        for (uint32_t m = word; m;) {
          const int bit = __ffs(m) - 1;
          m &= m - 1;
          int h = h0, b = b0 + bit;
          if (!aligned) { const int q = b / p.T_m; h += q; b -= q * p.T_m; }
          const int lo = bnd(b), hi = bnd(b + 1);
          const int wd = hi - lo;
          const int hb = h * p.T_enc;
          if (wd <= p.max_k) {
            if (inside) {                                           // the common case: no per-entry window test
              int* dst = s_out + (off - win);
              const int c0 = hb + hi - 1;
              for (int j = 0; j < wd; ++j) dst[j] = c0 - j;
            } else {
              for (int j = 0; j < wd; ++j) {
                const int o = off + j - win;
                if ((unsigned)o < (unsigned)EM_WIN) s_out[o] = hb + hi - 1 - j;
              }
            }
            off += wd;
          } else {                                                  // thinned pixel: the reference's fp32 stepping
            const float rs = (float)lo + (float)hb, re = (float)hi + (float)hb;
            const float step = __fdiv_rn(re - rs, (float)p.max_k);
            for (int j = 0; j < p.max_k; ++j) {
              const int o = off + j - win;
              if ((unsigned)o < (unsigned)EM_WIN) s_out[o] = (int)((re - (float)(int)__fmul_rn((float)j, step)) - 1.0f);
            }
            off += p.max_k;
          }
        }
      }
      __syncthreads();
      // flush the part of the window this pass has completed: [max(win, carry), min(win + EM_WIN, carry + total))
      const int fb = win > carry ? win : carry;
      const int fe = (win + EM_WIN < carry + total) ? win + EM_WIN : carry + total;
      for (int i = fb + tid; i < fe; i += TK_THREADS) {
        const int64_t dst = row_beg + i;
        if (dst < p.z_cap) {
          col[dst] = (I)s_out[i - win];
          if (vals) vals[dst] = 1.0f;
        }
      }
      __syncthreads();
    }
    carry += total;
  }
}

template <typename I>
__global__ __launch_bounds__(TK_THREADS) void csr_emit_kernel(EmitParams p) {
  csr_emit_row<I>(p, (int)blockIdx.x);
}

// ---- a decoding step's predictor CNN + tail + selection in ONE launch (round 5) ------------------------------------------------
// A position of a graph-replayed DecodeSession ran conv1, conv2 (each over the whole 25-row window: 10 - 12 us, all of it the
// launch and the 74 KB weight image staged into LDS to convolve rows nobody reads), tail + selection (13 us) and the window
// shift (4 us) as four launches for ONE new row per sequence.  Here one 4-wave workgroup per sequence
//   * computes the new row of conv1 from the rows t - 2 dil, t - dil (a ring of the MLP's previous outputs) and t (the row
//     the MLP launch has just written), and the new row of conv2 from a ring of conv1's previous rows -- conv_row_c8, bit for
//     bit the rows causal_conv_c8_kernel writes;
//   * runs tail_select_row on that row, unchanged;
//   * files the two new rows in their rings (no window to shift) and, as the LAST workgroup to finish (ticket), advances the
//     session's device counters -- what the shift launch did.
struct DecodeCnnParams {
  const void* x_new;      // (N, C/8, W, 8): the MLP's output row of this position
  void* x_ring;           // (N, RX, C/8, W, 8): rows of earlier positions, row of position p in slot p % RX
  void* y1_ring;          // (N, RY, C/8, W, 8): conv1's rows, slot p % RY
  void* y2;               // (N, C/8, W, 8): conv2's new row (the tail's input)
  const void *w1, *w2;    // packed (C, 9 * CinP)
  const float *b1, *b2;   // (C)
  int32_t* counters;      // [seen, tsrc, tsrc of the step just finished]
  int32_t* ticket;
  int C, W, RX, RY, dil, pad_w;
};

// EMIT: the selection's one CSR row is expanded into its column ids here too (csr_emit_row: the sea_csr_emit_at launch that
// followed); instantiated wherever the emit's 20 KB of LDS fit beside the weight image (all but the 80-channel form).
template <typename T, int EPT, int NT, int KCH, bool EMIT>
__global__ __launch_bounds__(TK_THREADS) void decode_cnn_tail_select_kernel(DecodeCnnParams dp, TailParams tp, TopkParams p, EmitParams ep) {
  extern __shared__ __attribute__((aligned(16))) float s_z[];
  const int n = (int)blockIdx.x;
  const int pos = dp.counters[0];                                  // rows the session has seen = index of the new position
  const int64_t row = (int64_t)dp.C * dp.W;                        // elements per C8 row
  const T* xn = reinterpret_cast<const T*>(dp.x_new) + n * row;
  T* xr = reinterpret_cast<T*>(dp.x_ring) + (int64_t)n * dp.RX * row;
  T* yr = reinterpret_cast<T*>(dp.y1_ring) + (int64_t)n * dp.RY * row;
  T* y2 = reinterpret_cast<T*>(dp.y2) + n * row;
  auto slot = [](int p_, int r_) { return ((p_ % r_) + r_) % r_; };
  T* y1_new = yr + slot(pos, dp.RY) * row;
  T* sW = reinterpret_cast<T*>(s_z);                               // the weight image lives where the tail's z tile will (dead until then)
#ifdef SEA_STAMP
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
#endif
  ConvRowC8<T, NT, KCH> c1, c2;
  // (what does not depend on the position leaves first: the weight image and the MLP's row are in flight while the counter's
  // scalar load returns)
  c1.load_weights(reinterpret_cast<const T*>(dp.w1), dp.C);
  c1.load_bias(dp.b1, dp.C);
  c2.load_bias(dp.b2, dp.C);
  c1.fetch_row(2, xn, dp.C, dp.W, dp.dil, dp.pad_w);
  c1.fetch_row(0, xr + slot(pos - 2 * dp.dil, dp.RX) * row, dp.C, dp.W, dp.dil, dp.pad_w);
  c1.fetch_row(1, xr + slot(pos - dp.dil, dp.RX) * row, dp.C, dp.W, dp.dil, dp.pad_w);
  c2.fetch_row(0, yr + slot(pos - 2 * dp.dil, dp.RY) * row, dp.C, dp.W, dp.dil, dp.pad_w);      // (conv1's rows of earlier positions)
  c2.fetch_row(1, yr + slot(pos - dp.dil, dp.RY) * row, dp.C, dp.W, dp.dil, dp.pad_w);
  c1.store_weights(sW);
  STAMP(5);   // decode: kernel start -> conv1's operands and weight image in place (all of it memory latency)
  c2.load_weights(reinterpret_cast<const T*>(dp.w2), dp.C);       // in flight while conv1's row is computed
  c1.run(sW, y1_new, dp.C, dp.W, 1);
  STAMP(10);  // decode: operand fetches + conv1's row
  c2.fetch_row(2, y1_new, dp.C, dp.W, dp.dil, dp.pad_w);           // (stored above, barrier passed)
  c2.store_weights(sW);
  c2.run(sW, y2, dp.C, dp.W, 1);
  STAMP(11);  // decode: conv2's row
  {                                                                // the MLP's row joins the ring (read by the next positions)
    const uint4* src = reinterpret_cast<const uint4*>(xn);
    uint4* dst = reinterpret_cast<uint4*>(xr + slot(pos, dp.RX) * row);
    for (int i = threadIdx.x; i < (int)(row / 8); i += TK_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  STAMP(12);  // decode: ring copy
  tail_select_row<T, EPT, false>(tp, p, s_z, n);                   // T = 1: row n of the call is batch item n
  __syncthreads();
#ifdef SEA_STAMP
  _tprev = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (EMIT) {
    if (ep.col != nullptr) {                                        // (grid-uniform)
      csr_emit_row<int32_t>(ep, n);                                 // reads the bits / crow this workgroup has just written
      __syncthreads();
    }
  }
  STAMP(13);  // decode: emit
  if (threadIdx.x == 0) {
    __threadfence();
    const int done = atomicAdd(dp.ticket, 1);
    if (done == (int)gridDim.x - 1) {                              // every workgroup has read the counters and finished
      const int ts = dp.counters[1];
      dp.counters[2] = ts;                                         // (the emit launch behind this one reads the step's T_src here)
      dp.counters[0] = pos + 1;
      dp.counters[1] = ts + 1;
      *dp.ticket = 0;
      __threadfence();
    }
  }
}

// ---- per-(row, head) offsets of a foreign flat CSR (rows grouped by ascending head) ----------------
template <typename I>
__global__ __launch_bounds__(TK_THREADS) void head_offsets_kernel(const I* crow_all, const I* col_all, int H, int T_dst,
                                                                 int T_src, int64_t col_stride_n, int32_t* head_off) {
  __shared__ int s_head[1024];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row = blockIdx.x;
  const int n = row / T_dst, t = row - n * T_dst;
  const I* crow = crow_all + (int64_t)n * (T_dst + 1);
  const I* col = col_all + n * col_stride_n;
  const int64_t beg = crow[t], end = crow[t + 1];
  for (int i = tid; i < H; i += TK_THREADS) s_head[i] = 0;
  __syncthreads();
  for (int64_t e0 = beg; e0 < end; e0 += TK_THREADS) {
    const int64_t e = e0 + tid;
    int h = -1;
    if (e < end) h = (int)(col[e] / T_src);
    // aggregate equal heads inside the wave (entries are grouped, so 1-3 rounds)
    unsigned long long active = __ballot(h >= 0);
    while (active) {
      const int src_lane = __ffsll((long long)active) - 1;
      const int h0 = __shfl(h, src_lane);
      const unsigned long long same = __ballot(h == h0);
      if (lane == src_lane) atomicAdd(&s_head[h0], __popcll(same));
      active &= ~same;
    }
  }
  __syncthreads();
  if (wv == 0) {
    int carry = 0;
    int32_t* ho = head_off + (int64_t)row * (H + 1);
    for (int h0 = 0; h0 < H; h0 += 64) {
      const int h = h0 + lane;
      const int v = h < H ? s_head[h] : 0;
      const int incl = wave_incl_scan(v);
      if (h < H) ho[h] = carry + incl - v;
      carry += __shfl(incl, 63);
    }
    if (lane == 0) ho[H] = carry;
  }
}

// ---- host side -----------------------------------------------------------------------------------
static int group_lanes(int T_m) {
  int q = T_m / 4, g = 1;
  while (g < 64 && (q % (g * 2)) == 0) g *= 2;
  return g;
}

template <typename T, bool FROM_MASK>
static int launch_select(const TopkParams& p, int64_t rows, hipStream_t s) {
  const int ept = ((p.nchunks + TK_THREADS - 1) / TK_THREADS) * 4;
  dim3 grid((unsigned)rows), block(TK_THREADS);
#define SEA_SEL(E)                                                                                      \
  do {                                                                                                  \
    const bool full = p.nchunks == ((E) / 4) * TK_THREADS; /* every register slot of THIS instantiation is live */ \
    if (full) hipLaunchKernelGGL((topk_select_kernel<T, E, FROM_MASK, true>), grid, block, 0, s, p);   \
    else hipLaunchKernelGGL((topk_select_kernel<T, E, FROM_MASK, false>), grid, block, 0, s, p);       \
  } while (0)
  if (ept <= 4) SEA_SEL(4);
  else if (ept <= 8) SEA_SEL(8);
  else if (ept <= 16) SEA_SEL(16);
  else if (ept <= 32) SEA_SEL(32);
  else if (ept <= 40) SEA_SEL(40);
  else SEA_SEL(64);
#undef SEA_SEL
  return 0;
}

template <bool FROM_MASK>
static int select_common(const char* name, const void* src, int dtype, int64_t N, int64_t H, int64_t T_dst, int64_t T_m,
                         int64_t sn, int64_t sh, int64_t st, const int32_t* keep, int64_t keep_stride_n, int64_t T_src,
                         int is_causal, int max_k, uint32_t* bits, float* mask_out, int32_t* row_nnz, int32_t* head_off,
                         hipStream_t s) {
  SEA_REQUIRE(src && bits && row_nnz && head_off, SEA_EINVAL, "%s: null pointer", name);
  SEA_REQUIRE(FROM_MASK || keep, SEA_EINVAL, "%s: keep is null", name);
  SEA_REQUIRE(N > 0 && H > 0 && T_dst > 0 && T_m > 0 && T_src >= T_dst, SEA_EINVAL, "%s: bad shape", name);
  SEA_REQUIRE(T_m % 4 == 0, SEA_EUNSUPPORTED, "%s: T_m=%lld must be a multiple of 4", name, (long long)T_m);
  SEA_REQUIRE(H * T_m <= 16384 && H <= 1024, SEA_EUNSUPPORTED, "%s: H*T_m=%lld > 16384", name, (long long)(H * T_m));
  SEA_REQUIRE(N * T_dst < (1ll << 31), SEA_EUNSUPPORTED, "%s: too many rows", name);
  const int esz = dtype == SEA_F32 ? 4 : 2;
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", name, dtype);
  SEA_REQUIRE(((uintptr_t)src % (4 * esz)) == 0 && sn % 4 == 0 && sh % 4 == 0 && st % 4 == 0, SEA_EUNSUPPORTED,
              "%s: rows must be 4-element aligned", name);
  SEA_REQUIRE(max_k > 0, SEA_EINVAL, "%s: max_k must be positive", name);
  TopkParams p;
  p.src = src; p.sn = sn; p.sh = sh; p.st = st;
  p.H = (int)H; p.T_dst = (int)T_dst; p.T_m = (int)T_m; p.T_src = (int)T_src;
  p.is_causal = is_causal; p.max_k = max_k;
  p.M = (int)(H * T_m); p.nchunks = p.M / 4; p.W = (p.M + 31) / 32; p.G = group_lanes((int)T_m);
  p.keep = keep; p.keep_stride_n = keep_stride_n;
  p.bits = bits; p.mask_out = mask_out; p.row_nnz = row_nnz; p.head_off = head_off; p.t_src_dev = nullptr; p.crow1 = nullptr;
  const int64_t rows = N * T_dst;
  if (dtype == SEA_F32) launch_select<float, FROM_MASK>(p, rows, s);
  else if (dtype == SEA_F16) launch_select<__half, FROM_MASK>(p, rows, s);
  else launch_select<__hip_bfloat16, FROM_MASK>(p, rows, s);
  SEA_CHECK_LAUNCH(name);
  return SEA_OK;
}

}  // namespace sea

using namespace sea;

#ifdef SEA_STAMP
extern "C" int sea_debug_stamps(unsigned long long* host16) {
  hipMemcpyFromSymbol(host16, HIP_SYMBOL(sea_dbg), sizeof(unsigned long long) * 16);
  unsigned long long z[16] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(sea_dbg), z, sizeof(z));
  return 0;
}
#endif

extern "C" int sea_topk_select(const void* probs, int dtype, int64_t N, int64_t H, int64_t T_dst, int64_t T_m,
                               int64_t stride_n, int64_t stride_h, int64_t stride_t, const int32_t* keep,
                               int64_t keep_stride_n, int64_t T_src, int is_causal, int max_k, uint32_t* bits,
                               float* mask_out, int32_t* row_nnz, int32_t* head_off, sea_stream_t stream) {
  return select_common<false>("sea_topk_select", probs, dtype, N, H, T_dst, T_m, stride_n, stride_h, stride_t, keep,
                              keep_stride_n, T_src, is_causal, max_k, bits, mask_out, row_nnz, head_off,
                              (hipStream_t)stream);
}

extern "C" int sea_mask_to_bits(const void* mask, int dtype, int64_t N, int64_t H, int64_t T_dst, int64_t T_m,
                                int64_t stride_n, int64_t stride_h, int64_t stride_t, int64_t T_src, int is_causal,
                                int max_k, uint32_t* bits, int32_t* row_nnz, int32_t* head_off, sea_stream_t stream) {
  return select_common<true>("sea_mask_to_bits", mask, dtype, N, H, T_dst, T_m, stride_n, stride_h, stride_t, nullptr, 0,
                             T_src, is_causal, max_k, bits, nullptr, row_nnz, head_off, (hipStream_t)stream);
}

template <typename T>
static int launch_tail_select(const TailParams& tp, const TopkParams& p, int64_t rows, hipStream_t s) {
  const int ept = ((p.nchunks + TK_THREADS - 1) / TK_THREADS) * 4;
  size_t lds = (size_t)(((tp.H + 15) / 16) * 16) * (tp.W4 + 3) * sizeof(float) + (size_t)TAIL_TAB_ROWS * 256 * sizeof(uint32_t);
  if (lds < 2 * TK_CAND_CAP * sizeof(uint32_t)) lds = 2 * TK_CAND_CAP * sizeof(uint32_t);     // the selection's candidate list re-uses it
  dim3 grid((unsigned)rows), block(TK_THREADS);
#define SEA_TSEL(EE)                                                                                              \
  do {                                                                                                            \
    /* always the ragged form (FULL = false): with every trip count known the compiler interleaves the eight heads of a wave */ \
    /* further and needs 88 registers where this form needs 78 (H = 32; spilling them to reach the same occupancy: 453 us   */ \
    /* against 399 us); what FULL saves is a compare per chunk                                                              */ \
    hipLaunchKernelGGL((predictor_tail_select_kernel<T, EE, false>), grid, block, lds, s, tp, p);                 \
  } while (0)
  if (ept <= 4) SEA_TSEL(4);
  else if (ept <= 8) SEA_TSEL(8);
  else if (ept <= 16) SEA_TSEL(16);
  else if (ept <= 32) SEA_TSEL(32);
  else if (ept <= 40) SEA_TSEL(40);
  else SEA_TSEL(64);
#undef SEA_TSEL
  return SEA_OK;
}

static int launch_tail_select_f32(const TailParams& tp, const TopkParams& p, int64_t rows, hipStream_t s) {
  const int ept = ((p.nchunks + TK_THREADS - 1) / TK_THREADS) * 4;
  size_t lds = (size_t)(((tp.H + 15) / 16) * 16) * (tp.W4 + 3) * sizeof(float) + (size_t)TAIL_TAB_ROWS * 256 * sizeof(uint32_t);
  if (lds < 2 * TK_CAND_CAP * sizeof(uint32_t)) lds = 2 * TK_CAND_CAP * sizeof(uint32_t);
  dim3 grid((unsigned)rows), block(TK_THREADS);
  if (ept <= 4) hipLaunchKernelGGL((predictor_tail_select_f32_kernel<4>), grid, block, lds, s, tp, p);
  else if (ept <= 8) hipLaunchKernelGGL((predictor_tail_select_f32_kernel<8>), grid, block, lds, s, tp, p);
  else if (ept <= 16) hipLaunchKernelGGL((predictor_tail_select_f32_kernel<16>), grid, block, lds, s, tp, p);
  else if (ept <= 32) hipLaunchKernelGGL((predictor_tail_select_f32_kernel<32>), grid, block, lds, s, tp, p);
  else return SEA_EUNSUPPORTED;
  return SEA_OK;
}

template <typename T>
static int launch_tail_select_gen(const TailParams& tp, const TopkParams& p, int64_t rows, hipStream_t s) {
  const int ept = ((p.nchunks + TK_THREADS - 1) / TK_THREADS) * 4;
  const int E = (tp.T_M + 63) / 64;
  const int EE = E <= 4 ? E : E <= 6 ? 6 : 8;        // the widths launch_tail_mfma instantiates (same lane <-> pixel map: same bits)
  // z tile + constants table (the candidate list of the selection re-uses them: at least its 8 KB), then the flat image
  size_t zt = (size_t)(((tp.H + 15) / 16) * 16) * (tp.W4 + 3) * sizeof(float) + (size_t)TAIL_TAB_ROWS * 64 * EE * sizeof(uint32_t);
  if (zt < 2 * TK_CAND_CAP * sizeof(uint32_t)) zt = 2 * TK_CAND_CAP * sizeof(uint32_t);
  const size_t lds = zt + (((size_t)p.M * 2 + 15) & ~(size_t)15);
  if (lds + 12 * 1024 > 160 * 1024) return SEA_EUNSUPPORTED;
  dim3 grid((unsigned)rows), block(TK_THREADS);
#define SEA_TSG(EV, PV)                                                                                            \
  do {                                                                                                             \
    static DevOnce once;                                                                                           \
    if (lds > 48 * 1024 && once.first()) SEA_MAX_LDS((predictor_tail_select_gen_kernel<T, EV, PV>), 148 * 1024);   \
    hipLaunchKernelGGL((predictor_tail_select_gen_kernel<T, EV, PV>), grid, block, lds, s, tp, p);                 \
  } while (0)
#define SEA_TSG_E(PV)                                                                                              \
  switch (EE) {                                                                                                    \
    case 1: SEA_TSG(1, PV); break; case 2: SEA_TSG(2, PV); break; case 3: SEA_TSG(3, PV); break; case 4: SEA_TSG(4, PV); break; \
    case 6: SEA_TSG(6, PV); break; default: SEA_TSG(8, PV); break;                                                 \
  }
  if (ept <= 8) { SEA_TSG_E(8) }
  else if (ept <= 16) { SEA_TSG_E(16) }
  else if (ept <= 32) { SEA_TSG_E(32) }
  else { SEA_TSG_E(64) }
#undef SEA_TSG_E
#undef SEA_TSG
  return SEA_OK;
}

static int tail_select_common(const char* nm, const float* z, const void* y, int dtype, int64_t N, int64_t C, int64_t H, int64_t T, int64_t W4,
                              int64_t up, int64_t T_m, const int64_t* y_strides, const void* conv_b,
                              const void* conv_w16, int64_t Cp, const void* gamma, const void* beta, float eps,
                              void* probs, void* scores, const int32_t* keep, int64_t keep_stride_n,
                              int64_t T_src, const int32_t* t_src_dev, int is_causal, int max_k, uint32_t* bits, int32_t* row_nnz,
                              int32_t* head_off, int32_t* crow1, const uint32_t* consts_tab, sea_stream_t stream) {
  SEA_REQUIRE(crow1 == nullptr || T == 1, SEA_EINVAL, "%s: crow_out goes with one row per batch item (T = %lld)", nm, (long long)T);
  SEA_REQUIRE((z || (y && y_strides && conv_w16)) && conv_b && gamma && beta && keep && bits && row_nnz && head_off, SEA_EINVAL,
              "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16 || dtype == SEA_F32, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && (z || C > 0) && H > 0 && T > 0 && T_src >= T && max_k > 0, SEA_EINVAL, "%s: bad shape", nm);
  const bool tm256 = T_m == 256 && W4 == 64 && up == 4 && H % 4 == 0;       // the register-resident form
  // fp32 data (round 5): the T_m = 256 form only, H <= 32, map written (probs != NULL), conv_w16 = the (C, Hpad) fp32 transposed weights
  SEA_REQUIRE(dtype != SEA_F32 || (tm256 && H <= 32 && z == nullptr && t_src_dev == nullptr && probs != nullptr), SEA_EUNSUPPORTED,
              "%s: fp32 data takes T_m = 256, H <= 32, H %% 4 == 0, a written map, no z / decode form", nm);
  SEA_REQUIRE(W4 * up == T_m && T_m % 4 == 0 && T_m <= 512 && H <= 64 && H * T_m <= 16384 && W4 + 1 < 1024 && W4 * up + 2 <= 2 * T_m,
              SEA_EUNSUPPORTED, "%s: needs W4 * up == T_m, T_m %% 4 == 0, T_m <= 512, H <= 64, H * T_m <= 16384", nm);
  SEA_REQUIRE(tm256 || t_src_dev == nullptr, SEA_EUNSUPPORTED, "%s: the decode form takes T_m = 256 (W4 = 64, up = 4), H %% 4 == 0", nm);
  if (z) {
    SEA_REQUIRE(W4 % 4 == 0 && (((uintptr_t)z | (uintptr_t)probs | (uintptr_t)scores) & 15) == 0, SEA_EUNSUPPORTED,
                "%s: z rows must be whole 16-byte vectors, 16-byte aligned", nm);
  } else {
    SEA_REQUIRE(y_strides[1] == 1 && C % 8 == 0 && y_strides[0] % 8 == 0 && y_strides[2] % 8 == 0 && y_strides[3] % 8 == 0 &&
                    y_strides[4] % 8 == 0 && (dtype == SEA_F32 || (Cp % 32 == 0 && Cp >= C)) &&
                    (((uintptr_t)y | (uintptr_t)conv_w16 | (uintptr_t)probs | (uintptr_t)scores) & 15) == 0,
                SEA_EUNSUPPORTED, "%s: y must be channels-last / C8 with 16-byte aligned vectors", nm);
  }
  SEA_REQUIRE(N * T < (1ll << 31), SEA_EUNSUPPORTED, "%s: too many rows", nm);
  TailParams tp;
  tp.y = y; tp.w4 = dtype == SEA_F32 ? conv_w16 : nullptr; tp.b4 = conv_b; tp.gamma = gamma; tp.beta = beta; tp.probs = probs; tp.scores = scores; tp.eps = eps;
  tp.N = (int)N; tp.C = (int)C; tp.H = (int)H; tp.T = (int)T; tp.W4 = (int)W4; tp.UP = (int)up; tp.T_M = (int)T_m;
  tp.ys_n = tp.ys_c = tp.ys_t = tp.ys_w = tp.ys_c8 = 0;
  if (!z) { tp.ys_n = y_strides[0]; tp.ys_c = y_strides[1]; tp.ys_t = y_strides[2]; tp.ys_w = y_strides[3]; tp.ys_c8 = y_strides[4]; }
  tp.w16 = conv_w16; tp.Cp = (int)Cp; tp.z = z;
  tp.tab = (consts_tab != nullptr && tm256 && (((uintptr_t)consts_tab) & 15) == 0) ? consts_tab : nullptr;   // (the register-resident form reads it)
  TopkParams p;
  p.src = nullptr; p.sn = H * T * T_m; p.sh = T * T_m; p.st = T_m;        // (the packed-key selection never re-reads the map)
  p.H = (int)H; p.T_dst = (int)T; p.T_m = (int)T_m; p.T_src = (int)T_src;
  p.is_causal = is_causal; p.max_k = max_k;
  p.M = (int)(H * T_m); p.nchunks = p.M / 4; p.W = (p.M + 31) / 32; p.G = group_lanes((int)T_m);
  p.keep = keep; p.keep_stride_n = keep_stride_n;
  p.bits = bits; p.mask_out = nullptr; p.row_nnz = row_nnz; p.head_off = head_off; p.t_src_dev = t_src_dev; p.crow1 = crow1;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (dtype == SEA_F32) rc = launch_tail_select_f32(tp, p, N * T, s);
  else if (tm256) rc = dtype == SEA_F16 ? launch_tail_select<__half>(tp, p, N * T, s) : launch_tail_select<__hip_bfloat16>(tp, p, N * T, s);
  else rc = dtype == SEA_F16 ? launch_tail_select_gen<__half>(tp, p, N * T, s) : launch_tail_select_gen<__hip_bfloat16>(tp, p, N * T, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: this (H, T_m) does not fit the fused kernel's LDS plan (run sea_predictor_tail + sea_topk_select)", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_predictor_tail_select(const void* y, int dtype, int64_t N, int64_t C, int64_t H, int64_t T, int64_t W4,
                                         int64_t up, int64_t T_m, const int64_t* y_strides, const void* conv_b,
                                         const void* conv_w16, int64_t Cp, const void* gamma, const void* beta, float eps,
                                         void* probs, void* scores, const int32_t* keep, int64_t keep_stride_n,
                                         int64_t T_src, int is_causal, int max_k, uint32_t* bits, int32_t* row_nnz,
                                         int32_t* head_off, const uint32_t* consts_tab, sea_stream_t stream) {
  return tail_select_common("sea_predictor_tail_select", nullptr, y, dtype, N, C, H, T, W4, up, T_m, y_strides, conv_b, conv_w16, Cp, gamma,
                            beta, eps, probs, scores, keep, keep_stride_n, T_src, nullptr, is_causal, max_k, bits, row_nnz,
                            head_off, nullptr, consts_tab, stream);
}

// The same launch fed with z = the 1x1 convolution's output (N, T, H, W4) fp32 as sea_causal_conv_c8_z's epilogue writes it:
// the z tile of a row is then a copy into LDS instead of loads + MFMAs (a third of a row's life in this issue-bound kernel).
extern "C" int sea_predictor_tail_select_z(const float* z, int dtype, int64_t N, int64_t H, int64_t T, int64_t W4, int64_t up,
                                           int64_t T_m, const float* conv_b, const void* gamma, const void* beta, float eps,
                                           void* probs, void* scores, const int32_t* keep, int64_t keep_stride_n, int64_t T_src,
                                           int is_causal, int max_k, uint32_t* bits, int32_t* row_nnz, int32_t* head_off,
                                           const uint32_t* consts_tab, sea_stream_t stream) {
  SEA_REQUIRE(z, SEA_EINVAL, "sea_predictor_tail_select_z: null pointer");
  return tail_select_common("sea_predictor_tail_select_z", z, nullptr, dtype, N, 0, H, T, W4, up, T_m, nullptr, conv_b, nullptr, 0,
                            gamma, beta, eps, probs, scores, keep, keep_stride_n, T_src, nullptr, is_causal, max_k, bits, row_nnz,
                            head_off, nullptr, consts_tab, stream);
}

// Decode form (a step captured as a HIP graph): the T new rows are the LAST rows of sequences of *t_src_dev tokens (device
// memory); keep_table[i] = K of the row with i+1 visible keys, for every position the session can reach.
extern "C" int sea_predictor_tail_select_at(const void* y, int dtype, int64_t N, int64_t C, int64_t H, int64_t T, int64_t W4,
                                            int64_t up, int64_t T_m, const int64_t* y_strides, const void* conv_b,
                                            const void* conv_w16, int64_t Cp, const void* gamma, const void* beta, float eps,
                                            void* probs, void* scores, const int32_t* keep_table, const int32_t* t_src_dev,
                                            int is_causal, int max_k, uint32_t* bits, int32_t* row_nnz, int32_t* head_off,
                                            int32_t* crow_out, const uint32_t* consts_tab, sea_stream_t stream) {
  SEA_REQUIRE(t_src_dev, SEA_EINVAL, "sea_predictor_tail_select_at: null pointer");
  return tail_select_common("sea_predictor_tail_select_at", nullptr, y, dtype, N, C, H, T, W4, up, T_m, y_strides, conv_b, conv_w16, Cp,
                            gamma, beta, eps, probs, scores, keep_table, 0, T, t_src_dev, is_causal, max_k, bits, row_nnz,
                            head_off, crow_out, consts_tab, stream);
}

// One launch for a decoding step's predictor CNN + tail + selection + state advance (DecodeCnnParams above; round 5).
template <typename T>
static int launch_decode_cnn(const DecodeCnnParams& dp, const TailParams& tp, const TopkParams& p, const EmitParams& ep, hipStream_t s) {
  const int ept = ((p.nchunks + TK_THREADS - 1) / TK_THREADS) * 4;
  size_t lds = (size_t)(((tp.H + 15) / 16) * 16) * (tp.W4 + 3) * sizeof(float) + (size_t)TAIL_TAB_ROWS * 256 * sizeof(uint32_t);
  if (lds < 2 * TK_CAND_CAP * sizeof(uint32_t)) lds = 2 * TK_CAND_CAP * sizeof(uint32_t);
  const int nt = (dp.C + 15) / 16, kch = (dp.C + 31) / 32;
  const size_t wimg = (size_t)(16 * nt) * 9 * kch * 64;                    // the weight image overlays the (later) z tile
  if (lds < wimg) lds = wimg;
  dim3 grid((unsigned)tp.N), block(TK_THREADS);
#define SEA_DCNN(EE, NTV, KV)                                                                                          \
  do {                                                                                                                 \
    constexpr bool EM = (NTV) <= 4;                                                                                    \
    if (!EM && ep.col != nullptr) return SEA_EUNSUPPORTED;                                                             \
    static DevOnce once;                                                                                               \
    if (lds > 32 * 1024 && once.first()) SEA_MAX_LDS((decode_cnn_tail_select_kernel<T, EE, NTV, KV, EM>), lds);        \
    hipLaunchKernelGGL((decode_cnn_tail_select_kernel<T, EE, NTV, KV, EM>), grid, block, lds, s, dp, tp, p, ep);       \
  } while (0)
  if (ept <= 8 && nt == 1 && kch == 1) { if (ept <= 4) SEA_DCNN(4, 1, 1); else SEA_DCNN(8, 1, 1); }
  else if (ept <= 16 && nt == 2 && kch == 1) SEA_DCNN(16, 2, 1);
  else if (ept <= 32 && nt == 3 && kch == 2) SEA_DCNN(32, 3, 2);
  else if (ept <= 32 && nt == 4 && kch == 2) SEA_DCNN(32, 4, 2);
  else if (ept <= 40 && nt == 5 && kch == 3) SEA_DCNN(40, 5, 3);
  else return SEA_EUNSUPPORTED;
#undef SEA_DCNN
  return SEA_OK;
}

extern "C" int sea_decode_cnn_tail_select(const void* x_new, void* x_ring, void* y1_ring, void* y2, int dtype, int64_t N, int64_t C,
                                          int64_t H, int64_t W4, int64_t ring_x, int64_t ring_y, const void* w1_packed,
                                          const float* bias1, const void* w2_packed, const float* bias2, int64_t CinP, int dilation,
                                          int pad_w, const void* conv_b, const void* conv_w16, int64_t Cp, const void* gamma,
                                          const void* beta, float eps, void* probs, const int32_t* keep_table, int32_t* counters,
                                          int32_t* ticket, int is_causal, int max_k, uint32_t* bits, int32_t* row_nnz,
                                          int32_t* head_off, int32_t* crow_out, int32_t* col, int64_t col_stride_n, int64_t z_cap,
                                          int64_t T_cap, const uint32_t* consts_tab, sea_stream_t stream) {
  const char* nm = "sea_decode_cnn_tail_select";
  SEA_REQUIRE(col == nullptr || (C <= 64 && col_stride_n >= z_cap && z_cap > 0 && T_cap > 0 && H * T_cap < (1ll << 31)), SEA_EUNSUPPORTED,
              "%s: the in-launch emit serves C <= 64 channels (beyond that the weight image leaves no LDS for it: pass col = NULL "
              "and call sea_csr_emit_at with t_src_dev = counters + 2)", nm);
  SEA_REQUIRE(x_new && x_ring && y1_ring && y2 && w1_packed && bias1 && w2_packed && bias2 && conv_b && conv_w16 && gamma && beta &&
                  keep_table && counters && ticket && bits && row_nnz && head_off && crow_out, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16, SEA_EUNSUPPORTED, "%s: 16-bit data only (dtype %d)", nm, dtype);
  SEA_REQUIRE(N > 0 && N < (1 << 20) && H > 0 && H <= 64 && H % 4 == 0 && C == 2 * H && C % 8 == 0 && W4 == 64 && max_k > 0, SEA_EUNSUPPORTED,
              "%s: needs T_m = 256 (W4 = 64), H %% 4 == 0, C = 2 H channels in whole blocks of 8", nm);
  SEA_REQUIRE(CinP == (C + 31) / 32 * 32 && Cp % 32 == 0 && Cp >= C && dilation > 0 && 2 * pad_w == 2 * dilation, SEA_EUNSUPPORTED,
              "%s: 3 x 3 width-preserving convolutions with CinP = C rounded up to 32", nm);
  SEA_REQUIRE(ring_x > 2 * dilation && ring_y > 2 * dilation, SEA_EINVAL,
              "%s: a ring must hold the rows t - 2 dil .. t in distinct slots (more than 2 * dilation of them)", nm);
  SEA_REQUIRE((((uintptr_t)x_new | (uintptr_t)x_ring | (uintptr_t)y1_ring | (uintptr_t)y2 | (uintptr_t)w1_packed | (uintptr_t)w2_packed |
                (uintptr_t)conv_w16 | (uintptr_t)probs) & 15) == 0, SEA_EUNSUPPORTED, "%s: 16-byte alignment", nm);
  DecodeCnnParams dp;
  dp.x_new = x_new; dp.x_ring = x_ring; dp.y1_ring = y1_ring; dp.y2 = y2; dp.w1 = w1_packed; dp.w2 = w2_packed; dp.b1 = bias1; dp.b2 = bias2;
  dp.counters = counters; dp.ticket = ticket;
  dp.C = (int)C; dp.W = (int)W4; dp.RX = (int)ring_x; dp.RY = (int)ring_y; dp.dil = dilation; dp.pad_w = pad_w;
  TailParams tp;
  tp.y = y2; tp.w4 = nullptr; tp.b4 = conv_b; tp.gamma = gamma; tp.beta = beta; tp.probs = probs; tp.scores = nullptr; tp.eps = eps;
  tp.N = (int)N; tp.C = (int)C; tp.H = (int)H; tp.T = 1; tp.W4 = (int)W4; tp.UP = 4; tp.T_M = 256;
  tp.ys_n = C * W4; tp.ys_c = 1; tp.ys_t = 0; tp.ys_w = 8; tp.ys_c8 = W4 * 8;          // one C8 row per batch item
  tp.w16 = conv_w16; tp.Cp = (int)Cp; tp.z = nullptr;
  tp.tab = (consts_tab != nullptr && (((uintptr_t)consts_tab) & 15) == 0) ? consts_tab : nullptr;
  TopkParams p;
  p.src = nullptr; p.sn = H * 256; p.sh = 256; p.st = 256;
  p.H = (int)H; p.T_dst = 1; p.T_m = 256; p.T_src = 1;
  p.is_causal = is_causal; p.max_k = max_k;
  p.M = (int)(H * 256); p.nchunks = p.M / 4; p.W = (p.M + 31) / 32; p.G = group_lanes(256);
  p.keep = keep_table; p.keep_stride_n = 0;
  p.bits = bits; p.mask_out = nullptr; p.row_nnz = row_nnz; p.head_off = head_off; p.t_src_dev = counters + 1; p.crow1 = crow_out;
  EmitParams ep;
  ep.bits = bits; ep.crow = crow_out; ep.H = (int)H; ep.T_dst = 1; ep.T_m = 256; ep.T_src = 1; ep.is_causal = is_causal; ep.max_k = max_k;
  ep.W = p.W; ep.col = col; ep.col_stride_n = col_stride_n; ep.z_cap = z_cap; ep.values_out = nullptr; ep.T_enc = (int)T_cap;
  ep.t_src_dev = counters + 1;
  hipStream_t s = (hipStream_t)stream;
  const int rc = dtype == SEA_F16 ? launch_decode_cnn<__half>(dp, tp, p, ep, s) : launch_decode_cnn<__hip_bfloat16>(dp, tp, p, ep, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: this head / channel count has no fused decode instantiation", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_csr_row_scan(const int32_t* row_nnz, int64_t N, int64_t T_dst, void* crow, int idx_bytes,
                                sea_stream_t stream) {
  SEA_REQUIRE(row_nnz && crow, SEA_EINVAL, "sea_csr_row_scan: null pointer");
  SEA_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SEA_EINVAL, "sea_csr_row_scan: idx_bytes must be 4 or 8");
  SEA_REQUIRE(N > 0 && T_dst > 0, SEA_EINVAL, "sea_csr_row_scan: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (idx_bytes == 4)
    hipLaunchKernelGGL((row_scan_kernel<int32_t>), dim3((unsigned)N), dim3(1024), 0, s, row_nnz, (int)T_dst, (int32_t*)crow);
  else
    hipLaunchKernelGGL((row_scan_kernel<int64_t>), dim3((unsigned)N), dim3(1024), 0, s, row_nnz, (int)T_dst, (int64_t*)crow);
  SEA_CHECK_LAUNCH("sea_csr_row_scan");
  return SEA_OK;
}

static int emit_common(const char* nm, const uint32_t* bits, const void* crow, int64_t N, int64_t H, int64_t T_dst, int64_t T_m,
                       int64_t T_src, const int32_t* t_src_dev, int64_t T_enc, int is_causal, int max_k, void* col, int idx_bytes,
                       int64_t col_stride_n, int64_t z_cap, float* values_out, sea_stream_t stream) {
  SEA_REQUIRE(bits && crow && col, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SEA_EINVAL, "%s: idx_bytes must be 4 or 8", nm);
  SEA_REQUIRE(N > 0 && H > 0 && T_dst > 0 && T_m > 0 && max_k > 0, SEA_EINVAL, "%s: bad shape", nm);
  SEA_REQUIRE(H * T_enc < (1ll << 24) && T_enc >= T_src, SEA_EUNSUPPORTED, "%s: H*T_src must stay below 2^24 (fp32-exact ids)", nm);
  if (z_cap == 0) return SEA_OK;
  EmitParams p;
  p.bits = bits; p.crow = crow; p.T_enc = (int)T_enc; p.t_src_dev = t_src_dev;
  p.H = (int)H; p.T_dst = (int)T_dst; p.T_m = (int)T_m; p.T_src = (int)T_src;
  p.is_causal = is_causal; p.max_k = max_k; p.W = (int)((H * T_m + 31) / 32);
  p.col = col; p.col_stride_n = col_stride_n; p.z_cap = z_cap; p.values_out = values_out;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(N * T_dst)), block(TK_THREADS);
  if (idx_bytes == 4) hipLaunchKernelGGL((csr_emit_kernel<int32_t>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((csr_emit_kernel<int64_t>), grid, block, 0, s, p);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_csr_emit(const uint32_t* bits, const void* crow, const int32_t* head_off, int64_t N, int64_t H,
                            int64_t T_dst, int64_t T_m, int64_t T_src, int is_causal, int max_k, void* col, int idx_bytes,
                            int64_t col_stride_n, int64_t z_cap, float* values_out, sea_stream_t stream) {
  (void)head_off;  // offsets follow from the flat (head-major) emission order; kept in the ABI for symmetry
  return emit_common("sea_csr_emit", bits, crow, N, H, T_dst, T_m, T_src, nullptr, T_src, is_causal, max_k, col, idx_bytes,
                     col_stride_n, z_cap, values_out, stream);
}

// Decode form (a step captured as a HIP graph): the rows' widths follow *t_src_dev (device memory: the current sequence
// length), the column ids are head * T_cap + key with a FIXED capacity T_cap >= *t_src_dev, so the attention launch that
// consumes them (K / V caches of T_cap rows) needs nothing position-dependent in its arguments.
extern "C" int sea_csr_emit_at(const uint32_t* bits, const void* crow, int64_t N, int64_t H, int64_t T_dst, int64_t T_m,
                               const int32_t* t_src_dev, int64_t T_cap, int is_causal, int max_k, void* col, int idx_bytes,
                               int64_t col_stride_n, int64_t z_cap, sea_stream_t stream) {
  SEA_REQUIRE(t_src_dev, SEA_EINVAL, "sea_csr_emit_at: null pointer");
  return emit_common("sea_csr_emit_at", bits, crow, N, H, T_dst, T_m, T_dst, t_src_dev, T_cap, is_causal, max_k, col, idx_bytes,
                     col_stride_n, z_cap, nullptr, stream);
}

extern "C" int sea_csr_head_offsets(const void* crow, const void* col, int idx_bytes, int64_t N, int64_t H,
                                    int64_t T_dst, int64_t T_src, int64_t col_stride_n, int32_t* head_off,
                                    sea_stream_t stream) {
  SEA_REQUIRE(crow && col && head_off, SEA_EINVAL, "sea_csr_head_offsets: null pointer");
  SEA_REQUIRE(idx_bytes == 4 || idx_bytes == 8, SEA_EINVAL, "sea_csr_head_offsets: idx_bytes must be 4 or 8");
  SEA_REQUIRE(H > 0 && H <= 1024, SEA_EUNSUPPORTED, "sea_csr_head_offsets: H must be in 1..1024");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(N * T_dst)), block(TK_THREADS);
  if (idx_bytes == 4)
    hipLaunchKernelGGL((head_offsets_kernel<int32_t>), grid, block, 0, s, (const int32_t*)crow, (const int32_t*)col, (int)H,
                       (int)T_dst, (int)T_src, col_stride_n, head_off);
  else
    hipLaunchKernelGGL((head_offsets_kernel<int64_t>), grid, block, 0, s, (const int64_t*)crow, (const int64_t*)col, (int)H,
                       (int)T_dst, (int)T_src, col_stride_n, head_off);
  SEA_CHECK_LAUNCH("sea_csr_head_offsets");
  return SEA_OK;
}
