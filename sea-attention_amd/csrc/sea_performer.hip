// Causal Performer (generalized ReLU feature map + causal linear attention) as ONE fp32-MFMA kernel, gfx950.
//
// Replaces step B of SEA's estimator (reference: src/models/perlin_attention/attention.py:497-514,556-590 ->
// performer_pytorch.FastAttention, causal=True, generalized_attention=True; not vendored, see
// perlin_attention/performer.py for the restated algorithm) together with the two concatenations around it:
//     v_for_atten = cat([v_eye_learned_causal[:T], v])          (attention.py:506-510)
//     ctx         = FastAttention(q, k, v_for_atten)            (fp32)
//     performer_value = cat([ctx, v])                           (attention.py:577-590)
// Output: performer_value (N,H,T,3D) = [ctx_pos | ctx_v | v] written directly; nothing else touches HBM.
//
//   phi(x) = relu(D^-1/4 * x W^T) + 1e-3                        W: (nb, D) projection
//   ctx_t  = sum_{s<=t} (phi(q_t).phi(k_s)) V_s / (phi(q_t).(sum_{s<=t} phi(k_s) + 1e-6))
//
// One 512-thread workgroup (8 waves) per (n, h) walks the sequence in chunks of C rows.  The running state
// S = sum phi(k)^T V (NBP x 2D) never leaves the accumulator registers of the wave that owns its columns:
// it is both the C operand of the update S += phi(K_c)^T V_c and -- register r of a 16x16 tile being row
// 4g+r of lane group g -- the B operand of the carry term phi(Q_c) S, with the k index of that product
// permuted accordingly on the A side.  All products run on v_mfma_f32_16x16x4_f32 (exact fp32).
#include "sea_common.hpp"
#include <cstdlib>
#include <type_traits>

#ifdef SEA_STAMP
__device__ unsigned long long sea_dbg_perf[16];
__device__ unsigned long long sea_dbg_wg[2048];   // [start, end] (s_memrealtime, 100 MHz) of every workgroup of the wide kernel's output pass
#define PSTAMP(i) do { if (threadIdx.x == 0) { unsigned long long _t = __builtin_amdgcn_s_memtime(); atomicAdd(&sea_dbg_perf[i], _t - _tprev); _tprev = _t; } } while (0)
// (the wide kernel keeps its deltas in registers and adds them once at the end: an atomic per phase sits in front of every vmcnt wait)
#define WSTAMP(i) do { if (threadIdx.x == 0 && (STATE_ONLY || blockIdx.y == 0 || blockIdx.y == gridDim.y - 1)) { unsigned long long _t = __builtin_amdgcn_s_memtime(); _tacc[i] += _t - _tprev; _tprev = _t; } } while (0)
#define WSTAMP_FLUSH() do { if (!STATE_ONLY && threadIdx.x == 0 && (blockIdx.y == 0 || blockIdx.y == gridDim.y - 1)) for (int _i = 0; _i < 4; ++_i) atomicAdd(&sea_dbg_perf[_i + (blockIdx.y ? 0 : 4)], _tacc[_i]); \
    if (STATE_ONLY && threadIdx.x == 0) for (int _i = 0; _i < 4; ++_i) atomicAdd(&sea_dbg_perf[10 + _i], _tacc[_i]); \
    if (threadIdx.x == 0) atomicAdd(&sea_dbg_perf[STATE_ONLY ? 8 : 9], __builtin_amdgcn_s_memtime() - _tstart); \
    if (!STATE_ONLY && threadIdx.x == 0) { const int _w = blockIdx.y * gridDim.x + blockIdx.x; if (_w < 1024) { sea_dbg_wg[2 * _w] = _rstart; sea_dbg_wg[2 * _w + 1] = __builtin_amdgcn_s_memrealtime(); } } } while (0)
#else
#define PSTAMP(i) do {} while (0)
#define WSTAMP(i) do {} while (0)
#define WSTAMP_FLUSH() do {} while (0)
#endif

namespace sea {

using f4 = __attribute__((ext_vector_type(4))) float;
#define SEA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct PerfParams {
  const void *q, *k, *v;   // (N,H,T,D)
  const void* pos;         // (>=T, D) learned causal value embedding
  const float* W;          // (nb, D) fp32
  void* out;               // (N,H,T,3D)
  void* avg;               // optional (N,H,T,D): cumulative average of v (bf16 kernel only)
  int64_t qs[3], ks[3], vs[3];
  int64_t pos_stride;
  int N, H, T, nb;
  // sequence-parallel form (few (n,h) pairs): the T rows are cut into nseg segments of seg_len rows (a multiple of the
  // chunk size).  Pass 1 (STATE_ONLY kernels, grid.y = nseg-1) leaves every segment's state increment in `carry`,
  // pass 2 (grid.y = nseg) starts each segment from the sum of the increments before it.  nseg = 1: one pass.
  float* carry;
  int nseg, seg_len;
  // stateful form (kv-cache decoding): the rows handed in continue a sequence.  state_in / state_out are per-(n,h)
  // images in the carry format (state registers, k-sum, column sum of v); t_base = rows the state has already seen
  // (only the cumulative average needs the absolute row index).  Either may be null.
  const float* state_in;
  float* state_out;
  int t_base;
  // device-resident form of t_base (a captured decode step replayed as a HIP graph: the position lives in memory).  When
  // set, `pos` is the BASE of the embedding table and the kernel reads its rows from row *t_base_dev on.
  const int32_t* t_base_dev;
  // chunk-aligned step (the 16-bit MFMA kernels): the state image is the state at the last CHUNK BOUNDARY c0 =
  // floor(t_base / C) * C, and the open chunk's rows c0 .. t_base-1 are walked again (k, v, pos from the caller's kv-cache,
  // no q, no output), so that every new row is computed by the very instruction sequence the stateless pass runs for it --
  // bitwise the stateless result for any split of the sequence into calls.  k / v / pos then point at row c0 (with
  // t_base_dev: at row 0 of the caches / the table, the kernel adds c0); q / out / avg at the first new row.
  int aligned;
};

typedef __attribute__((ext_vector_type(8))) __bf16 pbf8;
typedef __attribute__((ext_vector_type(8))) _Float16 ph8;
typedef __attribute__((ext_vector_type(4))) short ps4;

// 16-bit storage type of the split operands: bf16 data splits into bf16 terms (2 x 8 significand bits), fp16 data into
// fp16 terms (2 x 11 bits; the state S and the products stay well inside fp16's range for T up to tens of thousands)
template <typename T> struct S16;
template <> struct S16<__hip_bfloat16> {
  static constexpr unsigned short ONE = 0x3F80;
  __device__ static inline unsigned short bits(float x) { return __builtin_bit_cast(unsigned short, __float2bfloat16(x)); }
  __device__ static inline float val(unsigned short b) { return __uint_as_float((uint32_t)b << 16); }
  __device__ static inline f4 mfma(const uint4& a, const uint4& b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pbf8, a), __builtin_bit_cast(pbf8, b), c, 0, 0, 0);
  }
};
template <> struct S16<__half> {
  static constexpr unsigned short ONE = 0x3C00;
  __device__ static inline unsigned short bits(float x) { return __builtin_bit_cast(unsigned short, __float2half(x)); }
  __device__ static inline float val(unsigned short b) { return __half2float(__builtin_bit_cast(__half, b)); }
  __device__ static inline f4 mfma(const uint4& a, const uint4& b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ph8, a), __builtin_bit_cast(ph8, b), c, 0, 0, 0);
  }
};
template <> struct S16<float> {};     // (fp32 data never takes the 16-bit feature-map product)

// NW = waves per workgroup (8: two per SIMD, so one wave's LDS/MFMA latency hides behind the other's issue)
template <typename T, int D, int NBT, int C, int NW, bool STATE_ONLY>
__global__ __launch_bounds__(NW * 64) void performer_kernel(PerfParams p) {
  constexpr int NTH = NW * 64;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int E = 2 * D, NBP = NBT * 16;
  constexpr int RB = C / 16;                  // row blocks per chunk
  constexpr int EB = E / 16;                  // column blocks of V / S / O
  constexpr int JB = (EB + NW - 1) / NW;      // column blocks owned by one wave
  constexpr int LDQ = D + 2, LDV = E + 16, LDP = NBP + 2, LDA = C + 2, LDW = D + 2;
  static_assert(LDA <= LDQ, "the A tile is overlaid on the Q tile");
  // 16-bit data: q, k and the projection ARE 16-bit values, so the feature-map product runs exactly on the 16x16x32
  // 16-bit MFMA (8x the fp32 MFMA rate) from raw 16-bit LDS images [k-chunk of 8][row][8]; everything downstream stays fp32
  constexpr bool IS16 = !std::is_same<T, float>::value;
  constexpr int DK = (D + 31) / 32 * 32;      // feature-map contraction length, zero padded to whole 32-wide k-steps
  constexpr int F0 = IS16 ? 0 : NBP * LDW, F1 = IS16 ? C * LDA : C * LDQ, F2 = IS16 ? 0 : C * LDQ;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;               // NBP x LDW   W (rows >= nb are zero)                      [fp32 data only]
  float* sQ = sW + F0;            // C x LDQ     Q chunk, later the masked A tile (C x LDA)   [16-bit data: the A tile only]
  float* sK = sQ + F1;            // C x LDQ                                                  [fp32 data only]
  float* sV = sK + F2;            // C x LDV     [pos | v]
  float* sQp = sV + C * LDV;      // C x LDP     phi(Q)
  float* sKp = sQp + C * LDP;     // C x LDP     phi(K)
  float* sKsum = sKp + C * LDP;   // NBP         running sum of phi(k)
  float* sDen = sKsum + NBP;      // C           denominators of the current chunk
  constexpr int DSL = RB + NW * 64 / C;       // partial-denominator slots per row: RB key blocks + carry parts
  float* sDenP = sDen + C;        // C x DSL     partials, summed in a fixed order (bitwise reproducible)
  float* sA = sQ;
  static_assert((F0 + F1 + F2 + C * LDV + 2 * C * LDP + NBP + C + C * DSL) % 4 == 0, "16-bit images start 16-byte aligned");
  unsigned short* sW16 = reinterpret_cast<unsigned short*>(sDenP + C * DSL);   // [DK/8][NBP][8]
  unsigned short* sQ16 = sW16 + (DK / 8) * NBP * 8;                             // [DK/8][C][8]
  unsigned short* sK16 = sQ16 + (DK / 8) * C * 8;                               // [DK/8][C][8]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;   // MFMA lane coordinates
  const int nh = blockIdx.x;
  const int n = nh / p.H, h = nh - n * p.H;
  const int seg = blockIdx.y;
  const int t_begin = seg * p.seg_len, t_end = min(p.T, t_begin + p.seg_len);
  const T* qb = reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1];
  const T* kb = reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1];
  const T* vb = reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1];
  const int tb = p.t_base_dev ? *p.t_base_dev : p.t_base;       // rows the state has already seen (block-uniform)
  const T* pb = reinterpret_cast<const T*>(p.pos) + (p.t_base_dev ? (int64_t)tb * p.pos_stride : 0);
  T* ob = reinterpret_cast<T*>(p.out) + (int64_t)nh * p.T * (3 * D);
  const float cnorm = powf((float)D, -0.25f);

  if constexpr (IS16) {
    for (int i = tid; i < (DK / 8) * NBP * 8; i += NTH) {
      const int j = i & 7, f = (i >> 3) % NBP, dd = ((i >> 3) / NBP) * 8 + j;
      sW16[i] = (f < p.nb && dd < D) ? S16<T>::bits(p.W[f * D + dd]) : (unsigned short)0;
    }
    // padding k-chunks (D = 80: chunks 10, 11 of 12) are zeroed once; the staging below never writes them -- and must
    // not race with this loop, so the live chunks are left alone here
    if constexpr (DK > D)
      for (int i = tid; i < 2 * ((DK - D) / 8) * C * 8; i += NTH) {
        const int m = i / (((DK - D) / 8) * C * 8), rest = i - m * (((DK - D) / 8) * C * 8);
        sQ16[(m * (DK / 8) + D / 8) * C * 8 + rest] = 0;
      }
  } else {
    for (int i = tid; i < NBP * LDW; i += NTH) {
      const int r = i / LDW, c = i - r * LDW;
      sW[i] = (r < p.nb && c < D) ? p.W[r * D + c] : 0.f;
    }
  }

  // state = sum of the increments of the segments before this one (fixed order); raw per-thread register images
  constexpr int CARRY = JB * NBT * 4 * NTH + NBP;
  f4 S[JB][NBT];
#pragma unroll
  for (int a = 0; a < JB; ++a)
#pragma unroll
    for (int b = 0; b < NBT; ++b) S[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  {
    float ks0 = 0.f;
    if (!STATE_ONLY && p.state_in) {                       // increments of pass 1 do not include the incoming state
      const float* cr = p.state_in + (int64_t)nh * CARRY;
#pragma unroll
      for (int a = 0; a < JB; ++a)
#pragma unroll
        for (int b = 0; b < NBT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) S[a][b][r] = cr[((a * NBT + b) * 4 + r) * NTH + tid];
      if (tid < NBP) ks0 = cr[JB * NBT * 4 * NTH + tid];
    }
    if (!STATE_ONLY) {
      for (int s2 = 0; s2 < seg; ++s2) {
        const float* cr = p.carry + ((int64_t)nh * (p.nseg - 1) + s2) * CARRY;
#pragma unroll
        for (int a = 0; a < JB; ++a)
#pragma unroll
          for (int b = 0; b < NBT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[a][b][r] += cr[((a * NBT + b) * 4 + r) * NTH + tid];
        if (tid < NBP) ks0 += cr[JB * NBT * 4 * NTH + tid];
      }
    }
    if (tid < NBP) sKsum[tid] = ks0;
  }

#ifdef SEA_STAMP
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
#endif
  // chunk staging is software-pipelined: the global loads of chunk c+1 are issued before the MFMA phases of
  // chunk c and only written to LDS after them (the loads stay in flight across the barriers)
  constexpr int NCH = (C * (D / VEC) + NTH - 1) / NTH;     // 16-byte pieces per thread and tensor
  uint4 pq[NCH], pk[NCH], pv[NCH], pp[NCH];
  auto issue_loads = [&](int t0n) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = tid + i * NTH;
      const int r = ch / (D / VEC), c = (ch - r * (D / VEC)) * VEC;
      pq[i] = pk[i] = pv[i] = pp[i] = make_uint4(0, 0, 0, 0);
      if (ch < C * (D / VEC) && t0n + r < t_end) {
        const int64_t t = t0n + r;
        if (!STATE_ONLY) pq[i] = *reinterpret_cast<const uint4*>(qb + t * p.qs[2] + c);
        pk[i] = *reinterpret_cast<const uint4*>(kb + t * p.ks[2] + c);
        pv[i] = *reinterpret_cast<const uint4*>(vb + t * p.vs[2] + c);
        pp[i] = *reinterpret_cast<const uint4*>(pb + t * p.pos_stride + c);
      }
    }
  };
  issue_loads(t_begin);

  for (int t0 = t_begin; t0 < t_end; t0 += C) {
    const int rows = min(C, t_end - t0);
    // ---- (a) registers -> LDS as fp32: Q, K (C x D), V = [pos | v] (C x 2D); copy v into out[..., 2D:3D] ------
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = tid + i * NTH;
      if (ch < C * (D / VEC)) {
        const int r = ch / (D / VEC), c = (ch - r * (D / VEC)) * VEC;
        float fq[VEC], fk[VEC], fv[VEC], fp[VEC];
        if (!STATE_ONLY && r < rows) *reinterpret_cast<uint4*>(ob + (int64_t)(t0 + r) * (3 * D) + 2 * D + c) = pv[i];
        unpack16<T>(pv[i], fv); unpack16<T>(pp[i], fp);
        if constexpr (IS16) {                                // raw 16-byte pieces: k-chunk c/8, row r
          *reinterpret_cast<uint4*>(sQ16 + ((c / 8) * C + r) * 8) = pq[i];
          *reinterpret_cast<uint4*>(sK16 + ((c / 8) * C + r) * 8) = pk[i];
        } else {
          unpack16<T>(pq[i], fq); unpack16<T>(pk[i], fk);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if constexpr (!IS16) {
            sQ[r * LDQ + c + j] = fq[j];
            sK[r * LDQ + c + j] = fk[j];
          }
          sV[r * LDV + c + j] = fp[j];
          sV[r * LDV + D + c + j] = fv[j];
        }
      }
    }
    if (t0 + C < t_end) issue_loads(t0 + C);               // next chunk: in flight during phases (b)..(e)
    if (!STATE_ONLY) for (int i = tid; i < C * DSL; i += NTH) sDenP[i] = 0.f;
    __syncthreads();
    PSTAMP(0);   // (a) staging

    // ---- (b) feature maps phi(Q), phi(K): (C x D) @ W^T -> (C x NBP) -------------------------------------
    // a wave takes (matrix, row block) pairs: one A fragment feeds NBT independent accumulators
    for (int grp = (STATE_ONLY ? RB : 0) + wv; grp < 2 * RB; grp += NW) {   // the state-only pass needs phi(K) alone
      const int which = grp / RB, ib = grp - which * RB;        // 0: Q, 1: K
      if constexpr (IS16) {
        // transposed product X^T[f][t] = sum_d W[f][d] x[t][d]: a lane ends up with 4 consecutive features of one row
        const unsigned short* src16 = which ? sK16 : sQ16;
        f4 acc[NBT];
#pragma unroll
        for (int fb = 0; fb < NBT; ++fb) acc[fb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DK / 32; ++ks) {
          const uint4 bx = *reinterpret_cast<const uint4*>(src16 + ((4 * ks + lg) * C + ib * 16 + li) * 8);
#pragma unroll
          for (int fb = 0; fb < NBT; ++fb) {
            const uint4 aw = *reinterpret_cast<const uint4*>(sW16 + ((4 * ks + lg) * NBP + fb * 16 + li) * 8);
            acc[fb] = S16<T>::mfma(aw, bx, acc[fb]);
          }
        }
        float* dst = which ? sKp : sQp;
        const int row = ib * 16 + li;
#pragma unroll
        for (int fb = 0; fb < NBT; ++fb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = fb * 16 + lg * 4 + r;
            float val = fmaxf(cnorm * acc[fb][r], 0.f) + 1e-3f;
            if (col >= p.nb || row >= rows) val = 0.f;       // padded features / rows beyond T contribute nothing
            dst[row * LDP + col] = val;
          }
        continue;
      }
      const float* src = which ? sK : sQ;
      f4 acc[NBT];
#pragma unroll
      for (int jb = 0; jb < NBT; ++jb) acc[jb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int ks = 0; ks < D / 4; ++ks) {
        const float a = src[(ib * 16 + li) * LDQ + ks * 4 + lg];       // A[i][k]
#pragma unroll
        for (int jb = 0; jb < NBT; ++jb) {
          const float b = sW[(jb * 16 + li) * LDW + ks * 4 + lg];      // B[k][j] = W[j][k]
          acc[jb] = SEA_MFMA(a, b, acc[jb]);
        }
      }
      float* dst = which ? sKp : sQp;
#pragma unroll
      for (int jb = 0; jb < NBT; ++jb) {
        const int col = jb * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = ib * 16 + lg * 4 + r;
          float val = fmaxf(cnorm * acc[jb][r], 0.f) + 1e-3f;
          if (col >= p.nb || row >= rows) val = 0.f;       // padded features / rows beyond T contribute nothing
          dst[row * LDP + col] = val;
        }
      }
    }
    __syncthreads();
    PSTAMP(1);   // (b) feature maps

    // ---- (c) A = tril(phi(Q) phi(K)^T) (C x C, lower-triangular blocks), row sums into sDen -------------------
    if constexpr (!STATE_ONLY) {
    for (int tile = wv; tile < RB * (RB + 1) / 2; tile += NW) {
      int ib = 0, rem = tile;
      while (rem > ib) { rem -= ib + 1; ++ib; }          // tile -> (ib, jb) with jb <= ib
      const int jb = rem;
      f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NBP / 4; ++ks) {
        const float a = sQp[(ib * 16 + li) * LDP + ks * 4 + lg];   // A[i][r]
        const float b = sKp[(jb * 16 + li) * LDP + ks * 4 + lg];   // B[r][j] = Kp[j][r]
        acc = SEA_MFMA(a, b, acc);
      }
      const int col = jb * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ib * 16 + lg * 4 + r;
        const float val = (col <= row) ? acc[r] : 0.f;
        sA[row * LDA + col] = val;                       // NOTE: overlays sQ, which step (b) no longer needs
        float s = val;                                   // sum over the 16 columns held by lanes li = 0..15
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        if (li == 0) sDenP[row * DSL + jb] = s;            // one writer per (row, key block)
      }
    }
    // denominators' carry part: phi(q_i) . (ksum + eps)   (ksum = state BEFORE this chunk)
    {
      constexpr int PARTS = NTH / C;                       // threads per row
      const int row = tid % C, part = tid / C;
      const int per = (NBP + PARTS - 1) / PARTS;
      float s = 0.f;
      for (int r = part * per; r < min(p.nb, (part + 1) * per); ++r) s = fmaf(sQp[row * LDP + r], sKsum[r] + 1e-6f, s);
      sDenP[row * DSL + RB + part] = s;
    }
    __syncthreads();
    if (tid < C) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < DSL; ++i) s += sDenP[tid * DSL + i];
      sDen[tid] = s;
    }
    __syncthreads();
    }
    PSTAMP(2);   // (c) A + denominators
    // the diagonal part of the denominator also carries eps: sum_r phi(q)_r * eps is already in the carry term;
    // the intra-chunk part needs none (eps is added once to the k-sum, not per key).

    // ---- (d) O = A V + phi(Q) S, divided by the denominators; (e) S += phi(K)^T V -----------------------------
    // per owned column block: all RB row blocks of O are accumulated together (one V / S fragment feeds RB MFMAs)
#pragma unroll
    for (int a_ = 0; a_ < JB; ++a_) {
      const int jb = wv + NW * a_;
      if (jb < EB) {
        if constexpr (!STATE_ONLY) {
        f4 o[RB];
#pragma unroll
        for (int ib = 0; ib < RB; ++ib) o[ib] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb2 = 0; kb2 < RB; ++kb2) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const float b = sV[(kb2 * 16 + ks * 4 + lg) * LDV + jb * 16 + li];
#pragma unroll
            for (int ib = 0; ib < RB; ++ib) {
              if (ib >= kb2) {                                   // causal: key blocks above the diagonal are zero
                const float a = sA[(ib * 16 + li) * LDA + kb2 * 16 + ks * 4 + lg];
                o[ib] = SEA_MFMA(a, b, o[ib]);
              }
            }
          }
        }
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            // k-step `ks` of this product sums over state rows {rb*16 + 4g + ks}: register ks of the S tile
            const float b = S[a_][rb][ks];
#pragma unroll
            for (int ib = 0; ib < RB; ++ib) {
              const float a = sQp[(ib * 16 + li) * LDP + rb * 16 + 4 * lg + ks];
              o[ib] = SEA_MFMA(a, b, o[ib]);
            }
          }
        }
        const int col = jb * 16 + li;
#pragma unroll
        for (int ib = 0; ib < RB; ++ib) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = ib * 16 + lg * 4 + r;
            if (row < rows) ob[(int64_t)(t0 + row) * (3 * D) + col] = from_f<T>(o[ib][r] / sDen[row]);
          }
        }
        }
        // (e) state update: one V fragment feeds the NBT state tiles of this column block
#pragma unroll 4
        for (int ks = 0; ks < C / 4; ++ks) {
          const float b = sV[(ks * 4 + lg) * LDV + jb * 16 + li];           // B[k][j]
#pragma unroll
          for (int rb = 0; rb < NBT; ++rb) {
            const float a = sKp[(ks * 4 + lg) * LDP + rb * 16 + li];        // A[r][k] = Kp[k][r]
            S[a_][rb] = SEA_MFMA(a, b, S[a_][rb]);
          }
        }
      }
    }
    __syncthreads();
    PSTAMP(3);   // (d)+(e)
    // running sum of phi(k) (after every wave has used the old value in step (c))
    if (tid < NBP) {
      float s = sKsum[tid];
      for (int r = 0; r < C; ++r) s += sKp[r * LDP + tid];
      sKsum[tid] = s;
    }
    __syncthreads();
    PSTAMP(4);   // ksum
  }
  if constexpr (STATE_ONLY) {
    float* cw = p.carry + ((int64_t)nh * (p.nseg - 1) + seg) * CARRY;
#pragma unroll
    for (int a = 0; a < JB; ++a)
#pragma unroll
      for (int b = 0; b < NBT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cw[((a * NBT + b) * 4 + r) * NTH + tid] = S[a][b][r];
    if (tid < NBP) cw[JB * NBT * 4 * NTH + tid] = sKsum[tid];
  } else if (p.state_out && seg == p.nseg - 1) {           // the block that walked the last rows holds the final state
    float* cw = p.state_out + (int64_t)nh * CARRY;
#pragma unroll
    for (int a = 0; a < JB; ++a)
#pragma unroll
      for (int b = 0; b < NBT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cw[((a * NBT + b) * 4 + r) * NTH + tid] = S[a][b][r];
    if (tid < NBP) cw[JB * NBT * 4 * NTH + tid] = sKsum[tid];
  }
}


// ======================================================================================================
// bf16 variant (D = 64): the same algorithm on v_mfma_f32_16x16x32_bf16 -- 16x the fp32 MFMA rate.
//
// What keeps it accurate: q, k, v, pos and the projection ARE bf16 values, so the feature-map product is exact
// in the products and fp32 in the sums, as before.  Everything that is genuinely fp32 afterwards -- phi(q),
// phi(k), the masked tile A, the running state S -- enters the matrix cores SPLIT in two bf16 terms
// (x = hi + lo, hi = bf16(x), lo = bf16(x - hi): 16 significand bits), and a product of two split operands
// takes three MFMAs (hi.hi + hi.lo + lo.hi), one split against an exact bf16 operand takes two.  Relative
// error 2^-17 per term, accumulated in fp32, in front of a result that is rounded to bf16 (2^-9).
// The state S itself stays fp32 in the accumulators for the whole sequence.
//
// Layout notes (16x16x32 operand maps: A lane l = row l%16, k = 8(l/16)+j; B lane l = col l%16, same k):
//   * X^T = W . Q^T is computed transposed, so a lane holds 4 consecutive FEATURES of one row: the split
//     values leave as 8-byte row-major pieces, which is what the later A-operand reads want.
//   * the tile A is computed transposed too (A^T = phi(K) . phi(Q)^T): same reason, and its row sums (the
//     denominators) become an in-lane sum plus two cross-lane adds.
//   * products that contract over the chunk rows (A.V, phi(K)^T.V) need V and phi(K) "k-major"; both stay
//     row-major in LDS and are fetched with the transposing read ds_read_b64_tr_b16.
//   * S is, as in the fp32 kernel, the B operand of phi(Q).S straight from the accumulator registers; element j
//     of lane group g is feature 16*(2kk + j/4) + 4g + j%4, and the phi(Q) fragment is gathered in that order.
// ======================================================================================================
// x -> (hi, lo) 16-bit patterns with hi + lo ~ x to twice the type's significand bits
template <typename T> __device__ inline void split16(float x, unsigned short& hi, unsigned short& lo) {
  hi = S16<T>::bits(x);
  lo = S16<T>::bits(x - S16<T>::val(hi));
}
__device__ inline uint2 pack4(const unsigned short (&v)[4]) {
  return make_uint2((uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16));
}
__device__ inline uint4 cat8(uint2 a, uint2 b) { return make_uint4(a.x, a.y, b.x, b.y); }
__device__ inline uint2 lds_tr(const unsigned short* p) {      // ds_read_b64_tr_b16
  const ps4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ps4 __attribute__((address_space(3)))*)(p));
  return __builtin_bit_cast(uint2, v);
}

template <typename T, int NBT, bool STATE_ONLY>
__global__ __launch_bounds__(512) void performer_bf16_kernel(PerfParams p) {
  constexpr int D = 64, C = 64, NW = 8, NTH = 512, E = 2 * D;
  constexpr int NBP = NBT * 16;
  constexpr int KF = (NBT + 1) / 2;            // 32-wide k-steps over the (padded) features
  constexpr int FP = KF * 32;                  // padded feature count
  constexpr int RB = C / 16, EB = E / 16;
  static_assert(EB == NW, "one 16-column block of [pos | v] per wave");
  // LDS row strides picked with a bank model of the two access patterns (64 banks x 4 B; b128 row reads are served in
  // 4 lane groups of 16, transposing b64 reads in 2 of 32): 160 / 224 B rows make the operand row reads conflict-free
  // (4 cycles per wave instruction; the first version's 144 B rows: 8), 192 B rows halve the conflicts of the
  // transposing reads of phi(K) (4 cycles; 144 B: 8; the conflict-free 2 needs a swizzle); rows stay 16-byte multiples
  constexpr int LDQ2 = FP + 16;                // phi(Q) rows (elements): row reads only
  constexpr int LDK2 = (FP == 64) ? 96 : FP + 8; // phi(K) rows (16-byte multiples): transposing reads + a few row reads
  constexpr int LDA = C + 16;                  // A rows: row reads only
  constexpr int DSL = RB + NTH / C;            // denominator partial slots per row
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  unsigned short* sW = reinterpret_cast<unsigned short*>(smem_b);   // [D/8][NBP][8]   projection, k-chunked
  unsigned short* sQ = sW + (D / 8) * NBP * 8;                      // [D/8][C][8]
  unsigned short* sK = sQ + (D / 8) * C * 8;                        // [D/8][C][8]
  unsigned short* sV = sK + (D / 8) * C * 8;                        // [C][E] 256-byte rows, chunk-swizzled
  unsigned short* sQh = sV + C * E;                                 // [2][ [C][LDQ2] x 2, [C][LDK2] x 2 ]: the phi images of two
  unsigned short* sQl = sQh + C * LDQ2;                             //   chunks in flight (set stride PHI below)
  unsigned short* sKh = sQl + C * LDQ2;                             // [C][LDK2]
  unsigned short* sKl = sKh + C * LDK2;
  // two sets where they fit (FP = 64: 152 KB in all); with FP = 96 one set, and (b) of the next chunk waits for a fourth barrier
  constexpr int NSET = (FP == 64) ? 2 : 1;
  unsigned short* sAh = sKl + C * LDK2 + (NSET - 1) * (2 * C * LDQ2 + 2 * C * LDK2);   // [C][LDA]   (behind the last set)
  unsigned short* sAl = sAh + C * LDA;
  float* sKsum = reinterpret_cast<float*>(sAl + C * LDA);           // [FP]
  float* sDenP = sKsum + FP;                                        // [C][DSL]
  float* sKsP = sDenP + C * DSL;                                    // [NW][FP]   per-wave k-sum increments

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int nh = blockIdx.x;
  const int n = nh / p.H, h = nh - n * p.H;
  const int seg = blockIdx.y;                              // sequence-parallel form, see PerfParams
  const int tb = p.t_base_dev ? *p.t_base_dev : p.t_base;       // rows the state has already seen (block-uniform)
  // chunk-aligned step: local row 0 is the chunk boundary at or below tb; the `lead` rows up to tb are the open chunk's old
  // rows (k, v, pos only).  Otherwise lead = 0 and local row 0 is row tb.
  const int lead = p.aligned ? tb % C : 0;
  const int row_base = tb - lead;                          // absolute index of local row 0
  const int TL = p.T + lead;                               // local rows of this call
  const int t_begin = seg * p.seg_len, t_end = min(TL, t_begin + p.seg_len);
  const T* qb = reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] - (int64_t)lead * p.qs[2];   // rows < lead are never read
  const int64_t cache_row = (p.aligned && p.t_base_dev) ? row_base : 0;      // k / v given as cache bases: start at the boundary
  const T* kb = reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1] + cache_row * p.ks[2];
  const T* vb = reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1] + cache_row * p.vs[2];
  const T* pb = reinterpret_cast<const T*>(p.pos) + (p.t_base_dev ? (int64_t)row_base * p.pos_stride : 0);
  T* ob = reinterpret_cast<T*>(p.out) + (int64_t)nh * p.T * (3 * D) - (int64_t)lead * (3 * D);           // rows < lead are never stored
  const float cnorm = powf((float)D, -0.25f);

  // projection (its values are bf16-exact: the reference casts the buffer to the data dtype), zero padded rows
  for (int i = tid; i < (D / 8) * NBP * 8; i += NTH) {
    const int j = i & 7, f = (i >> 3) % NBP, kc = (i >> 3) / NBP;
    sW[i] = f < p.nb ? S16<T>::bits(p.W[f * D + kc * 8 + j]) : (unsigned short)0;
  }
  // phi and A images: padded features / upper-triangular tiles are written once (zero) and never again
  for (int i = tid; i < NSET * (2 * C * LDQ2 + 2 * C * LDK2) + 2 * C * LDA; i += NTH) sQh[i] = 0;
  for (int i = tid; i < C * DSL; i += NTH) sDenP[i] = 0.f;   // slots of key blocks above the diagonal stay zero

  // carry image of one (n, h, segment): per-thread state registers, the k-sum, the per-thread column sum of v
  constexpr int CARRY = NBT * 4 * NTH + FP + NTH;
  f4 S[NBT];
#pragma unroll
  for (int b = 0; b < NBT; ++b) S[b] = f4{0.f, 0.f, 0.f, 0.f};
  // running column sums of v (cumulative-average output): in the transposed accumulator layout of (d) a lane owns the four
  // columns e0 + 4 lg + r of its wave; the 16 lanes of a lane group carry the same four.  Image slot of column r: lane
  // (lg, li = r) of the wave, i.e. (tid & ~15) + r.
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  const int cslot = (tid & ~15);
  {
    float ks0 = 0.f;
    if (!STATE_ONLY && p.state_in) {                       // increments of pass 1 do not include the incoming state
      const float* cr = p.state_in + (int64_t)nh * CARRY;
#pragma unroll
      for (int b = 0; b < NBT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[b][r] = cr[(b * 4 + r) * NTH + tid];
      if (tid < FP) ks0 = cr[NBT * 4 * NTH + tid];
#pragma unroll
      for (int r = 0; r < 4; ++r) csum[r] = cr[NBT * 4 * NTH + FP + cslot + r];
    }
    if (!STATE_ONLY) {
      for (int s2 = 0; s2 < seg; ++s2) {                   // fixed order: bitwise reproducible
        const float* cr = p.carry + ((int64_t)nh * (p.nseg - 1) + s2) * CARRY;
#pragma unroll
        for (int b = 0; b < NBT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) S[b][r] += cr[(b * 4 + r) * NTH + tid];
        if (tid < FP) ks0 += cr[NBT * 4 * NTH + tid];
#pragma unroll
        for (int r = 0; r < 4; ++r) csum[r] += cr[NBT * 4 * NTH + FP + cslot + r];
      }
    }
    if (tid < FP) sKsum[tid] = ks0;
  }

  // one 16-byte piece of each tensor per thread and chunk; prefetched one chunk ahead
  const int sr = tid >> 3, sc = tid & 7;        // staging row, 8-element column chunk
  // All global traffic goes through buffer instructions with hardware range checking: a row beyond T reads zeros /
  // drops its store WITHOUT a branch.  (Branches around loads and stores make the compiler's vmcnt bookkeeping
  // pessimistic: the wait for the prefetched chunk then also waits for every output store of the previous one.)
  constexpr unsigned OOB = 0x7FFFFF00u;
  auto mk = [&](const T* base, int64_t row_stride) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (int)(((int64_t)(TL - 1) * row_stride + D) * 2), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rq = mk(qb, p.qs[2]), rk = mk(kb, p.ks[2]), rv = mk(vb, p.vs[2]), rp = mk(pb, p.pos_stride);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((int64_t)TL * 3 * D * 2), 0x00020000);
  typedef __attribute__((ext_vector_type(4))) unsigned int bu4;
  const bool want_avg = p.avg != nullptr;                  // block-uniform
  T* gb = reinterpret_cast<T*>(p.avg) + (want_avg ? (int64_t)nh * p.T * D - (int64_t)lead * D : 0);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(gb, 0, want_avg ? (int)((int64_t)TL * D * 2) : 0, 0x00020000);
  bu4 pq, pk, pv, pp;
  auto issue_qk = [&](int t0n) {
    const int t = t0n + sr;
    const bool ok = t < t_end;
    pq = __builtin_amdgcn_raw_buffer_load_b128(rq, (ok && !STATE_ONLY && t >= lead) ? (int)((t * p.qs[2] + sc * 8) * 2) : (int)OOB, 0, 0);
    pk = __builtin_amdgcn_raw_buffer_load_b128(rk, ok ? (int)((t * p.ks[2] + sc * 8) * 2) : (int)OOB, 0, 0);
  };
  auto issue_v = [&](int t0n) {
    const int t = t0n + sr;
    const bool ok = t < t_end;
    pv = __builtin_amdgcn_raw_buffer_load_b128(rv, ok ? (int)((t * p.vs[2] + sc * 8) * 2) : (int)OOB, 0, 0);
    pp = __builtin_amdgcn_raw_buffer_load_b128(rp, ok ? (int)((t * p.pos_stride + sc * 8) * 2) : (int)OOB, 0, 0);
  };
#ifdef SEA_STAMP
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
  const unsigned long long _rstart64 = __builtin_amdgcn_s_memrealtime();
#endif
  issue_qk(t_begin);
  issue_v(t_begin);
  // swizzled chunk position inside a 256-byte row of the V image (conflict-free transposing reads)
  auto vchunk = [](int row, int ch) { return ch ^ (((row & 3) << 2) | ((row >> 2) & 3)); };

  // cumulative average of v (step K's input) on the waves that own the v columns: prefix sum over the chunk rows =
  // tril(ones) . V on the matrix cores (1.0 and v are exact in bf16, fp32 accumulation) + the running column sum
  auto tril_frag = [&](int ib, int ks) {
    unsigned short o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (ks * 32 + lg * 8 + j <= ib * 16 + li) ? S16<T>::ONE : (unsigned short)0;
    const unsigned short (&o0)[4] = *reinterpret_cast<const unsigned short (*)[4]>(&o[0]);
    const unsigned short (&o1)[4] = *reinterpret_cast<const unsigned short (*)[4]>(&o[4]);
    return cat8(pack4(o0), pack4(o1));
  };

  uint4 tril[RB][C / 32];                                  // loop-invariant A operands of the prefix-sum product
#pragma unroll
  for (int ib = 0; ib < RB; ++ib)
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) tril[ib][ks] = tril_frag(ib, ks);

  auto write_state = [&]() {                               // this thread's part of the (n, h) image
    float* cw = p.state_out + (int64_t)nh * CARRY;
#pragma unroll
    for (int b = 0; b < NBT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) cw[(b * 4 + r) * NTH + tid] = S[b][r];
    if (tid < FP) cw[NBT * 4 * NTH + tid] = sKsum[tid];
    cw[NBT * 4 * NTH + FP + tid] = li == 0 ? csum[0] : li == 1 ? csum[1] : li == 2 ? csum[2] : li == 3 ? csum[3] : 0.f;
  };

  // ---- the chunk walk, software-pipelined over THREE barriers per chunk (the first version ran a chunk's five phases one after
  // the other behind five barriers; at two waves per SIMD every phase is a dependent chain -- LDS read -> MFMA -> vector ->
  // LDS write -- that nothing else covers: ablation builds put (b) at 28 %, (c) 12 %, (d) 31 %, (e) 14 %, the flush 8 % of the
  // kernel, adding up).  Iteration j:
  //   P1  (d_j)(e_j): output products and the state update of chunk j   +   (b_{j+1}): feature maps of the NEXT chunk, into the
  //       other phi image;
  //   P2  (c_{j+1}): A tiles and partials of chunk j+1;  [pos | v] of chunk j+1 and q, k of chunk j+2 go from the prefetch
  //       registers to LDS (their old images were last read in P1), the loads of chunks j+2 / j+3 are issued;
  //   (until round 4 a third phase P3 -- denominators, 1 / row index and the k-sum of chunk j+1 on a few threads, behind its own
  //   barrier; now the first lines of (d), see there: TWO barriers per chunk).
  // Results leave straight from the accumulators (8-byte pieces: phase (d) holds four consecutive columns of a row per lane).
  auto rows_of = [&](int t0) { return min(C, t_end - t0); };
  // aligned step, open chunk (always the call's last): its rows produce output but do NOT enter the state -- S, the k-sum
  // and the column sums stay the state AT THE BOUNDARY, which is the image the next call continues from (it walks this
  // chunk's rows again).  Block-uniform.
  auto upd_of = [&](int t0) { return !(p.aligned && rows_of(t0) < C); };
  auto stage_qk = [&]() {
    // row slot XOR 2 * chunk: the 16 lanes of a b128 store (2 rows x 8 chunks; chunk images are 1 KB = 0 mod 64 banks apart)
    // land in 16 different 4-bank groups instead of 2 (8-way conflicts); the reads of (b) permute inside their 16-row runs
    *reinterpret_cast<bu4*>(sQ + (sc * C + (sr ^ (2 * sc))) * 8) = pq;
    *reinterpret_cast<bu4*>(sK + (sc * C + (sr ^ (2 * sc))) * 8) = pk;
  };
  auto stage_v = [&](int t0) {                               // chunk t0's [pos | v] rows; v also is the third block of the output
    const int rows = rows_of(t0);
    if (!STATE_ONLY) __builtin_amdgcn_raw_buffer_store_b128(pv, ro, (sr < rows && t0 + sr >= lead) ? ((t0 + sr) * (3 * D) + 2 * D + sc * 8) * 2 : (int)OOB, 0, 0);
    *reinterpret_cast<bu4*>(sV + sr * E + vchunk(sr, sc) * 8) = pp;
    *reinterpret_cast<bu4*>(sV + sr * E + vchunk(sr, D / 8 + sc) * 8) = pv;
  };
  // ---- (b) feature maps, transposed: X^T[f][t] = sum_d W[f][d] x[t][d]; wave = (Q | K, row block) ----------
  auto phase_b = [&](int rows, unsigned short* sQh, unsigned short* sQl, unsigned short* sKh, unsigned short* sKl) {
    if (!STATE_ONLY || wv >= RB) {                           // the state-only pass needs phi(K) alone (wave-uniform)
      const int which = wv / RB, rb = wv - which * RB;
      const unsigned short* src = which ? sK : sQ;
      f4 acc[NBT];
#pragma unroll
      for (int fb = 0; fb < NBT; ++fb) acc[fb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < D / 32; ++ks) {
        const uint4 bx = *reinterpret_cast<const uint4*>(src + ((4 * ks + lg) * C + ((rb * 16 + li) ^ (2 * (4 * ks + lg)))) * 8);
#pragma unroll
        for (int fb = 0; fb < NBT; ++fb) {
          const uint4 aw = *reinterpret_cast<const uint4*>(sW + ((4 * ks + lg) * NBP + fb * 16 + li) * 8);
          acc[fb] = S16<T>::mfma(aw, bx, acc[fb]);
        }
      }
      unsigned short* dh = which ? sKh : sQh;
      unsigned short* dl = which ? sKl : sQl;
      const int row = rb * 16 + li;                          // lane: 4 consecutive features of one row
#pragma unroll
      for (int fb = 0; fb < NBT; ++fb) {
        unsigned short hh[4], ll[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = fb * 16 + lg * 4 + r;
          float val = fmaxf(cnorm * acc[fb][r], 0.f) + 1e-3f;
          if (f >= p.nb || row >= rows) val = 0.f;           // padded features / rows beyond T contribute nothing
          split16<T>(val, hh[r], ll[r]);
        }
        // feature 32kk + 16a + 4g + j is stored at position 32kk + 8g + 4a + j: the 8 features lane group g needs of
        // a 32-wide k-step in (d) (4 of tile 2kk, 4 of tile 2kk+1) are then one 16-byte piece
        const int pos = (fb >> 1) * 32 + lg * 8 + (fb & 1) * 4;
        const int ldx = which ? LDK2 : LDQ2;
        *reinterpret_cast<uint2*>(dh + row * ldx + pos) = pack4(hh);
        *reinterpret_cast<uint2*>(dl + row * ldx + pos) = pack4(ll);
      }
    }
  };
  // ---- (c) A^T tiles (lower triangle), denominator and k-sum partials ------------------------------------
  auto phase_c = [&](const unsigned short* sQh, const unsigned short* sQl, const unsigned short* sKh, const unsigned short* sKl) {
    if constexpr (!STATE_ONLY)
    for (int tile = wv; tile < RB * (RB + 1) / 2; tile += NW) {
      int ib = 0, rem = tile;
      while (rem > ib) { rem -= ib + 1; ++ib; }              // tile -> (query block ib, key block jb <= ib)
      const int jb = rem;
      f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KF; ++ks) {
        const int ko = ks * 32 + lg * 8;
        const uint4 kh = *reinterpret_cast<const uint4*>(sKh + (jb * 16 + li) * LDK2 + ko);
        const uint4 kl = *reinterpret_cast<const uint4*>(sKl + (jb * 16 + li) * LDK2 + ko);
        const uint4 qh = *reinterpret_cast<const uint4*>(sQh + (ib * 16 + li) * LDQ2 + ko);
        const uint4 ql = *reinterpret_cast<const uint4*>(sQl + (ib * 16 + li) * LDQ2 + ko);
        acc = S16<T>::mfma(kh, qh, acc);
        acc = S16<T>::mfma(kh, ql, acc);
        acc = S16<T>::mfma(kl, qh, acc);
      }
      const int trow = ib * 16 + li;                         // lane: A[trow][s0 .. s0+3]
      const int s0 = jb * 16 + lg * 4;
      unsigned short hh[4], ll[4];
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float val = (s0 + r <= trow) ? acc[r] : 0.f;   // causal mask inside the diagonal tiles
        rs += val;
        split16<T>(val, hh[r], ll[r]);
      }
      *reinterpret_cast<uint2*>(sAh + trow * LDA + s0) = pack4(hh);
      *reinterpret_cast<uint2*>(sAl + trow * LDA + s0) = pack4(ll);
      rs = xor32_sum(xor16_sum(rs));
      if (lg == 0) sDenP[trow * DSL + jb] = rs;              // one writer per (row, key block)
    }
    {
      // carry part of the denominators: phi(q_t) . (ksum + eps) over the (permuted) feature positions
      const int row = tid & (C - 1), part = tid / C;         // 8 parts x (FP/8) positions
      constexpr int PER = FP / (NTH / C);
      float s = 0.f;
      if constexpr (!STATE_ONLY)
#pragma unroll
      for (int g4 = 0; g4 < PER / 4; ++g4) {
        const int f = part * PER + g4 * 4;
        const uint2 qh = *reinterpret_cast<const uint2*>(sQh + row * LDQ2 + f);
        const uint2 ql = *reinterpret_cast<const uint2*>(sQl + row * LDQ2 + f);
        const float4 ks = *reinterpret_cast<const float4*>(sKsum + f);
        s = fmaf(S16<T>::val((unsigned short)(qh.x & 0xffff)) + S16<T>::val((unsigned short)(ql.x & 0xffff)), ks.x + 1e-6f, s);
        s = fmaf(S16<T>::val((unsigned short)(qh.x >> 16)) + S16<T>::val((unsigned short)(ql.x >> 16)), ks.y + 1e-6f, s);
        s = fmaf(S16<T>::val((unsigned short)(qh.y & 0xffff)) + S16<T>::val((unsigned short)(ql.y & 0xffff)), ks.z + 1e-6f, s);
        s = fmaf(S16<T>::val((unsigned short)(qh.y >> 16)) + S16<T>::val((unsigned short)(ql.y >> 16)), ks.w + 1e-6f, s);
      }
      sDenP[row * DSL + RB + part] = s;
      // k-sum increment: wave w owns rows 8w .. 8w+7; lane = (4-position group fc, row lane rr)
      constexpr int FG = FP / 4, RRN = 64 / FG, RPL = (C / NW) / RRN;
      const int fc = lane % FG, rr = lane / FG;
      float k4[4] = {0.f, 0.f, 0.f, 0.f};
      if (rr < RRN) {
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
          const int r2 = wv * (C / NW) + rr * RPL + j;
          const uint2 kh = *reinterpret_cast<const uint2*>(sKh + r2 * LDK2 + fc * 4);
          const uint2 kl = *reinterpret_cast<const uint2*>(sKl + r2 * LDK2 + fc * 4);
          k4[0] += S16<T>::val((unsigned short)(kh.x & 0xffff)) + S16<T>::val((unsigned short)(kl.x & 0xffff));
          k4[1] += S16<T>::val((unsigned short)(kh.x >> 16)) + S16<T>::val((unsigned short)(kl.x >> 16));
          k4[2] += S16<T>::val((unsigned short)(kh.y & 0xffff)) + S16<T>::val((unsigned short)(kl.y & 0xffff));
          k4[3] += S16<T>::val((unsigned short)(kh.y >> 16)) + S16<T>::val((unsigned short)(kl.y >> 16));
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (FG == 16) {
          k4[j] = xor16_sum(xor32_sum(k4[j]));                   // (l, l+32) then (+16): same association as the shuffle form
        } else {
#pragma unroll
          for (int o = FG * (RRN / 2); o >= FG; o >>= 1) k4[j] += __shfl_down(k4[j], o);   // fixed order
        }
      }
      if (lane < FG) *reinterpret_cast<float4*>(sKsP + wv * FP + fc * 4) = make_float4(k4[0], k4[1], k4[2], k4[3]);
    }
  };
  // ---- (d) O = A V + phi(Q) S over this wave's 16 columns; (e) S += phi(K)^T V ------------------------------
  auto phase_de = [&](int t0, int rows, bool upd, const unsigned short* sQh, const unsigned short* sQl, const unsigned short* sKh,
                      const unsigned short* sKl) {
    // what a phase of its own (and its barrier) did until round 4: the k-sum takes this chunk's increments (the partials of (c)
    // read it before, the next chunk's (c) runs behind the next barrier), and every lane sums the denominator partials of ITS
    // rows -- the same additions in the same order in every lane that holds the row, so nothing changes in the results
    if (tid < FP) {
      float s = sKsum[tid];
#pragma unroll
      for (int i = 0; i < NW; ++i) s += sKsP[i * FP + tid];       // fixed order: bitwise reproducible
      if (upd) sKsum[tid] = s;
    }
    // (a lane sums ONE row -- lane group g the rows of block g % RB -- and the reciprocals travel by lane permutes: four rows and
    // eight divisions per lane made the phase VALU-bound, 2.5 % slower than the barrier it replaces at 256 workgroups)
    float dn[RB], ri1 = 0.f;
    if constexpr (!STATE_ONLY) {
      const int myrow = (lg % RB) * 16 + li;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < DSL; ++i) s += sDenP[myrow * DSL + i];
      const float d1 = 1.0f / s;
      ri1 = 1.0f / (float)(row_base + t0 + myrow + 1);
#pragma unroll
      for (int ib = 0; ib < RB; ++ib) dn[ib] = __shfl(d1, ib * 16 + li);
    }
    {
      const int jb = wv, e0 = jb * 16;
      // V fragments (B operand, k = chunk row): rows 32ks + 8lg + {0..3 | 4..7}, columns e0 .. e0+15
      uint4 vf[C / 32];
      {
        const int q = li >> 2, pp_ = li & 3;
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks) {
          const int r0 = ks * 32 + lg * 8;
          const uint2 a = lds_tr(sV + (r0 + q) * E + vchunk(r0 + q, 2 * jb + (pp_ >> 1)) * 8 + 4 * (pp_ & 1));
          const uint2 b = lds_tr(sV + (r0 + 4 + q) * E + vchunk(r0 + 4 + q, 2 * jb + (pp_ >> 1)) * 8 + 4 * (pp_ & 1));
          vf[ks] = cat8(a, b);
        }
      }
      if constexpr (STATE_ONLY) {
        if (want_avg && jb >= EB / 2) {                    // column total of the chunk = the last row's prefix
          f4 cum = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks <= (RB - 1) / 2; ++ks) cum = S16<T>::mfma(vf[ks], tril[RB - 1][ks], cum);
#pragma unroll
          for (int r = 0; r < 4; ++r) csum[r] += __shfl(cum[r], (lane & 48) | 15);
        }
      } else {
      f4 o[RB];
#pragma unroll
      for (int ib = 0; ib < RB; ++ib) o[ib] = f4{0.f, 0.f, 0.f, 0.f};
      // A V: key k-step ks covers key blocks 2ks, 2ks+1; query block ib needs k-steps <= ib/2
#pragma unroll
      for (int ib = 0; ib < RB; ++ib) {
#pragma unroll
        for (int ks = 0; ks <= ib / 2; ++ks) {
          const uint4 ah = *reinterpret_cast<const uint4*>(sAh + (ib * 16 + li) * LDA + ks * 32 + lg * 8);
          const uint4 al = *reinterpret_cast<const uint4*>(sAl + (ib * 16 + li) * LDA + ks * 32 + lg * 8);
          o[ib] = S16<T>::mfma(vf[ks], ah, o[ib]);
          o[ib] = S16<T>::mfma(vf[ks], al, o[ib]);
        }
      }
      // phi(Q) S: the state tiles, split, are the B operand; k-step kk pairs feature blocks 2kk and 2kk+1
#pragma unroll
      for (int kk = 0; kk < KF; ++kk) {
        unsigned short h0[4], h1[4], l0[4], l1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          split16<T>(S[2 * kk][r], h0[r], l0[r]);
          if (2 * kk + 1 < NBT) split16<T>(S[(2 * kk + 1 < NBT) ? 2 * kk + 1 : 0][r], h1[r], l1[r]);
          else h1[r] = l1[r] = 0;
        }
        const uint4 bh = cat8(pack4(h0), pack4(h1)), bl = cat8(pack4(l0), pack4(l1));
#pragma unroll
        for (int ib = 0; ib < RB; ++ib) {
          const uint4 ah = *reinterpret_cast<const uint4*>(sQh + (ib * 16 + li) * LDQ2 + kk * 32 + lg * 8);
          const uint4 al = *reinterpret_cast<const uint4*>(sQl + (ib * 16 + li) * LDQ2 + kk * 32 + lg * 8);
          o[ib] = S16<T>::mfma(bh, ah, o[ib]);
          o[ib] = S16<T>::mfma(bl, ah, o[ib]);
          o[ib] = S16<T>::mfma(bh, al, o[ib]);
        }
      }
      // Every product above has its operands SWAPPED (O^T = V^T A^T + S^T phi(Q)^T; the A and B fragment layouts are mirror
      // images, so the same registers serve): the accumulator of query block ib then holds, per lane, row ib*16 + li and the
      // FOUR CONSECUTIVE COLUMNS e0 + 4 lg + r -- one 8-byte LDS store and one denominator read per block instead of four
      // 2-byte stores (two lanes per bank word) and four reads.
      const int col = e0 + 4 * lg;
#pragma unroll
      for (int ib = 0; ib < RB; ++ib) {
        const int row = ib * 16 + li;
        unsigned short ob[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[r] = S16<T>::bits(o[ib][r] * dn[ib]);
        const uint2 w2 = pack4(ob);
        typedef __attribute__((ext_vector_type(2))) unsigned int bu2;
        __builtin_amdgcn_raw_buffer_store_b64(bu2{w2.x, w2.y}, ro, (row < rows && t0 + row >= lead) ? ((t0 + row) * (3 * D) + col) * 2 : (int)OOB, 0, 0);
      }
      if (want_avg && jb >= EB / 2) {                      // wave-uniform: this wave's 16 columns are v features
        f4 cum[RB];
#pragma unroll
        for (int ib = 0; ib < RB; ++ib) {
          cum[ib] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks <= ib / 2; ++ks) cum[ib] = S16<T>::mfma(vf[ks], tril[ib][ks], cum[ib]);
        }
        const int gcol = col - D;
#pragma unroll
        for (int ib = 0; ib < RB; ++ib) {
          const int row = ib * 16 + li;
          const float ri = __shfl(ri1, ib * 16 + li);
          unsigned short gb4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) gb4[r] = S16<T>::bits((cum[ib][r] + csum[r]) * ri);
          const uint2 w2 = pack4(gb4);
          typedef __attribute__((ext_vector_type(2))) unsigned int bu2;
          __builtin_amdgcn_raw_buffer_store_b64(bu2{w2.x, w2.y}, rg, (row < rows && t0 + row >= lead) ? ((t0 + row) * D + gcol) * 2 : (int)OOB, 0, 0);
        }
        if (upd) {                                         // column totals of the chunk = its last row's prefix (lane li = 15)
#pragma unroll
          for (int r = 0; r < 4; ++r) csum[r] += __shfl(cum[RB - 1][r], (lane & 48) | 15);
        }
      }
      }
      // (e) S[f][e] += sum_s phi(k_s)[f] V[s][e]: A operand = phi(K)^T by transposing reads of the row-major images
      if (upd) {
        const int q = li >> 2, pp_ = li & 3;
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) {
#pragma unroll
          for (int ks = 0; ks < C / 32; ++ks) {
            const int r0 = ks * 32 + lg * 8;
            const int co = (rb >> 1) * 32 + 8 * pp_ + (rb & 1) * 4;      // positions of features 16rb + 4p .. +3
            const uint4 kh = cat8(lds_tr(sKh + (r0 + q) * LDK2 + co), lds_tr(sKh + (r0 + 4 + q) * LDK2 + co));
            const uint4 kl = cat8(lds_tr(sKl + (r0 + q) * LDK2 + co), lds_tr(sKl + (r0 + 4 + q) * LDK2 + co));
            S[rb] = S16<T>::mfma(kh, vf[ks], S[rb]);
            S[rb] = S16<T>::mfma(kl, vf[ks], S[rb]);
          }
        }
      }
    }
  };
  constexpr int PHI = 2 * C * LDQ2 + 2 * C * LDK2;          // elements of one phi image set (Qh, Ql, Kh, Kl)
  auto run_b = [&](int t0, int buf) { phase_b(rows_of(t0), sQh + buf * PHI, sQl + buf * PHI, sKh + buf * PHI, sKl + buf * PHI); };
  auto run_c = [&](int buf) { phase_c(sQh + buf * PHI, sQl + buf * PHI, sKh + buf * PHI, sKl + buf * PHI); };
  auto run_de = [&](int t0, int buf) {
    phase_de(t0, rows_of(t0), upd_of(t0), sQh + buf * PHI, sQl + buf * PHI, sKh + buf * PHI, sKl + buf * PHI);
  };
  if (t_begin < t_end) {
    // prologue: chunk 0 through (b), (c); chunk 1's q, k staged; loads of chunk 1 ([pos | v]) and chunk 2 (q, k) in flight
    stage_qk();
    stage_v(t_begin);
    issue_qk(t_begin + C);
    issue_v(t_begin + C);
    __syncthreads();
    run_b(t_begin, 0);
    __syncthreads();
    run_c(0);
    stage_qk();
    issue_qk(t_begin + 2 * C);
    __syncthreads();
    int buf = 0;
    for (int t0 = t_begin; t0 < t_end; t0 += C, buf ^= (NSET - 1)) {
      const bool has_next = t0 + C < t_end;                  // block-uniform
      // (measured and not taken: the two waves of a SIMD running P1's two parts in opposite order, 328 vs 306 us; (b) cut in two
      // and placed between the parts of (d), 320 vs 306 us -- DESIGN.md section 9)
      run_de(t0, buf);
      if constexpr (NSET == 1) __syncthreads();              // one image set: chunk j's must have been read
      if (has_next) run_b(t0 + C, buf ^ (NSET - 1));
      __syncthreads();
      PSTAMP(0);
      if (has_next) {
        run_c(buf ^ (NSET - 1));
        stage_v(t0 + C);
        issue_v(t0 + 2 * C);
        stage_qk();
        issue_qk(t0 + 3 * C);
      }
      __syncthreads();
      PSTAMP(1);
    }
  }
  if constexpr (STATE_ONLY) {
    float* cw = p.carry + ((int64_t)nh * (p.nseg - 1) + seg) * CARRY;
#pragma unroll
    for (int b = 0; b < NBT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) cw[(b * 4 + r) * NTH + tid] = S[b][r];
    if (tid < FP) cw[NBT * 4 * NTH + tid] = sKsum[tid];
    cw[NBT * 4 * NTH + FP + tid] = li == 0 ? csum[0] : li == 1 ? csum[1] : li == 2 ? csum[2] : li == 3 ? csum[3] : 0.f;
  } else {
    // the block that walked the last rows holds the final state (aligned step: the state at the last chunk boundary).  A step
    // that closed no chunk leaves the image as it found it: updated in place (a DecodeSession) there is nothing to write --
    // 25 KB of stores per (n, h) whose drain was the end of the launch on 63 of 64 positions
    const bool unchanged = p.aligned && p.state_in == p.state_out && t_end - t_begin <= C && !upd_of(t_begin);
    if (p.state_out && seg == p.nseg - 1 && !unchanged) write_state();
#ifdef SEA_STAMP
    if (threadIdx.x == 0) { const int _w = blockIdx.y * gridDim.x + blockIdx.x; if (_w < 1024) { sea_dbg_wg[2 * _w] = _rstart64; sea_dbg_wg[2 * _w + 1] = __builtin_amdgcn_s_memrealtime(); } }
#endif
  }
}

// ---- wide heads (D = 80, 128): the same algorithm with the chunk cut to C = 32 rows so that the LDS images fit (the
// D = 64 plan scaled up is 208 KB), up to two 16-column blocks of [pos | v] per wave (E / 16 = 16 or 10 blocks over 8
// waves: a wave keeps two state tiles S; at D = 80 only waves 0 and 1 own a second block), D / 8 staging chunks per row
// on the first C * D / 8 threads, the head dimension zero-padded to whole 32-wide k-steps (D = 80: 96) in the operand
// images of the feature-map product, and a V image of 512-byte rows whose chunk swizzle spreads the eight rows of a
// transposing read over the eight 8-bank groups.  Everything else -- split operands, transposed products, transposing
// reads, the cumulative average on the matrix cores, segments, state images, and (round 4) the chunk walk pipelined over
// three barriers with two sets of phi images and results stored straight from the accumulators -- is the D = 64 kernel's.
// The 25 KB of result tiles the first version flushed one chunk later paid for the second image set (d = 128: 120 KB).
template <typename T, int D, int C, int NBT, bool STATE_ONLY>
__global__ __launch_bounds__(512) void performer_bf16w_kernel(PerfParams p) {
  constexpr int NW = 8, NTH = 512, E = 2 * D, CPR = D / 8;
  constexpr int DP = (D + 31) / 32 * 32, KC = DP / 8;    // head dimension padded to whole k-steps, its 8-element chunks
  constexpr int EL = 256;                                // elements per row of the V image (512 bytes: room for the swizzle)
  constexpr int STG = C * CPR;                           // threads that stage one 16-byte piece of each tensor
  static_assert(STG <= NTH && D % 16 == 0 && E <= EL && C % 16 == 0 && C <= 32, "staging pieces fit the block, whole column blocks");
  constexpr int NBP = NBT * 16;
  constexpr int KF = (NBT + 1) / 2;            // 32-wide k-steps over the (padded) features
  constexpr int FP = KF * 32;                  // padded feature count
  constexpr int RB = C / 16, EB = E / 16, JB = (EB + NW - 1) / NW, NPART = 8;
  // (b) runs on all eight waves: a (Q | K, row block) pair is shared by BW waves, each taking FBP of the feature blocks
  constexpr int BW = NW / (2 * RB), FBP = (NBT + BW - 1) / BW;
  static_assert(2 * RB * BW == NW, "feature-map product: whole waves per (tensor, row block)");
  // LDS row strides: see the D = 64 kernel (bank model of the b128 row reads and the transposing b64 reads)
  constexpr int LDQ2 = FP + 16;                // phi(Q) rows (elements): row reads only
  constexpr int LDK2 = (FP == 64) ? 96 : FP + 8; // phi(K) rows (16-byte multiples): transposing reads + a few row reads
  constexpr int LDA = C + 16;                  // A rows: row reads only
  constexpr int DSL = RB + NPART;              // denominator partial slots per row
  constexpr int PHI = 2 * C * LDQ2 + 2 * C * LDK2;          // elements of one phi image set (Qh, Ql, Kh, Kl)
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  unsigned short* sW = reinterpret_cast<unsigned short*>(smem_b);   // [KC][NBP][8]  projection, k-chunked (zero padded)
  unsigned short* sQ = sW + KC * NBP * 8;                           // [KC][C][8]    (chunks >= D / 8: zero)
  unsigned short* sK = sQ + KC * C * 8;                             // [KC][C][8]
  unsigned short* sV = sK + KC * C * 8;                             // [C][EL] 512-byte rows, chunk-swizzled
  unsigned short* sPhi = sV + C * EL;                               // [2][ [C][LDQ2] x 2, [C][LDK2] x 2 ]: two chunks in flight
  unsigned short* sAh = sPhi + 2 * PHI;                             // [C][LDA]
  unsigned short* sAl = sAh + C * LDA;
  float* sKsum = reinterpret_cast<float*>(sAl + C * LDA);           // [FP]
  float* sDenP = sKsum + FP;                                        // [C][DSL]
  float* sKsP = sDenP + C * DSL;                                    // [NW][FP]   per-wave k-sum increments
  float* sKsP2 = sKsP + NW * FP;                                    // [NW][FP]   (state pass: the second chunk of an iteration)
  unsigned short* sV2 = reinterpret_cast<unsigned short*>(sKsP2 + NW * FP);   // [C][EL] (state pass: the second chunk's [pos | v])

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int nh = blockIdx.x;
  const int n = nh / p.H, h = nh - n * p.H;
  const int seg = blockIdx.y;                              // sequence-parallel form, see PerfParams
  const int tb = p.t_base_dev ? *p.t_base_dev : p.t_base;       // rows the state has already seen (block-uniform)
  // chunk-aligned step: local row 0 is the chunk boundary at or below tb; the `lead` rows up to tb are the open chunk's old
  // rows (k, v, pos only).  Otherwise lead = 0 and local row 0 is row tb.
  const int lead = p.aligned ? tb % C : 0;
  const int row_base = tb - lead;                          // absolute index of local row 0
  const int TL = p.T + lead;                               // local rows of this call
  const int t_begin = seg * p.seg_len, t_end = min(TL, t_begin + p.seg_len);
  const T* qb = reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1] - (int64_t)lead * p.qs[2];   // rows < lead are never read
  const int64_t cache_row = (p.aligned && p.t_base_dev) ? row_base : 0;      // k / v given as cache bases: start at the boundary
  const T* kb = reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1] + cache_row * p.ks[2];
  const T* vb = reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1] + cache_row * p.vs[2];
  const T* pb = reinterpret_cast<const T*>(p.pos) + (p.t_base_dev ? (int64_t)row_base * p.pos_stride : 0);
  T* ob = reinterpret_cast<T*>(p.out) + (int64_t)nh * p.T * (3 * D) - (int64_t)lead * (3 * D);           // rows < lead are never stored
  const float cnorm = powf((float)D, -0.25f);

  // projection (its values are bf16-exact: the reference casts the buffer to the data dtype), zero padded rows
  for (int i = tid; i < KC * NBP * 8; i += NTH) {
    const int j = i & 7, f = (i >> 3) % NBP, kc = (i >> 3) / NBP;
    sW[i] = (f < p.nb && kc < CPR) ? S16<T>::bits(p.W[f * D + kc * 8 + j]) : (unsigned short)0;
  }
  if constexpr (KC > CPR)                                  // padded k-chunks of the q / k images: zero, never written again
    for (int i = tid; i < (KC - CPR) * C * 8; i += NTH) sQ[CPR * C * 8 + i] = sK[CPR * C * 8 + i] = 0;
  // phi and A images: padded features / upper-triangular tiles are written once (zero) and never again
  for (int i = tid; i < 2 * PHI + 2 * C * LDA; i += NTH) sPhi[i] = 0;
  for (int i = tid; i < C * DSL; i += NTH) sDenP[i] = 0.f;   // slots of key blocks above the diagonal stay zero

  // carry image of one (n, h, segment): per-thread state registers, the k-sum, the column sums of v
  constexpr int CARRY = JB * NBT * 4 * NTH + FP + JB * NTH;
  f4 S[JB][NBT];
  // running column sums of v (cumulative-average output): in the transposed accumulator layout of (d) a lane owns the four
  // columns e0 + 4 lg + r of each of its wave's blocks; image slot of column r of block jq: jq * NTH + (tid & ~15) + r
  float csum[JB][4];
  const int cslot = (tid & ~15);
#pragma unroll
  for (int jq = 0; jq < JB; ++jq) {
#pragma unroll
    for (int r = 0; r < 4; ++r) csum[jq][r] = 0.f;
#pragma unroll
    for (int b = 0; b < NBT; ++b) S[jq][b] = f4{0.f, 0.f, 0.f, 0.f};
  }
  {
    float ks0 = 0.f;
    if (!STATE_ONLY && p.state_in) {                       // increments of pass 1 do not include the incoming state
      const float* cr = p.state_in + (int64_t)nh * CARRY;
#pragma unroll
      for (int jq = 0; jq < JB; ++jq) {
#pragma unroll
        for (int b = 0; b < NBT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) S[jq][b][r] = cr[((jq * NBT + b) * 4 + r) * NTH + tid];
#pragma unroll
        for (int r = 0; r < 4; ++r) csum[jq][r] = cr[JB * NBT * 4 * NTH + FP + jq * NTH + cslot + r];
      }
      if (tid < FP) ks0 = cr[JB * NBT * 4 * NTH + tid];
    }
    if (!STATE_ONLY) {
      for (int s2 = 0; s2 < seg; ++s2) {                   // fixed order: bitwise reproducible
        const float* cr = p.carry + ((int64_t)nh * (p.nseg - 1) + s2) * CARRY;
#pragma unroll
        for (int jq = 0; jq < JB; ++jq) {
#pragma unroll
          for (int b = 0; b < NBT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[jq][b][r] += cr[((jq * NBT + b) * 4 + r) * NTH + tid];
#pragma unroll
          for (int r = 0; r < 4; ++r) csum[jq][r] += cr[JB * NBT * 4 * NTH + FP + jq * NTH + cslot + r];
        }
        if (tid < FP) ks0 += cr[JB * NBT * 4 * NTH + tid];
      }
    }
    if (tid < FP) sKsum[tid] = ks0;
  }

  // one 16-byte piece of each tensor per thread and chunk; q, k prefetched two chunks ahead, [pos | v] one
  // (D = 80: the LAST C * D / 8 = 320 threads stage -- waves 0 and 1 carry a second column block in (d)(e))
  const bool stager = tid >= NTH - STG;
  const int stid = stager ? tid - (NTH - STG) : 0;
  const int sr = stid / CPR, sc = stid - sr * CPR;   // staging row, 8-element column chunk
  // All global traffic goes through buffer instructions with hardware range checking: a row beyond T reads zeros /
  // drops its store WITHOUT a branch.  (Branches around loads and stores make the compiler's vmcnt bookkeeping
  // pessimistic: the wait for the prefetched chunk then also waits for every output store of the previous one.)
  constexpr unsigned OOB = 0x7FFFFF00u;
  auto mk = [&](const T* base, int64_t row_stride) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (int)(((int64_t)(TL - 1) * row_stride + D) * 2), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rq = mk(qb, p.qs[2]), rk = mk(kb, p.ks[2]), rv = mk(vb, p.vs[2]), rp = mk(pb, p.pos_stride);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((int64_t)TL * 3 * D * 2), 0x00020000);
  typedef __attribute__((ext_vector_type(4))) unsigned int bu4;
  typedef __attribute__((ext_vector_type(2))) unsigned int bu2;
  const bool want_avg = p.avg != nullptr;                  // block-uniform
  T* gb = reinterpret_cast<T*>(p.avg) + (want_avg ? (int64_t)nh * p.T * D - (int64_t)lead * D : 0);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(gb, 0, want_avg ? (int)((int64_t)TL * D * 2) : 0, 0x00020000);
  // two sets of prefetch registers: every load has two chunk times between issue and use (d = 128: 202.7 -> 199.9 us; halves the lag
  // of the last segment's workgroups in the pipelined walk)
  struct Pre { bu4 q, k, v, p; };
  Pre pa, pn;
  auto issue_qk = [&](int t0n, Pre& r) {
    const int t = t0n + sr;
    const bool ok = t < t_end && stager;
    r.q = __builtin_amdgcn_raw_buffer_load_b128(rq, (ok && !STATE_ONLY && t >= lead) ? (int)((t * p.qs[2] + sc * 8) * 2) : (int)OOB, 0, 0);
    r.k = __builtin_amdgcn_raw_buffer_load_b128(rk, ok ? (int)((t * p.ks[2] + sc * 8) * 2) : (int)OOB, 0, 0);
  };
  auto issue_v = [&](int t0n, Pre& r) {
    const int t = t0n + sr;
    const bool ok = t < t_end && stager;
    r.v = __builtin_amdgcn_raw_buffer_load_b128(rv, ok ? (int)((t * p.vs[2] + sc * 8) * 2) : (int)OOB, 0, 0);
    r.p = __builtin_amdgcn_raw_buffer_load_b128(rp, ok ? (int)((t * p.pos_stride + sc * 8) * 2) : (int)OOB, 0, 0);
  };
  issue_qk(t_begin, pa);
  issue_v(t_begin, pa);
  // swizzled chunk position inside a row of the V image (conflict-free transposing reads):
  // 512-byte rows (32 chunks): rows {r..r+3, r+8..r+11} of one transposing read get the eight values of
  // (row & 3) | (bit 3 of row) << 2 XOR-ed into bits 1..3 of the chunk index -> eight disjoint 8-bank groups
  auto vchunk = [](int row, int ch) { return ch ^ ((((row & 3) | (((row >> 3) & 1) << 2))) << 1); };

  // cumulative average of v (step K's input) on the waves that own the v columns: prefix sum over the chunk rows =
  // tril(ones) . V on the matrix cores (1.0 and v are exact in bf16, fp32 accumulation) + the running column sum
  auto tril_frag = [&](int ib, int ks) {
    unsigned short o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (ks * 32 + lg * 8 + j <= ib * 16 + li) ? S16<T>::ONE : (unsigned short)0;
    const unsigned short (&o0)[4] = *reinterpret_cast<const unsigned short (*)[4]>(&o[0]);
    const unsigned short (&o1)[4] = *reinterpret_cast<const unsigned short (*)[4]>(&o[4]);
    return cat8(pack4(o0), pack4(o1));
  };

  uint4 tril[RB][C / 32];                                  // loop-invariant operands of the prefix-sum product
#pragma unroll
  for (int ib = 0; ib < RB; ++ib)
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) tril[ib][ks] = tril_frag(ib, ks);

  auto write_image = [&](float* cw) {                      // this thread's part of an (n, h[, segment]) image
#pragma unroll
    for (int jq = 0; jq < JB; ++jq) {
#pragma unroll
      for (int b = 0; b < NBT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cw[((jq * NBT + b) * 4 + r) * NTH + tid] = S[jq][b][r];
      float* cs = cw + JB * NBT * 4 * NTH + FP + jq * NTH + tid;    // lane li = 0 writes the group's four sums, lanes >= 4 zeros
      if (li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[r] = csum[jq][r];
      } else if (li >= 4) cs[0] = 0.f;
    }
    if (tid < FP) cw[JB * NBT * 4 * NTH + tid] = sKsum[tid];
  };

  // ---- the chunk walk: the D = 64 kernel's three-barrier pipeline (P1 = (d)(e) of chunk j + (b) of chunk j+1 into the other
  // image set; P2 = (c) of chunk j+1, staging of [pos | v] of chunk j+1 and q, k of chunk j+2, loads issued; P3 = denominators
  // and k-sum of chunk j+1) ----------------------------------------------------------------------------------------------------
  auto rows_of = [&](int t0) { return min(C, t_end - t0); };
  auto upd_of = [&](int t0) { return !(p.aligned && rows_of(t0) < C); };   // an open last chunk leaves the state at the boundary
  auto stage_qk = [&](const Pre& r) {
    // row slot XOR chunk: the 16 lanes of a b128 store (one row x 16 chunks at D = 128; chunk images are 512 B = 0 mod 64
    // banks apart) land in 16 different 4-bank groups instead of one; the reads of (b) permute inside their 16-row runs
    if (stager) {
      *reinterpret_cast<bu4*>(sQ + (sc * C + (sr ^ sc)) * 8) = r.q;
      *reinterpret_cast<bu4*>(sK + (sc * C + (sr ^ sc)) * 8) = r.k;
    }
  };
  auto stage_v = [&](int t0, const Pre& r) {                 // chunk t0's [pos | v] rows; v also is the third block of the output
    const int rows = rows_of(t0);
    if (!STATE_ONLY) __builtin_amdgcn_raw_buffer_store_b128(r.v, ro, (stager && sr < rows && t0 + sr >= lead) ? ((t0 + sr) * (3 * D) + 2 * D + sc * 8) * 2 : (int)OOB, 0, 0);
    if (stager) {
      *reinterpret_cast<bu4*>(sV + sr * EL + vchunk(sr, sc) * 8) = r.p;
      *reinterpret_cast<bu4*>(sV + sr * EL + vchunk(sr, CPR + sc) * 8) = r.v;
    }
  };
  // ---- (b) feature maps, transposed: X^T[f][t] = sum_d W[f][d] x[t][d]; wave = (Q | K, row block, part of the feature blocks) ----
  auto phase_b = [&](int rows, unsigned short* set) {
    const int which = wv / (NW / 2);
    if (!STATE_ONLY || which == 1) {                         // the state-only pass needs phi(K) alone (wave-uniform)
      // D = 80: waves 0 and 1 (SIMDs 0, 1) own a second column block in (d)(e), so the larger share of the feature blocks goes
      // to the waves of SIMDs 2 and 3
      const int rb = wv & (RB - 1), part = (BW - 1) - ((wv % (NW / 2)) / RB);
      const unsigned short* src = which ? sK : sQ;
      f4 acc[FBP];
#pragma unroll
      for (int fb = 0; fb < FBP; ++fb) acc[fb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DP / 32; ++ks) {
        const uint4 bx = *reinterpret_cast<const uint4*>(src + ((4 * ks + lg) * C + ((rb * 16 + li) ^ ((4 * ks + lg) & 15))) * 8);
#pragma unroll
        for (int fb = 0; fb < FBP; ++fb) {
          const int fbg = part * FBP + fb;                   // (wave-uniform)
          if (fbg < NBT) {
            const uint4 aw = *reinterpret_cast<const uint4*>(sW + ((4 * ks + lg) * NBP + fbg * 16 + li) * 8);
            acc[fb] = S16<T>::mfma(aw, bx, acc[fb]);
          }
        }
      }
      const int ldx = which ? LDK2 : LDQ2;
      unsigned short* dh = set + (which ? 2 * C * LDQ2 : 0);
      unsigned short* dl = dh + C * ldx;
      const int row = rb * 16 + li;                          // lane: 4 consecutive features of one row
#pragma unroll
      for (int fb = 0; fb < FBP; ++fb) {
        const int fbg = part * FBP + fb;
        if (fbg < NBT) {
          unsigned short hh[4], ll[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = fbg * 16 + lg * 4 + r;
            float val = fmaxf(cnorm * acc[fb][r], 0.f) + 1e-3f;
            if (f >= p.nb || row >= rows) val = 0.f;           // padded features / rows beyond T contribute nothing
            split16<T>(val, hh[r], ll[r]);
          }
          // feature 32kk + 16a + 4g + j is stored at position 32kk + 8g + 4a + j: the 8 features lane group g needs of
          // a 32-wide k-step in (d) (4 of tile 2kk, 4 of tile 2kk+1) are then one 16-byte piece
          const int pos = (fbg >> 1) * 32 + lg * 8 + (fbg & 1) * 4;
          *reinterpret_cast<uint2*>(dh + row * ldx + pos) = pack4(hh);
          *reinterpret_cast<uint2*>(dl + row * ldx + pos) = pack4(ll);
        }
      }
    }
  };
  // ---- (c) A^T tiles (lower triangle), denominator and k-sum partials ------------------------------------
  auto phase_c = [&](const unsigned short* set) {
    const unsigned short* sQh = set;
    const unsigned short* sQl = sQh + C * LDQ2;
    const unsigned short* sKh = sQl + C * LDQ2;
    const unsigned short* sKl = sKh + C * LDK2;
    if constexpr (!STATE_ONLY)
    for (int tile = wv; tile < RB * (RB + 1) / 2; tile += NW) {   // (the first waves: they stage nothing at D = 80)
      int ib = 0, rem = tile;
      while (rem > ib) { rem -= ib + 1; ++ib; }              // tile -> (query block ib, key block jb <= ib)
      const int jb = rem;
      f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KF; ++ks) {
        const int ko = ks * 32 + lg * 8;
        const uint4 kh = *reinterpret_cast<const uint4*>(sKh + (jb * 16 + li) * LDK2 + ko);
        const uint4 kl = *reinterpret_cast<const uint4*>(sKl + (jb * 16 + li) * LDK2 + ko);
        const uint4 qh = *reinterpret_cast<const uint4*>(sQh + (ib * 16 + li) * LDQ2 + ko);
        const uint4 ql = *reinterpret_cast<const uint4*>(sQl + (ib * 16 + li) * LDQ2 + ko);
        acc = S16<T>::mfma(kh, qh, acc);
        acc = S16<T>::mfma(kh, ql, acc);
        acc = S16<T>::mfma(kl, qh, acc);
      }
      const int trow = ib * 16 + li;                         // lane: A[trow][s0 .. s0+3]
      const int s0 = jb * 16 + lg * 4;
      unsigned short hh[4], ll[4];
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float val = (s0 + r <= trow) ? acc[r] : 0.f;   // causal mask inside the diagonal tiles
        rs += val;
        split16<T>(val, hh[r], ll[r]);
      }
      *reinterpret_cast<uint2*>(sAh + trow * LDA + s0) = pack4(hh);
      *reinterpret_cast<uint2*>(sAl + trow * LDA + s0) = pack4(ll);
      rs = xor32_sum(xor16_sum(rs));
      if (lg == 0) sDenP[trow * DSL + jb] = rs;              // one writer per (row, key block)
    }
    {
      // carry part of the denominators: phi(q_t) . (ksum + eps) over the (permuted) feature positions
      const int row = tid & (C - 1), part = tid / C;         // NPART parts x (FP / NPART) positions
      constexpr int PER = FP / NPART;
      static_assert(PER % 4 == 0, "four positions per step");
      float s = 0.f;
      if constexpr (!STATE_ONLY)
      if (part < NPART)
#pragma unroll
      for (int g4 = 0; g4 < PER / 4; ++g4) {
        const int f = part * PER + g4 * 4;
        const uint2 qh = *reinterpret_cast<const uint2*>(sQh + row * LDQ2 + f);
        const uint2 ql = *reinterpret_cast<const uint2*>(sQl + row * LDQ2 + f);
        const float4 ks = *reinterpret_cast<const float4*>(sKsum + f);
        s = fmaf(S16<T>::val((unsigned short)(qh.x & 0xffff)) + S16<T>::val((unsigned short)(ql.x & 0xffff)), ks.x + 1e-6f, s);
        s = fmaf(S16<T>::val((unsigned short)(qh.x >> 16)) + S16<T>::val((unsigned short)(ql.x >> 16)), ks.y + 1e-6f, s);
        s = fmaf(S16<T>::val((unsigned short)(qh.y & 0xffff)) + S16<T>::val((unsigned short)(ql.y & 0xffff)), ks.z + 1e-6f, s);
        s = fmaf(S16<T>::val((unsigned short)(qh.y >> 16)) + S16<T>::val((unsigned short)(ql.y >> 16)), ks.w + 1e-6f, s);
      }
      if (part < NPART) sDenP[row * DSL + RB + part] = s;
      // k-sum increment: wave w owns rows 4w .. 4w+3; lane = (4-position group fc, row lane rr)
      constexpr int FG = FP / 4, RRN = 64 / FG, RPL = (C / NW) / RRN;
      const int fc = lane % FG, rr = lane / FG;
      float k4[4] = {0.f, 0.f, 0.f, 0.f};
      if (rr < RRN) {
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
          const int r2 = wv * (C / NW) + rr * RPL + j;
          const uint2 kh = *reinterpret_cast<const uint2*>(sKh + r2 * LDK2 + fc * 4);
          const uint2 kl = *reinterpret_cast<const uint2*>(sKl + r2 * LDK2 + fc * 4);
          k4[0] += S16<T>::val((unsigned short)(kh.x & 0xffff)) + S16<T>::val((unsigned short)(kl.x & 0xffff));
          k4[1] += S16<T>::val((unsigned short)(kh.x >> 16)) + S16<T>::val((unsigned short)(kl.x >> 16));
          k4[2] += S16<T>::val((unsigned short)(kh.y & 0xffff)) + S16<T>::val((unsigned short)(kl.y & 0xffff));
          k4[3] += S16<T>::val((unsigned short)(kh.y >> 16)) + S16<T>::val((unsigned short)(kl.y >> 16));
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (FG == 16) {
          k4[j] = xor16_sum(xor32_sum(k4[j]));                   // (l, l+32) then (+16): same association as the shuffle form
        } else {
#pragma unroll
          for (int o = FG * (RRN / 2); o >= FG; o >>= 1) k4[j] += __shfl_down(k4[j], o);   // fixed order
        }
      }
      if (lane < FG) *reinterpret_cast<float4*>(sKsP + wv * FP + fc * 4) = make_float4(k4[0], k4[1], k4[2], k4[3]);
    }
  };
  // ---- (d) O = A V + phi(Q) S over this wave's column blocks; (e) S += phi(K)^T V ------------------------------
  auto phase_de = [&](int t0, int rows, bool upd, const unsigned short* set) {
    const unsigned short* sQh = set;
    const unsigned short* sQl = sQh + C * LDQ2;
    const unsigned short* sKh = sQl + C * LDQ2;
    const unsigned short* sKl = sKh + C * LDK2;
    // what a phase of its own (and its barrier) did in the first version: the k-sum takes this chunk's increments (the partials
    // of (c) read it before, the next chunk's (c) runs behind the next barrier), and every lane sums the denominator partials of
    // ITS rows -- the same additions in the same order in every lane that holds the row
    if (tid < FP) {
      float s = sKsum[tid];
#pragma unroll
      for (int i = 0; i < NW; ++i) s += sKsP[i * FP + tid];       // fixed order: bitwise reproducible
      if (upd) sKsum[tid] = s;
    }
    // (a lane sums ONE row -- lane group g the rows of block g % RB -- and the reciprocals travel by lane permutes: four rows and
    // eight divisions per lane made the phase VALU-bound, 2.5 % slower than the barrier it replaces at 256 workgroups)
    float dn[RB], ri1 = 0.f;
    if constexpr (!STATE_ONLY) {
      const int myrow = (lg % RB) * 16 + li;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < DSL; ++i) s += sDenP[myrow * DSL + i];
      const float d1 = 1.0f / s;
      ri1 = 1.0f / (float)(row_base + t0 + myrow + 1);
#pragma unroll
      for (int ib = 0; ib < RB; ++ib) dn[ib] = __shfl(d1, ib * 16 + li);
    }
#pragma unroll
    for (int jq = 0; jq < JB; ++jq) {
      const int jb = wv + jq * NW, e0 = jb * 16;
      if (EB % NW != 0 && jb >= EB) break;                 // wave-uniform: no such column block
      // V fragments (k = chunk row): rows 32ks + 8lg + {0..3 | 4..7}, columns e0 .. e0+15
      uint4 vf[C / 32];
      {
        const int q = li >> 2, pp_ = li & 3;
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks) {
          const int r0 = ks * 32 + lg * 8;
          const uint2 a = lds_tr(sV + (r0 + q) * EL + vchunk(r0 + q, 2 * jb + (pp_ >> 1)) * 8 + 4 * (pp_ & 1));
          const uint2 b = lds_tr(sV + (r0 + 4 + q) * EL + vchunk(r0 + 4 + q, 2 * jb + (pp_ >> 1)) * 8 + 4 * (pp_ & 1));
          vf[ks] = cat8(a, b);
        }
      }
      if constexpr (STATE_ONLY) {
        if (want_avg && jb >= EB / 2) {                    // column total of the chunk = the last row's prefix
          f4 cum = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks <= (RB - 1) / 2; ++ks) cum = S16<T>::mfma(vf[ks], tril[RB - 1][ks], cum);
#pragma unroll
          for (int r = 0; r < 4; ++r) csum[jq][r] += __shfl(cum[r], (lane & 48) | 15);
        }
      } else {
      f4 o[RB];
#pragma unroll
      for (int ib = 0; ib < RB; ++ib) o[ib] = f4{0.f, 0.f, 0.f, 0.f};
      // Operands SWAPPED as in the D = 64 kernel (O^T = V^T A^T + S^T phi(Q)^T; the A and B fragment layouts mirror each other):
      // the accumulator of query block ib holds, per lane, row ib*16 + li and the four consecutive columns e0 + 4 lg + r.
      // A V: key k-step ks covers key blocks 2ks, 2ks+1; query block ib needs k-steps <= ib/2
#pragma unroll
      for (int ib = 0; ib < RB; ++ib) {
#pragma unroll
        for (int ks = 0; ks <= ib / 2; ++ks) {
          const uint4 ah = *reinterpret_cast<const uint4*>(sAh + (ib * 16 + li) * LDA + ks * 32 + lg * 8);
          const uint4 al = *reinterpret_cast<const uint4*>(sAl + (ib * 16 + li) * LDA + ks * 32 + lg * 8);
          o[ib] = S16<T>::mfma(vf[ks], ah, o[ib]);
          o[ib] = S16<T>::mfma(vf[ks], al, o[ib]);
        }
      }
      // phi(Q) S: the state tiles, split; k-step kk pairs feature blocks 2kk and 2kk+1
#pragma unroll
      for (int kk = 0; kk < KF; ++kk) {
        unsigned short h0[4], h1[4], l0[4], l1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          split16<T>(S[jq][2 * kk][r], h0[r], l0[r]);
          if (2 * kk + 1 < NBT) split16<T>(S[jq][(2 * kk + 1 < NBT) ? 2 * kk + 1 : 0][r], h1[r], l1[r]);
          else h1[r] = l1[r] = 0;
        }
        const uint4 bh = cat8(pack4(h0), pack4(h1)), bl = cat8(pack4(l0), pack4(l1));
#pragma unroll
        for (int ib = 0; ib < RB; ++ib) {
          const uint4 ah = *reinterpret_cast<const uint4*>(sQh + (ib * 16 + li) * LDQ2 + kk * 32 + lg * 8);
          const uint4 al = *reinterpret_cast<const uint4*>(sQl + (ib * 16 + li) * LDQ2 + kk * 32 + lg * 8);
          o[ib] = S16<T>::mfma(bh, ah, o[ib]);
          o[ib] = S16<T>::mfma(bl, ah, o[ib]);
          o[ib] = S16<T>::mfma(bh, al, o[ib]);
        }
      }
      const int col = e0 + 4 * lg;
#pragma unroll
      for (int ib = 0; ib < RB; ++ib) {
        const int row = ib * 16 + li;
        unsigned short o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = S16<T>::bits(o[ib][r] * dn[ib]);
        const uint2 w2 = pack4(o4);
        __builtin_amdgcn_raw_buffer_store_b64(bu2{w2.x, w2.y}, ro, (row < rows && t0 + row >= lead) ? ((t0 + row) * (3 * D) + col) * 2 : (int)OOB, 0, 0);
      }
      if (want_avg && jb >= EB / 2) {                      // wave-uniform: this wave's 16 columns are v features
        f4 cum[RB];
#pragma unroll
        for (int ib = 0; ib < RB; ++ib) {
          cum[ib] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks <= ib / 2; ++ks) cum[ib] = S16<T>::mfma(vf[ks], tril[ib][ks], cum[ib]);
        }
        const int gcol = col - D;
#pragma unroll
        for (int ib = 0; ib < RB; ++ib) {
          const int row = ib * 16 + li;
          const float ri = __shfl(ri1, ib * 16 + li);
          unsigned short g4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) g4[r] = S16<T>::bits((cum[ib][r] + csum[jq][r]) * ri);
          const uint2 w2 = pack4(g4);
          __builtin_amdgcn_raw_buffer_store_b64(bu2{w2.x, w2.y}, rg, (row < rows && t0 + row >= lead) ? ((t0 + row) * D + gcol) * 2 : (int)OOB, 0, 0);
        }
        if (upd) {                                         // column totals of the chunk = its last row's prefix (lane li = 15)
#pragma unroll
          for (int r = 0; r < 4; ++r) csum[jq][r] += __shfl(cum[RB - 1][r], (lane & 48) | 15);
        }
      }
      }
      // (e) S[f][e] += sum_s phi(k_s)[f] V[s][e]: A operand = phi(K)^T by transposing reads of the row-major images
      if (upd) {
        const int q = li >> 2, pp_ = li & 3;
#pragma unroll
        for (int rb = 0; rb < NBT; ++rb) {
#pragma unroll
          for (int ks = 0; ks < C / 32; ++ks) {
            const int r0 = ks * 32 + lg * 8;
            const int co = (rb >> 1) * 32 + 8 * pp_ + (rb & 1) * 4;      // positions of features 16rb + 4p .. +3
            const uint4 kh = cat8(lds_tr(sKh + (r0 + q) * LDK2 + co), lds_tr(sKh + (r0 + 4 + q) * LDK2 + co));
            const uint4 kl = cat8(lds_tr(sKl + (r0 + q) * LDK2 + co), lds_tr(sKl + (r0 + 4 + q) * LDK2 + co));
            S[jq][rb] = S16<T>::mfma(kh, vf[ks], S[jq][rb]);
            S[jq][rb] = S16<T>::mfma(kl, vf[ks], S[jq][rb]);
          }
        }
      }
    }
  };
  // ---- pass 1 (STATE_ONLY) walks TWO chunks per iteration on three barriers (round 5, late) --------------------------------------
  // The state pass needs phi(K) and S += phi(K)^T [pos | v] only: no causal structure, nothing to store per row.  Walked like
  // the output pass it spent its time between barriers with half the waves idle ((b): the four "K" waves; stamps: 1.65 us per
  // 32-row chunk at d = 80 on four barriers, 2.7 us at d = 128 on two).  Here the four "Q" waves of (b) take the NEXT chunk's k
  // rows (staged where q goes, phi images into the second image set), the k-sum partials are produced beside the state
  // update and folded in one barrier later, and (e) runs for both chunks back to back.  Same operand placement and k order per
  // chunk as before: the state tiles S are bit for bit the output pass's; the k-sum's partials are summed per chunk in the
  // same fixed order.
  if constexpr (STATE_ONLY) {
    auto stage_k_to = [&](unsigned short* dst, const bu4& r) {
      if (stager) *reinterpret_cast<bu4*>(dst + (sc * C + (sr ^ sc)) * 8) = r;
    };
    auto stage_v_to = [&](unsigned short* dst, const bu4& rp_, const bu4& rv_) {
      if (stager) {
        *reinterpret_cast<bu4*>(dst + sr * EL + vchunk(sr, sc) * 8) = rp_;
        *reinterpret_cast<bu4*>(dst + sr * EL + vchunk(sr, CPR + sc) * 8) = rv_;
      }
    };
    auto load_k = [&](int t0n) {
      const int t = t0n + sr;
      return __builtin_amdgcn_raw_buffer_load_b128(rk, (t < t_end && stager) ? (int)((t * p.ks[2] + sc * 8) * 2) : (int)OOB, 0, 0);
    };
    // phi(K) of one chunk by HALF the workgroup (the waves whose `which` selects it): src = staged k rows, dh = hi image
    auto phase_bk = [&](int rows, const unsigned short* src, unsigned short* dh) {
      const int rb = wv & (RB - 1), part = (BW - 1) - ((wv % (NW / 2)) / RB);
      f4 acc[FBP];
#pragma unroll
      for (int fb = 0; fb < FBP; ++fb) acc[fb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DP / 32; ++ks) {
        const uint4 bx = *reinterpret_cast<const uint4*>(src + ((4 * ks + lg) * C + ((rb * 16 + li) ^ ((4 * ks + lg) & 15))) * 8);
#pragma unroll
        for (int fb = 0; fb < FBP; ++fb) {
          const int fbg = part * FBP + fb;                   // (wave-uniform)
          if (fbg < NBT) {
            const uint4 aw = *reinterpret_cast<const uint4*>(sW + ((4 * ks + lg) * NBP + fbg * 16 + li) * 8);
            acc[fb] = S16<T>::mfma(aw, bx, acc[fb]);
          }
        }
      }
      unsigned short* dl = dh + C * LDK2;
      const int row = rb * 16 + li;
#pragma unroll
      for (int fb = 0; fb < FBP; ++fb) {
        const int fbg = part * FBP + fb;
        if (fbg < NBT) {
          unsigned short hh[4], ll[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = fbg * 16 + lg * 4 + r;
            float val = fmaxf(cnorm * acc[fb][r], 0.f) + 1e-3f;
            if (f >= p.nb || row >= rows) val = 0.f;
            split16<T>(val, hh[r], ll[r]);
          }
          const int pos = (fbg >> 1) * 32 + lg * 8 + (fbg & 1) * 4;
          *reinterpret_cast<uint2*>(dh + row * LDK2 + pos) = pack4(hh);
          *reinterpret_cast<uint2*>(dl + row * LDK2 + pos) = pack4(ll);
        }
      }
    };
    // per-wave k-sum increments of one chunk (phase (c)'s code) into `dst` [NW][FP]
    auto ksum_partial = [&](const unsigned short* sKh, float* dst) {
      const unsigned short* sKl = sKh + C * LDK2;
      constexpr int FG = FP / 4, RRN = 64 / FG, RPL = (C / NW) / RRN;
      const int fc = lane % FG, rr = lane / FG;
      float k4[4] = {0.f, 0.f, 0.f, 0.f};
      if (rr < RRN) {
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
          const int r2 = wv * (C / NW) + rr * RPL + j;
          const uint2 kh = *reinterpret_cast<const uint2*>(sKh + r2 * LDK2 + fc * 4);
          const uint2 kl = *reinterpret_cast<const uint2*>(sKl + r2 * LDK2 + fc * 4);
          k4[0] += S16<T>::val((unsigned short)(kh.x & 0xffff)) + S16<T>::val((unsigned short)(kl.x & 0xffff));
          k4[1] += S16<T>::val((unsigned short)(kh.x >> 16)) + S16<T>::val((unsigned short)(kl.x >> 16));
          k4[2] += S16<T>::val((unsigned short)(kh.y & 0xffff)) + S16<T>::val((unsigned short)(kl.y & 0xffff));
          k4[3] += S16<T>::val((unsigned short)(kh.y >> 16)) + S16<T>::val((unsigned short)(kl.y >> 16));
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (FG == 16) {
          k4[j] = xor16_sum(xor32_sum(k4[j]));
        } else {
#pragma unroll
          for (int o = FG * (RRN / 2); o >= FG; o >>= 1) k4[j] += __shfl_down(k4[j], o);
        }
      }
      if (lane < FG) *reinterpret_cast<float4*>(dst + wv * FP + fc * 4) = make_float4(k4[0], k4[1], k4[2], k4[3]);
    };
    auto ksum_fold = [&](const float* src) {                 // fixed order: bitwise reproducible
      if (tid < FP) {
        float s_ = sKsum[tid];
#pragma unroll
        for (int i = 0; i < NW; ++i) s_ += src[i * FP + tid];
        sKsum[tid] = s_;
      }
    };
    // column totals of v and S += phi(K)^T [pos | v] for one chunk: phase (d)(e)'s STATE_ONLY code
    auto phase_e = [&](const unsigned short* sVi, const unsigned short* sKh) {
      const unsigned short* sKl = sKh + C * LDK2;
      const int q = li >> 2, pp_ = li & 3;
      // V fragments of BOTH column blocks of the wave first: a phi(K) fragment, which depends on (feature block, k-step) only, is
      // then read ONCE and feeds both blocks' state tiles (the output pass's loop reads it per block: 40 transposing reads per
      // chunk and wave at d = 128, all eight waves reading the same 12 KB of images -- the phase is bound by the LDS array)
      uint4 vf[JB][C / 32];
      bool has[JB];
#pragma unroll
      for (int jq = 0; jq < JB; ++jq) {
        const int jb = wv + jq * NW;
        has[jq] = !(EB % NW != 0 && jb >= EB);               // (wave-uniform) no such column block
        const int jbc = has[jq] ? jb : wv;
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks) {
          const int r0 = ks * 32 + lg * 8;
          const uint2 a = lds_tr(sVi + (r0 + q) * EL + vchunk(r0 + q, 2 * jbc + (pp_ >> 1)) * 8 + 4 * (pp_ & 1));
          const uint2 b = lds_tr(sVi + (r0 + 4 + q) * EL + vchunk(r0 + 4 + q, 2 * jbc + (pp_ >> 1)) * 8 + 4 * (pp_ & 1));
          vf[jq][ks] = cat8(a, b);
        }
        if (has[jq] && want_avg && jb >= EB / 2) {
          f4 cum = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks <= (RB - 1) / 2; ++ks) cum = S16<T>::mfma(vf[jq][ks], tril[RB - 1][ks], cum);
#pragma unroll
          for (int r = 0; r < 4; ++r) csum[jq][r] += __shfl(cum[r], (lane & 48) | 15);
        }
      }
#pragma unroll
      for (int rb = 0; rb < NBT; ++rb) {
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks) {
          const int r0 = ks * 32 + lg * 8;
          const int co = (rb >> 1) * 32 + 8 * pp_ + (rb & 1) * 4;
          const uint4 kh = cat8(lds_tr(sKh + (r0 + q) * LDK2 + co), lds_tr(sKh + (r0 + 4 + q) * LDK2 + co));
          const uint4 kl = cat8(lds_tr(sKl + (r0 + q) * LDK2 + co), lds_tr(sKl + (r0 + 4 + q) * LDK2 + co));
#pragma unroll
          for (int jq = 0; jq < JB; ++jq) {
            if (has[jq]) {                                   // per tile: hi then lo, k-steps ascending -- the output pass's order
              S[jq][rb] = S16<T>::mfma(kh, vf[jq][ks], S[jq][rb]);
              S[jq][rb] = S16<T>::mfma(kl, vf[jq][ks], S[jq][rb]);
            }
          }
        }
      }
    };
    unsigned short* kA = sPhi + 2 * C * LDQ2;                // K images of the two sets
    unsigned short* kB = sPhi + PHI + 2 * C * LDQ2;
    // prefetch registers: pa = chunk A (k in .k, [pos | v] in .p / .v), pn = chunk B; the loads of the next iteration leave
    // right after this iteration's registers went to LDS
    pa.k = load_k(t_begin);  issue_v(t_begin, pa);            // (issue_qk / issue_v of the prologue above already asked for
    pn.k = load_k(t_begin + C);  issue_v(t_begin + C, pn);    //  chunk A: the duplicates are dropped by the compiler or hit L1)
    bool pending = false;                                    // k-sum partials of the previous iteration not folded yet
    for (int t0 = t_begin; t0 < t_end; t0 += 2 * C) {
      const bool two = t0 + C < t_end;                       // (block-uniform) the last iteration may hold one chunk
      stage_k_to(sK, pa.k);  stage_v_to(sV, pa.p, pa.v);
      stage_k_to(sQ, pn.k);  stage_v_to(sV2, pn.p, pn.v);
      pa.k = load_k(t0 + 2 * C);  issue_v(t0 + 2 * C, pa);
      pn.k = load_k(t0 + 3 * C);  issue_v(t0 + 3 * C, pn);
      if (pending) { ksum_fold(sKsP); }
      __syncthreads();                                       // staged rows visible; the fold above has read sKsP
      if (pending) { ksum_fold(sKsP2); }                     // (second chunk of the previous iteration: after the first, same order)
      if (wv >= NW / 2) phase_bk(rows_of(t0), sK, kA);
      else if (two) phase_bk(rows_of(t0 + C), sQ, kB);
      __syncthreads();                                       // phi(K) images visible
      ksum_partial(kA, sKsP);
      if (two) ksum_partial(kB, sKsP2);
      else if (lane < FP / 4) *reinterpret_cast<float4*>(sKsP2 + wv * FP + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      phase_e(sV, kA);
      if (two) phase_e(sV2, kB);
      pending = true;
      __syncthreads();                                       // images, staging buffers and partials are free / complete
    }
    if (pending) { ksum_fold(sKsP); }
    __syncthreads();
    if (pending) { ksum_fold(sKsP2); }
    __syncthreads();
    write_image(p.carry + ((int64_t)nh * (p.nseg - 1) + seg) * CARRY);
    return;
  }
  // D = 80 keeps the phases one after the other (four barriers): pipelined, the median workgroup is 7 % faster (105 vs 113.6 us on
  // 1 x 32 x 8192) but the workgroups of the LAST segment run 10 us behind (every chunk's P2 is longer; not a cold-cache effect: a
  // warming walk changes nothing), and the launch ends later (128.5 vs 121.9 us; two-chunk prefetch: 5 us behind) -- DESIGN.md section 9
  constexpr bool PIPE = D != 80;
#ifdef SEA_STAMP
  unsigned long long _tacc[4] = {0, 0, 0, 0};
  unsigned long long _tprev = __builtin_amdgcn_s_memtime();
  const unsigned long long _tstart = _tprev, _rstart = __builtin_amdgcn_s_memrealtime();
#endif
  if constexpr (!PIPE) {
    for (int t0 = t_begin; t0 < t_end; t0 += C) {
      stage_qk(pa);
      stage_v(t0, pa);
      issue_qk(t0 + C, pa);
      issue_v(t0 + C, pa);
      __syncthreads();
      WSTAMP(0);
      phase_b(rows_of(t0), sPhi);
      __syncthreads();
      WSTAMP(1);
      phase_c(sPhi);
      __syncthreads();
      WSTAMP(2);
      phase_de(t0, rows_of(t0), upd_of(t0), sPhi);
      __syncthreads();
      WSTAMP(3);
    }
  } else
  if (t_begin < t_end) {
    // prologue: chunk 0 through (b), (c); chunk 1's q, k staged; in flight: [pos | v] of chunks 1 (set A) and 2 (set B), q, k of
    // chunks 2 (B) and 3 (A)
    stage_qk(pa);
    stage_v(t_begin, pa);
    issue_qk(t_begin + C, pa);
    issue_v(t_begin + C, pa);
    issue_qk(t_begin + 2 * C, pn);
    issue_v(t_begin + 2 * C, pn);
    __syncthreads();
    phase_b(rows_of(t_begin), sPhi);
    __syncthreads();
    phase_c(sPhi);
    stage_qk(pa);
    issue_qk(t_begin + 3 * C, pa);
    __syncthreads();
    // iteration j: [pos | v] of chunk j+1 leaves set `rv_` (refilled with chunk j+3), q, k of chunk j+2 leave set `rqk_` (refilled
    // with chunk j+4); the sets swap roles every iteration, so the walk is unrolled by two
    auto iter = [&](int t0, int buf, Pre& rv_, Pre& rqk_) {
      const bool has_next = t0 + C < t_end;                  // block-uniform
      phase_de(t0, rows_of(t0), upd_of(t0), sPhi + buf * PHI);
      WSTAMP(0);
      if (has_next) phase_b(rows_of(t0 + C), sPhi + (buf ^ 1) * PHI);
      __syncthreads();
      WSTAMP(1);
      if (has_next) {
        phase_c(sPhi + (buf ^ 1) * PHI);
        stage_v(t0 + C, rv_);
        issue_v(t0 + 3 * C, rv_);
        stage_qk(rqk_);
        issue_qk(t0 + 4 * C, rqk_);
      }
      __syncthreads();
      WSTAMP(2);
    };
    for (int t0 = t_begin; t0 < t_end; t0 += 2 * C) {
      iter(t0, 0, pa, pn);
      if (t0 + C < t_end) iter(t0 + C, 1, pn, pa);
    }
  }
  WSTAMP_FLUSH();
  if constexpr (STATE_ONLY) {
    write_image(p.carry + ((int64_t)nh * (p.nseg - 1) + seg) * CARRY);
  } else {
    // the block that walked the last rows holds the final state (aligned step: the state at the last chunk boundary); a step that
    // closed no chunk and updates the image in place has nothing to write (see the D = 64 kernel)
    const bool unchanged = p.aligned && p.state_in == p.state_out && t_end - t_begin <= C && !upd_of(t_begin);
    if (p.state_out && seg == p.nseg - 1 && !unchanged) write_image(p.state_out + (int64_t)nh * CARRY);
  }
}

}  // namespace sea

using namespace sea;

template <typename T, int D, int NBT, int C, int NW = 8>
static int launch_perf(const PerfParams& p, hipStream_t s) {
  constexpr int E = 2 * D, NBP = NBT * 16;
  constexpr bool IS16 = !std::is_same<T, float>::value;
  constexpr int DK = (D + 31) / 32 * 32;
  constexpr size_t lds = IS16
      ? sizeof(float) * (C * (C + 2) + C * (E + 16) + 2 * C * (NBP + 2) + NBP + C + C * (C / 16 + NW * 64 / C)) +
            2 * ((DK / 8) * NBP * 8 + 2 * (DK / 8) * C * 8)
      : sizeof(float) * (NBP * (D + 2) + 2 * C * (D + 2) + C * (E + 16) + 2 * C * (NBP + 2) + NBP + C +
                         C * (C / 16 + NW * 64 / C));
  static_assert(lds <= 160 * 1024, "LDS budget");
  static DevOnce once;              // one per template instantiation and device; the attribute call is a slow driver round trip
  if (lds > 64 * 1024 && once.first()) {
    SEA_MAX_LDS((performer_kernel<T, D, NBT, C, NW, false>), lds);
    SEA_MAX_LDS((performer_kernel<T, D, NBT, C, NW, true>), lds);
  }
  if (p.seg_len % C != 0) return SEA_EINVAL;
  if (p.nseg > 1)
    hipLaunchKernelGGL((performer_kernel<T, D, NBT, C, NW, true>), dim3((unsigned)(p.N * p.H), (unsigned)(p.nseg - 1)), dim3(NW * 64), lds, s, p);
  hipLaunchKernelGGL((performer_kernel<T, D, NBT, C, NW, false>), dim3((unsigned)(p.N * p.H), (unsigned)p.nseg), dim3(NW * 64), lds, s, p);
  return SEA_OK;
}

// floats of carry per (n, h, segment) of the fp32 kernel: the per-thread state registers + the k-sum
template <int D, int NBT, int NW = 8>
constexpr int64_t perf_carry_floats() { return (int64_t)(((2 * D / 16) + NW - 1) / NW) * NBT * 4 * NW * 64 + NBT * 16; }

template <typename T, int NBT>
static int launch_perf_bf16(const PerfParams& p, hipStream_t s) {
  constexpr int D = 64, C = 64, NTH = 512, E = 2 * D, NBP = NBT * 16, FP = ((NBT + 1) / 2) * 32, LDQ2 = FP + 16, LDK2 = (FP == 64) ? 96 : FP + 8, LDA = C + 16;
  constexpr int NSET = (FP == 64) ? 2 : 1;                   // phi image sets (performer_bf16_kernel)
  constexpr size_t lds = 2 * ((D / 8) * NBP * 8 + 2 * (D / 8) * C * 8 + C * E + NSET * (2 * C * LDQ2 + 2 * C * LDK2) + 2 * C * LDA) +
                         sizeof(float) * (FP + C * (C / 16 + NTH / C) + 8 * FP);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static DevOnce once;
  if (once.first()) {
    SEA_MAX_LDS((performer_bf16_kernel<T, NBT, false>), lds);
    SEA_MAX_LDS((performer_bf16_kernel<T, NBT, true>), lds);
  }
  if (p.seg_len % C != 0) return SEA_EINVAL;
  if (p.nseg > 1)
    hipLaunchKernelGGL((performer_bf16_kernel<T, NBT, true>), dim3((unsigned)(p.N * p.H), (unsigned)(p.nseg - 1)), dim3(NTH), lds, s, p);
  hipLaunchKernelGGL((performer_bf16_kernel<T, NBT, false>), dim3((unsigned)(p.N * p.H), (unsigned)p.nseg), dim3(NTH), lds, s, p);
  return SEA_OK;
}

template <int NBT>
constexpr int64_t perf_bf16_carry_floats() { return (int64_t)NBT * 4 * 512 + ((NBT + 1) / 2) * 32 + 512; }

// the wide-head form of the 16-bit kernel (d = 128: 32-row chunks, two column blocks per wave)
template <typename T, int D, int C, int NBT>
static int launch_perf_bf16w(const PerfParams& p, hipStream_t s) {
  constexpr int NTH = 512, NBP = NBT * 16, FP = ((NBT + 1) / 2) * 32, LDQ2 = FP + 16, LDK2 = (FP == 64) ? 96 : FP + 8, LDA = C + 16;
  constexpr int KC = (D + 31) / 32 * 4, EL = 256;
  constexpr size_t lds = 2 * (KC * NBP * 8 + 2 * KC * C * 8 + C * EL + 2 * (2 * C * LDQ2 + 2 * C * LDK2) + 2 * C * LDA) +
                         sizeof(float) * (FP + C * (C / 16 + 8) + 8 * FP) +
                         sizeof(float) * 8 * FP + 2 * C * EL;     // the state pass's second k-sum partials and second [pos | v] image
  static_assert(lds <= 160 * 1024, "LDS budget");
  static DevOnce once;
  if (once.first()) {
    SEA_MAX_LDS((performer_bf16w_kernel<T, D, C, NBT, false>), lds);
    SEA_MAX_LDS((performer_bf16w_kernel<T, D, C, NBT, true>), lds);
  }
  if (p.seg_len % C != 0) return SEA_EINVAL;
  if (p.nseg > 1)
    hipLaunchKernelGGL((performer_bf16w_kernel<T, D, C, NBT, true>), dim3((unsigned)(p.N * p.H), (unsigned)(p.nseg - 1)), dim3(NTH), lds, s, p);
  hipLaunchKernelGGL((performer_bf16w_kernel<T, D, C, NBT, false>), dim3((unsigned)(p.N * p.H), (unsigned)p.nseg), dim3(NTH), lds, s, p);
  return SEA_OK;
}

template <int D, int NBT>
constexpr int64_t perf_bf16w_carry_floats() { return (int64_t)((2 * D / 16 + 7) / 8) * NBT * 4 * 512 + ((NBT + 1) / 2) * 32 + ((2 * D / 16 + 7) / 8) * 512; }

template <typename T>
static int dispatch_perf(const PerfParams& p, int D, int nbt, hipStream_t s) {
  if constexpr (!std::is_same<T, float>::value) {           // 16-bit data, d = 64 / 128: split-operand 16-bit MFMA kernels
    if (D == 64) {
      if (nbt <= 3) return launch_perf_bf16<T, 3>(p, s);
      if (nbt <= 5) return launch_perf_bf16<T, 5>(p, s);
    }
    if (D == 80 && nbt <= 3) return launch_perf_bf16w<T, 80, 32, 3>(p, s);
    if (D == 80 && nbt <= 5) return launch_perf_bf16w<T, 80, 32, 5>(p, s);
    if (D == 128 && nbt <= 5) return launch_perf_bf16w<T, 128, 32, 5>(p, s);
  }
  if (D == 64 && nbt <= 3) return launch_perf<T, 64, 3, 64>(p, s);
  if (D == 64 && nbt <= 5) return launch_perf<T, 64, 5, 64>(p, s);
  if (D == 80 && nbt <= 3) return launch_perf<T, 80, 3, 64>(p, s);
  if (D == 80 && nbt <= 5) return launch_perf<T, 80, 5, 32>(p, s);
  if (D == 128 && nbt <= 5) return launch_perf<T, 128, 5, 32>(p, s);
  return SEA_EUNSUPPORTED;
}

#ifdef SEA_STAMP
extern "C" int sea_debug_perf_wg(unsigned long long* host2048) {
  (void)hipMemcpyFromSymbol(host2048, HIP_SYMBOL(sea_dbg_wg), sizeof(unsigned long long) * 2048);
  return 0;
}
extern "C" int sea_debug_perf_stamps(unsigned long long* host8) {
  (void)hipMemcpyFromSymbol(host8, HIP_SYMBOL(sea_dbg_perf), sizeof(unsigned long long) * 16);
  unsigned long long z[16] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(sea_dbg_perf), z, sizeof(z));
  return 0;
}
#endif

// carry floats per (n, h, segment) of the kernel dispatch_perf picks (0: unsupported shape)
static int64_t perf_carry_floats_for(int dtype, int D, int nbt) {
  if (dtype != SEA_F32 && D == 64) {
    if (nbt <= 3) return perf_bf16_carry_floats<3>();
    if (nbt <= 5) return perf_bf16_carry_floats<5>();
  }
  if (dtype != SEA_F32 && D == 80 && nbt <= 3) return perf_bf16w_carry_floats<80, 3>();
  if (dtype != SEA_F32 && D == 80 && nbt <= 5) return perf_bf16w_carry_floats<80, 5>();
  if (dtype != SEA_F32 && D == 128 && nbt <= 5) return perf_bf16w_carry_floats<128, 5>();
  if (D == 64 && nbt <= 3) return perf_carry_floats<64, 3>();
  if (D == 64 && nbt <= 5) return perf_carry_floats<64, 5>();
  if (D == 80 && nbt <= 3) return perf_carry_floats<80, 3>();
  if (D == 80 && nbt <= 5) return perf_carry_floats<80, 5>();
  if (D == 128 && nbt <= 5) return perf_carry_floats<128, 5>();
  return 0;
}

// rows per chunk of the 16-bit MFMA kernel dispatch_perf picks (0: no chunk-aligned step for this shape / dtype)
static int perf_chunk_rows_for(int dtype, int D, int nbt) {
  if (dtype == SEA_F32) return 0;
  if (D == 64 && nbt <= 5) return 64;
  if ((D == 80 || D == 128) && nbt <= 5) return 32;
  return 0;
}

extern "C" int64_t sea_performer_chunk_rows(int64_t D, int64_t nb, int dtype) {
  if (D <= 0 || nb <= 0) return 0;
  return perf_chunk_rows_for(dtype, (int)D, (int)((nb + 15) / 16));
}

// rows per segment: whole 64-row chunks (a multiple of every kernel's chunk size), segments as even as possible
static int64_t perf_seg_len(int64_t T, int64_t nseg) {
  const int64_t chunks = (T + 63) / 64;
  return ((chunks + nseg - 1) / nseg) * 64;
}

// avg_out comes from the 16-bit MFMA kernels only (16-bit data, d = 64, 80 or 128, up to 80 features)
extern "C" int sea_performer_avg_supported(int64_t D, int64_t nb, int dtype) {
  return (dtype == SEA_BF16 || dtype == SEA_F16) && (D == 64 || D == 80 || D == 128) && nb > 0 && nb <= 80;
}

extern "C" int sea_performer_plan(int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb, int dtype,
                                  int64_t* n_segments, int64_t* workspace_bytes) {
  const char* nm = "sea_performer_plan";
  SEA_REQUIRE(n_segments && workspace_bytes, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(N > 0 && H > 0 && T > 0 && D > 0 && nb > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int64_t cf = perf_carry_floats_for(dtype, (int)D, (int)((nb + 15) / 16));
  SEA_REQUIRE(cf > 0, SEA_EUNSUPPORTED, "%s: unsupported head size D=%lld / feature count nb=%lld", nm, (long long)D, (long long)nb);
  // One workgroup walks one (n, h) pair's rows in order; with fewer pairs than compute units the rows are cut into
  // segments (pass 1 re-does the phi(K) / state part of all but the last: ~1/3 extra work for nseg x the parallelism).
  const int64_t pairs = N * H, chunks = (T + 63) / 64;
  int64_t nseg = 1;
  if (pairs <= 128 && chunks >= 8) {
    nseg = 256 / pairs;                                      // all workgroups of a pass resident at once (1 per CU: LDS)
    if (nseg > chunks / 4) nseg = chunks / 4;
    if (nseg > 16) nseg = 16;
    const int64_t len = perf_seg_len(T, nseg);
    nseg = (T + len - 1) / len;                              // no empty segment
  }
  *n_segments = nseg;
  *workspace_bytes = pairs * (nseg - 1) * cf * (int64_t)sizeof(float);
  return SEA_OK;
}

static int perf_entry(const char* nm, const void* q, const void* k, const void* v, const void* pos, int dtype,
                      const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                      const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                      int64_t pos_stride, void* out, void* avg_out, int64_t n_segments, void* workspace,
                      int64_t workspace_bytes, const void* state_in, void* state_out, int64_t state_bytes, int64_t t_base,
                      const int32_t* t_base_dev, int aligned,
                      sea_stream_t stream) {
  SEA_REQUIRE(q && k && v && pos && proj && out && q_strides && k_strides && v_strides, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && H > 0 && T > 0 && D > 0 && nb > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int vec = dtype == SEA_F32 ? 4 : 8;
  auto ok3 = [&](const int64_t* s) { return s[0] % vec == 0 && s[1] % vec == 0 && s[2] % vec == 0; };
  SEA_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)pos | (uintptr_t)out) & 15) == 0 && ok3(q_strides) &&
                  ok3(k_strides) && ok3(v_strides) && pos_stride % vec == 0,
              SEA_EUNSUPPORTED, "%s: rows must be 16-byte aligned", nm);
  PerfParams p;
  p.q = q; p.k = k; p.v = v; p.pos = pos; p.W = proj; p.out = out; p.avg = avg_out;
  SEA_REQUIRE(avg_out == nullptr || (sea_performer_avg_supported(D, nb, dtype) && ((uintptr_t)avg_out & 15) == 0),
              SEA_EUNSUPPORTED, "%s: avg_out needs the 16-bit MFMA kernel (bf16 / fp16 data, D = 64)", nm);
  for (int i = 0; i < 3; ++i) { p.qs[i] = q_strides[i]; p.ks[i] = k_strides[i]; p.vs[i] = v_strides[i]; }
  p.pos_stride = pos_stride;
  p.N = (int)N; p.H = (int)H; p.T = (int)T; p.nb = (int)nb;
  const int nbt = (int)((nb + 15) / 16);
  SEA_REQUIRE(n_segments >= 1 && n_segments <= 64, SEA_EINVAL, "%s: n_segments %lld outside 1..64", nm, (long long)n_segments);
  p.nseg = (int)n_segments;
  // local rows of the call: the new rows plus, for a chunk-aligned step, the open chunk's old rows in front of them (with the
  // position in device memory the host cannot know how many: up to a chunk -- one segment, any length covers it)
  int64_t TL = T;
  p.aligned = aligned;
  if (aligned) {
    const int C = perf_chunk_rows_for(dtype, (int)D, nbt);
    SEA_REQUIRE(C > 0, SEA_EUNSUPPORTED, "%s: the chunk-aligned step runs on the 16-bit MFMA kernels (bf16 / fp16 data, D = 64, 80, 128)", nm);
    TL = T + (t_base_dev ? C - 1 : t_base % C);
  }
  p.seg_len = (int)perf_seg_len(TL, n_segments);
  p.carry = reinterpret_cast<float*>(workspace);
  p.state_in = reinterpret_cast<const float*>(state_in);
  p.state_out = reinterpret_cast<float*>(state_out);
  p.t_base = (int)t_base;
  p.t_base_dev = t_base_dev;
  if (state_in || state_out) {
    const int64_t need = N * H * perf_carry_floats_for(dtype, (int)D, nbt) * (int64_t)sizeof(float);
    SEA_REQUIRE(need > 0 && state_bytes >= need && ((((uintptr_t)state_in) | ((uintptr_t)state_out)) & 15) == 0 && t_base >= 0,
                SEA_EINVAL, "%s: state images of %lld bytes needed (16-byte aligned), got %lld", nm, (long long)need, (long long)state_bytes);
    SEA_REQUIRE(state_in != nullptr || t_base == 0, SEA_EINVAL, "%s: t_base %lld without state_in", nm, (long long)t_base);
    SEA_REQUIRE(n_segments == 1 || state_in != state_out, SEA_EINVAL, "%s: state_in and state_out may alias only with one segment", nm);
  }
  if (n_segments > 1) {
    SEA_REQUIRE((n_segments - 1) * (int64_t)p.seg_len < TL, SEA_EINVAL, "%s: %lld segments leave one empty at T=%lld (use sea_performer_plan)",
                nm, (long long)n_segments, (long long)TL);
    const int64_t need = N * H * (n_segments - 1) * perf_carry_floats_for(dtype, (int)D, nbt) * (int64_t)sizeof(float);
    SEA_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= need, SEA_EINVAL,
                "%s: workspace of %lld bytes needed (16-byte aligned), got %lld", nm, (long long)need, (long long)workspace_bytes);
  }
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (dtype == SEA_F32) rc = dispatch_perf<float>(p, (int)D, nbt, s);
  else if (dtype == SEA_F16) rc = dispatch_perf<__half>(p, (int)D, nbt, s);
  else rc = dispatch_perf<__hip_bfloat16>(p, (int)D, nbt, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported head size D=%lld / feature count nb=%lld", nm, (long long)D, (long long)nb);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_performer_causal(const void* q, const void* k, const void* v, const void* pos, int dtype,
                                    const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                                    const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                    int64_t pos_stride, void* out, void* avg_out, sea_stream_t stream) {
  return perf_entry("sea_performer_causal", q, k, v, pos, dtype, proj, N, H, T, D, nb, q_strides, k_strides, v_strides,
                    pos_stride, out, avg_out, 1, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 0, stream);
}

extern "C" int sea_performer_causal_segmented(const void* q, const void* k, const void* v, const void* pos, int dtype,
                                              const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                                              const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                              int64_t pos_stride, void* out, void* avg_out, int64_t n_segments,
                                              void* workspace, int64_t workspace_bytes, sea_stream_t stream) {
  return perf_entry("sea_performer_causal_segmented", q, k, v, pos, dtype, proj, N, H, T, D, nb, q_strides, k_strides,
                    v_strides, pos_stride, out, avg_out, n_segments, workspace, workspace_bytes, nullptr, nullptr, 0, 0, nullptr, 0, stream);
}

extern "C" int64_t sea_performer_state_bytes(int64_t N, int64_t H, int64_t D, int64_t nb, int dtype) {
  if (N <= 0 || H <= 0 || D <= 0 || nb <= 0) return 0;
  return N * H * perf_carry_floats_for(dtype, (int)D, (int)((nb + 15) / 16)) * (int64_t)sizeof(float);
}

// Stateful step, CHUNK ALIGNED (kv-cache decoding that reproduces the stateless pass bit for bit, attention_state.py:43-140 /
// test_perlin_opt_cache.py:7-32): the sequences have seen `t_base` rows, the image is the state at the last chunk boundary
// c0 = floor(t_base / C) * C (C = sea_performer_chunk_rows), and the call walks the open chunk again from c0.
//   k, v, pos  point at ROW c0 (the caller's kv-cache / embedding table hold those rows), T + t_base % C rows are read;
//   q, out, avg_out point at the first NEW row, T rows.
// state_out = the image at the last chunk boundary at or below t_base + T.  16-bit data, D in {64, 80, 128}.
extern "C" int sea_performer_causal_step(const void* q, const void* k, const void* v, const void* pos, int dtype,
                                         const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                                         const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                         int64_t pos_stride, void* out, void* avg_out, const void* state_in,
                                         void* state_out, int64_t state_bytes, int64_t t_base, int64_t n_segments,
                                         void* workspace, int64_t workspace_bytes, sea_stream_t stream) {
  return perf_entry("sea_performer_causal_step", q, k, v, pos, dtype, proj, N, H, T, D, nb, q_strides, k_strides,
                    v_strides, pos_stride, out, avg_out, n_segments, workspace, workspace_bytes, state_in, state_out,
                    state_bytes, t_base, nullptr, 1, stream);
}

// The same step with the position in DEVICE memory (SURVEY 8f-3 / opt_generate.py:131: the decode loop captured once as
// a HIP graph and replayed per token -- nothing position-dependent may live in kernel arguments).  *t_base_dev = rows the
// state has seen; k_cache / v_cache are the BASES (row 0) of the kv-caches, which already hold the new rows, pos_table the
// BASE of the value-embedding table: the kernel finds the chunk boundary itself.  q / out / avg_out: the T new rows.
// state_in and state_out may be the same image (updated in place); one segment.
extern "C" int sea_performer_causal_step_at(const void* q, const void* k_cache, const void* v_cache, const void* pos_table, int dtype,
                                            const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                                            const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                            int64_t pos_stride, void* out, void* avg_out, const void* state_in,
                                            void* state_out, int64_t state_bytes, const int32_t* t_base_dev,
                                            sea_stream_t stream) {
  SEA_REQUIRE(t_base_dev && state_in && state_out, SEA_EINVAL, "sea_performer_causal_step_at: null pointer");
  return perf_entry("sea_performer_causal_step_at", q, k_cache, v_cache, pos_table, dtype, proj, N, H, T, D, nb, q_strides, k_strides,
                    v_strides, pos_stride, out, avg_out, 1, nullptr, 0, state_in, state_out, state_bytes, 0, t_base_dev, 1, stream);
}
