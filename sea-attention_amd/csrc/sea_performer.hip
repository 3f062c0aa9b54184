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

#ifdef SEA_STAMP
__device__ unsigned long long sea_dbg_perf[8];
#define PSTAMP(i) do { if (threadIdx.x == 0) { unsigned long long _t = __builtin_amdgcn_s_memtime(); atomicAdd(&sea_dbg_perf[i], _t - _tprev); _tprev = _t; } } while (0)
#else
#define PSTAMP(i) do {} while (0)
#endif

namespace sea {

using f4 = __attribute__((ext_vector_type(4))) float;
#define SEA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct PerfParams {
  const void *q, *k, *v;   // (N,H,T,D)
  const void* pos;         // (>=T, D) learned causal value embedding
  const float* W;          // (nb, D) fp32
  void* out;               // (N,H,T,3D)
  int64_t qs[3], ks[3], vs[3];
  int64_t pos_stride;
  int N, H, T, nb;
};

// NW = waves per workgroup (8: two per SIMD, so one wave's LDS/MFMA latency hides behind the other's issue)
template <typename T, int D, int NBT, int C, int NW>
__global__ __launch_bounds__(NW * 64) void performer_kernel(PerfParams p) {
  constexpr int NTH = NW * 64;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int E = 2 * D, NBP = NBT * 16;
  constexpr int RB = C / 16;                  // row blocks per chunk
  constexpr int EB = E / 16;                  // column blocks of V / S / O
  constexpr int JB = (EB + NW - 1) / NW;      // column blocks owned by one wave
  constexpr int LDQ = D + 2, LDV = E + 16, LDP = NBP + 2, LDA = C + 2, LDW = D + 2;
  static_assert(LDA <= LDQ, "the A tile is overlaid on the Q tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;               // NBP x LDW   W (rows >= nb are zero)
  float* sQ = sW + NBP * LDW;     // C x LDQ     Q chunk, later the masked A tile (C x LDA)
  float* sK = sQ + C * LDQ;       // C x LDQ
  float* sV = sK + C * LDQ;       // C x LDV     [pos | v]
  float* sQp = sV + C * LDV;      // C x LDP     phi(Q)
  float* sKp = sQp + C * LDP;     // C x LDP     phi(K)
  float* sKsum = sKp + C * LDP;   // NBP         running sum of phi(k)
  float* sDen = sKsum + NBP;      // C           denominators of the current chunk
  constexpr int DSL = RB + NW * 64 / C;       // partial-denominator slots per row: RB key blocks + carry parts
  float* sDenP = sDen + C;        // C x DSL     partials, summed in a fixed order (bitwise reproducible)
  float* sA = sQ;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;   // MFMA lane coordinates
  const int nh = blockIdx.x;
  const int n = nh / p.H, h = nh - n * p.H;
  const T* qb = reinterpret_cast<const T*>(p.q) + n * p.qs[0] + h * p.qs[1];
  const T* kb = reinterpret_cast<const T*>(p.k) + n * p.ks[0] + h * p.ks[1];
  const T* vb = reinterpret_cast<const T*>(p.v) + n * p.vs[0] + h * p.vs[1];
  const T* pb = reinterpret_cast<const T*>(p.pos);
  T* ob = reinterpret_cast<T*>(p.out) + (int64_t)nh * p.T * (3 * D);
  const float cnorm = powf((float)D, -0.25f);

  for (int i = tid; i < NBP * LDW; i += NTH) {
    const int r = i / LDW, c = i - r * LDW;
    sW[i] = (r < p.nb && c < D) ? p.W[r * D + c] : 0.f;
  }
  for (int i = tid; i < NBP; i += NTH) sKsum[i] = 0.f;

  f4 S[JB][NBT];
#pragma unroll
  for (int a = 0; a < JB; ++a)
#pragma unroll
    for (int b = 0; b < NBT; ++b) S[a][b] = f4{0.f, 0.f, 0.f, 0.f};

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
      if (ch < C * (D / VEC) && t0n + r < p.T) {
        const int64_t t = t0n + r;
        pq[i] = *reinterpret_cast<const uint4*>(qb + t * p.qs[2] + c);
        pk[i] = *reinterpret_cast<const uint4*>(kb + t * p.ks[2] + c);
        pv[i] = *reinterpret_cast<const uint4*>(vb + t * p.vs[2] + c);
        pp[i] = *reinterpret_cast<const uint4*>(pb + t * p.pos_stride + c);
      }
    }
  };
  issue_loads(0);

  for (int t0 = 0; t0 < p.T; t0 += C) {
    const int rows = min(C, p.T - t0);
    // ---- (a) registers -> LDS as fp32: Q, K (C x D), V = [pos | v] (C x 2D); copy v into out[..., 2D:3D] ------
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = tid + i * NTH;
      if (ch < C * (D / VEC)) {
        const int r = ch / (D / VEC), c = (ch - r * (D / VEC)) * VEC;
        float fq[VEC], fk[VEC], fv[VEC], fp[VEC];
        if (r < rows) *reinterpret_cast<uint4*>(ob + (int64_t)(t0 + r) * (3 * D) + 2 * D + c) = pv[i];
        unpack16<T>(pq[i], fq); unpack16<T>(pk[i], fk); unpack16<T>(pv[i], fv); unpack16<T>(pp[i], fp);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          sQ[r * LDQ + c + j] = fq[j];
          sK[r * LDQ + c + j] = fk[j];
          sV[r * LDV + c + j] = fp[j];
          sV[r * LDV + D + c + j] = fv[j];
        }
      }
    }
    if (t0 + C < p.T) issue_loads(t0 + C);                 // next chunk: in flight during phases (b)..(e)
    for (int i = tid; i < C * DSL; i += NTH) sDenP[i] = 0.f;
    __syncthreads();
    PSTAMP(0);   // (a) staging

    // ---- (b) feature maps phi(Q), phi(K): (C x D) @ W^T -> (C x NBP) -------------------------------------
    // a wave takes (matrix, row block) pairs: one A fragment feeds NBT independent accumulators
    for (int grp = wv; grp < 2 * RB; grp += NW) {
      const int which = grp / RB, ib = grp - which * RB;        // 0: Q, 1: K
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
    PSTAMP(2);   // (c) A + denominators
    // the diagonal part of the denominator also carries eps: sum_r phi(q)_r * eps is already in the carry term;
    // the intra-chunk part needs none (eps is added once to the k-sum, not per key).

    // ---- (d) O = A V + phi(Q) S, divided by the denominators; (e) S += phi(K)^T V -----------------------------
    // per owned column block: all RB row blocks of O are accumulated together (one V / S fragment feeds RB MFMAs)
#pragma unroll
    for (int a_ = 0; a_ < JB; ++a_) {
      const int jb = wv + NW * a_;
      if (jb < EB) {
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
}

}  // namespace sea

using namespace sea;

template <typename T, int D, int NBT, int C, int NW = 8>
static int launch_perf(const PerfParams& p, hipStream_t s) {
  constexpr int E = 2 * D, NBP = NBT * 16;
  constexpr size_t lds = sizeof(float) * (NBP * (D + 2) + 2 * C * (D + 2) + C * (E + 16) + 2 * C * (NBP + 2) + NBP + C +
                                          C * (C / 16 + NW * 64 / C));
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool configured = false;   // one per template instantiation; the attribute call is a slow driver round trip
  if (lds > 64 * 1024 && !configured) {
    (void)hipFuncSetAttribute((const void*)performer_kernel<T, D, NBT, C, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = true;
  }
  hipLaunchKernelGGL((performer_kernel<T, D, NBT, C, NW>), dim3((unsigned)(p.N * p.H)), dim3(NW * 64), lds, s, p);
  return SEA_OK;
}

template <typename T>
static int dispatch_perf(const PerfParams& p, int D, int nbt, hipStream_t s) {
  if (D == 64 && nbt <= 3) return launch_perf<T, 64, 3, 64>(p, s);
  if (D == 64 && nbt <= 5) return launch_perf<T, 64, 5, 64>(p, s);
  if (D == 80 && nbt <= 3) return launch_perf<T, 80, 3, 64>(p, s);
  if (D == 80 && nbt <= 5) return launch_perf<T, 80, 5, 32>(p, s);
  if (D == 128 && nbt <= 5) return launch_perf<T, 128, 5, 32>(p, s);
  return SEA_EUNSUPPORTED;
}

#ifdef SEA_STAMP
extern "C" int sea_debug_perf_stamps(unsigned long long* host8) {
  (void)hipMemcpyFromSymbol(host8, HIP_SYMBOL(sea_dbg_perf), sizeof(unsigned long long) * 8);
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(sea_dbg_perf), z, sizeof(z));
  return 0;
}
#endif

extern "C" int sea_performer_causal(const void* q, const void* k, const void* v, const void* pos, int dtype,
                                    const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                                    const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                    int64_t pos_stride, void* out, sea_stream_t stream) {
  const char* nm = "sea_performer_causal";
  SEA_REQUIRE(q && k && v && pos && proj && out && q_strides && k_strides && v_strides, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F32 || dtype == SEA_F16 || dtype == SEA_BF16, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && H > 0 && T > 0 && D > 0 && nb > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int vec = dtype == SEA_F32 ? 4 : 8;
  auto ok3 = [&](const int64_t* s) { return s[0] % vec == 0 && s[1] % vec == 0 && s[2] % vec == 0; };
  SEA_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)pos | (uintptr_t)out) & 15) == 0 && ok3(q_strides) &&
                  ok3(k_strides) && ok3(v_strides) && pos_stride % vec == 0,
              SEA_EUNSUPPORTED, "%s: rows must be 16-byte aligned", nm);
  PerfParams p;
  p.q = q; p.k = k; p.v = v; p.pos = pos; p.W = proj; p.out = out;
  for (int i = 0; i < 3; ++i) { p.qs[i] = q_strides[i]; p.ks[i] = k_strides[i]; p.vs[i] = v_strides[i]; }
  p.pos_stride = pos_stride;
  p.N = (int)N; p.H = (int)H; p.T = (int)T; p.nb = (int)nb;
  const int nbt = (int)((nb + 15) / 16);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (dtype == SEA_F32) rc = dispatch_perf<float>(p, (int)D, nbt, s);
  else if (dtype == SEA_F16) rc = dispatch_perf<__half>(p, (int)D, nbt, s);
  else rc = dispatch_perf<__hip_bfloat16>(p, (int)D, nbt, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported head size D=%lld / feature count nb=%lld", nm, (long long)D, (long long)nb);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
