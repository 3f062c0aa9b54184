// SEA predictor CNN in channel-blocked ("C8") form, hand-written for gfx950 (16-bit data, bf16/f16 MFMA).
//
// Replaces, for 16-bit tensors (reference: src/models/perlin_attention/attention.py:266-281,
// modules.py:96-192):
//   ChannelSplit + cnn.lnorm1                         -> split_layernorm_c8_kernel   (writes C8)
//   cnn.keepres.conv1 / conv2 (+ the ReLU after each) -> causal_conv_c8_kernel       (C8 -> C8)
//
// C8 layout of an activation with logical shape (N, C, T, W):  memory (N, T, C/8, W, 8) -- blocks of 8 channels
// (16 B) are the unit, consecutive pixels of one block are adjacent.  It is what the MFMA operand wants: a lane
// holds 8 consecutive K (channels) of one pixel, the 16 lanes of a fragment column-group hold 16 consecutive
// pixels, so every wave-wide load or store touches four contiguous 256-byte runs (plain NHWC puts the 16 pixels
// 2*C bytes apart: 16 half-used cache lines per pass, which kept the texture addresser ~80 % busy and the MFMA
// pipe at 17 %).  A tap shift is an offset of whole 16-byte units.  MIOpen's implicit GEMM needs NHWC and brackets
// every NCHW conv with two transposes, a padded copy and separate bias / ReLU passes; here the tensors stay C8
// between the LayerNorm and the predictor tail, padding is hardware range-checking and bias + ReLU live in the
// epilogue.
//
// causal_conv_c8_kernel: implicit GEMM, M = C_out, N = pixels, K = taps x C_in, v_mfma_f32_16x16x32.
//   one wave = 64 consecutive pixels of one (n, t) row  x  all C_out   (NT M-tiles x 4 N-tiles)
#include "sea_common.hpp"

namespace sea {

typedef __attribute__((ext_vector_type(4))) float cf4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) unsigned int cu4;

template <typename T> struct Mfma16;
template <> struct Mfma16<__hip_bfloat16> {
  __device__ static inline cf4 run(const uint4& a, const cu4& b, cf4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma16<__half> {
  __device__ static inline cf4 run(const uint4& a, const cu4& b, cf4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
  }
};

struct ConvParams {
  const void* x;    // (N, T, Cin/8, W, 8)   C8
  const void* w;    // (Cout, taps*CinP) packed [co][tap][ci], ci padded to CinP (multiple of 32), 16-bit
  const float* b;   // (Cout) fp32
  void* y;          // (N, T, Cout/8, W, 8)  C8
  int N, T, W, Cin, Cout, CinP;
  int KS, dil, pad_w, relu;
};

// NT = number of 16-wide output-channel tiles (Cout <= 16*NT); KS = square kernel size (compile time: the
// per-lane tap offsets live in registers); 8 waves per workgroup share one weight image in LDS.
//
// A operand = weights, B operand = pixels: in the accumulator (col = lane%16, row = 4*(lane/16) + r) a lane then
// owns ONE pixel and 4 consecutive channels per M-tile.  Weight row (nt, 4*g + r) of the LDS image is channel
// g*4*NT + nt*4 + r, so the 4*NT values of a lane are one contiguous run of channels -> 16-byte C8 stores.
// LDS weight image: [k-step][lg][row][8 elements]; a lane's 16 bytes sit at bank 4*li for every lg, which is
// conflict-free for ds_read_b128's lane groups without any padding.
// k-steps run (tap row, 32-channel chunk, tap column); the three column shifts of one chunk re-read the same
// lines back to back.  Pixel fragments come through buffer loads: a lane whose tap falls outside the row (or whose
// channel block is K padding) carries an offset beyond num_records and the hardware returns zeros, so the k-loop
// is branch-free and the next step's loads are in flight under the current step's MFMAs.
constexpr int CONV_WAVES = 6;
constexpr unsigned CONV_OOB = 0x7FFFFF00u;       // > any valid byte offset (launcher checks the image is < 1 GiB)

// ONESEG: W <= 64, every row is one segment: the tap-validity tests and lane offsets are then kernel invariants.
template <typename T, int NT, int KS, bool ONESEG>
__global__ __launch_bounds__(CONV_WAVES * 64, (NT <= 4 ? 3 : 2)) void causal_conv_c8_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTH = CONV_WAVES * 64;
  constexpr int ROWS = 16 * NT;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int kchunks = p.CinP / 32;
  const int nsteps = KS * KS * kchunks;          // k-steps, ordered (ti, chunk, tj)
  const int KP = KS * KS * p.CinP;               // packed K extent of the global weight rows (elements)
  T* sW = reinterpret_cast<T*>(smem);            // nsteps x 4 x ROWS x 8
  float* sBias = reinterpret_cast<float*>(smem + (size_t)nsteps * 4 * ROWS * 8 * sizeof(T));   // ROWS, by channel
  // ---- stage the weights once per workgroup ----------------------------------------------------------
  {
    const T* wg = reinterpret_cast<const T*>(p.w);
    for (int ch = threadIdx.x; ch < nsteps * 4 * ROWS; ch += NTH) {
      const int row = ch % ROWS, sl = ch / ROWS;           // sl = st*4 + lg
      const int g = sl & 3, st = sl >> 2;
      const int tj = st % KS, tc = st / KS;
      const int cci = tc % kchunks, ti = tc / kchunks;
      const int nt = row >> 4, rr = row & 15;
      const int co = (rr >> 2) * (4 * NT) + nt * 4 + (rr & 3);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (co < p.Cout) v = *reinterpret_cast<const uint4*>(wg + (int64_t)co * KP + (ti * KS + tj) * p.CinP + cci * 32 + g * 8);
      *reinterpret_cast<uint4*>(sW + (int64_t)ch * 8) = v;
    }
    for (int c = threadIdx.x; c < ROWS; c += NTH) sBias[c] = c < p.Cout ? p.b[c] : 0.f;
  }
  __syncthreads();

  const int segs = (p.W + 63) / 64;                          // 64-pixel segments per row
  const int C8i = p.Cin >> 3, C8o = p.Cout >> 3;
  const unsigned img_bytes = (unsigned)p.T * (unsigned)p.W * (unsigned)p.Cin * (unsigned)sizeof(T);
  const T* wlane = sW + (lg * ROWS + li) * 8;                // + st*4*ROWS*8 + nt*16*8
  const int nwork = p.N * p.T * segs;                        // launcher: < 2^31
  // a workgroup owns a contiguous run of rows: a row's two upper tap rows were fetched by the previous pass of
  // the same workgroup (same XCD, same L2), only the newest row is a compulsory miss
  const int per = (nwork + (int)gridDim.x - 1) / (int)gridDim.x;
  const int wend = min(nwork, ((int)blockIdx.x + 1) * per);

  // A cursor walks this wave's (row segment, tap-row x channel-chunk group) sequence.  The load cursor runs
  // ahead of the MFMA cursor ACROSS row boundaries, so the first fragments of the next row are already in flight
  // while the current row's epilogue stores drain.  Everything in it is wave-uniform (kept in SGPRs).
  struct Cursor {
    int work, grp, ti, cci, t, w0;
    int row_soff;                                            // byte offset of (tap row tr, block 0) in the image
    __amdgpu_buffer_rsrc_t rsrc;                             // the image of batch entry n
    int64_t ybase;                                           // element offset of output row (n, t)
    bool live;
  };
  const int row_bytes = p.W * p.Cin * (int)sizeof(T);
  auto open_work = [&](Cursor& c) {                          // position on the first live group of c.work
    c.live = c.work < wend;
    const int wk = c.live ? c.work : 0;
    const int nt_ = wk / segs;
    const int seg = wk - nt_ * segs;
    const int n = __builtin_amdgcn_readfirstlane(nt_ / p.T);
    c.t = __builtin_amdgcn_readfirstlane(nt_ - n * p.T);
    c.w0 = __builtin_amdgcn_readfirstlane(seg * 64);
    const T* xn = reinterpret_cast<const T*>(p.x) + (int64_t)n * p.T * p.W * p.Cin;
    c.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xn), 0, (int)img_bytes, 0x00020000);
    c.ybase = ((int64_t)n * p.T + c.t) * p.W * p.Cout;
    int ti0 = 0;                                             // tap rows entirely in the causal padding: skipped
    while (ti0 < KS - 1 && c.t + p.dil * (ti0 - (KS - 1)) < 0) ++ti0;
    c.ti = ti0; c.cci = 0; c.grp = ti0 * kchunks;
    c.row_soff = (c.t + p.dil * (ti0 - (KS - 1))) * row_bytes;   // tr >= 0 for every group the cursor visits
  };
  auto advance = [&](Cursor& c) {
    ++c.grp;
    if (++c.cci == kchunks) {
      c.cci = 0; c.row_soff += p.dil * row_bytes;
      if (++c.ti == KS) { c.work += CONV_WAVES; open_work(c); }
    }
  };
  auto issue = [&](const Cursor& c, int tj, cu4 (&a)[4]) {   // request the 4 pixel fragments of step (c.grp, tj)
    const int soff = c.row_soff + c.cci * 4 * p.W * 16;      // byte offset of the (row, 4-block chunk) slab
    // dead cursor / K-padding blocks: an all-ones-ish mask OR-ed into the offset keeps it beyond num_records
    // (pure arithmetic on purpose: a boolean here gets jump-threaded into divergent load paths)
    const unsigned dead = (c.live && (c.cci * 4 + lg < C8i)) ? 0u : CONV_OOB;
    const int px = (ONESEG ? 0 : c.w0) + li;                 // this lane's pixel of N-tile 0
    const unsigned lbase = (unsigned)((lg * p.W + px) * 16) | dead;
    const int dw = p.dil * tj - p.pad_w;                     // column shift of this tap (pixels outside the row: zeros;
#pragma unroll                                               //  pixels >= W of a ragged last segment are never stored)
    for (int mt = 0; mt < 4; ++mt) {
      const bool in_row = (unsigned)(px + mt * 16 + dw) < (unsigned)p.W;
      a[mt] = __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)(in_row ? lbase + (unsigned)((mt * 16 + dw) * 16) : CONV_OOB), soff, 0);
    }
  };

  cf4 acc[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = cf4{0.f, 0.f, 0.f, 0.f};
  // one k-step = NT x 4 MFMAs.  wf0 carries the first weight fragment across steps: it is read from LDS during
  // the previous step, so a step starts its MFMAs without an exposed LDS round trip.
  auto wptr = [&](int grp, int tj) { return wlane + (grp * KS + tj) * (4 * ROWS * 8); };
  auto compute = [&](const T* wst, const T* wnext, const cu4 (&a)[4], uint4& wf0) {
    uint4 wf[NT];
    wf[0] = wf0;
#pragma unroll
    for (int nt = 1; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const uint4*>(wst + nt * 128);
    wf0 = *reinterpret_cast<const uint4*>(wnext);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = Mfma16<T>::run(wf[nt], a[mt], acc[nt][mt]);
  };
  // bias (+ ReLU), C8 store: lane = pixel (mt, li), channels lg*4*NT + nt*4 + r; then clear the accumulators
  auto epilogue = [&](const Cursor& c) {
    T* yn = reinterpret_cast<T*>(p.y) + c.ybase;
    const int cbase = lg * 4 * NT;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int wpix = (ONESEG ? 0 : c.w0) + mt * 16 + li;
      unsigned pk[2 * NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 b4 = *reinterpret_cast<const float4*>(sBias + cbase + nt * 4);
        float v0 = acc[nt][mt][0] + b4.x, v1 = acc[nt][mt][1] + b4.y, v2 = acc[nt][mt][2] + b4.z, v3 = acc[nt][mt][3] + b4.w;
        if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        pk[2 * nt] = pack2<T>(v0, v1);
        pk[2 * nt + 1] = pack2<T>(v2, v3);
        acc[nt][mt] = cf4{0.f, 0.f, 0.f, 0.f};
      }
      if (wpix < p.W) {
        if constexpr ((NT & 1) == 0) {           // whole 8-channel blocks per lane: 16-byte stores
#pragma unroll
          for (int q = 0; q < NT / 2; ++q) {
            const int blk = (cbase >> 3) + q;
            if (blk < C8o)
              *reinterpret_cast<uint4*>(yn + ((int64_t)blk * p.W + wpix) * 8) = make_uint4(pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]);
          }
        } else {                                 // odd NT: lanes own half blocks -> 8-byte stores
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int c0 = cbase + nt * 4;
            if (c0 < p.Cout)
              *reinterpret_cast<uint2*>(yn + ((int64_t)(c0 >> 3) * p.W + wpix) * 8 + (c0 & 7)) = make_uint2(pk[2 * nt], pk[2 * nt + 1]);
          }
        }
      }
    }
  };

  Cursor cc;                                                 // MFMA cursor
  cc.work = (int)blockIdx.x * per + wv;
  open_work(cc);
  if constexpr (KS == 3) {
    // three register buffers, one per column tap: a fragment is requested two steps (32 MFMAs) before its use
    Cursor cl = cc;                                          // load cursor, one group ahead after the prologue
    cu4 a0[4], a1[4], a2[4];
    issue(cl, 0, a0);  __builtin_amdgcn_sched_barrier(0);   // (program order = queue order the waits are counted in)
    issue(cl, 1, a1);  __builtin_amdgcn_sched_barrier(0);
    uint4 wf0 = *reinterpret_cast<const uint4*>(wptr(cc.grp, 0));
#pragma unroll 1
    while (cc.live) {
      // (scheduling fences: the loads must leave BEFORE the MFMA block they overlap, not sink below it)
      const T* w0p = wptr(cc.grp, 0);
      issue(cl, 2, a2);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p, w0p + 4 * ROWS * 8, a0, wf0);  __builtin_amdgcn_sched_barrier(0);
      advance(cl);
      issue(cl, 0, a0);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p + 4 * ROWS * 8, w0p + 8 * ROWS * 8, a1, wf0);  __builtin_amdgcn_sched_barrier(0);
      issue(cl, 1, a1);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p + 8 * ROWS * 8, wptr(cl.grp, 0), a2, wf0);  __builtin_amdgcn_sched_barrier(0);   // cl == next(cc)
      if (cc.ti == KS - 1 && cc.cci == kchunks - 1) epilogue(cc);   // wave-uniform
      advance(cc);
    }
  } else {
    cu4 a0[4];
#pragma unroll 1
    while (cc.live) {
#pragma unroll
      for (int tj = 0; tj < KS; ++tj) {
        issue(cc, tj, a0);
        uint4 wf0 = *reinterpret_cast<const uint4*>(wptr(cc.grp, tj));
        compute(wptr(cc.grp, tj), wptr(cc.grp, tj), a0, wf0);
      }
      if (cc.ti == KS - 1 && cc.cci == kchunks - 1) epilogue(cc);
      advance(cc);
    }
  }
}

// ChannelSplit + LayerNorm with a C8 result:
//   x (N, C, T, S*W) -> out[n, t, (c*S+i)/8, w, (c*S+i)%8] = LN(x[n, c, t, i*W:(i+1)*W])[w] * gamma[w] + beta[w]
// One workgroup per (n, t): the C*S rows are normalised by 8-lane groups, transposed through LDS and written
// as one contiguous (C*S/8) x W x 8 block.
template <typename T>
__global__ __launch_bounds__(256) void split_layernorm_c8_kernel(const T* x, T* out, const T* gamma, const T* beta, float eps,
                                                                  int C, int Tn, int S, int W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC = 8;
  T* tile = reinterpret_cast<T*>(smem);        // W x (CS + 8)
  const int CS = C * S;
  const int ldt = CS + 8;
  const int nt = blockIdx.x;
  const int n = nt / Tn, t = nt - n * Tn;
  const int lpr = W / VEC;                      // lanes that carry data per row (W % 8 == 0, W <= 512)
  int lprp = 1;                                 // lanes per row: the next power of two (xor butterflies); W = 24, 96: 3 of 4, 12 of 16
  while (lprp < lpr) lprp <<= 1;
  const int rows_per_pass = 256 / lprp;
  const int sub = threadIdx.x % lprp, rloc = threadIdx.x / lprp;
  const bool lact = sub < lpr;
  float g[VEC], b[VEC];
  unpack16<T>(lact ? *reinterpret_cast<const uint4*>(gamma + sub * VEC) : make_uint4(0, 0, 0, 0), g);
  unpack16<T>(lact ? *reinterpret_cast<const uint4*>(beta + sub * VEC) : make_uint4(0, 0, 0, 0), b);
  const float invW = 1.0f / (float)W;
  for (int r0 = 0; r0 < CS; r0 += rows_per_pass) {
    const int r = r0 + rloc;                    // output channel c*S + i
    const bool ok = r < CS && lact;
    float f[VEC];
    {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        const int c = r / S, i = r - c * S;
        v = *reinterpret_cast<const uint4*>(x + (((int64_t)n * C + c) * Tn + t) * ((int64_t)S * W) + i * W + sub * VEC);
      }
      unpack16<T>(v, f);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += f[j];
    for (int o = 1; o < lprp; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * invW;
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { const float d = lact ? f[j] - mean : 0.f; q2 += d * d; }
    for (int o = 1; o < lprp; o <<= 1) q2 += __shfl_xor(q2, o);
    const float rstd = rsqrtf(q2 * invW + eps);
    if (ok) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) tile[(sub * VEC + j) * ldt + r] = from_f<T>((f[j] - mean) * rstd * g[j] + b[j]);
    }
  }
  __syncthreads();
  T* on = out + (int64_t)nt * W * CS;
  for (int ch = threadIdx.x; ch < W * (CS / VEC); ch += 256) {     // ch = block*W + w: the C8 order, coalesced
    const int blk = ch / W, w = ch - blk * W;
    *reinterpret_cast<uint4*>(on + (int64_t)ch * VEC) = *reinterpret_cast<const uint4*>(tile + w * ldt + blk * VEC);
  }
}

}  // namespace sea

using namespace sea;

template <typename T>
static int launch_conv(const ConvParams& p, hipStream_t s) {
  const int nt = (p.Cout + 15) / 16;
  const size_t lds = (size_t)(p.KS * p.KS * (p.CinP / 32)) * 4 * (16 * nt) * 8 * sizeof(T) + (size_t)(16 * nt) * sizeof(float);
  if (lds > 160 * 1024) return SEA_EUNSUPPORTED;
  if ((int64_t)p.T * p.W * p.Cin * (int64_t)sizeof(T) >= (int64_t)(1u << 30)) return SEA_EUNSUPPORTED;   // 32-bit buffer offsets
  const int64_t nwork = (int64_t)p.N * p.T * ((p.W + 63) / 64);
  if (nwork >= (int64_t)1 << 30) return SEA_EUNSUPPORTED;
  int64_t blocks = (nwork + CONV_WAVES - 1) / CONV_WAVES;
  if (blocks > 256 * 2) blocks = 256 * 2;      // persistent: two resident workgroups per CU, weights staged once each
  dim3 grid((unsigned)blocks), block(CONV_WAVES * 64);
#define SEA_CONV(NTV, KSV)                                                                                          \
  do {                                                                                                              \
    static bool configured = false;                                                                                 \
    if (lds > 64 * 1024 && !configured) {                                                                           \
      (void)hipFuncSetAttribute((const void*)causal_conv_c8_kernel<T, NTV, KSV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
      (void)hipFuncSetAttribute((const void*)causal_conv_c8_kernel<T, NTV, KSV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      configured = true;                                                                                            \
    }                                                                                                               \
    if (p.W <= 64) hipLaunchKernelGGL((causal_conv_c8_kernel<T, NTV, KSV, true>), grid, block, lds, s, p);          \
    else hipLaunchKernelGGL((causal_conv_c8_kernel<T, NTV, KSV, false>), grid, block, lds, s, p);                   \
  } while (0)
#define SEA_CONV_NT(KSV)                                                                                            \
  switch (nt) {                                                                                                     \
    case 1: SEA_CONV(1, KSV); break; case 2: SEA_CONV(2, KSV); break; case 3: SEA_CONV(3, KSV); break;              \
    case 4: SEA_CONV(4, KSV); break; case 5: SEA_CONV(5, KSV); break; case 6: SEA_CONV(6, KSV); break;              \
    case 8: SEA_CONV(8, KSV); break;                                                                                \
    default: return SEA_EUNSUPPORTED;                                                                               \
  }
  if (p.KS == 3) { SEA_CONV_NT(3) }
  else if (p.KS == 1) { SEA_CONV_NT(1) }
  else return SEA_EUNSUPPORTED;
#undef SEA_CONV_NT
#undef SEA_CONV
  return SEA_OK;
}

extern "C" int sea_causal_conv_c8(const void* x, int dtype, int64_t N, int64_t T, int64_t W, int64_t Cin, int64_t Cout,
                                    const void* w_packed, int64_t CinP, const float* bias, int ksize, int dilation,
                                    int pad_w, int relu, void* y, sea_stream_t stream) {
  const char* nm = "sea_causal_conv_c8";
  SEA_REQUIRE(x && w_packed && bias && y, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16, SEA_EUNSUPPORTED, "%s: 16-bit data only (dtype %d)", nm, dtype);
  SEA_REQUIRE(N > 0 && T > 0 && W > 0 && Cin > 0 && Cout > 0 && ksize > 0 && dilation > 0 && pad_w >= 0, SEA_EINVAL,
              "%s: bad shape", nm);
  SEA_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && CinP % 32 == 0 && CinP >= Cin && CinP - Cin < 32 && Cout <= 128, SEA_EUNSUPPORTED,
              "%s: needs Cin %% 8 == 0, Cout %% 8 == 0, CinP = Cin rounded up to 32, Cout <= 128", nm);
  SEA_REQUIRE(W + dilation * (ksize - 1) - 2 * pad_w == W, SEA_EUNSUPPORTED, "%s: width-preserving padding only", nm);
  SEA_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed) & 15) == 0, SEA_EUNSUPPORTED, "%s: 16-byte alignment", nm);
  ConvParams p;
  p.x = x; p.w = w_packed; p.b = bias; p.y = y;
  p.N = (int)N; p.T = (int)T; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout; p.CinP = (int)CinP;
  p.KS = ksize; p.dil = dilation; p.pad_w = pad_w; p.relu = relu;
  hipStream_t s = (hipStream_t)stream;
  const int rc = dtype == SEA_BF16 ? launch_conv<__hip_bfloat16>(p, s) : launch_conv<__half>(p, s);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported channel count / kernel size (1 or 3) / image size for the LDS weight tile", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_split_layernorm_c8(const void* x, int dtype, int64_t N, int64_t C, int64_t T, int64_t S, int64_t W,
                                        const void* gamma, const void* beta, float eps, void* out, sea_stream_t stream) {
  const char* nm = "sea_split_layernorm_c8";
  SEA_REQUIRE(x && gamma && beta && out, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16, SEA_EUNSUPPORTED, "%s: 16-bit data only (dtype %d)", nm, dtype);
  SEA_REQUIRE(N > 0 && C > 0 && T > 0 && S > 0 && W > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int64_t lpr = W / 8;
  SEA_REQUIRE(W % 8 == 0 && lpr <= 64 && (C * S) % 8 == 0, SEA_EUNSUPPORTED,
              "%s: needs W a multiple of 8 (<= 512) and C*S %% 8 == 0", nm);
  const size_t lds = (size_t)W * (size_t)(C * S + 8) * 2;
  SEA_REQUIRE(lds <= 64 * 1024, SEA_EUNSUPPORTED, "%s: tile needs %zu B of LDS", nm, lds);
  SEA_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, SEA_EUNSUPPORTED,
              "%s: 16-byte alignment", nm);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(N * T)), block(256);
  if (dtype == SEA_BF16)
    hipLaunchKernelGGL((split_layernorm_c8_kernel<__hip_bfloat16>), grid, block, lds, s, (const __hip_bfloat16*)x,
                       (__hip_bfloat16*)out, (const __hip_bfloat16*)gamma, (const __hip_bfloat16*)beta, eps, (int)C, (int)T, (int)S, (int)W);
  else
    hipLaunchKernelGGL((split_layernorm_c8_kernel<__half>), grid, block, lds, s, (const __half*)x, (__half*)out,
                       (const __half*)gamma, (const __half*)beta, eps, (int)C, (int)T, (int)S, (int)W);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
