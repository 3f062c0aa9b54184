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
#include "sea_convfrag.hpp"

namespace sea {

struct ConvParams {
  const void* x;    // (N, T, Cin/8, W, 8)   C8
  const void* w;    // (Cout, taps*CinP) packed [co][tap][ci], ci padded to CinP (multiple of 32), 16-bit
  const float* b;   // (Cout) fp32
  void* y;          // (N, T, Cout/8, W, 8)  C8
  int N, T, W, Cin, Cout, CinP;
  int KS, dil, pad_w, relu;
  // ZEPI kernels: the 1x1 convolution that follows this layer's ReLU (KeepRes has no residual and the nearest x4 upsample
  // commutes with a 1x1 kernel: modules.py:42-55, attention.py:266-281), evaluated on the tile while it is in registers
  const void* w1;   // (16 * HT, Cp1) 16-bit row-major, zero padded: TailParams.w16, the pack the predictor tail uses
  const float* b1;  // (>= H1) fp32
  float* z;         // (N, T, H1, W) fp32: z = W1 . act(y) + b1, act(y) rounded to the data type as the y store rounds it
  int H1, Cp1;      // heads; Cout rounded up to 32
};

// NT = number of 16-wide output-channel tiles (Cout <= 16*NT); KS = square kernel size (compile time: the
// per-lane tap offsets live in registers); 8 waves per workgroup share one weight image in LDS.
//
// A operand = weights, B operand = pixels: in the accumulator (col = lane%16, row = 4*(lane/16) + r) a lane then
// owns ONE pixel and 4 consecutive channels per M-tile.  Weight row (nt, 4*g + r) of the LDS image is channel
// conv_chan<NT>(nt, g) + r: a PAIR of M-tiles (2q, 2q+1) gives lane group g the 8 channels q*32 + g*8 .. +8 -- one whole C8
// block (16-byte stores) and, packed to 16 bits, exactly the MFMA operand fragment of k-step q of a product over the
// channels (lane <-> pixel, lane group <-> 8 consecutive K): the z epilogue below feeds the packed registers straight back
// into the matrix pipe.  An odd last tile holds channels (NT-1)*16 + g*4 .. +4 (8-byte stores).
// LDS weight image: [k-step][lg][row][8 elements]; a lane's 16 bytes sit at bank 4*li for every lg, which is
// conflict-free for ds_read_b128's lane groups without any padding.
// k-steps run (tap row, 32-channel chunk, tap column); the three column shifts of one chunk re-read the same
// lines back to back.  Pixel fragments come through buffer loads: a lane whose tap falls outside the row (or whose
// channel block is K padding) carries an offset beyond num_records and the hardware returns zeros, so the k-loop
// is branch-free and the next step's loads are in flight under the current step's MFMAs.
#ifndef SEA_CONV_FIXED_OFFSETS
#define SEA_CONV_FIXED_OFFSETS 1
#endif
constexpr int CONV_WAVES = 6;                     // waves per workgroup of the big launches (NW below: 6 or 8)
constexpr unsigned CONV_OOB = 0x7FFFFF00u;       // > any valid byte offset (launcher checks the image is < 1 GiB)

// ONESEG: W <= 64, every row is one segment: the tap-validity tests and lane offsets are then kernel invariants.
// ZEPI: also (or only, y = null) write z = W1 . relu(conv + b) + b1, the 1x1 convolution of the predictor tail, (N, T, H1, W)
// fp32.  Per 16-pixel tile that is HT x ceil(NT/2) MFMAs on operands that are already in registers (+5.5 % at 64 -> 64
// channels, 32 heads) and it takes the "z tile" phase -- a third of a row's life -- out of the issue-bound tail + selection
// kernel.  Same operand placement, k order and rounding point as tail_z_tile (sea_tail.hpp): the same bits.
// NW: waves per workgroup.  6 for the big launches (two workgroups of 6 per CU at 64 -> 64 channels); 8 for launches of few
// rows and for narrow layers (round 5, same-box A/B `scripts/ab_conv_small.py`: one sequence of LLaMA-13B 62 -> 41 us together
// with one workgroup per CU, OPT-2.7B 51 -> 48, OPT-125m x 8 33.5 -> 30.4; OPT-1.3B x 8 keeps 6: 180 vs 181 - 187 us).
template <typename T, int NT, int KS, bool ONESEG, bool ZEPI = false, int NW = CONV_WAVES>
__global__ __launch_bounds__(NW * 64, (NT <= 4 ? 3 : 2)) void causal_conv_c8_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTH = NW * 64;
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
      const int co = conv_chan<NT>(nt, rr >> 2) + (rr & 3);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (co < p.Cout) v = *reinterpret_cast<const uint4*>(wg + (int64_t)co * KP + (ti * KS + tj) * p.CinP + cci * 32 + g * 8);
      *reinterpret_cast<uint4*>(sW + (int64_t)ch * 8) = v;
    }
    for (int c = threadIdx.x; c < ROWS; c += NTH) sBias[c] = c < p.Cout ? p.b[c] : 0.f;
  }
  // ZEPI: B fragments of the 1x1 weights, [k-step q][head tile][lane] x 16 bytes (lane <-> head li, channels q*32 + lg*8 .. +8)
  constexpr int KC1 = (NT + 1) / 2;
  const int HT1 = ZEPI ? (p.H1 + 15) / 16 : 0;
  T* sW1 = reinterpret_cast<T*>(sBias + ROWS);
  float* sB1 = reinterpret_cast<float*>(sW1 + (size_t)KC1 * HT1 * 64 * 8);
  if constexpr (ZEPI) {
    const T* w1 = reinterpret_cast<const T*>(p.w1);
    for (int ch = threadIdx.x; ch < KC1 * HT1 * 64; ch += NTH) {
      const int l = ch & 63, qh = ch >> 6;
      const int ht = qh % HT1, q = qh / HT1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q * 32 + (l >> 4) * 8 < p.Cp1) v = *reinterpret_cast<const uint4*>(w1 + (int64_t)(ht * 16 + (l & 15)) * p.Cp1 + q * 32 + (l >> 4) * 8);
      *reinterpret_cast<uint4*>(sW1 + (int64_t)ch * 8) = v;
    }
    for (int h = threadIdx.x; h < HT1 * 16; h += NTH) sB1[h] = h < p.H1 ? p.b1[h] : 0.f;   // (a global read in the epilogue stalls it)
  }
  __syncthreads();

#ifdef SEA_CONV_STAGGER
  {  // experiment: the three waves of a SIMD start a third of a row apart, so that their row boundaries (epilogue, stores, the
     // newest row's compulsory misses) do not coincide.  Placement model: waves round-robin over the SIMDs, second workgroup of
     // a CU continues where the first stopped.
    const bool first = blockIdx.x < 256;
    const int ph = NW == 6 ? (first ? wv / 4 : (wv < 2 ? 1 : 2)) : (wv / 4);
    for (int i = 0; i < ph * SEA_CONV_STAGGER; ++i) __builtin_amdgcn_s_sleep(100);
  }
#endif
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
    int64_t zbase;                                           // ZEPI: element offset of row (n, t) in z (N, T, H1, W)
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
    if constexpr (ZEPI) c.zbase = ((int64_t)n * p.T + c.t) * p.W * p.H1;
    int ti0 = 0;                                             // tap rows entirely in the causal padding: skipped
    while (ti0 < KS - 1 && c.t + p.dil * (ti0 - (KS - 1)) < 0) ++ti0;
    c.ti = ti0; c.cci = 0; c.grp = ti0 * kchunks;
    c.row_soff = (c.t + p.dil * (ti0 - (KS - 1))) * row_bytes;   // tr >= 0 for every group the cursor visits
  };
  auto advance = [&](Cursor& c) {
    ++c.grp;
    if (++c.cci == kchunks) {
      c.cci = 0; c.row_soff += p.dil * row_bytes;
      if (++c.ti == KS) { c.work += NW; open_work(c); }
    }
  };
#if SEA_CONV_FIXED_OFFSETS
  // ONESEG, 3 x 3: the per-lane byte offsets of the 12 (tap column, pixel tile) fragments are kernel invariants -- a tap that
  // leaves the row carries an offset beyond num_records -- and so is the K-padding mask of the last channel chunk.  (Computed
  // per request they were ~10 vector instructions in front of every 4 loads: 23 % of the waves' cycles were vector issue in a
  // kernel whose vector work is its epilogue.)  A cursor past the end re-reads work item 0 (valid memory, never used).
  unsigned voff[3][4];
  const unsigned lastdead = ((kchunks - 1) * 4 + lg < C8i) ? 0u : CONV_OOB;
  if constexpr (ONESEG && KS == 3) {
#pragma unroll
    for (int tj = 0; tj < 3; ++tj)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int dw = p.dil * tj - p.pad_w;
        const bool in_row = (unsigned)(li + mt * 16 + dw) < (unsigned)p.W;
        voff[tj][mt] = in_row ? (unsigned)((lg * p.W + li + mt * 16 + dw) * 16) : CONV_OOB;
      }
  }
#endif
  auto issue = [&](const Cursor& c, int tj, cu4 (&a)[4]) {   // request the 4 pixel fragments of step (c.grp, tj)
    const int soff = c.row_soff + c.cci * 4 * p.W * 16;      // byte offset of the (row, 4-block chunk) slab
#if SEA_CONV_FIXED_OFFSETS
    if constexpr (ONESEG && KS == 3) {
      const unsigned m = (c.cci == kchunks - 1) ? lastdead : 0u;       // (the compare is scalar: one v_or per request)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        a[mt] = __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)(voff[tj][mt] | m), soff, 0);
      return;
    }
#endif
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
  // bias (+ ReLU), C8 store: lane = pixel (mt, li), channels conv_chan<NT>(nt, lg) + r; then clear the accumulators
  auto epilogue = [&](const Cursor& c) {
    T* yn = reinterpret_cast<T*>(p.y) + c.ybase;
    const bool want_y = !ZEPI || p.y != nullptr;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int wpix = (ONESEG ? 0 : c.w0) + mt * 16 + li;
      unsigned pk[2 * NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 b4 = *reinterpret_cast<const float4*>(sBias + conv_chan<NT>(nt, lg));
        float v0 = acc[nt][mt][0] + b4.x, v1 = acc[nt][mt][1] + b4.y, v2 = acc[nt][mt][2] + b4.z, v3 = acc[nt][mt][3] + b4.w;
        if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        pk[2 * nt] = pack2<T>(v0, v1);
        pk[2 * nt + 1] = pack2<T>(v2, v3);
        acc[nt][mt] = cf4{0.f, 0.f, 0.f, 0.f};
      }
#ifdef SEA_CONV_EXP_NOSTORE
      if (want_y && wpix < p.W && pk[0] == 0x12345678u) {
#else
      if (want_y && wpix < p.W) {
#endif
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {         // tile pairs: whole 8-channel blocks per lane, 16-byte stores
          const int blk = q * 4 + lg;
          if (blk < C8o)
            *reinterpret_cast<uint4*>(yn + ((int64_t)blk * p.W + wpix) * 8) = make_uint4(pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]);
        }
        if constexpr (NT & 1) {                    // odd last tile: half blocks, 8-byte stores
          const int c0 = conv_chan<NT>(NT - 1, lg);
          if (c0 < p.Cout)
            *reinterpret_cast<uint2*>(yn + ((int64_t)(c0 >> 3) * p.W + wpix) * 8 + (c0 & 7)) = make_uint2(pk[2 * NT - 2], pk[2 * NT - 1]);
        }
      }
      if constexpr (ZEPI) {
        // z[h][pixel] = sum_c W1[h][c] * y[c][pixel] + b1[h]: A = this tile's packed activations (lane <-> pixel li, 8 channels of
        // k-step q), B = W1 (lane <-> head li); D: col = li -> head, row = lg*4 + r -> pixel mt*16 + lg*4 + r
        cu4 af[KC1];
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) af[q] = cu4{pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]};
        if constexpr (NT & 1) {
          // the odd tile spreads its 16 channels over the four lane groups (4 each); k-step KC1-1 wants 8 per group in groups
          // 0 and 1 (channels (NT-1)*16 + lg*8 .. +8, the rest of the 32 is padding): fetch them from groups 2lg, 2lg + 1
          const int s0 = li + 32 * (lg & 1), s1 = s0 + 16;
          const unsigned a0_ = __shfl(pk[2 * NT - 2], s0), a1_ = __shfl(pk[2 * NT - 1], s0);
          const unsigned a2_ = __shfl(pk[2 * NT - 2], s1), a3_ = __shfl(pk[2 * NT - 1], s1);
          af[KC1 - 1] = lg < 2 ? cu4{a0_, a1_, a2_, a3_} : cu4{0u, 0u, 0u, 0u};
        }
        const int px0 = (ONESEG ? 0 : c.w0) + mt * 16 + lg * 4;
        float* zrow = p.z + c.zbase + px0;
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {                     // H1 <= 64 (launcher)
          if (ht < HT1) {                                    // wave-uniform
            cf4 zc = cf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KC1; ++q)
              zc = Mfma16<T>::run(__builtin_bit_cast(uint4, af[q]), __builtin_bit_cast(cu4, *reinterpret_cast<const uint4*>(sW1 + ((int64_t)(q * HT1 + ht) * 64 + lane) * 8)), zc);
            const int h = ht * 16 + li;
            const float bz = sB1[h];
            if (h < p.H1 && px0 < p.W)
              *reinterpret_cast<float4*>(zrow + (int64_t)h * p.W) = make_float4(zc[0] + bz, zc[1] + bz, zc[2] + bz, zc[3] + bz);
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
#ifdef SEA_CONV_EARLY_A2
    issue(cl, 2, a2);  __builtin_amdgcn_sched_barrier(0);
#endif
    uint4 wf0 = *reinterpret_cast<const uint4*>(wptr(cc.grp, 0));
#pragma unroll 1
    while (cc.live) {
      // (scheduling fences: the loads must leave BEFORE the MFMA block they overlap, not sink below it)
      const T* w0p = wptr(cc.grp, 0);
#if defined(SEA_CONV_EXP_ONETAP)     // timing experiment (wrong results): one load per chunk instead of three column taps
      for (int i = 0; i < 4; ++i) { a1[i] = a0[i]; a2[i] = a0[i]; }
      compute(w0p, w0p + 4 * ROWS * 8, a0, wf0);  __builtin_amdgcn_sched_barrier(0);
      advance(cl);
      issue(cl, 0, a0);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p + 4 * ROWS * 8, w0p + 8 * ROWS * 8, a1, wf0);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p + 8 * ROWS * 8, wptr(cl.grp, 0), a2, wf0);  __builtin_amdgcn_sched_barrier(0);
#elif defined(SEA_CONV_EXP_NOLOAD)   // timing experiment (wrong results): no pixel loads at all
      compute(w0p, w0p + 4 * ROWS * 8, a0, wf0);  __builtin_amdgcn_sched_barrier(0);
      advance(cl);
      compute(w0p + 4 * ROWS * 8, w0p + 8 * ROWS * 8, a1, wf0);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p + 8 * ROWS * 8, wptr(cl.grp, 0), a2, wf0);  __builtin_amdgcn_sched_barrier(0);
#elif defined(SEA_CONV_EARLY_A2)     // the third tap's fragments leave at the END of the previous step, i.e. before a row's epilogue stores
      compute(w0p, w0p + 4 * ROWS * 8, a0, wf0);  __builtin_amdgcn_sched_barrier(0);
      advance(cl);
      issue(cl, 0, a0);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p + 4 * ROWS * 8, w0p + 8 * ROWS * 8, a1, wf0);  __builtin_amdgcn_sched_barrier(0);
      issue(cl, 1, a1);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p + 8 * ROWS * 8, wptr(cl.grp, 0), a2, wf0);  __builtin_amdgcn_sched_barrier(0);
      issue(cl, 2, a2);  __builtin_amdgcn_sched_barrier(0);
#else
      issue(cl, 2, a2);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p, w0p + 4 * ROWS * 8, a0, wf0);  __builtin_amdgcn_sched_barrier(0);
      advance(cl);
      issue(cl, 0, a0);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p + 4 * ROWS * 8, w0p + 8 * ROWS * 8, a1, wf0);  __builtin_amdgcn_sched_barrier(0);
      issue(cl, 1, a1);  __builtin_amdgcn_sched_barrier(0);
      compute(w0p + 8 * ROWS * 8, wptr(cl.grp, 0), a2, wf0);  __builtin_amdgcn_sched_barrier(0);   // cl == next(cc)
#endif
#ifndef SEA_CONV_EXP_NOEPI
      if (cc.ti == KS - 1 && cc.cci == kchunks - 1) epilogue(cc);   // wave-uniform
#endif
      advance(cc);
    }
#ifdef SEA_CONV_EXP_NOEPI     // timing experiment: keep the accumulators alive without an epilogue
    float sink = 0.f;
    for (int nt = 0; nt < NT; ++nt) for (int mt = 0; mt < 4; ++mt) for (int r = 0; r < 4; ++r) sink += acc[nt][mt][r];
    if (sink == 12345.678f) reinterpret_cast<float*>(p.y)[threadIdx.x] = sink;
#endif
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

// ---- fp32 DATA (round 5): the same implicit GEMM on v_mfma_f32_16x16x4_f32 -------------------------------------------------
// The reference's measurement protocol is fp32 (src/main/benchmark_bert.py:196-239) and until now fp32 tensors took the
// framework's dilated convolutions (MIOpen: 2 x 0.63 ms at BASELINE config 2, a quarter of that step each).  Same layout idea
// (C8: blocks of 8 channels, here 32 bytes per pixel), same work split (one wave = 64 consecutive pixels of one row x all
// C_out), exact fp32 products and accumulation.  The fp32 MFMA takes ONE K value per lane (k = lane / 16): a lane's 16-byte
// load holds 4 consecutive channels c0 .. c0+3 of its pixel, c0 = 16 b + 4 g, and feeds 4 MFMAs -- in MFMA j the k-slot of
// lane group g is channel 16 b + 4 g + j, and the weights' LDS image is packed the same way ([step][tile][lane] x 4 floats:
// W[16 nt + li][16 b + 4 g + j], one ds_read_b128 per (step, tile)).  Accumulator: col = li -> pixel, row = 4 g + r -> 4
// consecutive output channels: a 16-byte store into the C8 result.
struct ConvF32Params {
  const float* x;   // (N, T, Cin/8, W, 8) fp32
  const float* w;   // (Cout, KS*KS, CinP) fp32, ci zero-padded to CinP (multiple of 16)
  const float* b;   // (Cout)
  float* y;         // (N, T, Cout/8, W, 8) fp32
  int N, T, W, Cin, Cout, CinP;
  int KS, dil, pad_w, relu;
};
constexpr int CONVF_WAVES = 4;

template <int NT, int KS>
__global__ __launch_bounds__(CONVF_WAVES * 64) void causal_conv_c8f_kernel(ConvF32Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTH = CONVF_WAVES * 64;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int kch = p.CinP / 16;
  const int nsteps = KS * KS * kch;                     // ordered (ti, tj, b)
  float* sW = reinterpret_cast<float*>(smem);           // nsteps x NT x 64 lanes x 4
  float* sBias = sW + (size_t)nsteps * NT * 256;        // 16 NT
  for (int ch = threadIdx.x; ch < nsteps * NT * 64; ch += NTH) {
    const int l = ch & 63, sn = ch >> 6;
    const int nt = sn % NT, st = sn / NT;
    const int b = st % kch, tap = st / kch;
    const int co = nt * 16 + (l & 15), ci = 16 * b + 4 * (l >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (co < p.Cout) v = *reinterpret_cast<const float4*>(p.w + ((int64_t)co * KS * KS + tap) * p.CinP + ci);   // (zero padded past Cin)
    *reinterpret_cast<float4*>(sW + (int64_t)ch * 4) = v;
  }
  for (int c = threadIdx.x; c < 16 * NT; c += NTH) sBias[c] = c < p.Cout ? p.b[c] : 0.f;
  __syncthreads();

  const int segs = (p.W + 63) / 64;
  const int C8i = p.Cin >> 3, C8o = p.Cout >> 3;
  const unsigned img_bytes = (unsigned)p.T * (unsigned)p.W * (unsigned)p.Cin * 4u;
  const int row_bytes = p.W * p.Cin * 4;
  const int nwork = p.N * p.T * segs;
  const int per = (nwork + (int)gridDim.x - 1) / (int)gridDim.x;     // contiguous rows per workgroup: the upper tap rows are L2 hits
  const int wend = min(nwork, ((int)blockIdx.x + 1) * per);
  const float* wl = sW + lane * 4;                                   // + (st * NT + nt) * 256

  for (int work = (int)blockIdx.x * per + wv; work < wend; work += CONVF_WAVES) {
    const int nt_ = work / segs, seg = work - nt_ * segs;
    const int n = nt_ / p.T, t = nt_ - n * p.T;
    const int w0 = seg * 64;
    const float* xn = p.x + (int64_t)n * p.T * p.W * p.Cin;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, (int)img_bytes, 0x00020000);
    cf4 acc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = cf4{0.f, 0.f, 0.f, 0.f};
    int ti0 = 0;                                                     // tap rows entirely in the causal padding: skipped
    while (ti0 < KS - 1 && t + p.dil * (ti0 - (KS - 1)) < 0) ++ti0;
    const int s_begin = ti0 * KS * kch, s_end = nsteps;
    // fragments of step s: 4 pixel tiles x 16 bytes (channels 16 b + 4 lg .. +4 of pixel w0 + 16 mt + li + shift)
    auto issue = [&](int s, cf4 (&f)[4]) {
      const int b = s % kch, tap = s / kch;
      const int tj = tap % KS, ti = tap / KS;
      const int tr = t + p.dil * (ti - (KS - 1));                    // >= 0 for every step visited
      const int c0 = 16 * b + 4 * lg;
      const int dw = p.dil * tj - p.pad_w;
      const unsigned base = (unsigned)(tr * row_bytes) + (unsigned)(((c0 >> 3) * p.W) * 32 + (c0 & 7) * 4);
      const bool cok = (c0 >> 3) < C8i && s < s_end;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int px = w0 + mt * 16 + li + dw;
        const bool ok = cok && (unsigned)px < (unsigned)p.W;
        f[mt] = __builtin_bit_cast(cf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok ? base + (unsigned)(px * 32) : CONV_OOB), 0, 0));
      }
    };
    auto compute = [&](int s, const cf4 (&f)[4]) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const cf4 wf = *reinterpret_cast<const cf4*>(wl + (int64_t)(s * NT + nt) * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], f[mt][j], acc[nt][mt], 0, 0, 0);
      }
    };
    cf4 fa[4], fb[4];
    issue(s_begin, fa);
    for (int s = s_begin; s < s_end; s += 2) {                       // two register sets: the next step's loads fly under this step's MFMAs
      issue(s + 1, fb);
      compute(s, fa);
      if (s + 1 < s_end) {
        issue(s + 2, fa);
        compute(s + 1, fb);
      }
    }
    float* yn = p.y + ((int64_t)n * p.T + t) * p.W * p.Cout;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int wpix = w0 + mt * 16 + li;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c0 = nt * 16 + 4 * lg;
        const float4 b4 = *reinterpret_cast<const float4*>(sBias + c0);
        float v0 = acc[nt][mt][0] + b4.x, v1 = acc[nt][mt][1] + b4.y, v2 = acc[nt][mt][2] + b4.z, v3 = acc[nt][mt][3] + b4.w;
        if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        if (wpix < p.W && (c0 >> 3) < C8o)
          *reinterpret_cast<float4*>(yn + ((int64_t)(c0 >> 3) * p.W + wpix) * 8 + (c0 & 7)) = make_float4(v0, v1, v2, v3);
      }
    }
  }
}

// ChannelSplit + LayerNorm with a C8 result:
//   x (N, C, T, S*W) -> out[n, t, (c*S+i)/8, w, (c*S+i)%8] = LN(x[n, c, t, i*W:(i+1)*W])[w] * gamma[w] + beta[w]
// One workgroup per (n, t): the C*S rows are normalised by 8-lane groups, transposed through LDS and written
// as one contiguous (C*S/8) x W x 8 block.
// 8 consecutive elements of a row <-> 8 floats (16-bit: one 16-byte vector; fp32: two)
template <typename T> __device__ inline void load8(const T* p, float* f) {
  if constexpr (sizeof(T) == 4) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    unpack16<T>(*reinterpret_cast<const uint4*>(p), f);
  }
}

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
#pragma unroll
  for (int j = 0; j < VEC; ++j) { g[j] = 0.f; b[j] = 0.f; }
  if (lact) { load8<T>(gamma + sub * VEC, g); load8<T>(beta + sub * VEC, b); }
  const float invW = 1.0f / (float)W;
  for (int r0 = 0; r0 < CS; r0 += rows_per_pass) {
    const int r = r0 + rloc;                    // output channel c*S + i
    const bool ok = r < CS && lact;
    float f[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) f[j] = 0.f;
    if (ok) {
      const int c = r / S, i = r - c * S;
      load8<T>(x + (((int64_t)n * C + c) * Tn + t) * ((int64_t)S * W) + i * W + sub * VEC, f);
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
  constexpr int PER = (int)(VEC * sizeof(T) / 16);                 // 16-byte vectors per (block, pixel): 1 (16-bit) or 2 (fp32)
  for (int ch = threadIdx.x; ch < W * (CS / VEC) * PER; ch += 256) {     // ch / PER = block*W + w: the C8 order, coalesced
    const int bw = ch / PER, half = ch - bw * PER;
    const int blk = bw / W, w = bw - blk * W;
    *(reinterpret_cast<uint4*>(on + (int64_t)bw * VEC) + half) = *(reinterpret_cast<const uint4*>(tile + w * ldt + blk * VEC) + half);
  }
}

}  // namespace sea

using namespace sea;

template <typename T>
static int launch_conv(const ConvParams& p, hipStream_t s, bool zepi = false) {
  const int nt = (p.Cout + 15) / 16;
  size_t lds = (size_t)(p.KS * p.KS * (p.CinP / 32)) * 4 * (16 * nt) * 8 * sizeof(T) + (size_t)(16 * nt) * sizeof(float);
  if (zepi) lds += (size_t)((nt + 1) / 2) * ((p.H1 + 15) / 16) * 64 * 16 + (size_t)((p.H1 + 15) / 16) * 16 * sizeof(float);   // the 1x1 weights' operand fragments + biases
  if (lds > 160 * 1024) return SEA_EUNSUPPORTED;
  if ((int64_t)p.T * p.W * p.Cin * (int64_t)sizeof(T) >= (int64_t)(1u << 30)) return SEA_EUNSUPPORTED;   // 32-bit buffer offsets
  const int64_t nwork = (int64_t)p.N * p.T * ((p.W + 63) / 64);
  if (nwork >= (int64_t)1 << 30) return SEA_EUNSUPPORTED;
  // 8-wave workgroups for narrow layers and for launches of few rows (measured, see the kernel's comment)
  const bool w8 = nt <= 3 || nt == 5 || nwork <= 16384;
  const int nw = w8 ? 8 : CONV_WAVES;
  int64_t blocks = (nwork + nw - 1) / nw;
  if (blocks > 256 * 2) blocks = 256 * 2;      // persistent: two resident workgroups per CU, weights staged once each
  if (lds > 80 * 1024 && blocks > 256) blocks = 256;     // an image this large leaves room for ONE workgroup per CU: one staging round
  dim3 grid((unsigned)blocks), block(nw * 64);
#define SEA_CONV_L(NTV, KSV, ZV, NWV)                                                                               \
  do {                                                                                                              \
    static DevOnce once;                                                                                            \
    if (lds > 64 * 1024 && once.first()) {                                                                          \
      SEA_MAX_LDS((causal_conv_c8_kernel<T, NTV, KSV, true, ZV, NWV>), 160 * 1024);                                 \
      SEA_MAX_LDS((causal_conv_c8_kernel<T, NTV, KSV, false, ZV, NWV>), 160 * 1024);                                \
    }                                                                                                               \
    if (p.W <= 64) hipLaunchKernelGGL((causal_conv_c8_kernel<T, NTV, KSV, true, ZV, NWV>), grid, block, lds, s, p); \
    else hipLaunchKernelGGL((causal_conv_c8_kernel<T, NTV, KSV, false, ZV, NWV>), grid, block, lds, s, p);          \
  } while (0)
#define SEA_CONV_K(NTV, KSV, ZV)                                                                                    \
  do {                                                                                                              \
    if (w8) SEA_CONV_L(NTV, KSV, ZV, 8);                                                                            \
    else SEA_CONV_L(NTV, KSV, ZV, CONV_WAVES);                                                                      \
  } while (0)
#define SEA_CONV(NTV, KSV) SEA_CONV_K(NTV, KSV, false)
#define SEA_CONV_NT(KSV)                                                                                            \
  switch (nt) {                                                                                                     \
    case 1: SEA_CONV(1, KSV); break; case 2: SEA_CONV(2, KSV); break; case 3: SEA_CONV(3, KSV); break;              \
    case 4: SEA_CONV(4, KSV); break; case 5: SEA_CONV(5, KSV); break; case 6: SEA_CONV(6, KSV); break;              \
    case 7: SEA_CONV(7, KSV); break; case 8: SEA_CONV(8, KSV); break;                                               \
    default: return SEA_EUNSUPPORTED;                                                                               \
  }
  if (zepi) {
    if (p.KS != 3) return SEA_EUNSUPPORTED;
    switch (nt) {
      case 1: SEA_CONV_K(1, 3, true); break; case 2: SEA_CONV_K(2, 3, true); break; case 3: SEA_CONV_K(3, 3, true); break;
      case 4: SEA_CONV_K(4, 3, true); break; case 5: SEA_CONV_K(5, 3, true); break;
      default: return SEA_EUNSUPPORTED;
    }
  } else if (p.KS == 3) { SEA_CONV_NT(3) }
  else if (p.KS == 1) { SEA_CONV_NT(1) }
  else return SEA_EUNSUPPORTED;
#undef SEA_CONV_NT
#undef SEA_CONV
#undef SEA_CONV_K
#undef SEA_CONV_L
  return SEA_OK;
}

static int conv_common(const char* nm, const void* x, int dtype, int64_t N, int64_t T, int64_t W, int64_t Cin, int64_t Cout,
                       const void* w_packed, int64_t CinP, const float* bias, int ksize, int dilation, int pad_w, int relu, void* y,
                       const void* w1, const float* b1, int64_t H1, int64_t Cp1, float* z, bool zepi, sea_stream_t stream) {
  SEA_REQUIRE(x && w_packed && bias && (y || zepi), SEA_EINVAL, "%s: null pointer", nm);
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
  p.w1 = nullptr; p.b1 = nullptr; p.z = nullptr; p.H1 = 0; p.Cp1 = 0;
  if (zepi) {
    SEA_REQUIRE(w1 && b1 && z, SEA_EINVAL, "%s: null pointer", nm);
    SEA_REQUIRE(H1 > 0 && H1 <= 64 && Cp1 == (Cout + 31) / 32 * 32 && W % 4 == 0 && ksize == 3 && Cout <= 80, SEA_EUNSUPPORTED,
                "%s: needs H <= 64, Cp1 = Cout rounded up to 32, W %% 4 == 0, a 3 x 3 kernel, Cout <= 80", nm);
    SEA_REQUIRE((((uintptr_t)w1 | (uintptr_t)z) & 15) == 0, SEA_EUNSUPPORTED, "%s: 16-byte alignment", nm);
    p.w1 = w1; p.b1 = b1; p.z = z; p.H1 = (int)H1; p.Cp1 = (int)Cp1;
  }
  hipStream_t s = (hipStream_t)stream;
  const int rc = dtype == SEA_BF16 ? launch_conv<__hip_bfloat16>(p, s, zepi) : launch_conv<__half>(p, s, zepi);
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported channel count / kernel size (1 or 3) / image size for the LDS weight tile", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_causal_conv_c8(const void* x, int dtype, int64_t N, int64_t T, int64_t W, int64_t Cin, int64_t Cout,
                                    const void* w_packed, int64_t CinP, const float* bias, int ksize, int dilation,
                                    int pad_w, int relu, void* y, sea_stream_t stream) {
  return conv_common("sea_causal_conv_c8", x, dtype, N, T, W, Cin, Cout, w_packed, CinP, bias, ksize, dilation, pad_w, relu, y,
                     nullptr, nullptr, 0, 0, nullptr, false, stream);
}

// The last (conv, ReLU) pair of the predictor CNN with the 1x1 convolution of the tail in its epilogue: writes
// z (N, T, H, W) fp32 = W1 . relu(conv(x) + bias) + b1 and, when y is not null, the activation itself as sea_causal_conv_c8.
extern "C" int sea_causal_conv_c8_z(const void* x, int dtype, int64_t N, int64_t T, int64_t W, int64_t Cin, int64_t Cout,
                                      const void* w_packed, int64_t CinP, const float* bias, int ksize, int dilation,
                                      int pad_w, int relu, void* y, const void* conv1x1_w16, int64_t Cp1, const float* conv1x1_b,
                                      int64_t H, float* z, sea_stream_t stream) {
  return conv_common("sea_causal_conv_c8_z", x, dtype, N, T, W, Cin, Cout, w_packed, CinP, bias, ksize, dilation, pad_w, relu, y,
                     conv1x1_w16, conv1x1_b, H, Cp1, z, true, stream);
}

template <int KS>
static int launch_conv_f32(const ConvF32Params& p, hipStream_t s) {
  const int nt = (p.Cout + 15) / 16;
  const size_t lds = (size_t)(KS * KS * (p.CinP / 16)) * nt * 256 * sizeof(float) + (size_t)(16 * nt) * sizeof(float);
  if (lds > 160 * 1024) return SEA_EUNSUPPORTED;
  if ((int64_t)p.T * p.W * p.Cin * 4 >= (int64_t)(1u << 30)) return SEA_EUNSUPPORTED;   // 32-bit buffer offsets
  const int64_t nwork = (int64_t)p.N * p.T * ((p.W + 63) / 64);
  if (nwork >= (int64_t)1 << 30) return SEA_EUNSUPPORTED;
  const int per_cu = lds > 80 * 1024 ? 1 : lds > 40 * 1024 ? 2 : 4;                    // resident workgroups per CU (by LDS)
  int64_t blocks = (nwork + CONVF_WAVES - 1) / CONVF_WAVES;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;                                     // persistent: weights staged once each
  dim3 grid((unsigned)blocks), block(CONVF_WAVES * 64);
#define SEA_CONVF(NTV)                                                                                  \
  do {                                                                                                  \
    static DevOnce once;                                                                                \
    if (lds > 64 * 1024 && once.first()) SEA_MAX_LDS((causal_conv_c8f_kernel<NTV, KS>), 160 * 1024);    \
    hipLaunchKernelGGL((causal_conv_c8f_kernel<NTV, KS>), grid, block, lds, s, p);                      \
  } while (0)
  switch (nt) {
    case 1: SEA_CONVF(1); break; case 2: SEA_CONVF(2); break; case 3: SEA_CONVF(3); break; case 4: SEA_CONVF(4); break;
    case 5: SEA_CONVF(5); break;
    default: return SEA_EUNSUPPORTED;
  }
#undef SEA_CONVF
  return SEA_OK;
}

// fp32 twin of sea_causal_conv_c8 (exact fp32 products on the fp32 MFMA): x / y in the same C8 layout with 4-byte elements,
// w_packed (Cout, ksize*ksize, CinP) fp32 with CinP = Cin rounded up to 16.
extern "C" int sea_causal_conv_c8_f32(const float* x, int64_t N, int64_t T, int64_t W, int64_t Cin, int64_t Cout,
                                        const float* w_packed, int64_t CinP, const float* bias, int ksize, int dilation,
                                        int pad_w, int relu, float* y, sea_stream_t stream) {
  const char* nm = "sea_causal_conv_c8_f32";
  SEA_REQUIRE(x && w_packed && bias && y, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(N > 0 && T > 0 && W > 0 && Cin > 0 && Cout > 0 && ksize > 0 && dilation > 0 && pad_w >= 0, SEA_EINVAL, "%s: bad shape", nm);
  SEA_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && CinP % 16 == 0 && CinP >= Cin && CinP - Cin < 16 && Cout <= 80, SEA_EUNSUPPORTED,
              "%s: needs Cin %% 8 == 0, Cout %% 8 == 0, CinP = Cin rounded up to 16, Cout <= 80", nm);
  SEA_REQUIRE(W + dilation * (ksize - 1) - 2 * pad_w == W, SEA_EUNSUPPORTED, "%s: width-preserving padding only", nm);
  SEA_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed) & 15) == 0, SEA_EUNSUPPORTED, "%s: 16-byte alignment", nm);
  ConvF32Params p;
  p.x = x; p.w = w_packed; p.b = bias; p.y = y;
  p.N = (int)N; p.T = (int)T; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout; p.CinP = (int)CinP;
  p.KS = ksize; p.dil = dilation; p.pad_w = pad_w; p.relu = relu;
  hipStream_t s = (hipStream_t)stream;
  const int rc = ksize == 3 ? launch_conv_f32<3>(p, s) : ksize == 1 ? launch_conv_f32<1>(p, s) : SEA_EUNSUPPORTED;
  SEA_REQUIRE(rc == SEA_OK, rc, "%s: unsupported channel count / kernel size (1 or 3) / the fp32 weight image exceeds the LDS", nm);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}

extern "C" int sea_split_layernorm_c8(const void* x, int dtype, int64_t N, int64_t C, int64_t T, int64_t S, int64_t W,
                                        const void* gamma, const void* beta, float eps, void* out, sea_stream_t stream) {
  const char* nm = "sea_split_layernorm_c8";
  SEA_REQUIRE(x && gamma && beta && out, SEA_EINVAL, "%s: null pointer", nm);
  SEA_REQUIRE(dtype == SEA_F16 || dtype == SEA_BF16 || dtype == SEA_F32, SEA_EINVAL, "%s: bad dtype %d", nm, dtype);
  SEA_REQUIRE(N > 0 && C > 0 && T > 0 && S > 0 && W > 0, SEA_EINVAL, "%s: bad shape", nm);
  const int64_t lpr = W / 8;
  SEA_REQUIRE(W % 8 == 0 && lpr <= 64 && (C * S) % 8 == 0, SEA_EUNSUPPORTED,
              "%s: needs W a multiple of 8 (<= 512) and C*S %% 8 == 0", nm);
  const size_t lds = (size_t)W * (size_t)(C * S + 8) * (dtype == SEA_F32 ? 4 : 2);
  SEA_REQUIRE(lds <= 64 * 1024, SEA_EUNSUPPORTED, "%s: tile needs %zu B of LDS", nm, lds);
  SEA_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, SEA_EUNSUPPORTED,
              "%s: 16-byte alignment", nm);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(N * T)), block(256);
  if (dtype == SEA_F32)
    hipLaunchKernelGGL((split_layernorm_c8_kernel<float>), grid, block, lds, s, (const float*)x, (float*)out, (const float*)gamma,
                       (const float*)beta, eps, (int)C, (int)T, (int)S, (int)W);
  else if (dtype == SEA_BF16)
    hipLaunchKernelGGL((split_layernorm_c8_kernel<__hip_bfloat16>), grid, block, lds, s, (const __hip_bfloat16*)x,
                       (__hip_bfloat16*)out, (const __hip_bfloat16*)gamma, (const __hip_bfloat16*)beta, eps, (int)C, (int)T, (int)S, (int)W);
  else
    hipLaunchKernelGGL((split_layernorm_c8_kernel<__half>), grid, block, lds, s, (const __half*)x, (__half*)out,
                       (const __half*)gamma, (const __half*)beta, eps, (int)C, (int)T, (int)S, (int)W);
  SEA_CHECK_LAUNCH(nm);
  return SEA_OK;
}
