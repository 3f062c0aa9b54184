/*
 * sea_hip.h -- C ABI of libsea_hip.so: SEA sparse-attention hot path on MI355X (gfx950).
 *
 * Drop-in boundary for the reference's operator package
 *   src/models/perlin_attention/ops/__init__.py:1-7   (7 exported operators)
 * and for the sparse branch of PerlinAttention.forward
 *   src/models/perlin_attention/attention.py:774-947  (grouped top-k)
 *   src/models/perlin_attention/attention.py:1034-1042 (mask -> flat CSR)
 *   src/models/perlin_attention/attention.py:1158-1173 (SDDMM/softmax/elmul/SpMM)
 *   src/models/perlin_attention/attention.py:1236-1244 (average-pool mix)
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller;
 *   - the library allocates nothing persistent, frees nothing, keeps no global state except a
 *     thread-local last-error string; every call is asynchronous on `stream` (a hipStream_t);
 *   - return 0 on success, negative SEA_E* otherwise (no exception crosses the ABI);
 *   - tensors are addressed by ELEMENT strides; the innermost (feature / pixel) stride must be 1;
 *   - "flat CSR" is the reference's wire format (causal_resize_m_to_t.py:757-762): one CSR row per
 *     (batch, query) with column id = head*T_src + key, entries of a row grouped by ascending head,
 *     pixels ascending inside a head, keys DESCENDING inside a pixel (causal_resize_m_to_t.py:569).
 *     Internally indices are int32; `idx_bytes` = 8 selects int64 at the API edge (torch CSR).
 */
#ifndef SEA_HIP_H
#define SEA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version, returned by sea_version(); the Python binding refuses a library of another version (a stale .so next to newer
 * Python, or the other way round, through SEA_HIP_LIB).  History of INCOMPATIBLE changes of existing entry points:
 *   1  rounds 1-2
 *   2  round 3: sea_performer_causal_step / _at read k / v / pos FROM THE LAST CHUNK BOUNDARY (T + t_base % C rows) and take the
 *      state image at that boundary;  round 4: sea_predictor_mlp's w2_packed / vectors pad every decoder half to whole
 *      16-row tiles (identical for Wd % 16 == 0), sea_predictor_tail_select accepts probs = NULL and any T_m % 4 == 0 <= 512
 *   3  round 5: new entry points the binding requires (sea_causal_conv_c8_z, sea_predictor_tail_z, sea_predictor_tail_select_z,
 *      sea_causal_conv_c8_f32, sea_decode_cnn_tail_select, sea_predictor_tail_consts, sea_sparse_attention_fused_at);
 *      sea_predictor_tail_select / _at take a
 *      trailing `consts_tab` argument (NULL = the round-4 behaviour)
 *      */
#define SEA_ABI_VERSION 3

enum sea_dtype { SEA_F32 = 0, SEA_F16 = 1, SEA_BF16 = 2 };

enum sea_error {
  SEA_OK = 0,
  SEA_EINVAL = -1,      /* bad argument (null pointer, bad dtype, stride) */
  SEA_EUNSUPPORTED = -2,/* shape outside what the kernels are built for */
  SEA_ELAUNCH = -3      /* HIP launch error (see sea_last_error) */
};

typedef void* sea_stream_t; /* hipStream_t */

int sea_version(void);
const char* sea_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * a6  grouped top-k selection.
 * Replaces: PerlinAttention.forward "mask" region, attention.py:774-947 (torch.sort + int64 rank
 * scatter + compare) and the kernel-test helper ops/kernels/causal_topk_masking.py:3-77.
 *
 * For every (n, t) keep the keep[n*keep_stride_n + t] largest of the H*T_m pooled values
 * probs[n, :, t, :] (ties: lower flat index h*T_m+b first).  Writes
 *   bits      (N, T_dst, W) uint32, W = ceil(H*T_m/32): bit f of a row = pixel f kept;
 *   mask_out  optional (N, H, T_dst, T_m) fp32 0/1, contiguous  (= partial_attention_mask_before_interp);
 * and, for the interpolation that follows (target widths as causal_resize_m_to_t.py:951-955,
 * boundaries round_half_away(b*fp32(w/T_m)), per-pixel count clamped to max_k, :657-659),
 *   row_nnz   (N, T_dst) int32      entries row t will emit,
 *   head_off  (N, T_dst, H+1) int32 exclusive per-head offsets inside the row.
 * Limits: H*T_m <= 16384, T_m % 4 == 0.
 */
int sea_topk_select(const void* probs, int dtype,
                    int64_t N, int64_t H, int64_t T_dst, int64_t T_m,
                    int64_t stride_n, int64_t stride_h, int64_t stride_t,
                    const int32_t* keep, int64_t keep_stride_n,
                    int64_t T_src, int is_causal, int max_k,
                    uint32_t* bits, float* mask_out,
                    int32_t* row_nnz, int32_t* head_off,
                    sea_stream_t stream);

/* Same outputs as sea_topk_select, but from an existing 0/1 mask (N,H,T_dst,T_m) of `dtype`
 * (nonzero = kept).  Replaces the `n_pixels` pass of scan_col, causal_resize_m_to_t.py:657-659. */
int sea_mask_to_bits(const void* mask, int dtype,
                     int64_t N, int64_t H, int64_t T_dst, int64_t T_m,
                     int64_t stride_n, int64_t stride_h, int64_t stride_t,
                     int64_t T_src, int is_causal, int max_k,
                     uint32_t* bits, int32_t* row_nnz, int32_t* head_off,
                     sea_stream_t stream);

/* crow[n, 0..T_dst] = exclusive scan of row_nnz[n, :]  (replaces cumsum + `crow_indices[:,1:] = ...`,
 * causal_resize_m_to_t.py:664,672).  crow is int32 or int64 per idx_bytes. */
int sea_csr_row_scan(const int32_t* row_nnz, int64_t N, int64_t T_dst,
                     void* crow, int idx_bytes, sea_stream_t stream);

/* a7  emit the column indices (replaces nonzero() + __scan_col_4_compute,
 * causal_resize_m_to_t.py:493-572,724-746).  col has room for z_cap entries per batch item
 * (col_stride_n elements apart); entries at or beyond crow[n,T_dst] are left untouched.
 * values_out (optional, fp32, same shape as col) receives 1.0 for every emitted entry. */
int sea_csr_emit(const uint32_t* bits, const void* crow, const int32_t* head_off,
                 int64_t N, int64_t H, int64_t T_dst, int64_t T_m,
                 int64_t T_src, int is_causal, int max_k,
                 void* col, int idx_bytes, int64_t col_stride_n, int64_t z_cap,
                 float* values_out,
                 sea_stream_t stream);

/* Per-(row, head) offsets of a foreign flat CSR whose rows are grouped by ascending head
 * (replaces __flat_csr_sdbmm_tch_compute, flat_csr_sdbmm.py:48-127). */
int sea_csr_head_offsets(const void* crow, const void* col, int idx_bytes,
                         int64_t N, int64_t H, int64_t T_dst, int64_t T_src,
                         int64_t col_stride_n, int32_t* head_off, sea_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a9..a12  the four unfused CSR operators (1:1 with ops/__init__.py), values are fp32 (N, Z).
 */
/* flat_csr_masked_bmm, flat_csr_masked_bmm.py:137-195: values[n,e] = q[n,h,row,:] . k[n,h,key,:] */
int sea_csr_sddmm(const void* q, const void* k, int dtype,
                  int64_t N, int64_t H, int64_t T_dst, int64_t T_src, int64_t D,
                  const int64_t* q_strides /*[n,h,t]*/, const int64_t* k_strides /*[n,h,t]*/,
                  const void* crow, const void* col, int idx_bytes, int64_t col_stride_n,
                  float* values, sea_stream_t stream);

/* flat_csr_softmax, flat_csr_softmax.py:127-176: softmax over each (row, head) group of entries. */
int sea_csr_softmax(const float* in_values, float* out_values,
                    int64_t N, int64_t H, int64_t T_dst, int64_t T_src,
                    const void* crow, const void* col, int idx_bytes, int64_t col_stride_n,
                    sea_stream_t stream);

/* flat_csr_elmul, flat_csr_elmul.py:110-162: values[n,e] *= other[n,h,row,key]  (element strides,
 * stride 0 allowed -- the module passes a row-broadcast view, attention.py:1170-1171). */
int sea_csr_elmul(const float* in_values, float* out_values,
                  const void* other, int dtype, const int64_t* other_strides /*[n,h,t,s]*/,
                  int64_t N, int64_t H, int64_t T_dst, int64_t T_src,
                  const void* crow, const void* col, int idx_bytes, int64_t col_stride_n,
                  sea_stream_t stream);

/* flat_csr_sdbmm, flat_csr_sdbmm.py:323-439: out[n,h,row,:] = sum_e values[e] * v[n,h,key,:];
 * out is fp32 (N,H,T_dst,D) contiguous, as the reference (`torch.zeros` w/o dtype, :347). */
int sea_csr_spmm(const float* values, const void* v, int dtype,
                 int64_t N, int64_t H, int64_t T_dst, int64_t T_src, int64_t D,
                 const int64_t* v_strides /*[n,h,t]*/,
                 const void* crow, const void* col, int idx_bytes, int64_t col_stride_n,
                 const int32_t* head_off,
                 float* out, sea_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused row-indexed sparse attention (the hot kernel): SDDMM + per-(row,head) softmax +
 * row scale + SpMM (+ optional average-pool mix) in one pass, one wavefront per (n, h, t).
 * Replaces the four launches of attention.py:1158-1173 and, with `avg`/`mix`, :1236-1237.
 *
 *   out[n,h,t,:] = rs * sum_e softmax_e(q.k_e) v_e            rs  = row_scale[n,h,t]  (or 1)
 *   if mix:  out = out*a + (1-a)*avg[n,h,t,:]                  a   = mix[n,h,t]
 *
 * crow/col are int32 (internal format), head_off as produced by sea_topk_select.
 * out has dtype out_dtype and arbitrary [n,h,t] element strides (so it can be written straight
 * into the (N, T, H*D) layout of attention.py:1279-1282).
 */
int sea_sparse_attention(const void* q, const void* k, const void* v, int dtype,
                         int64_t N, int64_t H, int64_t T_dst, int64_t T_src, int64_t D,
                         const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                         const int32_t* crow, const int32_t* col, int64_t col_stride_n,
                         const int32_t* head_off,
                         const float* row_scale, /* (N,H,T_dst) contiguous or NULL */
                         const void* avg, const int64_t* avg_strides, /* dtype `dtype`, or NULL */
                         const float* mix,       /* (N,H,T_dst) contiguous or NULL */
                         void* out, int out_dtype, const int64_t* out_strides,
                         sea_stream_t stream);

/* The same operator with two more knobs.
 *
 * flags, low byte = kernel path:
 *   SEA_ATTN_AUTO    with a `block_path` plan (sea_attention_plan): the ONE kernel the plan's statistics favour for this
 *                    launch, decided on the device; without a plan: the gather kernels (the tile kernel pays off only
 *                    where neighbouring query rows share most of their keys);
 *   SEA_ATTN_GATHER  row-indexed gather kernels (sea_attn.hip): one lane group per (n,h,t) row walks the row's entries,
 *                    every K / V row is fetched per entry (L2-served); any dtype, any D <= 64*vec, duplicates counted;
 *   SEA_ATTN_TILE    MFMA tile kernel (sea_attn_tile.hip): a wave owns 16 (or 32) consecutive query rows of one (n,h),
 *                    turns their entries into a key bitmap in LDS, and for every 16-key tile that holds a kept key
 *                    runs K.Q^T, the masked online softmax and V^T.P^T on v_mfma_f32_16x16x32 with K/V rows fetched
 *                    once per tile -- the shape of the reference's flat_csr_sdbmm.py:141-313.  16-bit data,
 *                    D in {64, 80, 128}, no duplicate (row, column) pairs.  SEA_EUNSUPPORTED otherwise.
 *                    CONTRACT (differs from the gather kernels): every V row below T_src must be FINITE.  A key that
 *                    shares a 16-key tile with a kept key is staged and multiplied by an exact 0, and 0 * Inf = NaN
 *                    (the reference's dense branch matmul(probs, v), attention.py:1128, has the same property); K rows
 *                    may hold anything (their scores are replaced before use).  The gather kernels read kept keys only:
 *                    a kv-cache whose unused slots are uninitialised takes SEA_ATTN_GATHER (as DecodeSession does) or
 *                    zero-fills them;
 *   bits 8..11: row tiles per wave for the tile kernel (1 or 2; 0 = default for the head size);
 *   bits 12..15: log2 of its key window (6..12; 0 = default 2048).
 *
 * probs_out (optional): fp32, laid out like `col` (row n at probs_out + n*probs_stride_n): entry e receives
 *   rs * softmax_e -- the values of `partial_attention_probs` after flat_csr_softmax + flat_csr_elmul
 *   (attention.py:1162-1171).  Served by the gather kernels (SEA_ATTN_TILE + probs_out is SEA_EUNSUPPORTED;
 *   SEA_ATTN_AUTO falls back to them). */
enum sea_attn_path { SEA_ATTN_AUTO = 0, SEA_ATTN_GATHER = 1, SEA_ATTN_TILE = 2 };
int sea_sparse_attention_ex(const void* q, const void* k, const void* v, int dtype,
                            int64_t N, int64_t H, int64_t T_dst, int64_t T_src, int64_t D,
                            const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                            const int32_t* crow, const int32_t* col, int64_t col_stride_n,
                            const int32_t* head_off,
                            const float* row_scale, const void* avg, const int64_t* avg_strides,
                            const float* mix,
                            void* out, int out_dtype, const int64_t* out_strides,
                            float* probs_out, int64_t probs_stride_n,
                            const uint8_t* block_path, /* buffer filled by sea_attention_plan, or NULL */
                            int flags, sea_stream_t stream);

/* Steps I + J of the hot path in ONE launch (round 3): the nearest-neighbour interpolation of the kept pixels
 * (causal_resize_m_to_t.py:493-572,631-762) INSIDE the row-indexed sparse attention (flat_csr_masked_bmm / softmax / elmul /
 * sdbmm).  Each lane group of the gather kernels expands ITS (row, head)'s kept pixels to key columns -- sea_csr_emit's
 * arithmetic bit for bit: fp32 scale = w_t / T_m, bounds round_half_away(b * scale), keys descending inside a pixel, the
 * reference's fp32 stepping for a pixel wider than max_k -- into a group-private list in LDS, writes the list to `col` and
 * walks it from LDS.  No separate sea_csr_emit launch, no column re-read from memory.
 *   bits      (N, T_dst, ceil(H*T_m/32)) kept-pixel masks of sea_topk_select / sea_predictor_tail_select (T_m % 32 == 0);
 *   crow      from sea_csr_row_scan over that launch's row_nnz; head_off from the same launch;
 *   col       (N, col_stride_n) int32: OUTPUT -- after the launch it holds exactly what sea_csr_emit would have written.
 *   write_columns  0 (round 4): the expanded columns stay in LDS and `col` is NOT written (blocks whose key lists exceed the
 *             kernel's LDS list still pass through their part of it): the column array is an output nobody on the hot path
 *             reads -- 266 MB and ~50 us of the headline launch.  A caller that wants it later runs sea_csr_emit on the same
 *             bits / crow (bit-identical); the Python handle keeps its columns pending and does that on first access.
 * Other arguments as sea_sparse_attention_ex (gather path; probs_out allowed).  Rows of 4 lanes and rows wider than 16
 * lanes are SEA_EUNSUPPORTED: run sea_csr_emit + sea_sparse_attention_ex there.
 * sea_attention_few_rows(): for a launch of at most that many rows (N * H * T_dst: a decoding step) sea_csr_emit +
 * sea_sparse_attention_ex is the faster pair -- with T_dst <= 8 that kernel's idle lane groups first touch every K / V row
 * the step will gather, so the rows' dependent walks find them in cache (same arithmetic, bit for bit). */
int64_t sea_attention_few_rows(void);
int sea_sparse_attention_fused(const void* q, const void* k, const void* v, int dtype,
                               int64_t N, int64_t H, int64_t T_dst, int64_t T_src, int64_t D,
                               const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                               const int32_t* crow, int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                               const float* row_scale, const void* avg, const int64_t* avg_strides, const float* mix,
                               void* out, int out_dtype, const int64_t* out_strides,
                               float* probs_out, int64_t probs_stride_n,
                               const uint32_t* bits, int64_t T_m, int is_causal, int max_k, int write_columns,
                               sea_stream_t stream);

/* The DECODE form of the fused launch (round 5; SURVEY 8f-3, src/main/opt_generate.py:131, PA/attention.py:410-426): a position
 * of a graph-replayed decoding session has static kernel arguments, so the sequence length the row widths follow is read
 * from device memory (*t_src_dev, what sea_csr_emit_at reads) while T_cap -- the row count of the K / V caches -- is the
 * stride the column ids are encoded with (head * T_cap + key) and the T_src the operator is called with.  T_dst <= 8 new
 * rows per sequence; 16-bit d = 64 / 80 / 128 or fp32 d = 32 / 64 (the fused forms), else SEA_EUNSUPPORTED (run
 * sea_csr_emit_at + sea_sparse_attention_ex).  The lane groups of a workgroup that have no row touch the K / V rows of the
 * expanded lists before the one group per row starts its dependent walk (what the unfused kernel does from `col`).  Same
 * arithmetic in the same order as sea_csr_emit_at + sea_sparse_attention_ex: the step stays bitwise the stateless forward;
 * the emit launch (or the emit phase of sea_decode_cnn_tail_select: pass col = NULL there) and the crow -> col -> K / V
 * load chain leave the position's critical path.  `col`: scratch / output as in sea_sparse_attention_fused. */
int sea_sparse_attention_fused_at(const void* q, const void* k, const void* v, int dtype,
                                  int64_t N, int64_t H, int64_t T_dst, int64_t T_cap, int64_t D,
                                  const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                  const int32_t* crow, int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                                  const float* row_scale, const void* avg, const int64_t* avg_strides, const float* mix,
                                  void* out, int out_dtype, const int64_t* out_strides,
                                  const uint32_t* bits, int64_t T_m, const int32_t* t_src_dev, int is_causal, int max_k,
                                  int write_columns, sea_stream_t stream);

/* Backward of the fused operator WITHOUT its epilogue (o = sum_e softmax_e(q.k_e) v_e; the caller applies row scale and mix
 * in its autograd framework): dQ, dK, dV from dO.  Reference shape: masked_mm.py:169-267 + the dense branch's autograd
 * (attention.py:1061-1133).  probs = the forward's probs_out with row_scale = NULL; out / dout / dq fp32 (N,H,T_dst,D)
 * contiguous; dk / dv fp32 (N,H,T_src,D) contiguous, ZEROED by the caller (rows are accumulated with fp32 atomics). */
int sea_sparse_attention_bwd(const void* q, const void* k, const void* v, int dtype,
                             int64_t N, int64_t H, int64_t T_dst, int64_t T_src, int64_t D,
                             const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                             const int32_t* crow, const int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                             const float* probs, int64_t probs_stride_n, const float* out, const float* dout,
                             float* dq, float* dk, float* dv, sea_stream_t stream);

/* The same backward WITHOUT float atomics (round 3): dK / dV are gathered over the transposed pattern.  The library counts
 * the entries of every (n, head, key) column (int32 atomics), scans them into list starts, lets the row pass (dQ, plain
 * stores) drop a 16-byte record {row, p, ds} into its key's list, and a column pass (one lane group per key) sums
 * dK = sum ds q_row, dV = sum p dO_row with plain stores -- every dk / dv row is written, the caller does NOT zero them.
 * Same arguments as sea_sparse_attention_bwd plus a caller-owned scratch buffer of
 * sea_sparse_attention_bwd_workspace_bytes(N, H, T_src, col_stride_n) bytes (16-byte aligned; contents undefined afterwards).
 * Rows wider than 16 lanes (fp32 with D > 64) are SEA_EUNSUPPORTED here and take the atomic form.  The records of a key are
 * in the order the row pass reached them: sums agree between runs to fp32 rounding, like the atomic form.
 * Reference shape: masked_mm.py:169-267 (gradients flow through the kept entries only). */
int64_t sea_sparse_attention_bwd_workspace_bytes(int64_t N, int64_t H, int64_t T_src, int64_t col_stride_n);
int sea_sparse_attention_bwd_gather(const void* q, const void* k, const void* v, int dtype,
                                    int64_t N, int64_t H, int64_t T_dst, int64_t T_src, int64_t D,
                                    const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                    const int32_t* crow, const int32_t* col, int64_t col_stride_n, const int32_t* head_off,
                                    const float* probs, int64_t probs_stride_n, const float* out, const float* dout,
                                    float* dq, float* dk, float* dv, void* workspace, int64_t workspace_bytes,
                                    sea_stream_t stream);

/* Kernel-choice plan for SEA_ATTN_AUTO, made on the device (no host round trip, graph-capturable).  One byte per (n, h,
 * 16-row block): 1 where the block's entries per staged 16-key tile favour the MFMA tile kernel -- estimated from the
 * kept-pixel bit masks of sea_topk_select / sea_predictor_tail_select (`bits`, (N,T_dst,ceil(H*T_m/32))): entries the block
 * walks against the 16-key tiles it would stage, favourable when entries >= entries_per_tile * tiles (<= 0: 30) -- followed,
 * 4-byte aligned, by the int32 COUNT of such blocks: `block_path` holds ((N*H*ceil(T_dst/16) + 3) & ~3) + 4 bytes.
 * With a plan, sea_sparse_attention_ex launches BOTH kernels over all rows; each reads the count, and ONE of them runs
 * the launch while the other's workgroups exit at once: the tile kernel when the favourable blocks' share exceeds 1/2
 * (D <= 80) or 13/20 (D = 128), else the gather kernels.  (Round 2 split a launch per block between the two kernels; since
 * the gather kernels deal their rows by length that mix is slower than the better kernel alone -- scripts/
 * sweep_plan_threshold.py -- and the bytes only feed the count.)  T_m % 32 == 0, H <= 64. */
int sea_attention_plan(const uint32_t* bits, int64_t N, int64_t H, int64_t T_dst, int64_t T_src, int64_t T_m,
                       int is_causal, float entries_per_tile, uint8_t* block_path, sea_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Bandwidth-bound estimator / epilogue pieces (the callers either side of the hot kernels).
 */
/* ChannelSplit + LayerNorm (+ optional GELU): out[n, c*S+i, t, :] = act(LN(x[n, c, t, i*W:(i+1)*W]) * gamma + beta).
 * S=2, activation=0 replaces ChannelSplit (attention.py:123-131) + cnn.lnorm1 (attention.py:266);
 * S=1, activation=1 (exact-erf GELU) replaces LayerNorm + GELU of attention_predictor_enc (attention.py:190-196).
 * x (N,C,T,S*W) and out (N,C*S,T,W) contiguous, dtype `dtype`; gamma/beta (W) of the same dtype. */
int sea_split_layernorm(const void* x, int dtype, int64_t N, int64_t C, int64_t T, int64_t S, int64_t W,
                        const void* gamma, const void* beta, float eps, int activation, void* out,
                        sea_stream_t stream);

/* Predictor tail in one pass: nearest upsample x`up` along the width + 1x1 conv (C -> H channels, zero pad 1
 * on the width) + area resize (T_m+2 -> T_m) + LayerNorm(T_m) + softmax(T_m).
 * Replaces cnn.keepres.upsam / conv4 / the KeepRes resize / cnn.lnorm2 (attention.py:271-281,
 * modules.py:42-55,77-92) and the softmax of attention.py:670-673.
 * y (N,C,T,W4) of `dtype` with FIVE element strides y_strides = {n, c, t, w, c8}: element (n,c,t,w) lives at
 * n*s[0] + (c/8)*s[4] + (c%8)*s[1] + t*s[2] + w*s[3].  Plain 4-D layouts have s[4] = 8*s[1] (NCHW: s[3] == 1;
 * NHWC: s[1] == 1); the C8 layout of sea_causal_conv_c8 has s[1] = 1, s[3] = 8, s[4] = 8*W4, s[2] = C*W4;
 * conv_wT (C, Hpad) FP32 = the conv weight transposed with the head axis zero-padded to Hpad = 8*ceil(H/8),
 * conv_b (Hpad) FP32 (both are read through the scalar cache); gamma/beta (T_m) of `dtype`;
 * conv_w16 (optional, may be NULL): (16*ceil(H/16), Cp) row-major copy of the weight in `dtype`, channels zero-padded
 * to Cp (multiple of 32) -- with it, 16-bit NHWC / C8 input takes the MFMA variant of the kernel;
 * probs and optional scores (pre-softmax) are (N,H,T,T_m) contiguous of `dtype`.
 * Requires W4*up == T_m, T_m <= 512, W4 a multiple of the 16-byte vector width. */
int sea_predictor_tail(const void* y, int dtype, int64_t N, int64_t C, int64_t H, int64_t T, int64_t W4,
                       int64_t up, int64_t T_m, const int64_t* y_strides,
                       const void* conv_wT, const void* conv_b, const void* conv_w16, int64_t Cp,
                       const void* gamma, const void* beta, float eps,
                       void* probs, void* scores, sea_stream_t stream);

/* The same tail from z = the 1x1 convolution's output (N, T, H, W4) fp32, as sea_causal_conv_c8_z writes it (16-bit maps;
 * conv_b (>= H) fp32 is still needed: the zero-padded border pixels of the padded 1x1 convolution are the bias alone). */
int sea_predictor_tail_z(const float* z, int dtype, int64_t N, int64_t H, int64_t T, int64_t W4, int64_t up, int64_t T_m,
                         const float* conv_b, const void* gamma, const void* beta, float eps, void* probs, void* scores,
                         sea_stream_t stream);

/* Predictor tail + grouped top-k selection in one launch: sea_predictor_tail (MFMA variant, 16-bit
 * channels-last / C8 input) followed by sea_topk_select on the probability map it produces, with the map's values
 * handed over on chip -- the (N,H,T,T_m) map is never re-read, and `probs` MAY BE NULL (round 4): the map is then not
 * written at all (537 MB per step at OPT-1.3B x 8, stored only because the module returns it, attention.py:1343); a caller
 * that wants it later runs sea_predictor_tail on the same y (bit-identical values).  The selection's slow path (overfull
 * threshold bin, all-equal rows) works on the on-chip keys too.
 * Bit-identical to the two separate calls.  Requires W4 * up == T_m, T_m % 4 == 0, T_m <= 512, H <= 64, H * T_m <= 16384
 * (round 4: any predictor length -- the reference's own grid runs 64 / 96 / 128 / 256 / 384,
 * src/main/benchmark_opt_ablation.py:160-186, exp_long_context.py:152; T_m = 256 with H % 4 == 0 keeps the map in
 * registers, every other shape passes it through a flat LDS image of the row).  SEA_EUNSUPPORTED when the row's LDS plan
 * does not fit.  Arguments as in sea_predictor_tail (conv_w16 mandatory, no FP32 weight copy) and sea_topk_select
 * (keep / keep_stride_n / T_src / is_causal / max_k -> bits / row_nnz / head_off).
 * fp32 data (round 5, the reference's measurement protocol): dtype = SEA_F32 with T_m = 256 (W4 = 64, up = 4), H % 4 == 0,
 * H <= 32; `conv_w16` then carries sea_predictor_tail's conv_wT, the (C, Hpad) FP32 transposed weights (Cp ignored), probs is
 * mandatory (the fp32 map is always written) and the 1x1 convolution runs on the fp32 MFMA -- bit-identical to
 * sea_predictor_tail (fp32, channels-last / C8 input: the same device code) followed by sea_topk_select. */
int sea_predictor_tail_select(const void* y, int dtype, int64_t N, int64_t C, int64_t H, int64_t T, int64_t W4,
                              int64_t up, int64_t T_m, const int64_t* y_strides, const void* conv_b,
                              const void* conv_w16, int64_t Cp, const void* gamma, const void* beta, float eps,
                              void* probs, void* scores, const int32_t* keep, int64_t keep_stride_n,
                              int64_t T_src, int is_causal, int max_k, uint32_t* bits, int32_t* row_nnz,
                              int32_t* head_off, const uint32_t* consts_tab, sea_stream_t stream);

/* The per-pixel constants of the tail -- for every output pixel the <= 3 taps of the area resize, gamma, beta: (W4, up, T_m,
 * gamma, beta) only, the same for every row -- computed ONCE into tab (3 * 256 uint32, 16-byte aligned) instead of by every
 * row's workgroup (round 5; T_m = 256).  The `consts_tab` argument of sea_predictor_tail_select / _at / _z and
 * sea_decode_cnn_tail_select takes it (NULL = each row computes the table itself, as before): -3.5 % of the launch at
 * OPT-1.3B x 8, -8 % at H = 12. */
int sea_predictor_tail_consts(int dtype, int64_t W4, int64_t up, int64_t T_m, const void* gamma, const void* beta,
                              uint32_t* tab, sea_stream_t stream);

/* The same launch fed with z (N, T, H, W4) fp32 = the 1x1 convolution's output from sea_causal_conv_c8_z (round 5): the "z
 * tile" of a row -- loads of y and of the weights, MFMAs, LDS stores: a third of a row's life in this issue-bound kernel --
 * becomes a 16-byte-per-lane copy into LDS.  Everything else (and every bit of the result) as sea_predictor_tail_select;
 * a caller that wants the map later runs sea_predictor_tail_z on the same z. */
int sea_predictor_tail_select_z(const float* z, int dtype, int64_t N, int64_t H, int64_t T, int64_t W4, int64_t up,
                                int64_t T_m, const float* conv_b, const void* gamma, const void* beta, float eps,
                                void* probs, void* scores, const int32_t* keep, int64_t keep_stride_n, int64_t T_src,
                                int is_causal, int max_k, uint32_t* bits, int32_t* row_nnz, int32_t* head_off,
                                const uint32_t* consts_tab, sea_stream_t stream);

/* A decoding step's predictor CNN + tail + selection + state advance in ONE launch (round 5; perlin_attention/decode.py).
 * A graph-replayed position used to run conv1, conv2 (each over the session's whole 25-row window), sea_predictor_tail_select_at
 * and sea_c8_window_shift: four launches for one new row per sequence, each at its fixed cost.  Here one workgroup per sequence
 *   - computes conv1's new row from x rows t - 2 dil, t - dil (ring of the MLP's earlier rows) and t (`x_new`, which the MLP
 *     launch has just written), conv2's new row from the ring of conv1's rows -- bit for bit the rows sea_causal_conv_c8 writes
 *     (same operand placement and k order; weights read straight from the packed images, w1_packed / w2_packed / bias as for
 *     sea_causal_conv_c8 with Cin = Cout = C, 3 x 3, `dilation`, pad_w = dilation);
 *   - runs the tail + selection of sea_predictor_tail_select_at on that row (T_m = 256, W4 = 64, up = 4; keep_table over
 *     absolute rows; outputs bits (N,1,W), row_nnz (N,1), head_off (N,1,H+1), crow_out (N,2), optional probs (N,H,1,256));
 *   - files x_new and conv1's new row in their rings (slot = position % ring size: nothing is shifted) and, as the last
 *     workgroup to finish, advances counters = {rows seen (the new row's position), T_src of the step, T_src of the step JUST
 *     FINISHED}: counters[2] = counters[1], then counters[0] += 1, counters[1] += 1.  Launches behind this one in the same
 *     step (sea_csr_emit_at) read their T_src from counters + 2.  `ticket` is one zero-initialised int32 the library owns
 *     between calls.
 * `col` (optional, C <= 64): the step's CSR columns (N, col_stride_n) int32, ids = head * T_cap + key as sea_csr_emit_at writes
 *   them, at most z_cap per item -- the emit of the one new row runs inside this launch too (one launch less per position);
 *   NULL = the caller runs sea_csr_emit_at.
 * x_new (N, C/8, 64, 8); x_ring (N, ring_x, C/8, 64, 8); y1_ring (N, ring_y, C/8, 64, 8); y2 (N, C/8, 64, 8) scratch, all `dtype`
 * (16-bit); ring sizes > 2 * dilation.  C = 2 H <= 80, H % 4 == 0. */
int sea_decode_cnn_tail_select(const void* x_new, void* x_ring, void* y1_ring, void* y2, int dtype, int64_t N, int64_t C,
                               int64_t H, int64_t W4, int64_t ring_x, int64_t ring_y, const void* w1_packed,
                               const float* bias1, const void* w2_packed, const float* bias2, int64_t CinP, int dilation,
                               int pad_w, const void* conv_b, const void* conv_w16, int64_t Cp, const void* gamma,
                               const void* beta, float eps, void* probs, const int32_t* keep_table, int32_t* counters,
                               int32_t* ticket, int is_causal, int max_k, uint32_t* bits, int32_t* row_nnz,
                               int32_t* head_off, int32_t* crow_out, int32_t* col, int64_t col_stride_n, int64_t z_cap,
                               int64_t T_cap, const uint32_t* consts_tab, sea_stream_t stream);

/* Causal cumulative average out[n,h,t,:] = sum_{s<=t} v[n,h,s,:] / (t+1), fp32 accumulation.
 * Replaces `avg_v.cumsum(-2) / arange(1..T)` (attention.py:1220-1222).  out (N,H,T,D) contiguous. */
int sea_cumavg(const void* v, int dtype, int64_t N, int64_t H, int64_t T, int64_t D, const int64_t* v_strides,
               void* out, sea_stream_t stream);
/* The same with the T rows cut into n_slices slices (two launches: column totals per slice, then the averages
 * starting from the totals before a slice) -- for the few (n,h) pairs of a one-sequence-per-GPU shard.
 * workspace: N*H*n_slices*D floats, caller-owned.  16-bit data, D in {32,64,80,128}. */
int sea_cumavg_sliced(const void* v, int dtype, int64_t N, int64_t H, int64_t T, int64_t D, const int64_t* v_strides,
                      void* out, int64_t n_slices, void* workspace, int64_t workspace_bytes, sea_stream_t stream);

/* Channel-blocked ("C8") predictor CNN for 16-bit data.
 * C8 layout of a logical (N, C, T, W) activation, C % 8 == 0:  memory (N, T, C/8, W, 8) -- 16-byte blocks of 8
 * channels, consecutive pixels of a block adjacent (what an MFMA operand fragment reads contiguously).
 * sea_split_layernorm_c8: as sea_split_layernorm (no activation) but the result is written C8,
 *   out (N, T, C*S/8, W, 8) -- the layout the conv kernels below consume (16-bit data, and fp32 since round 5).
 * sea_causal_conv_c8: y = act(conv2d(x) + bias), square kernel `ksize` (1 or 3), dilation `dilation`, zero padding
 *   (ksize-1)*dilation rows on TOP only (causal along T, = CausalConv2d of modules.py:96-192 whose lower kernel
 *   rows are masked) and pad_w columns on both sides (width preserving).  x (N,T,Cin/8,W,8), y (N,T,Cout/8,W,8);
 *   w_packed (Cout, ksize*ksize*CinP) 16-bit = weight[co, ci, i, j] laid out [co][i*ksize+j][ci], ci zero-padded to
 *   CinP (Cin rounded up to 32); bias (Cout) FP32; relu != 0 fuses the ReLU that follows conv1/conv2
 *   (attention.py:271-276). */
int sea_split_layernorm_c8(const void* x, int dtype, int64_t N, int64_t C, int64_t T, int64_t S, int64_t W,
                           const void* gamma, const void* beta, float eps, void* out, sea_stream_t stream);
int sea_causal_conv_c8(const void* x, int dtype, int64_t N, int64_t T, int64_t W, int64_t Cin, int64_t Cout,
                       const void* w_packed, int64_t CinP, const float* bias, int ksize, int dilation, int pad_w,
                       int relu, void* y, sea_stream_t stream);

/* fp32 twin of sea_causal_conv_c8 (round 5): exact fp32 products and accumulation on v_mfma_f32_16x16x4_f32, for callers that
 * keep the reference's fp32 measurement protocol (src/main/benchmark_bert.py:196-239).  x (N,T,Cin/8,W,8), y (N,T,Cout/8,W,8)
 * fp32 in the same channel-blocked layout; w_packed (Cout, ksize*ksize, CinP) fp32 = weight[co, ci, i, j] laid out
 * [co][i*ksize+j][ci], ci zero-padded to CinP = Cin rounded up to 16; bias (Cout) fp32.  ksize 1 or 3, Cout <= 80;
 * SEA_EUNSUPPORTED when the fp32 weight image (16*ceil(Cout/16) x ksize^2 x CinP x 4 B) exceeds the 160 KB LDS. */
int sea_causal_conv_c8_f32(const float* x, int64_t N, int64_t T, int64_t W, int64_t Cin, int64_t Cout,
                           const float* w_packed, int64_t CinP, const float* bias, int ksize, int dilation, int pad_w,
                           int relu, float* y, sea_stream_t stream);

/* The LAST (conv, ReLU) pair of the predictor CNN with the tail's 1x1 convolution in its epilogue (round 5).  `KeepRes` adds
 * no residual (modules.py:42-55) and a 1x1 kernel commutes with the nearest x`up` upsample that sits between the two
 * (attention.py:266-281), so  z = W1 . relu(conv(x) + bias) + b1  can be formed while the activation tile is still in the
 * matrix pipe's registers: + ceil(H/16) * ceil(Cout/32) MFMAs per 16 pixels (+5.5 % at 64 -> 64 channels, 32 heads).
 *   arguments up to `y` as sea_causal_conv_c8 (ksize = 3, Cout <= 80); y may be NULL (the activation is then not written);
 *   conv1x1_w16 (16*ceil(H/16), Cp1) 16-bit row-major, zero padded, Cp1 = Cout rounded up to 32 (= sea_predictor_tail's
 *   conv_w16); conv1x1_b (>= H) fp32; z (N, T, H, W) fp32 -- what sea_predictor_tail_z / sea_predictor_tail_select_z read.
 * The activation enters the product rounded to `dtype` exactly as the y store rounds it, with the operand placement and k
 * order of the tail kernels' own z stage: z is bit for bit what they compute from y. */
int sea_causal_conv_c8_z(const void* x, int dtype, int64_t N, int64_t T, int64_t W, int64_t Cin, int64_t Cout,
                         const void* w_packed, int64_t CinP, const float* bias, int ksize, int dilation, int pad_w,
                         int relu, void* y, const void* conv1x1_w16, int64_t Cp1, const float* conv1x1_b, int64_t H,
                         float* z, sea_stream_t stream);

/* Predictor MLP of SEA's estimator in one launch, 16-bit data, bf16/f16 MFMA:
 *   enc  = GELU(LayerNorm_D1(x W1^T + b1))                      attention_predictor_enc      (attention.py:190-196)
 *   dec  = enc W2^T + b2, split into S = 2 halves of Wd = D2/2      attention_predictor_dec_row  (attention.py:123-131,623)
 *   y    = LayerNorm_Wd(dec half) * g2 + be2, written in the C8 layout of sea_causal_conv_c8 with channel = 2*h + half
 *                                                                   cnn.lnorm1                   (attention.py:266)
 *   gate = sigmoid(enc Wsc^T + bsc)  (2 values per row)             attention_predictor_dec_scaler (attention.py:1158-1166)
 * Every Linear / LayerNorm output is rounded to `dtype` exactly where the reference's module chain rounds it.
 * x (N,H,T,Din) of `dtype`, element strides x_strides[n,h,t], Din contiguous.
 * w1_packed: W1 (D1,Din) as MFMA A fragments  [ks][tile][lane][j] = W1[16*tile + lane%16][32*ks + 8*(lane/16) + j]
 *            (ks < ceil(Din/32), tile < D1/16, zero beyond Din);
 * w2_packed: [ks][tile][lane][j] = W2'[16*tile + lane%16][f(ks, lane/16, j)],  f(ks,g,j) = 16*(2*ks + j/4) + 4*g + j%4,
 *            ks < D1/32, tile <= 2*HT, HT = ceil(Wd/16), WdP = 16*HT, where W2' = the rows of W2 (D2,D1) with each half
 *            zero-padded to WdP rows (half s at rows s*WdP .. s*WdP+Wd-1; identical to W2 when Wd % 16 == 0), followed by
 *            one extra tile whose rows 0,1 are Wsc (2,D1);
 * vectors (fp32): b1[D1] g1[D1] be1[D1] b2[2*WdP] (padded like W2') g2[WdP] be2[WdP] (zero past Wd) bsc[2].
 * Outputs: x_c8 (N, T, H*2/8, Wd, 8) of `dtype`, batch items x_c8_stride_n elements apart (0 = dense; a decode session
 *   lets the one new row of every item land behind that item's CNN window); optional tpred (N,H,T,D1) of `dtype` (= enc); optional
 * row_scale / avg_scale (N,H,T) FP32 = gate[...,0] / gate[...,1].
 * Supported (D1, D2): D1 = 128 with any D2 % 16 == 0 up to 256 (round 4: every predictor length T_M = 2*D2 with
 * T_M % 32 == 0, the reference's grid of src/main/benchmark_opt_ablation.py:160-186 included), (160,128), (256,128)
 * (the weights must fit 160 KB of LDS, or stream: D1 = 256); H % 4 == 0; Din % 8 == 0. */
int sea_predictor_mlp(const void* x, int dtype, int64_t N, int64_t H, int64_t T, int64_t Din, const int64_t* x_strides,
                      int64_t D1, int64_t D2, const void* w1_packed, const void* w2_packed, const float* vectors,
                      float eps1, float eps2, void* x_c8, int64_t x_c8_stride_n, void* tpred, float* row_scale, float* avg_scale,
                      sea_stream_t stream);

/* Causal Performer of SEA's estimator in one launch, fp32 MFMA:
 *   phi(x) = relu(D^-1/4 x W^T) + 1e-3;  ctx_t = sum_{s<=t} (phi(q_t).phi(k_s)) V_s / (phi(q_t).(sum_{s<=t} phi(k_s) + 1e-6))
 * with V = [pos | v] (the learned causal value embedding concatenated in front of v, attention.py:506-510).
 * Replaces performer_pytorch.FastAttention(causal, generalized) as called at attention.py:556-572 plus the
 * concatenations at :506-510 and :577-590.  q,k,v (N,H,T,D) of `dtype` (element strides [n,h,t]), pos (>=T, D)
 * with row stride pos_stride, proj (nb, D) FP32.  out (N,H,T,3D) contiguous of `dtype` = [ctx_pos | ctx_v | v].
 * Supported: D in {64,80,128}, nb <= 80.
 * 16-bit data runs on 16-bit MFMA with split (hi+lo) operands (64-row chunks at D = 64, 32-row chunks
 * and two column blocks per wave at D = 80 / 128); those kernels can also emit avg_out (N,H,T,D) = cumsum_t(v)/(t+1),
 * the input of the mix step (attention.py:1220-1222) -- pass NULL otherwise (sea_performer_avg_supported says when it
 * may be non-NULL).  FP32 data: fp32 MFMA throughout. */
int sea_performer_causal(const void* q, const void* k, const void* v, const void* pos, int dtype,
                         const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                         const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                         int64_t pos_stride, void* out, void* avg_out, sea_stream_t stream);

/* Sequence-parallel form of sea_performer_causal.  One workgroup walks one (n, h) pair's rows in order, so N*H pairs
 * fill N*H compute units: the reference's configurations that put ONE sequence on a GPU (BASELINE configs 4-5: 32-40
 * pairs on 256 CUs) leave most of the chip idle.  With n_segments > 1 the T rows are cut into segments of whole
 * 64-row chunks; a first launch leaves every segment's state increment (sum phi(k)^T [pos|v], sum phi(k), sum v) in
 * `workspace`, the second starts each segment from the sum of the increments before it (added in segment order:
 * results are reproducible run to run; they differ from the one-segment kernel in fp32 summation order only).
 * sea_performer_plan proposes n_segments for a shape (1 when N*H already fills the chip) and the workspace size;
 * the caller owns the workspace (16-byte aligned, no initialisation needed).  n_segments = 1 is
 * sea_performer_causal (workspace may be NULL).  Same role in the reference as sea_performer_causal. */
/* 1 when sea_performer_causal* can also write `avg_out` (the cumulative average of v) for this head size, feature
 * count and dtype, else 0 -- the one predicate both sides of the ABI use (host arithmetic only). */
int sea_performer_avg_supported(int64_t D, int64_t nb, int dtype);
int sea_performer_plan(int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb, int dtype,
                       int64_t* n_segments, int64_t* workspace_bytes);
int sea_performer_causal_segmented(const void* q, const void* k, const void* v, const void* pos, int dtype,
                                   const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                                   const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                   int64_t pos_stride, void* out, void* avg_out, int64_t n_segments,
                                   void* workspace, int64_t workspace_bytes, sea_stream_t stream);

/* Stateful, CHUNK-ALIGNED form of the causal Performer for kv-cache decoding (role of the reference's
 * StatefulCausalPerformer, attention_state.py:43-140, called from attention.py:559-566 when pconfig.use_cache; parity
 * protocol test_perlin_opt_cache.py:7-32: cached decoding reproduces the stateless forward).  The T rows handed in CONTINUE
 * sequences of which t_base rows have been seen.  The kernels walk the rows in chunks of C = sea_performer_chunk_rows(...)
 * rows (64 for D = 64, 32 for D = 80 / 128); an image -- sea_performer_state_bytes(N,H,D,nb,dtype) bytes holding, per
 * (n,h), the kernel's own FP32 accumulators sum phi(k)^T [pos|v], sum phi(k), sum v -- is always the state at a CHUNK
 * BOUNDARY: state_in at c0 = floor(t_base / C) * C (NULL only at t_base = 0), state_out at floor((t_base + T) / C) * C.
 * The open chunk's old rows c0 .. t_base-1 are walked AGAIN:
 *   k, v, pos   point at ROW c0 of the caller's kv-cache / value embedding: T + t_base % C rows are read;
 *   q, out, avg_out point at the first NEW row: T rows.
 * Every new row is thereby computed by the very instruction sequence the one-pass kernel runs for it (same chunk, same
 * operand tiles, same summation order): outputs and images are BITWISE those of sea_performer_causal over the whole
 * sequence, however the sequence is cut into calls.  16-bit data, D in {64, 80, 128} (fp32 data: SEA_EUNSUPPORTED -- the
 * torch-side state of attention_state.py serves it).  state_in and state_out may alias when n_segments = 1.
 * n_segments / workspace as in sea_performer_causal_segmented (1 / NULL for the few rows of a decode step; a prefill may
 * cut: its image then sums the segments' increments in segment order). */
int64_t sea_performer_chunk_rows(int64_t D, int64_t nb, int dtype);
int64_t sea_performer_state_bytes(int64_t N, int64_t H, int64_t D, int64_t nb, int dtype);
int sea_performer_causal_step(const void* q, const void* k, const void* v, const void* pos, int dtype,
                              const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                              const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                              int64_t pos_stride, void* out, void* avg_out, const void* state_in, void* state_out,
                              int64_t state_bytes, int64_t t_base, int64_t n_segments, void* workspace,
                              int64_t workspace_bytes, sea_stream_t stream);

/* ---- decode step with the position in DEVICE memory ------------------------------------------------------------------
 * The reference's generation loop (src/main/opt_generate.py:131 -> attention.py use_cache branches + attention_state.py)
 * runs one position per forward.  For a step that is captured ONCE as a HIP graph and replayed per token, nothing that
 * changes with the position may sit in kernel arguments: these three entry points read it from device memory instead
 * (an int32 the captured step itself increments).  Everything else of the step -- predictor MLP, the two convolutions
 * over the cached window, row scan, fused attention over K / V caches of fixed capacity -- has static arguments already.
 *
 * sea_performer_causal_step_at: sea_performer_causal_step with t_base = *t_base_dev.  k_cache / v_cache are the BASES (row 0)
 *   of the kv-caches -- they already hold the new rows -- and pos_table the BASE of the value embedding: the kernel finds
 *   the chunk boundary itself and walks the open chunk from there.  q / out / avg_out: the T new rows.  state_in / state_out
 *   may be one image (updated in place; it changes only when a chunk completes).
 * sea_predictor_tail_select_at: sea_predictor_tail_select for the LAST T rows of sequences of *t_src_dev tokens;
 *   keep_table[i] = K of the row that sees i+1 keys, for every position the session can reach (attention.py:849-866).
 *   crow_out (optional; T == 1 only): (N, 2) int32 = [0, row total] per batch item, i.e. the one-row CSR's crow -- the step then
 *   needs no sea_csr_row_scan launch.
 * sea_csr_emit_at: sea_csr_emit with the row widths following *t_src_dev and column ids = head * T_cap + key for a FIXED
 *   capacity T_cap >= *t_src_dev (the K / V caches' row count), so sea_sparse_attention is called with T_src = T_cap. */
int sea_performer_causal_step_at(const void* q, const void* k_cache, const void* v_cache, const void* pos_table, int dtype,
                                 const float* proj, int64_t N, int64_t H, int64_t T, int64_t D, int64_t nb,
                                 const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                 int64_t pos_stride, void* out, void* avg_out, const void* state_in, void* state_out,
                                 int64_t state_bytes, const int32_t* t_base_dev, sea_stream_t stream);
int sea_predictor_tail_select_at(const void* y, int dtype, int64_t N, int64_t C, int64_t H, int64_t T, int64_t W4,
                                 int64_t up, int64_t T_m, const int64_t* y_strides, const void* conv_b,
                                 const void* conv_w16, int64_t Cp, const void* gamma, const void* beta, float eps,
                                 void* probs, void* scores, const int32_t* keep_table, const int32_t* t_src_dev,
                                 int is_causal, int max_k, uint32_t* bits, int32_t* row_nnz, int32_t* head_off,
                                 int32_t* crow_out, const uint32_t* consts_tab,
                                 sea_stream_t stream);
int sea_csr_emit_at(const uint32_t* bits, const void* crow, int64_t N, int64_t H, int64_t T_dst, int64_t T_m,
                    const int32_t* t_src_dev, int64_t T_cap, int is_causal, int max_k, void* col, int idx_bytes,
                    int64_t col_stride_n, int64_t z_cap, sea_stream_t stream);

/* Algorithmic bytes of one sea_sparse_attention launch:
 * Z*(2*D*s + 4) + N*H*T_dst*(2*D*s + 4).  Host-side helper, no device work. */
int64_t sea_sparse_attention_bytes(int64_t Z, int64_t N, int64_t H, int64_t T_dst, int64_t D, int elem_bytes);

/* Glue of a graph-replayed decoding step (round 4; perlin_attention/decode.py, reference loop src/main/opt_generate.py:131).
 * sea_decode_stage: the ONE launch of a step whose arguments change (the caller's new q / k / v rows, (N,H,1,D) with element
 *   strides {n, h}, feature stride 1, 16-byte aligned rows): q is copied into q_in (N,H,D), k / v are written into
 *   kv_cache (2,N,H,capacity,D) at row counters[0] (device int32: the rows the session's state has seen).  Replaces three
 *   input copies and an index_copy_ of the framework (four launches of ~4.5 us).
 * sea_c8_window_shift: xs (N, rows, row_bytes) moved up by one row in place (xs[n, r] = xs[n, r + 1]): the predictor CNN's
 *   window after a step whose MLP wrote the new row behind it (sea_predictor_mlp with x_c8_stride_n).  `counters` (optional):
 *   two device int32 advanced by one by the same launch -- the LAST of a step, so every reader of the step is done. */
int sea_decode_stage(const void* q, const void* k, const void* v, int dtype, int64_t N, int64_t H, int64_t D,
                     const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                     void* q_in, void* kv_cache, int64_t capacity, const int32_t* counters, sea_stream_t stream);
int sea_c8_window_shift(void* xs, int64_t N, int64_t rows, int64_t row_bytes, int32_t* counters, sea_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEA_HIP_H */
