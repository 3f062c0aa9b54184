"""CPU oracle for SEA's sparse-attention hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain restatement, in CPU torch/numpy, of the algorithm the
reference (gmlwns2000/sea-attention) runs on its sparse path.  It exists to
CHECK the HIP implementation; nothing in the product package
(`sea-attention_amd/`) imports it.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module.

Parity status: PINNED.  Every function below that has a counterpart among the
reference's operators is checked in `tests/test_oracle_golden.py` against
golden vectors produced by running the reference's own operators in the build
container (`tests/golden/make_golden.py`, Triton interpreter).  Functions that
restate parts of `PerlinAttention.forward` which cannot be imported from the
reference here (module-level top-k, mix epilogue) say so in their docstring;
they are pinned only by their equivalence to the pinned pieces.

Not restated here, and how each is pinned instead:
  * predictor CNN (steps E-F): the reference's own `PA/modules.py` classes, run in place -> tests/golden/estimator.npz;
  * the callers (`OPTAttention`, `OPTDecoderLayer`, `benchmark_bert.exam`'s surgery): the reference's own
    `src/models/perlin_opt/perlin_opt.py`, run in place ON THIS PACKAGE -> tests/golden/callers.npz;
  * the causal Performer (step B, third-party `performer-pytorch==1.1.4`, absent from the tree): HALF pinned.  The prefix
    sums / eps placement / denominator are pinned to the reference's own restatement of that arithmetic
    (`StatefulCausalPerformer`, PA/attention_state.py:43-122, run in place -> tests/golden/performer.npz).  The FEATURE MAP
    `phi(x) = relu(d^-1/4 x W^T) + 1e-3` (`generalized_kernel`) is PARITY UNPINNED: no code under /root/reference evaluates
    it (the reference's stateful path takes features as given, :84-98), so it follows the package's published formula only.

Citations are `path:line` under the reference root; shorthand
  PA/ = src/models/perlin_attention/      K/ = PA/ops/kernels/

Index arithmetic is integer / fp32 exactly as the reference computes it:
  * interpolation boundaries: fp32 `round_half_away(b * fp32((t+1)/T_M))`
    (K/causal_resize_m_to_t.py:642,654-655, libdevice roundf via triton_round)
  * per-row keep count, module path: fp32
    `clamp_min(round_half_even(H * (k*os*T_M / (t+1))), 1)` (PA/attention.py:849,856,866)
  * per-row keep count, kernel-test path:
    `clamp(H*floor(k*T_M/(t+1)), 1, H*T_M)` (K/causal_topk_masking.py:31)
Tie rule (the reference leaves it to torch.sort(stable=False)): among equal
probabilities the LOWER flat index (head-major, then pixel) is kept first.
"""
import math
import numpy as np
import torch

FP_MIN_F16_HALF = torch.finfo(torch.float16).min * 0.5


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def round_half_away(x: torch.Tensor) -> torch.Tensor:
    """libdevice roundf (Triton 2.0 tl.math.round), K/causal_resize_m_to_t.py:215-238."""
    return torch.where(x >= 0, torch.floor(x + 0.5), torch.ceil(x - 0.5))


def target_widths(T_DST: int, T_SRC: int, is_causal: bool) -> torch.Tensor:
    """K/causal_resize_m_to_t.py:951-955: width of the key window of every query row."""
    if is_causal:
        return torch.arange(1, T_SRC + 1)[-T_DST:]
    return torch.full((T_SRC,), T_SRC)[-T_DST:]


def pixel_bounds(T_DST: int, T_SRC: int, T_M: int, is_causal: bool):
    """v_starts / v_ends tables, (T_DST, T_M) fp32 -> int64.

    K/causal_resize_m_to_t.py:642 `scales = target_width / original_width` (int64 / int
    -> fp32 true division), :652-655 `triton_round(b*scales)`, `triton_round((b+1)*scales)`.
    """
    tw = target_widths(T_DST, T_SRC, is_causal)
    scales = tw / T_M                                   # fp32
    b = torch.arange(0, T_M).view(1, T_M)
    v_starts = round_half_away(b * scales.view(T_DST, 1))
    v_ends = round_half_away((b + 1) * scales.view(T_DST, 1))
    return v_starts, v_ends                             # fp32 tables, as in the reference


def keep_counts_module(H, T_DST, T_M, k, k_oversample=1.0):
    """Per-row keep count K_t of the module path, causal / 'causal_batch'.

    PA/attention.py:800 `causal_token_length = arange(1, T_DST+1)` (int64),
    :849 `H * (k * k_oversample * T_M / causal_token_length)` (python float / int64
    tensor -> fp32), :856 torch.round (half-to-even), :866 clamp_min(1).
    Returned as fp32 (the comparison at :916 is `int64_rank < fp32_K`).
    """
    ctl = torch.arange(1, T_DST + 1, dtype=torch.long)
    per = H * (k * k_oversample * T_M / ctl)
    per = torch.round(per)
    per = torch.clamp_min(per, 1)
    return per


def keep_counts_kernel_test(H, T_DST, T_M, k):
    """K/causal_topk_masking.py:26-37 (causal branch, causal mask = lower triangle)."""
    ctl = torch.arange(1, T_DST + 1, dtype=torch.long)
    per = torch.clamp(H * torch.floor(k * T_M / ctl), 1, H * T_M)
    per = torch.clamp_min(per, 1)
    return per


def keep_counts_kernel_test_noncausal(H, T_M, k, token_length):
    """K/causal_topk_masking.py:33-34 (non-causal): one count per batch item."""
    per = H * torch.round(k * T_M / token_length)
    return torch.clamp_min(per, 1)


# --------------------------------------------------------------------------
# a6: grouped top-k  (PA/attention.py:774-947, K/causal_topk_masking.py:3-77)
# --------------------------------------------------------------------------
def grouped_topk_mask(probs: torch.Tensor, keep: torch.Tensor) -> torch.Tensor:
    """0/1 fp32 mask (N,H,T,T_M): per (n,t) keep the `keep[n or 0, t]` largest of the
    H*T_M pooled probabilities.

    probs (N,H,T,T_M); keep: (T,) or (N,T) or (N,1) counts (any real dtype).
    Restates: t = probs.transpose(1,2).reshape(N,T,H*T_M) (PA/attention.py:844);
    descending sort, rank scatter (:879-907); alive = rank < per_item_top_k (:916-917);
    view back (N,T,H,T_M)->(N,H,T,T_M) (:921).
    Tie rule: stable sort => lower flat index first.
    """
    N, H, T, T_M = probs.shape
    t = probs.transpose(1, 2).reshape(N, T, H * T_M)
    _, indices = torch.sort(t.float(), dim=-1, descending=True, stable=True)
    rank = torch.empty_like(indices)
    rank.scatter_(-1, indices, torch.arange(H * T_M).view(1, 1, -1).expand_as(indices))
    keep = keep.to(torch.float32)
    if keep.ndim == 1:
        keep = keep.view(1, -1, 1)
    elif keep.ndim == 2:
        keep = keep.view(keep.shape[0], -1, 1)
    alive = rank < keep
    return alive.float().view(N, T, H, T_M).transpose(1, 2).contiguous()


# --------------------------------------------------------------------------
# a8: dense twin  (K/resize_m_to_t.py:6-73), no training jitter, no oversample thinning
# --------------------------------------------------------------------------
def resize_m_to_t_dense(x: torch.Tensor, masked_fill_value: float, attention_mask: torch.Tensor,
                        target_width=None, is_causal=True) -> torch.Tensor:
    N, H, T1, T_M = x.shape
    T2 = target_width if target_width is not None else T1
    if not is_causal:
        attention_mask = attention_mask.expand(N, 1, T1, T2)
    mask = (attention_mask > -1).float()
    mask_cs = mask.cumsum(-1)
    token_length = mask_cs[:, :, :, -1].unsqueeze(-1)
    idx = torch.floor(((mask_cs - 1) + 0.5) / token_length * T_M - 1e-4).to(torch.long) \
        + ((1 - mask) * T_M).to(torch.long)                                     # :46
    idx = torch.clamp(idx, 0, T_M).expand(N, H, T1, T2)
    grid_input = torch.nn.functional.pad(x, pad=(0, 1), value=masked_fill_value)
    return grid_input.gather(dim=-1, index=idx)


# --------------------------------------------------------------------------
# a7: mask -> flat CSR  (K/causal_resize_m_to_t.py:631-762, :493-572)
# --------------------------------------------------------------------------
def resize_m_to_t_csr(mask_m: torch.Tensor, k: int, target_width=None, is_causal=True):
    """Return (crow int64 (N,T_DST+1), col int64 (N,Z)) exactly as the reference lays them out.

    * n_pixels = (v_end - v_start) * mask, clamped to max_k = k   (:657-659)
    * inclusive cumsum over the flattened (t, h, b) order          (:664)
    * Z = max over batch of the total, shorter items zero-padded   (:667-669)
    * crow[:,1:] = cumsum at row ends                              (:672)
    * entry i of a pixel: range_end - int32(i * ((range_end-range_start)/col_len)) - 1,
      range_* = v_* + h*T_SRC, all fp32                             (:565-569)
    """
    N, H, T_DST, T_M = mask_m.shape
    T_SRC = target_width if target_width is not None else T_DST
    vs, ve = pixel_bounds(T_DST, T_SRC, T_M, is_causal)               # (T_DST,T_M) fp32
    x = mask_m.transpose(1, 2).reshape(N, T_DST, H, T_M)
    n_pixels = (ve - vs).view(1, T_DST, 1, T_M).to(torch.int32) * x.to(torch.int32)
    n_pixels = torch.clamp_max(n_pixels, k)
    flat = n_pixels.reshape(N, -1).to(torch.long)
    pixel_indices = flat.cumsum(-1)                                    # inclusive
    row_end = pixel_indices.view(N, T_DST, -1)[:, :, -1]
    Z = int(row_end.max().item()) if row_end.numel() else 0
    crow = torch.zeros((N, T_DST + 1), dtype=torch.long)
    crow[:, 1:] = row_end
    col = torch.zeros((N, Z), dtype=torch.long)
    hb = H * T_M
    for n in range(N):
        nz = flat[n].nonzero().view(-1)                                # flat pixel ids, ascending
        if nz.numel() == 0:
            continue
        cnt = flat[n][nz]
        start = pixel_indices[n][nz] - cnt
        t_idx = nz // hb
        h_idx = (nz % hb) // T_M
        b_idx = nz % T_M
        rs = vs[t_idx, b_idx] + (h_idx * T_SRC).float()
        re = ve[t_idx, b_idx] + (h_idx * T_SRC).float()
        pix = torch.repeat_interleave(torch.arange(nz.numel()), cnt)
        i = torch.arange(pix.numel()) - start[pix]
        step = (re - rs) / cnt.float()
        val = re[pix] - (i.float() * step[pix]).to(torch.int32) - 1    # fp32 - int32 -> fp32
        col[n, start[pix] + i] = val.to(torch.long)
    return crow, col


def head_offsets(crow: torch.Tensor, col: torch.Tensor, H: int, T_SRC: int) -> torch.Tensor:
    """Per-(row, head) start offsets inside a row, (N,T_DST,H+1) int64.
    The reference derives the same thing with a counting pass
    (`__flat_csr_sdbmm_tch_compute`, K/flat_csr_sdbmm.py:48-127)."""
    N, R1 = crow.shape
    R = R1 - 1
    out = torch.zeros((N, R, H + 1), dtype=torch.long)
    for n in range(N):
        z = int(crow[n, -1])
        rows = torch.repeat_interleave(torch.arange(R), crow[n, 1:] - crow[n, :-1])
        heads = col[n, :z] // T_SRC
        cnt = torch.zeros(R * H, dtype=torch.long)
        cnt.index_add_(0, rows * H + heads, torch.ones(z, dtype=torch.long))
        out[n, :, 1:] = cnt.view(R, H).cumsum(-1)
    return out


# --------------------------------------------------------------------------
# a14: flat_csr_to_dense  (K/flat_csr_to_dense.py:3-36) -- of the VALID entries
# --------------------------------------------------------------------------
def flat_csr_to_dense(crow, col, values, T_SRC, H):
    """(N,H,T_DST,T_SRC).  Only entries below crow[n,-1] are scattered; the reference's
    helper also pushes a shorter item's zero padding through torch's to_dense()
    (landing on [n,0,last_row,0]) -- that artefact of the debug helper is not restated."""
    N, R1 = crow.shape
    R = R1 - 1
    out = torch.zeros((N, H, R, T_SRC), dtype=values.dtype)
    for n in range(N):
        z = int(crow[n, -1])
        rows = torch.repeat_interleave(torch.arange(R), crow[n, 1:] - crow[n, :-1])
        c = col[n, :z]
        out[n].index_put_((c // T_SRC, rows, c % T_SRC), values[n, :z], accumulate=True)
    return out


# --------------------------------------------------------------------------
# a9..a12: the four CSR operators
# --------------------------------------------------------------------------
def _rows_of(crow_n):
    R = crow_n.numel() - 1
    return torch.repeat_interleave(torch.arange(R), crow_n[1:] - crow_n[:-1])


def csr_sddmm(q, k, crow, col):
    """K/flat_csr_masked_bmm.py:8-27: values[n,ic] = q[n,h,row] . k[n,h,colidx]; fp32 accumulate.
    Padding entries (>= crow[n,-1]) keep the mask's value 1.0 (values().clone(), :151)."""
    N, H, T_DST, D = q.shape
    T_SRC = k.shape[2]
    out = torch.ones(col.shape, dtype=torch.float32)
    for n in range(N):
        z = int(crow[n, -1])
        rows = _rows_of(crow[n])
        c = col[n, :z]
        h = c // T_SRC
        out[n, :z] = (q[n, h, rows].float() * k[n, h, c % T_SRC].float()).sum(-1)
    return out


def csr_softmax(values, crow, col, H, T_SRC):
    """K/flat_csr_softmax.py:12-43,55-125: softmax over each (row, head) segment."""
    out = values.clone()
    N = crow.shape[0]
    R = crow.shape[1] - 1
    for n in range(N):
        z = int(crow[n, -1])
        if z == 0:
            continue
        rows = _rows_of(crow[n])
        seg = rows * H + col[n, :z] // T_SRC
        v = values[n, :z].float()
        mx = torch.full((R * H,), -float("inf"))
        mx.scatter_reduce_(0, seg, v, reduce="amax")
        e = torch.exp(v - mx[seg])
        den = torch.zeros(R * H)
        den.index_add_(0, seg, e)
        out[n, :z] = e / den[seg]
    return out


def csr_elmul(values, crow, col, other, T_SRC):
    """K/flat_csr_elmul.py:6-28: values *= other[n, head, row, colidx]."""
    out = values.clone()
    for n in range(crow.shape[0]):
        z = int(crow[n, -1])
        rows = _rows_of(crow[n])
        c = col[n, :z]
        out[n, :z] = values[n, :z] * other[n, c // T_SRC, rows, c % T_SRC]
    return out


def csr_spmm(values, crow, col, v, T_SRC):
    """K/flat_csr_sdbmm.py:141-313: out[n,h,row,:] = sum_j p_j v[n,h,col_j,:]; output fp32
    (`torch.zeros` without dtype, :347).  The reference drops a head's entries beyond
    MAX_ROW_T (:382-388); that truncation is a launch heuristic, not restated."""
    N, H, _, D = v.shape
    R = crow.shape[1] - 1
    out = torch.zeros((N, H, R, D), dtype=torch.float32)
    for n in range(N):
        z = int(crow[n, -1])
        rows = _rows_of(crow[n])
        c = col[n, :z]
        h = c // T_SRC
        contrib = values[n, :z].float().unsqueeze(-1) * v[n, h, c % T_SRC].float()
        out[n].view(H * R, D).index_add_(0, h * R + rows, contrib)
    return out


def sparse_attention(q, k, v, crow, col, row_scale=None):
    """Composition used by the module's sparse branch (PA/attention.py:1158-1173):
    SDDMM -> per-(row,head) softmax -> * sigmoid-scale[n,h,row] -> SpMM.  fp32 out."""
    N, H, T_DST, D = q.shape
    T_SRC = k.shape[2]
    s = csr_sddmm(q, k, crow, col)
    p = csr_softmax(s, crow, col, H, T_SRC)
    if row_scale is not None:
        other = row_scale.view(N, H, T_DST, 1).expand(N, H, T_DST, T_SRC)
        p = csr_elmul(p, crow, col, other, T_SRC)
    return csr_spmm(p, crow, col, v, T_SRC)


# --------------------------------------------------------------------------
# a13: average-pool mix (PA/attention.py:1236-1244) -- not importable from the reference
# --------------------------------------------------------------------------
def cumavg(v):
    """avg_v.cumsum(-2) / arange(1..T)  (PA/attention.py:1220-1222), no padding."""
    T = v.shape[-2]
    return (v.float().cumsum(-2) / torch.arange(1, T + 1).view(1, 1, -1, 1)).to(v.dtype)


def mix(partial, v, scale1):
    """partial*sigma(s1) + (1-sigma(s1))*cumavg(v)  (PA/attention.py:1236-1237)."""
    a = torch.sigmoid(scale1).unsqueeze(-1)
    return partial * a + (1 - a) * cumavg(v)


# --------------------------------------------------------------------------
# Estimator pieces that became HIP kernels: plain fp32 torch references of the same ops
# (floating-point kernels: "plain PyTorch fp32 reference"), following the reference's module chain.
# --------------------------------------------------------------------------
def split_layernorm(x, splits, weight, bias, eps=1e-5):
    """ChannelSplit (PA/attention.py:123-131) + nn.LayerNorm over the last dim (cnn.lnorm1, :266)."""
    N, C, H, W = x.shape
    y = x.float().view(N, C, H, splits, W // splits).permute(0, 1, 3, 2, 4).reshape(N, C * splits, H, W // splits)
    return torch.nn.functional.layer_norm(y, (W // splits,), weight.float(), bias.float(), eps)


def predictor_tail(y, conv_w_full, conv_b, ln_w, ln_b, up, T_m, eps=1e-5):
    """UpsampleFP32((1,up)) -> CausalConv2d(C->H, k=1, padding=1) -> KeepRes resize ('area', since
    T_m < T_m+2) -> LayerNorm(T_m) -> softmax  (PA/attention.py:271-281,670-673; PA/modules.py:12-55,77-92,96-192).
    conv_w_full is the module's (H, C, 1, 1) weight.  Returns (probs, scores) in fp32."""
    F = torch.nn.functional
    x = F.interpolate(y.float(), scale_factor=(1, up), mode='nearest')
    x = F.conv2d(x, conv_w_full.float(), conv_b.float(), 1, (0, 1), 1)
    x = F.interpolate(x, (x.shape[-2], T_m), mode='area')
    s = F.layer_norm(x, (T_m,), ln_w.float(), ln_b.float(), eps)
    return torch.softmax(s, -1), s


# --------------------------------------------------------------------------
# Dense restatement of the whole kernel-level path: this is the CPU BASELINE
# (BASELINE.md section 2): sort-based grouped top-k, gather-based interpolation,
# dense matmul + additive mask + softmax + matmul (PA/attention.py:1066-1133).
# --------------------------------------------------------------------------
def dense_path(probs, q, k, v, row_scale, kk, k_oversample=1.0, head_chunk=None):
    """probs (N,H,T,T_M) -> context (N,H,T,D) fp32 through the reference's dense branch."""
    N, H, T, T_M = probs.shape
    keep = keep_counts_module(H, T, T_M, kk, k_oversample)
    mask_m = grouped_topk_mask(probs, keep)                                  # 0/1
    fp_min = torch.finfo(torch.float32).min / 2
    add_m = (1.0 - mask_m) * fp_min                                          # PA/attention.py:913
    causal = ((torch.arange(T).view(1, T) > torch.arange(T).view(T, 1)) * fp_min).view(1, 1, T, T)
    out = torch.empty((N, H, T, q.shape[-1]), dtype=torch.float32)
    hc = head_chunk or H
    for h0 in range(0, H, hc):
        sl = slice(h0, h0 + hc)
        pm = resize_m_to_t_dense(add_m[:, sl], fp_min, causal.expand(N, 1, T, T), T, True)  # :956
        pm = pm.masked_fill(causal < -1, fp_min)                                           # :958
        s = torch.matmul(q[:, sl].float(), k[:, sl].float().transpose(-1, -2)) + pm         # :1066,:1113
        p = torch.softmax(s, -1).masked_fill_(pm < -1, 0)                                   # :1115-1117
        if row_scale is not None:
            p = p * row_scale[:, sl].unsqueeze(-1)                                          # :1125
        out[:, sl] = torch.matmul(p, v[:, sl].float())                                      # :1129
    return out, mask_m
