"""-m gpu: the clamp and scale edges at T = 32 768.

With T / T_M = 128 > k = 64 every pixel of a late row is wider than `max_k` and is THINNED by the reference's fp32 stepping
(`K/causal_resize_m_to_t.py:565-569,657-659`): the branch the golden `clamp` fixture pins at T = 96 only.  Context extension
to such lengths is the reference's own use (`src/trainer/perlin_trainer.py:533-566`).  Here, at full size:

  * the LAYER (OPT-125m shape, one 32 768-token sequence, bf16) through the fused estimator and the fused interpolation +
    attention launch: its CSR equals the oracle's grouped top-k + interpolation on the layer's own map bit for bit on row
    blocks from every regime (first rows, the last unthinned rows, both sides of the thinning onset t + 1 = k T_M, the last
    rows), the capacity bound holds, and sampled context rows match the oracle to 1e-3;
  * the operator path on a map that puts a row's whole budget into ONE head: the (row, head) key lists of a block overflow
    the fused kernel's LDS list (8192 entries) WITH thinned pixels, so `fused_expand` runs its fp32 stepping into the memory
    path -- columns == the emit launch's == the oracle's, outputs equal."""
import pytest
import torch

import sea_attention_amd as S
from oracle import sea_oracle as O
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention, ops
from sea_attention_amd.perlin_attention import attention as A

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T, T_M, K = 32768, 256, 64


class Cfg:
    def __init__(self, hidden, heads, max_pos):
        self.hidden_size, self.num_attention_heads, self.max_position_embeddings = hidden, heads, max_pos


def oracle_rows(probs_rows, H, t0, t1):
    """Oracle CSR of rows t0 .. t1-1 of a T-token causal sequence from their probability rows (N=1): per-row column lists
    (encoded against T) in the reference's order."""
    keep = O.keep_counts_module(H, T, T_M, K)[t0:t1]
    mask = O.grouped_topk_mask(probs_rows, keep)
    crow, col = O.resize_m_to_t_csr(mask, K, target_width=t1, is_causal=True)      # rows are the tail of a t1-token prefix
    col = (col // t1) * T + col % t1                                                  # head * t1 + key -> head * T + key
    return crow[0], col[0]


def blocks():
    on = K * T_M                                   # first row with a pixel wider than max_k: t + 1 > k T_M
    return [(0, 192), (on - 320, on - 64), (on - 64, on + 192), (T // 2 + 4000, T // 2 + 4128), (T - 256, T)]


def test_layer_at_32768_tokens_thins_every_late_pixel(monkeypatch):
    N, H, d, dtype = 1, 12, 64, torch.bfloat16
    S.seed(42)
    pc = PerlinAttentionConfig(k=K, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                               k_flatten_dim='causal_batch', context_output_method='mix')
    layer = PerlinSelfAttention(Cfg(H * d, H, T), pc).to(DEV).to(dtype).eval()
    for m in layer.modules():
        if hasattr(m, 'benchmarking'):
            m.benchmarking = True
    layer.attention.assume_not_padded = True
    S.seed(7)
    x = torch.randn((N, H, T, d), device=DEV)
    q, k, v = (x * d ** -0.5).to(dtype), torch.randn_like(x).to(dtype), torch.randn_like(x).to(dtype)
    del x
    fp_min = torch.finfo(torch.float16).min / 2
    ar = torch.arange(T, device=DEV)
    mask = ((ar.view(1, T) > ar.view(T, 1)).to(dtype) * fp_min).view(1, 1, T, T)
    seen = {}
    real = A.ops.sparse_attention

    def spy(q_, k_, v_, csr, **kw):
        seen.update(q=q_, k=k_, v=v_, csr=csr, kw=kw, pending=csr.col_is_pending)
        return real(q_, k_, v_, csr, **kw)
    monkeypatch.setattr(A.ops, "sparse_attention", spy)
    with torch.no_grad():
        out = layer(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=mask)
    torch.cuda.synchronize()
    del mask
    assert seen["pending"], "steps I + J fused: the attention launch expands (and thins) the pixels itself"
    csr = out.partial_attention_mask
    crow = csr.crow[0].cpu().long()
    z = int(crow[-1])
    assert z <= csr.col.shape[1], "analytic capacity bound (ops.z_capacity) below the real entry count"
    col = csr.col[0, :z].cpu().long()
    # late rows: K_t pixels of exactly max_k entries each
    keep = O.keep_counts_module(H, T, T_M, K)
    t_late = torch.arange(K * T_M + T_M, T)          # every pixel of these rows is wider than k
    assert torch.equal(crow[t_late + 1] - crow[t_late], (keep[t_late] * K).long())
    probs = out.estimated_attention_probs_m            # lazy: computed here
    for t0, t1 in blocks():
        ocrow, ocol = oracle_rows(probs[:, :, t0:t1].float().cpu(), H, t0, t1)
        assert torch.equal(crow[t0:t1 + 1] - crow[t0], ocrow), (t0, t1)
        assert torch.equal(col[crow[t0]:crow[t1]], ocol[:int(ocrow[-1])]), (t0, t1)
    # context rows of the last block and of the onset block against the oracle on the layer's own rounded inputs
    kw = seen["kw"]
    for t0, t1 in (blocks()[2], (T - 64, T)):
        rows = torch.arange(t0, t1)
        sub_crow = (crow[t0:t1 + 1] - crow[t0]).view(1, -1)
        sub_col = col[crow[t0]:crow[t1]].view(1, -1)
        qh, kh, vh = (seen[n_].float().cpu() for n_ in ("q", "k", "v"))
        rs = kw["row_scale"].cpu()[:, :, rows] if kw.get("row_scale") is not None else None
        sparse = O.sparse_attention(qh[:, :, rows], kh, vh, sub_crow, sub_col, rs)
        a = kw["mix"].cpu()[:, :, rows].unsqueeze(-1)
        ref = sparse * a + (1.0 - a) * kw["avg"].float().cpu()[:, :, rows]
        got = out.context_layer.view(N, T, H, d).permute(0, 2, 1, 3).cpu()[:, :, rows]
        rel = ((got - ref).norm() / ref.norm()).item()
        assert torch.isfinite(got).all() and rel < 1e-3, (t0, rel)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fused_expand_overflows_its_lds_list_with_thinned_pixels(dtype):
    """One head takes a row's whole budget: 64 rows x K_t pixels x max_k entries do not fit the block's 8192-entry list."""
    N, H, d = 1, 4, 64
    g = torch.Generator().manual_seed(5)
    T_dst = 1024                                                        # the LAST 1024 rows of a 32768-token sequence
    probs = torch.rand((N, H, T_dst, T_M), generator=g) * 1e-3
    probs[:, 1] += 1.0                                                  # head 1 outranks every other head's pixels
    probs = torch.softmax(probs * 8, -1).to(dtype if dtype != torch.float32 else torch.float32)
    # a denser budget than the module's (k_oversample-like): 200 pixels per row, all of them land in head 1
    keep = torch.full((T_dst,), 200, dtype=torch.int32)
    q = (torch.randn((N, H, T_dst, d), generator=g) * d ** -0.5).to(dtype)
    kk = torch.randn((N, H, T, d), generator=g).to(dtype)
    v = torch.randn((N, H, T, d), generator=g).to(dtype)
    z_cap = ops.z_capacity(keep, H, T_dst, T, T_M, K, True)
    pd, kd = probs.to(DEV), keep.to(DEV)
    c_emit, _ = ops.topk_to_csr(pd, kd, K, target_width=T, is_causal=True, z_cap=z_cap)
    o_emit = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), c_emit, path="gather")
    c_fused = ops.csr_from_selection(c_emit.bits, c_emit.row_nnz, c_emit.head_off, H, T_M, T, K, True, z_cap, defer_emit=True)
    assert c_fused.col_is_pending
    o_fused = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), c_fused, path="gather")
    torch.cuda.synchronize()
    assert not c_fused.col_is_pending
    zz = int(c_emit.crow[0, -1])
    assert zz == 1024 * 200 * K                                         # every kept pixel thinned to max_k entries
    assert zz / (T_dst / 64) > 8192 * 4                                 # per 64-row block: far beyond the LDS list
    assert torch.equal(c_fused.col[0, :zz], c_emit.col[0, :zz])
    assert torch.equal(o_fused, o_emit)
    # the oracle on a slice of rows (their absolute widths: the tail of a T-token sequence)
    t0, t1 = T - 96, T
    mask_m = O.grouped_topk_mask(probs[:, :, -96:].float(), keep[-96:])
    ocrow, ocol = O.resize_m_to_t_csr(mask_m, K, target_width=T, is_causal=True)
    cr = c_emit.crow[0].cpu().long()
    r0 = T_dst - 96
    assert torch.equal(cr[r0:] - cr[r0], ocrow[0])
    assert torch.equal(c_emit.col[0, cr[r0]:zz].cpu().long(), ocol[0, :int(ocrow[0, -1])])
    ref = O.sparse_attention(q[:, :, r0:].float(), kk.float(), v.float(), ocrow, ocol, None)
    got = o_fused[:, :, r0:].float().cpu()
    assert ((got - ref).norm() / ref.norm()).item() < (1e-3 if dtype != torch.float32 else 1e-5)
