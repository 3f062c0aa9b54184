"""-m gpu: HIP path vs the CPU oracle on seeded inputs (sizes the oracle finishes in seconds), and
size-independent properties at the full BASELINE configurations.

Tolerances (north_star): top-k / CSR indices bit-exact; attention output within 1e-3 relative
(fp32 inputs: 1e-4 abs; bf16/fp16 inputs are compared against the oracle run on the SAME rounded
inputs, so only accumulation order differs: 2e-3 abs on O(1) outputs)."""
import math

import numpy as np
import pytest
import torch

from oracle import sea_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from sea_attention_amd.perlin_attention import ops
    return ops


def _inputs(N, H, T, T_M, d, seed=42, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    probs = torch.softmax(torch.randn((N, H, T, T_M), generator=g), -1)
    q = (torch.randn((N, H, T, d), generator=g) * d ** -0.5).to(dtype)
    k = torch.randn((N, H, T, d), generator=g).to(dtype)
    v = torch.randn((N, H, T, d), generator=g).to(dtype)
    rs = torch.sigmoid(torch.randn((N, H, T), generator=g))
    return probs, q, k, v, rs


@pytest.mark.parametrize("N,H,T,T_M,k,d", [(1, 12, 512, 64, 16, 64), (2, 12, 1024, 256, 64, 64), (1, 5, 300, 96, 8, 80),
                                           (1, 32, 1024, 256, 64, 64), (1, 40, 768, 256, 64, 128)])
def test_topk_csr_vs_oracle(ops, N, H, T, T_M, k, d):
    probs, *_ = _inputs(N, H, T, T_M, 8)
    keep = O.keep_counts_module(H, T, T_M, k)
    mask_ref = O.grouped_topk_mask(probs, keep)
    crow_ref, col_ref = O.resize_m_to_t_csr(mask_ref, k, T, True)
    keep_dev = ops.keep_table_causal(H, T, T_M, k, device=DEV)
    csr, mask = ops.topk_to_csr(probs.to(DEV), keep_dev, k, target_width=T, want_mask=True)
    assert torch.equal(mask.cpu(), mask_ref)
    assert torch.equal(csr.crow.cpu().long(), crow_ref)
    z = crow_ref[:, -1]
    for n in range(N):
        assert torch.equal(csr.col[n, :z[n]].cpu().long(), col_ref[n, :z[n]])
    assert csr.col.shape[1] >= int(z.max())                    # analytic capacity really is an upper bound
    assert torch.equal(csr.head_off.cpu().long(), O.head_offsets(crow_ref, col_ref, H, T))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_topk_low_precision_probs_and_ties(ops, dtype):
    """bf16/fp16 probability maps carry many exact ties: the tie rule (lower flat index first) must hold."""
    N, H, T, T_M, k = 1, 8, 256, 64, 8
    probs, *_ = _inputs(N, H, T, T_M, 8, seed=3)
    probs = probs.to(dtype)
    keep = O.keep_counts_module(H, T, T_M, k)
    mask_ref = O.grouped_topk_mask(probs.float(), keep)
    mask = ops.topk_mask(probs.to(DEV), ops.keep_table_causal(H, T, T_M, k, device=DEV), k)
    assert torch.equal(mask.cpu(), mask_ref)
    # degenerate rows: all equal, and a two-valued row
    flat = torch.full((1, 2, 16, 32), 0.25)
    flat[0, :, 8:, ::2] = 0.5
    keep2 = torch.tensor([5] * 8 + [40] * 8, dtype=torch.int32)
    ref2 = O.grouped_topk_mask(flat, keep2)
    got2 = ops.topk_mask(flat.to(DEV), keep2.to(DEV), 4)
    assert torch.equal(got2.cpu(), ref2)


@pytest.mark.parametrize("levels", [1, 2, 4, 8, 16, 37])
def test_topk_overfull_threshold_bin(ops, levels):
    """Few distinct values over H*T_m = 8192 pixels: the threshold bin holds thousands of equal keys, which
    forces the multi-pass fallback and the ordered tie scan of the select kernel."""
    N, H, T, T_M, k = 1, 32, 96, 256, 64
    g = torch.Generator().manual_seed(11)
    vals = torch.rand(levels, generator=g) + 0.01
    probs = vals[torch.randint(0, levels, (N, H, T, T_M), generator=g)]
    keep = O.keep_counts_module(H, T, T_M, k)
    keep[40:] = torch.tensor([1, 2, 100, 512, 1023, 1024, 1025, 5000] * 7)       # also ranks around the list capacity (1024; 8 / 16 levels: ~1024 / ~512 keys per bin)
    ref = O.grouped_topk_mask(probs, keep)
    got = ops.topk_mask(probs.to(DEV), keep.to(torch.int32).to(DEV), k)
    assert torch.equal(got.cpu(), ref)
    # negative values and signed zeros order like floats
    x = torch.randn((1, 4, 33, 64), generator=g)
    x[0, 0, :, :8] = 0.0
    x[0, 1, :, :8] = -0.0
    keep2 = torch.full((33,), 100, dtype=torch.int32)
    assert torch.equal(ops.topk_mask(x.to(DEV), keep2.to(DEV), 4).cpu(), O.grouped_topk_mask(x, keep2))


@pytest.mark.parametrize("T,T_M", [(4096, 256), (1000, 96), (130, 256), (8192, 256)])
def test_interpolation_bounds_every_pixel(ops, T, T_M):
    """An all-ones mask exercises every (t, b) boundary: the kernel's fp32 bound arithmetic must equal the
    reference's round_half_away(b * fp32((t+1)/T_m)) table everywhere (T=8192: widths up to 32)."""
    H, k = 1, 64
    mask = torch.ones((1, H, T, T_M))
    crow_ref, col_ref = O.resize_m_to_t_csr(mask, k, T, True)
    csr = ops.resize_from_m_to_t_csr(mask.to(DEV), 0, k, target_width=T)
    assert torch.equal(csr.crow_indices().cpu(), crow_ref)
    assert torch.equal(csr.col_indices().cpu(), col_ref)


@pytest.mark.parametrize("dtype,atol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-3), (torch.float16, 2e-3)])
@pytest.mark.parametrize("N,H,T,T_M,k,d", [(2, 12, 1024, 256, 64, 64), (1, 4, 512, 64, 16, 128), (1, 3, 256, 32, 8, 80),
                                           (1, 2, 256, 32, 8, 32)])
def test_sparse_attention_vs_oracle(ops, dtype, atol, N, H, T, T_M, k, d):
    probs, q, kk, v, rs = _inputs(N, H, T, T_M, d, dtype=dtype)
    keep = O.keep_counts_module(H, T, T_M, k)
    crow, col = O.resize_m_to_t_csr(O.grouped_topk_mask(probs, keep), k, T, True)
    ref = O.sparse_attention(q.float(), kk.float(), v.float(), crow, col, rs)
    csr, _ = ops.topk_to_csr(probs.to(DEV), ops.keep_table_causal(H, T, T_M, k, device=DEV), k, target_width=T)
    out = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, row_scale=rs.to(DEV))
    assert out.dtype == torch.float32
    err = (out.cpu() - ref).abs().max().item()
    assert err < atol, err
    rel = ((out.cpu() - ref).norm() / ref.norm()).item()
    assert rel < 1e-3, rel
    # strided q/k/v views (as cut from a fused qkv projection) give the same result
    big = torch.randn((N, T, 3, H, d), dtype=dtype)
    big[:, :, 0] = q.permute(0, 2, 1, 3); big[:, :, 1] = kk.permute(0, 2, 1, 3); big[:, :, 2] = v.permute(0, 2, 1, 3)
    bd = big.to(DEV)
    qv, kv, vv = (bd[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    out2 = ops.sparse_attention(qv, kv, vv, csr, row_scale=rs.to(DEV))
    assert torch.equal(out2, out)


def test_unfused_chain_equals_fused(ops):
    N, H, T, T_M, k, d = 1, 12, 512, 64, 16, 64
    probs, q, kk, v, rs = _inputs(N, H, T, T_M, d)
    keep_dev = ops.keep_table_causal(H, T, T_M, k, device=DEV)
    csr, mask = ops.topk_to_csr(probs.to(DEV), keep_dev, k, target_width=T, want_mask=True)
    fused = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, row_scale=rs.to(DEV))
    m = ops.resize_from_m_to_t_csr(mask, 0, k, target_width=T)
    s = ops.flat_csr_masked_bmm(q.to(DEV), kk.to(DEV), m)
    p = ops.flat_csr_softmax(s, H, T)
    p = ops.flat_csr_elmul(p, rs.to(DEV).view(N, H, T, 1).expand(N, H, T, T))
    o = ops.flat_csr_sdbmm(p, v.to(DEV), T_M)
    assert (o - fused).abs().max().item() < 1e-5


# ------------------------------------------------------------------------------------------------
# full BASELINE sizes: size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,T,T_M,k,d,dtype", [
    (1, 32, 4096, 256, 64, 64, torch.bfloat16),      # cfg 3  OPT-1.3B
    (2, 12, 2048, 256, 64, 64, torch.float32),       # cfg 2  synthetic (2 of the 8 batch items)
    (1, 32, 8192, 256, 64, 80, torch.bfloat16),      # cfg 4  OPT-2.7B shape, one item
    (1, 40, 4096, 256, 64, 128, torch.bfloat16),     # cfg 5  LLaMA-13B shape
    (8, 12, 2048, 256, 64, 64, torch.float32),       # cfg 2 exactly as BASELINE.json writes it (B = 8)
    (8, 32, 4096, 256, 64, 64, torch.bfloat16),      # cfg 3 at the batch bench.py times (8 sequences per GPU)
])
def test_full_size_properties(ops, N, H, T, T_M, k, d, dtype):
    g = torch.Generator(device=DEV).manual_seed(42)
    probs = torch.softmax(torch.randn((N, H, T, T_M), generator=g, device=DEV), -1)
    keep = ops.keep_table_causal(H, T, T_M, k, device=DEV)
    csr, mask = ops.topk_to_csr(probs, keep, k, target_width=T, want_mask=True)
    # (1) exactly K_t pixels survive in every row, and they are the K_t largest: min kept >= max dropped
    kept = mask.transpose(1, 2).reshape(N, T, H * T_M)
    assert torch.equal(kept.sum(-1).to(torch.int32), keep.view(1, T).expand(N, T))
    flat = probs.transpose(1, 2).reshape(N, T, H * T_M)
    lo_kept = torch.where(kept > 0, flat, torch.full_like(flat, 2.0)).min(-1).values
    hi_drop = torch.where(kept > 0, torch.full_like(flat, -1.0), flat).max(-1).values
    assert torch.all(lo_kept >= hi_drop)
    # (2) CSR structure: crow monotone, row nnz == head_off totals, heads ascending, columns causal & unique
    crow = csr.crow.long()
    assert torch.all(crow[:, 1:] >= crow[:, :-1]) and torch.all(crow[:, 0] == 0)
    assert torch.equal(csr.head_off[:, :, -1].long(), crow[:, 1:] - crow[:, :-1])
    assert torch.all(csr.head_off[:, :, 1:] >= csr.head_off[:, :, :-1])
    Z = int(crow[0, -1])
    col = csr.col[0, :Z].long()
    rows = torch.repeat_interleave(torch.arange(T, device=DEV), crow[0, 1:] - crow[0, :-1])
    head, key = col // T, col % T
    assert torch.all(key <= rows)                                        # causal: key index <= query index
    code = (rows * H + head) * T + key
    assert torch.unique(code).numel() == Z                               # no duplicate (row, head, key)
    seg = rows * H + head
    assert torch.all(seg[1:] >= seg[:-1])                                # head-grouped, rows ascending
    # densified CSR == dense twin of the compressed mask (max_k clamp idle at these shapes), per sampled rows
    from sea_attention_amd.perlin_attention.ops import resize_from_m_to_t
    rsel = torch.tensor([0, 1, 63, 64, 255, 256, 257, T // 2, T - 2, T - 1], device=DEV)
    fp_min = torch.finfo(torch.float16).min / 2
    cm = ((torch.arange(T, device=DEV).view(1, T) > rsel.view(-1, 1)) * fp_min).view(1, 1, len(rsel), T)
    dense_rows = resize_from_m_to_t(mask[:1, :, rsel], 0, cm, target_width=T, is_causal=True, k=k, oversampled=1.0)
    dense_rows = dense_rows.masked_fill(cm < -1, 0)
    for i, t in enumerate(rsel.tolist()):
        s, e = int(crow[0, t]), int(crow[0, t + 1])
        got = torch.zeros((H, T), device=DEV)
        got[csr.col[0, s:e].long() // T, csr.col[0, s:e].long() % T] = 1
        assert torch.equal(got, dense_rows[0, :, i])
    # (3) attention: rows are convex combinations -> with V = 1 every non-empty (row, head) returns exactly rs
    q = (torch.randn((N, H, T, d), generator=g, device=DEV) * d ** -0.5).to(dtype)
    kk = torch.randn((N, H, T, d), generator=g, device=DEV).to(dtype)
    ones = torch.ones((N, H, T, d), device=DEV, dtype=dtype)
    rs = torch.sigmoid(torch.randn((N, H, T), generator=g, device=DEV))
    out1 = ops.sparse_attention(q, kk, ones, csr, row_scale=rs)
    nonempty = (csr.head_off[:, :, 1:] > csr.head_off[:, :, :-1]).transpose(1, 2)       # (N,H,T)
    expect = (rs * nonempty).unsqueeze(-1).expand_as(out1)
    assert (out1 - expect).abs().max().item() < 2e-5
    # (4) linearity in V: attn(a*V1 + V2) == a*attn(V1) + attn(V2)  (same probabilities)
    v1 = torch.randn((N, H, T, d), generator=g, device=DEV).to(dtype)
    v2 = torch.randn((N, H, T, d), generator=g, device=DEV).to(dtype)
    o1, o2 = ops.sparse_attention(q, kk, v1, csr), ops.sparse_attention(q, kk, v2, csr)
    o12 = ops.sparse_attention(q, kk, (v1.float() * 0.5 + v2.float()).to(dtype), csr)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert (o12 - (0.5 * o1 + o2)).abs().max().item() < tol
    # (5) determinism: two launches are bit-identical
    assert torch.equal(o1, ops.sparse_attention(q, kk, v1, csr))
    # (5b) batch independence: item n of the batched launch == the same item run alone, bit for bit; and for 16-bit data
    #      the two kernel paths (MFMA tile / row gather) agree to rounding on the whole output
    if N > 1:
        from sea_attention_amd.perlin_attention.ops.flat_csr import FlatCSR
        for n in (0, N - 1):
            one = FlatCSR(csr.crow[n:n + 1], csr.col[n:n + 1], csr.head_off[n:n + 1], H, T)
            alone = ops.sparse_attention(q[n:n + 1], kk[n:n + 1], v1[n:n + 1], one)
            assert torch.equal(alone, o1[n:n + 1])
    if dtype != torch.float32:
        og = ops.sparse_attention(q, kk, v1, csr, path="gather")
        ot = ops.sparse_attention(q, kk, v1, csr, path="tile")
        assert (og - ot).abs().max().item() < 2e-3
        assert ((og - ot).norm() / og.norm()).item() < 1e-3
    # (6) rows 0..k-1 keep everything: full causal attention for the first k queries
    sc = torch.matmul(q[:, :, :k].float(), kk[:, :, :k].float().transpose(-1, -2))
    sc = sc.masked_fill(torch.arange(k, device=DEV).view(1, k) > torch.arange(k, device=DEV).view(k, 1), float("-inf"))
    full = torch.matmul(torch.softmax(sc, -1), v1[:, :, :k].float())
    assert (o1[:, :, :k] - full).abs().max().item() < (1e-4 if dtype == torch.float32 else 2e-3)


def test_empty_rows_and_zero_keep(ops):
    """K_t = 0 rows (padded queries) emit nothing and the attention output of an empty (row, head) is 0."""
    N, H, T, T_M, k, d = 1, 4, 64, 16, 4, 16
    probs, q, kk, v, rs = _inputs(N, H, T, T_M, d)
    keep = ops.keep_table_causal(H, T, T_M, k)
    keep[10:20] = 0
    csr, mask = ops.topk_to_csr(probs.to(DEV), keep.to(DEV), k, target_width=T, want_mask=True)
    assert mask[:, :, 10:20].sum() == 0
    assert torch.all(csr.crow[0, 11:21] == csr.crow[0, 10])
    out = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr)
    assert torch.all(out[:, :, 10:20] == 0) and torch.isfinite(out).all()


@pytest.mark.parametrize("N,H,T_dst,T_src,T_M,k,d", [(2, 4, 5, 200, 32, 8, 32), (1, 8, 1, 777, 64, 16, 64), (1, 3, 40, 64, 16, 4, 64),
                                                     (1, 12, 17, 1030, 256, 64, 64)])
def test_tail_rows_of_a_longer_prefix_match_the_oracle(ops, N, H, T_dst, T_src, T_M, k, d):
    """The `T_dst < T_src` contract kv-cache decoding and row sharding rest on: the query rows are the LAST T_dst rows of
    a T_src-long causal sequence (row width = T_src - T_dst + t + 1, `target_width = arange(1..T_src)[-T_dst:]`,
    causal_resize_m_to_t.py:951-955).  Top-k mask and CSR bit-exact, attention within tolerance."""
    g = torch.Generator().manual_seed(5)
    probs = torch.softmax(torch.randn((N, H, T_dst, T_M), generator=g), -1)
    q = torch.randn((N, H, T_dst, d), generator=g) * d ** -0.5
    kk, v = torch.randn((N, H, T_src, d), generator=g), torch.randn((N, H, T_src, d), generator=g)
    rs = torch.sigmoid(torch.randn((N, H, T_dst), generator=g))
    keep = O.keep_counts_module(H, T_src, T_M, k)[-T_dst:].contiguous()
    mask = O.grouped_topk_mask(probs, keep)
    crow, col = O.resize_m_to_t_csr(mask, k, T_src, True)
    ref = O.sparse_attention(q, kk, v, crow, col, rs)
    csr, m_dev = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T_src, want_mask=True)
    assert torch.equal(m_dev.cpu(), mask.float())
    assert torch.equal(csr.crow.cpu().long(), crow)
    for n in range(N):
        z = int(crow[n, -1])
        assert torch.equal(csr.col[n, :z].cpu().long(), col[n, :z])
    out = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, row_scale=rs.to(DEV))
    torch.testing.assert_close(out.cpu(), ref, atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float16, 2e-3)])
@pytest.mark.parametrize("N,H,T_dst,T_src,T_M,k", [(1, 6, 9, 500, 64, 16), (2, 4, 33, 300, 32, 8), (1, 5, 257, 257, 64, 16)])
def test_d80_rows_on_eight_lanes_match_the_oracle(ops, dtype, tol, N, H, T_dst, T_src, T_M, k):
    """d = 80 heads in 16-bit data take `sparse_attn_rows80_kernel` (8 lanes x (8 + 2) elements per row).  Against the
    oracle on the same 16-bit inputs: full epilogue (row scale, mix with the cumulative average), the last T_dst rows
    of a longer prefix, row counts that leave the last workgroup / wave partly empty, fp32 and 16-bit outputs."""
    d = 80
    g = torch.Generator().manual_seed(8)
    probs = torch.softmax(torch.randn((N, H, T_dst, T_M), generator=g), -1)
    q = (torch.randn((N, H, T_dst, d), generator=g) * d ** -0.5).to(dtype)
    kk, v = torch.randn((N, H, T_src, d), generator=g).to(dtype), torch.randn((N, H, T_src, d), generator=g).to(dtype)
    rs = torch.sigmoid(torch.randn((N, H, T_dst), generator=g))
    mx = torch.sigmoid(torch.randn((N, H, T_dst), generator=g))
    avg = torch.randn((N, H, T_dst, d), generator=g).to(dtype)
    keep = O.keep_counts_module(H, T_src, T_M, k)[-T_dst:].contiguous()
    mask = O.grouped_topk_mask(probs, keep)
    crow, col = O.resize_m_to_t_csr(mask, k, T_src, True)
    sparse = O.sparse_attention(q.float(), kk.float(), v.float(), crow, col, rs)
    ref = sparse * mx.unsqueeze(-1) + (1.0 - mx.unsqueeze(-1)) * avg.float()
    csr, _ = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T_src)
    out32 = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, row_scale=rs.to(DEV), avg=avg.to(DEV), mix=mx.to(DEV))
    assert out32.dtype == torch.float32
    torch.testing.assert_close(out32.cpu(), ref, atol=2e-4, rtol=1e-3)          # fp32 accumulation on exact 16-bit inputs
    out16 = torch.empty((N, T_dst, H * d), dtype=dtype, device=DEV)               # the layer's (N, T, H*d) layout
    ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, row_scale=rs.to(DEV), avg=avg.to(DEV), mix=mx.to(DEV),
                         out=out16.view(N, T_dst, H, d).permute(0, 2, 1, 3))
    got = out16.view(N, T_dst, H, d).permute(0, 2, 1, 3).float().cpu()
    torch.testing.assert_close(got, ref, atol=tol, rtol=tol)
    plain = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr)           # no epilogue inputs at all
    torch.testing.assert_close(plain.cpu(), O.sparse_attention(q.float(), kk.float(), v.float(), crow, col, None), atol=2e-4, rtol=1e-3)
