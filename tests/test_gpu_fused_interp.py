"""-m gpu: steps I + J in one launch (`sea_sparse_attention_fused`, round 3): the gather attention kernels do the
nearest-neighbour interpolation of their own (row, head) themselves -- expand the kept pixels to key columns, WRITE the
CSR's column array and walk it from LDS -- instead of reading what a separate emit launch wrote.

Bars: the column array the fused launch leaves behind equals the emit launch's bit for bit, order included (and through
it the reference's: the emit is pinned to the golden fixtures), and the attention output equals the unfused launch's bit
for bit (same entries in the same order through the same arithmetic)."""
import pytest
import torch

from oracle import sea_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from sea_attention_amd.perlin_attention import ops
    return ops


def _inputs(N, H, T_dst, T_src, d, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn((N, H, T_dst, d), generator=g) * d ** -0.5).to(dtype).to(DEV)
    k = torch.randn((N, H, T_src, d), generator=g).to(dtype).to(DEV)
    v = torch.randn((N, H, T_src, d), generator=g).to(dtype).to(DEV)
    rs = torch.sigmoid(torch.randn((N, H, T_dst), generator=g)).to(DEV)
    mx = torch.sigmoid(torch.randn((N, H, T_dst), generator=g)).to(DEV)
    avg = torch.randn((N, H, T_dst, d), generator=g).to(dtype).to(DEV)
    return q, k, v, rs, mx, avg


def _selection(ops, probs, keep, k, T_src, defer):
    """selection launch + row scan (+ emit unless deferred) -> FlatCSR"""
    N, H, T_dst, T_m = probs.shape
    csr0, _ = ops.topk_to_csr(probs, keep, k, target_width=T_src)              # reference handle (emit launch)
    csr1 = ops.csr_from_selection(csr0.bits, csr0.row_nnz, csr0.head_off, H, T_m, T_src, k, True, csr0._col.shape[1], defer_emit=defer)
    return csr0, csr1


@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 64), (torch.float16, 128), (torch.float32, 32), (torch.float32, 64), (torch.bfloat16, 128),
                                     (torch.bfloat16, 80), (torch.float16, 80), (torch.float32, 128)])
@pytest.mark.parametrize("N,H,T_dst,T_src,T_M,k", [(2, 6, 300, 300, 64, 16),        # ragged T, pixel widths 1 .. 5
                                                  (1, 12, 1024, 1024, 256, 64),    # BASELINE-like proportions (widths <= 4)
                                                  (1, 3, 70, 2100, 64, 16),        # last rows of a long prefix: widths ~ 32
                                                  (1, 4, 40, 40, 64, 8),           # T < T_m: zero-width pixels
                                                  (1, 9, 200, 200, 32, 2)])        # max_k clamp active: thinned pixels
def test_fused_columns_and_output_equal_the_two_launch_path(ops, dtype, d, N, H, T_dst, T_src, T_M, k):
    g = torch.Generator().manual_seed(11)
    probs = torch.softmax(torch.randn((N, H, T_dst, T_M), generator=g), -1).to(DEV)
    keep = O.keep_counts_module(H, T_src, T_M, k)[-T_dst:].clamp_max(H * T_M).to(torch.int32).contiguous().to(DEV)
    q, kk, v, rs, mx, avg = _inputs(N, H, T_dst, T_src, d, dtype, 5)
    ref_csr, csr = _selection(ops, probs, keep, k, T_src, defer=True)
    assert csr.col_is_pending and not ref_csr.col_is_pending
    csr._col.fill_(-7)                                                       # whatever the allocator handed out
    ref = ops.sparse_attention(q, kk, v, ref_csr, row_scale=rs, avg=avg, mix=mx, path="gather")
    out = ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, path="gather")
    assert not csr.col_is_pending                                             # the fused launch (or, for rows wider than 16 lanes,
    #                                                                           the emit launch `.col` triggered) wrote the columns ...
    assert torch.equal(out, ref)
    for n in range(N):
        z = int(csr.crow[n, -1])
        assert torch.equal(csr.col[n, :z], ref_csr.col[n, :z]), n             # ... exactly the emit launch's
    # per-entry probabilities through the fused launch too
    _, csr2 = _selection(ops, probs, keep, k, T_src, defer=True)
    o2, p2 = ops.sparse_attention(q, kk, v, csr2, row_scale=rs, path="gather", want_probs=True)
    o1, p1 = ops.sparse_attention(q, kk, v, ref_csr, row_scale=rs, path="gather", want_probs=True)
    assert torch.equal(o1, o2) and torch.equal(p1, p2)


def test_fused_overflowing_block_takes_the_memory_path(ops):
    """A head that takes (nearly) all kept pixels of every row: the block's key lists exceed the LDS budget (8192 entries per
    64 rows), the kernel expands straight into `col` and reads it back -- same columns, same output."""
    N, H, T, T_M, k, d = 1, 4, 1024, 64, 64, 64
    g = torch.Generator().manual_seed(2)
    probs = torch.softmax(torch.randn((N, H, T, T_M), generator=g), -1)
    probs[:, 1] += 10.0                                                        # head 1 wins every pixel the row may keep
    probs = probs.to(DEV)
    keep = O.keep_counts_module(H, T, T_M, k).clamp_max(H * T_M).to(torch.int32).contiguous().to(DEV)
    q, kk, v, rs, mx, avg = _inputs(N, H, T, T, d, torch.bfloat16, 9)
    ref_csr, csr = _selection(ops, probs, keep, k, T, defer=True)
    ho = ref_csr.head_off.long()
    per_block = (ho[0, :, 2] - ho[0, :, 1]).view(-1, 64).sum(-1)              # head 1, 64-row blocks
    assert int(per_block.max()) > 8192, int(per_block.max())
    ref = ops.sparse_attention(q, kk, v, ref_csr, row_scale=rs, avg=avg, mix=mx, path="gather")
    out = ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, path="gather")
    assert not csr.col_is_pending and torch.equal(out, ref)
    z = int(csr.crow[0, -1])
    assert torch.equal(csr.col[0, :z], ref_csr.col[0, :z])


def test_pending_columns_materialise_for_every_other_consumer(ops):
    """A handle with pending columns is safe to hand to anything: the tile kernel, a plan that may choose it, the backward,
    `.to_sparse_csr()`, rows wider than 16 lanes -- whoever reads `.col` first runs the emit launch."""
    N, H, T, T_M, k, d = 1, 4, 256, 64, 16, 64
    g = torch.Generator().manual_seed(4)
    probs = torch.softmax(torch.randn((N, H, T, T_M), generator=g), -1).to(DEV)
    keep = O.keep_counts_module(H, T, T_M, k).clamp_max(H * T_M).to(torch.int32).contiguous().to(DEV)
    q, kk, v, rs, mx, avg = _inputs(N, H, T, T, d, torch.bfloat16, 3)
    ref_csr, _ = _selection(ops, probs, keep, k, T, defer=False)
    ref = ops.sparse_attention(q, kk, v, ref_csr, row_scale=rs, path="gather")
    for how in ("tile", "auto+plan", "wire", "no_fuse"):
        _, csr = _selection(ops, probs, keep, k, T, defer=True)
        assert csr.col_is_pending
        if how == "tile":
            out = ops.sparse_attention(q, kk, v, csr, row_scale=rs, path="tile")
            assert (out - ref).abs().max().item() < 2e-3
        elif how == "auto+plan":
            out = ops.sparse_attention(q, kk, v, csr, row_scale=rs, path="auto", plan=ops.attention_plan(csr, T_M))
            assert (out - ref).abs().max().item() < 2e-3
        elif how == "wire":
            t = csr.to_sparse_csr()
            assert torch.equal(t.col_indices(), ref_csr.to_sparse_csr().col_indices())
        else:
            out = ops.sparse_attention(q, kk, v, csr, row_scale=rs, path="gather", fuse_emit=False)
            assert torch.equal(out, ref)
        assert not csr.col_is_pending
        z = int(csr.crow[0, -1])
        assert torch.equal(csr.col[0, :z], ref_csr.col[0, :z])


def test_layer_gather_path_fuses_and_returns_the_reference_csr():
    """The layer with sparse_kernel = "gather": no emit launch in the step (the attention launch writes the columns), and the
    CSR it returns equals the oracle's top-k + interpolation on the layer's own map; context equal to the unfused layer's."""
    import sea_attention_amd as S
    from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
    from sea_attention_amd.perlin_attention import attention as A

    class Cfg:
        hidden_size, num_attention_heads, max_position_embeddings = 8 * 64, 8, 1024
    N, H, T, d, T_M, k = 2, 8, 1024, 64, 256, 32
    S.seed(42)
    pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                               k_flatten_dim='causal_batch', context_output_method='mix')
    layer = PerlinSelfAttention(Cfg(), pc).to(DEV).to(torch.bfloat16).eval()
    for m in layer.modules():
        if hasattr(m, 'benchmarking'):
            m.benchmarking = True
    layer.attention.assume_not_padded = True
    S.seed(3)
    x = torch.randn((N, H, T, d), device=DEV).to(torch.bfloat16)
    fp_min = torch.finfo(torch.float16).min / 2
    mask = ((torch.arange(T, device=DEV).view(1, T) > torch.arange(T, device=DEV).view(T, 1)) * fp_min).view(1, 1, T, T)
    mask = mask.to(torch.bfloat16).expand(N, 1, T, T).contiguous()
    outs = {}
    for mode in ("gather", "auto"):
        layer.attention.sparse_kernel = mode
        with torch.no_grad():
            outs[mode] = layer(None, None, None, query_layer=x * d ** -0.5, key_layer=x, value_layer=x, attention_mask=mask)
    a, b = outs["gather"], outs["auto"]
    # round 4: the layer leaves the column array unwritten (lazy_csr_columns) -- the handle it returns is still pending and
    # the first reader below runs the emit launch; with the switch off the step writes it
    assert a.partial_attention_mask.col_is_pending and b.partial_attention_mask.col_is_pending
    layer.attention.lazy_csr_columns = False
    with torch.no_grad():
        c = layer(None, None, None, query_layer=x * d ** -0.5, key_layer=x, value_layer=x, attention_mask=mask)
    assert not c.partial_attention_mask.col_is_pending and torch.equal(c.context_layer, b.context_layer)
    for n in range(N):
        zc = int(c.partial_attention_mask.crow[n, -1])
        assert torch.equal(c.partial_attention_mask.col[n, :zc], b.partial_attention_mask.col[n, :zc])
    assert torch.equal(a.estimated_attention_probs_m, b.estimated_attention_probs_m)
    assert torch.equal(a.partial_attention_mask.crow, b.partial_attention_mask.crow)
    probs = a.estimated_attention_probs_m.float().cpu()
    keep = O.keep_counts_module(H, T, T_M, k)
    crow, col = O.resize_m_to_t_csr(O.grouped_topk_mask(probs, keep), k, T, True)
    for n in range(N):
        z = int(crow[n, -1])
        assert torch.equal(a.partial_attention_mask.crow[n].cpu().long(), crow[n])
        assert torch.equal(a.partial_attention_mask.col[n, :z].cpu().long(), col[n, :z])
        assert torch.equal(b.partial_attention_mask.col[n, :z].cpu().long(), col[n, :z])
    rel = ((a.context_layer.float() - b.context_layer.float()).norm() / b.context_layer.float().norm()).item()
    assert rel < 2e-3, rel                                   # auto may pick the tile kernel; gather vs gather would be bitwise


@pytest.mark.parametrize("dtype,d,fused", [(torch.bfloat16, 64, True), (torch.float16, 128, True), (torch.bfloat16, 64, False)])
def test_launches_over_groups_of_sequences_equal_the_whole_launch(ops, dtype, d, fused):
    """`FlatCSR.items(n0, n1)` (round 4: one attention launch per chunk of a chunked all-gather pipeline): launches over the
    parts write the same columns and the same context rows as ONE launch over the batch, bit for bit, whether the columns
    are still pending (fused interpolation: every part expands its own) or already there."""
    N, H, T, T_M, k = 4, 8, 700, 64, 16
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(torch.randn((N, H, T, T_M), generator=g), -1).to(DEV)
    keep = O.keep_counts_module(H, T, T_M, k).clamp_max(H * T_M).to(torch.int32).contiguous().to(DEV)
    q, kk, v, rs, mx, avg = _inputs(N, H, T, T, d, dtype, 9)
    _, whole = _selection(ops, probs, keep, k, T, defer=fused)
    _, parts = _selection(ops, probs, keep, k, T, defer=fused)
    ctx_w = torch.empty((N, T, H * d), dtype=torch.float32, device=DEV)
    ctx_p = torch.full((N, T, H * d), float("nan"), dtype=torch.float32, device=DEV)
    ops.sparse_attention(q, kk, v, whole, row_scale=rs, avg=avg, mix=mx, out=ctx_w.view(N, T, H, d).permute(0, 2, 1, 3), path="gather")
    if fused:
        parts._col.fill_(-7)
    for n0, n1 in ((0, 1), (1, 3), (3, 4)):                                 # ragged groups
        sub = parts.items(n0, n1)
        assert sub.col_is_pending == fused and (sub.N, sub.T_dst) == (n1 - n0, T)
        ops.sparse_attention(q[n0:n1], kk[n0:n1], v[n0:n1], sub, row_scale=rs[n0:n1], avg=avg[n0:n1], mix=mx[n0:n1],
                             out=ctx_p[n0:n1].view(n1 - n0, T, H, d).permute(0, 2, 1, 3), path="gather")
        assert not sub.col_is_pending
    torch.cuda.synchronize()
    assert torch.equal(ctx_w, ctx_p)
    for n in range(N):
        z = int(whole.crow[n, -1])
        assert torch.equal(whole._col[n, :z], parts._col[n, :z])
    if fused:                     # the parent still believes its columns pending: reading them runs the emit launch -- same values
        assert parts.col_is_pending
        assert torch.equal(parts.col[0, :int(whole.crow[0, -1])], whole._col[0, :int(whole.crow[0, -1])])


@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 64), (torch.float16, 80), (torch.float32, 64), (torch.bfloat16, 128)])
def test_columns_can_stay_pending_through_the_fused_launch(ops, dtype, d):
    """`keep_columns_pending` (round 4: the layer's hot path returns the CSR, nobody reads its columns): the fused launch
    keeps the expanded columns in LDS and does not write the column array -- same output bit for bit, the handle stays
    pending, and the first reader gets exactly the emit launch's columns.  A block whose lists overflow the LDS list still
    works (it passes through its part of the array)."""
    N, H, T, T_M, k = 2, 6, 600, 64, 16
    g = torch.Generator().manual_seed(21)
    probs = torch.softmax(torch.randn((N, H, T, T_M), generator=g), -1)
    probs[1, 2] += 5.0                                                       # item 1: head 2 takes (almost) every kept pixel
    probs = probs.to(DEV)
    keep = O.keep_counts_module(H, T, T_M, k).clamp_max(H * T_M).to(torch.int32).contiguous().to(DEV)
    q, kk, v, rs, mx, avg = _inputs(N, H, T, T, d, dtype, 13)
    ref_csr, csr = _selection(ops, probs, keep, k, T, defer=True)
    ref = ops.sparse_attention(q, kk, v, ref_csr, row_scale=rs, avg=avg, mix=mx, path="gather")
    csr._col.fill_(-7)
    out = ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, path="gather", keep_columns_pending=True)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert csr.col_is_pending, "the handle keeps its columns pending"
    z0 = int(ref_csr.crow[0, -1])
    assert bool((csr._col[0, :z0] == -7).all()), "item 0's lists fit the LDS list: nothing of its column array was written"
    for n in range(N):                                                       # first reader: the emit launch, bit-identical
        z = int(ref_csr.crow[n, -1])
        assert torch.equal(csr.col[n, :z], ref_csr.col[n, :z])
    assert not csr.col_is_pending
