"""-m gpu: the MFMA tile path of the fused sparse attention (csrc/sea_attn_tile.hip) and the per-entry probabilities
output of the gather path, through the C ABI (`sea_sparse_attention_ex`), against the CPU oracle and the reference's
golden fixtures.

Tolerances: 16-bit inputs are compared with the oracle evaluated in fp32 on the SAME rounded inputs.  bf16: P enters
the matrix cores split in two bf16 terms (16 significand bits), so the bar is the gather path's (2e-3 abs, 1e-3 relative
norm = north_star's 1e-3); fp16: one 11-bit term, same bar."""
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


def _case(N, H, T_dst, T_src, T_M, k, d, dtype, seed=7, structured=False):
    g = torch.Generator().manual_seed(seed)
    if structured:
        from sea_attention_amd import synthetic
        probs = synthetic.structured_probs(N, H, T_dst, T_M, "cpu", seed=seed, T_src=T_src)
    else:
        probs = torch.softmax(torch.randn((N, H, T_dst, T_M), generator=g), -1)
    q = (torch.randn((N, H, T_dst, d), generator=g) * d ** -0.5).to(dtype)
    kk = torch.randn((N, H, T_src, d), generator=g).to(dtype)
    v = torch.randn((N, H, T_src, d), generator=g).to(dtype)
    rs = torch.sigmoid(torch.randn((N, H, T_dst), generator=g))
    mx = torch.sigmoid(torch.randn((N, H, T_dst), generator=g))
    avg = torch.randn((N, H, T_dst, d), generator=g).to(dtype)
    keep = O.keep_counts_module(H, T_src, T_M, k)[-T_dst:].contiguous()
    crow, col = O.resize_m_to_t_csr(O.grouped_topk_mask(probs, keep), k, T_src, True)
    return probs, q, kk, v, rs, mx, avg, keep, crow, col


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T_dst,T_src,T_M,k,d,rt,kw", [
    (2, 12, 1024, 1024, 256, 64, 64, 0, 0),        # defaults
    (1, 4, 512, 512, 64, 16, 128, 1, 0),
    (1, 4, 512, 512, 64, 16, 128, 2, 0),
    (1, 3, 256, 256, 32, 8, 80, 0, 0),
    (1, 5, 300, 300, 96, 8, 64, 2, 64),            # T not a multiple of 16; five 64-key windows
    (1, 5, 300, 300, 96, 8, 64, 1, 256),
    (2, 4, 33, 300, 32, 8, 80, 2, 128),            # last rows of a longer prefix, partial row tiles
    (1, 8, 1, 777, 64, 16, 64, 0, 512),            # a decode step: one query row
    (1, 6, 9, 500, 64, 16, 128, 0, 0),
    (1, 2, 70, 4100, 256, 64, 64, 2, 4096),        # keys beyond one 4096-key window
])
def test_tile_path_vs_oracle(ops, dtype, N, H, T_dst, T_src, T_M, k, d, rt, kw):
    probs, q, kk, v, rs, mx, avg, keep, crow, col = _case(N, H, T_dst, T_src, T_M, k, d, dtype)
    sparse = O.sparse_attention(q.float(), kk.float(), v.float(), crow, col, rs)
    csr, _ = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T_src)
    qd, kd, vd = q.to(DEV), kk.to(DEV), v.to(DEV)
    out = ops.sparse_attention(qd, kd, vd, csr, row_scale=rs.to(DEV), path="tile", row_tiles=rt, key_window=kw)
    assert out.dtype == torch.float32
    err = (out.cpu() - sparse).abs().max().item()
    rel = ((out.cpu() - sparse).norm() / sparse.norm()).item()
    assert err < 2e-3 and rel < 1e-3, (err, rel)
    # the gather path on the same inputs: the two kernels agree to rounding
    ref_g = ops.sparse_attention(qd, kd, vd, csr, row_scale=rs.to(DEV), path="gather")
    assert (out - ref_g).abs().max().item() < 2e-3
    # full epilogue into the layer's (N, T, H*d) layout, 16-bit output
    ref = sparse * mx.unsqueeze(-1) + (1.0 - mx.unsqueeze(-1)) * avg.float()
    ctx = torch.empty((N, T_dst, H * d), dtype=dtype, device=DEV)
    ops.sparse_attention(qd, kd, vd, csr, row_scale=rs.to(DEV), avg=avg.to(DEV), mix=mx.to(DEV),
                         out=ctx.view(N, T_dst, H, d).permute(0, 2, 1, 3), path="tile", row_tiles=rt, key_window=kw)
    got = ctx.view(N, T_dst, H, d).permute(0, 2, 1, 3).float().cpu()
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    torch.testing.assert_close(got, ref, atol=tol, rtol=tol)
    # determinism
    again = ops.sparse_attention(qd, kd, vd, csr, row_scale=rs.to(DEV), path="tile", row_tiles=rt, key_window=kw)
    assert torch.equal(out, again)


def test_tile_path_strided_inputs_and_window_choice_agree(ops):
    """q/k/v as strided views of a fused projection; every (row tiles, key window) choice gives the same numbers up to the
    order the tiles are summed in (a window boundary moves no key, it only regroups tile pairs)."""
    N, H, T, T_M, k, d = 1, 6, 640, 128, 32, 64
    probs, q, kk, v, rs, *_rest, keep, crow, col = _case(N, H, T, T, T_M, k, d, torch.bfloat16, structured=True)
    ref = O.sparse_attention(q.float(), kk.float(), v.float(), crow, col, rs)
    csr, _ = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T)
    big = torch.randn((N, T, 3, H, d), dtype=torch.bfloat16)
    big[:, :, 0] = q.permute(0, 2, 1, 3); big[:, :, 1] = kk.permute(0, 2, 1, 3); big[:, :, 2] = v.permute(0, 2, 1, 3)
    bd = big.to(DEV)
    qv, kv, vv = (bd[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    for rt in (1, 2):
        for kw in (64, 512, 2048):
            out = ops.sparse_attention(qv, kv, vv, csr, row_scale=rs.to(DEV), path="tile", row_tiles=rt, key_window=kw)
            assert ((out.cpu() - ref).norm() / ref.norm()).item() < 1e-3, (rt, kw)
            assert (out.cpu() - ref).abs().max().item() < 2e-3, (rt, kw)


def test_tile_path_rejects_what_it_cannot_do(ops):
    N, H, T, T_M, k, d = 1, 2, 64, 16, 4, 32
    probs, q, kk, v, rs, *_r, keep, crow, col = _case(N, H, T, T, T_M, k, d, torch.bfloat16)
    csr, _ = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T)
    with pytest.raises(RuntimeError, match="tile kernel"):
        ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, path="tile")              # d = 32
    with pytest.raises(RuntimeError, match="tile kernel"):
        ops.sparse_attention(q.float().to(DEV), kk.float().to(DEV), v.float().to(DEV), csr, path="tile")   # fp32
    # auto falls back to the gather kernels for those
    out = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr)
    ref = O.sparse_attention(q.float(), kk.float(), v.float(), crow, col, None)
    assert (out.cpu() - ref).abs().max().item() < 2e-3


@pytest.mark.parametrize("path", ["gather", "tile"])
def test_empty_rows_and_non_finite_padding_keys(ops, path):
    """Rows that keep nothing give exact zeros, and a key no row keeps may hold garbage (an uninitialised kv-cache
    slot) without reaching any output."""
    N, H, T, T_M, k, d = 1, 2, 96, 16, 4, 64
    dtype = torch.bfloat16
    probs, q, kk, v, rs, *_r, keep, crow, col = _case(N, H, T, T, T_M, k, d, dtype)
    keepz = keep.clone()
    keepz[10:30] = 0                                                   # padded query rows
    mask = O.grouped_topk_mask(probs, keepz)
    crow, col = O.resize_m_to_t_csr(mask, k, T, True)
    z = int(crow[0, -1])
    used = torch.zeros(H * T, dtype=torch.bool)
    used[col[0, :z]] = True
    free = [(h, key) for h in range(H) for key in range(T) if not used[h * T + key]]
    assert len(free) >= 4
    kk = kk.clone(); v = v.clone()
    kz, vz = kk.clone(), v.clone()
    for h, key in free:
        # gather path: never touches a key no row keeps.  Tile path: a key that shares a 16-key tile with a kept key IS
        # multiplied (by an exact 0), so its K row may be anything but its V row must be finite (0 x NaN; the reference's
        # dense branch, `matmul(probs, v)` at attention.py:1128, has the same property) -- huge finite garbage there.
        kk[:, h, key] = float("nan")
        v[:, h, key] = float("nan") if path == "gather" else 3.0e38
        kz[:, h, key] = 0; vz[:, h, key] = 0
    csr, _ = ops.topk_to_csr(probs.to(DEV), keepz.to(torch.int32).to(DEV), k, target_width=T)
    assert torch.equal(csr.crow.cpu().long(), crow)
    out = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, row_scale=rs.to(DEV), path=path).cpu()
    assert torch.isfinite(out).all()
    assert torch.all(out[:, :, 10:30] == 0)
    ref = O.sparse_attention(q.float(), kz.float(), vz.float(), crow, col, rs)
    assert (out - ref).abs().max().item() < 2e-3


@pytest.mark.parametrize("dtype,d", [(torch.float32, 64), (torch.bfloat16, 64), (torch.float16, 80), (torch.float32, 256),
                                     (torch.bfloat16, 128)])
def test_probs_output_matches_the_operator_chain(ops, dtype, d):
    """`want_probs`: per-entry rs * softmax (the reference's partial_attention_probs values, attention.py:1162-1171)
    == oracle SDDMM -> softmax -> elmul on the same CSR, and the context is unchanged by asking for them."""
    N, H, T, T_M, k = 2, 5, 200, 32, 8
    probs, q, kk, v, rs, *_r, keep, crow, col = _case(N, H, T, T, T_M, k, d, dtype)
    s = O.csr_sddmm(q.float(), kk.float(), crow, col)
    p = O.csr_softmax(s, crow, col, H, T)
    p = O.csr_elmul(p, crow, col, rs.view(N, H, T, 1).expand(N, H, T, T), T)
    csr, _ = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T)
    out, pv = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, row_scale=rs.to(DEV), want_probs=True)
    plain = ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, row_scale=rs.to(DEV), path="gather")
    assert torch.equal(out, plain)
    for n in range(N):
        z = int(crow[n, -1])
        np.testing.assert_allclose(pv[n, :z].cpu().numpy(), p[n, :z].numpy(), atol=2e-6, rtol=2e-5)
    # as a torch CSR tensor, like the reference returns it
    t = csr.to_sparse_csr(pv)
    assert t.is_sparse_csr and tuple(t.shape) == (N, T, H * T)
    dense = ops.flat_csr_to_dense(t, T, H).cpu()
    ref_dense = O.flat_csr_to_dense(crow, col, p, T, H)
    assert (dense - ref_dense).abs().max().item() < 2e-6


def test_probs_output_matches_golden_elmul(golden, ops):
    """Fixture F6 (`elmul` = the reference's flat_csr_softmax + flat_csr_elmul output on its own CSR)."""
    from test_gpu_golden import _keep, _meta
    for case in ("tiny", "mid", "ragged", "short", "big", "clamp"):
        g = golden(case)
        N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
        dev = torch.device(DEV)
        probs = torch.from_numpy(g["probs"]).to(dev)
        csr, _ = ops.topk_to_csr(probs, _keep(ops, g, dev), k, target_width=T_SRC, is_causal=causal)
        q, kk, v = (torch.from_numpy(g[n]).to(dev) for n in ("q", "k", "v"))
        scaler = torch.from_numpy(g["scaler"]).to(dev).contiguous()
        _out, pv = ops.sparse_attention(q, kk, v, csr, row_scale=scaler, want_probs=True)
        valid = np.arange(g["col"].shape[1])[None, :] < g["crow"][:, -1:]
        got = pv[:, :g["col"].shape[1]].cpu().numpy()
        np.testing.assert_allclose(got[valid], g["elmul"][valid], atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 64), (torch.float16, 80), (torch.bfloat16, 128)])
def test_per_block_dispatch_any_plan_gives_the_same_answer(ops, dtype, d):
    """path="auto" with a plan launches both kernels; the plan's count of tile-favouring blocks decides on the device which
    ONE of them runs the launch (the other exits at once).  Whatever the plan says (all gather, all tile, random shares on
    either side of the cut, the library's own estimate), every row is written exactly once and the output is the oracle's."""
    N, H, T_dst, T_src, T_M, k = 2, 6, 200, 333, 64, 16
    probs, q, kk, v, rs, mx, avg, keep, crow, col = _case(N, H, T_dst, T_src, T_M, k, d, dtype, structured=True)
    sparse = O.sparse_attention(q.float(), kk.float(), v.float(), crow, col, rs)
    ref = sparse * mx.unsqueeze(-1) + (1.0 - mx.unsqueeze(-1)) * avg.float()
    csr, _ = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T_src)
    TB16 = (T_dst + 15) // 16
    g = torch.Generator().manual_seed(3)
    own = ops.attention_plan(csr, T_M)
    assert own is not None and own.dtype == torch.uint8
    ob = ops.plan_blocks(own, N, H, T_dst)
    assert ob.shape == (N, H, TB16) and int(ob.max()) <= 1
    assert int(own[-4:].view(torch.int32)[0]) == int(ob.sum())            # the count behind the bytes
    half = torch.zeros((N, H, TB16), dtype=torch.uint8); half[:, :, : TB16 // 2 - 1] = 1   # just under half: the gather kernels run
    plans = {"gather": ops.make_plan(torch.zeros((N, H, TB16), dtype=torch.uint8)),
             "tile": ops.make_plan(torch.ones((N, H, TB16), dtype=torch.uint8)),
             "random": ops.make_plan((torch.rand((N, H, TB16), generator=g) < 0.4).to(torch.uint8)),
             "mostly tile": ops.make_plan((torch.rand((N, H, TB16), generator=g) < 0.8).to(torch.uint8)),
             "first half": ops.make_plan(half), "own": own.cpu()}
    for name, pl in plans.items():
        out = torch.full((N, H, T_dst, d), float("nan"), device=DEV)            # every row must be written by someone
        ops.sparse_attention(q.to(DEV), kk.to(DEV), v.to(DEV), csr, row_scale=rs.to(DEV), avg=avg.to(DEV), mix=mx.to(DEV),
                             out=out, path="auto", plan=pl.to(DEV).contiguous())
        assert torch.isfinite(out).all(), name
        assert (out.cpu() - ref).abs().max().item() < 2e-3, name
        assert ((out.cpu() - ref).norm() / ref.norm()).item() < 1e-3, name


def test_plan_follows_the_structure_of_the_map(ops):
    """Structured maps (rows share their keys) send most late blocks to the tile kernel, independent random rows almost none;
    the first rows (t < k: every key kept, a dense triangle) go to the tile kernel in both."""
    from sea_attention_amd import synthetic
    N, H, T, T_M, k = 1, 8, 4096, 256, 64
    keep = ops.keep_table_causal(H, T, T_M, k, device=DEV)
    shares = {}
    for name, gen in (("random", synthetic.random_probs), ("structured", synthetic.structured_probs)):
        csr, _ = ops.topk_to_csr(gen(N, H, T, T_M, DEV, torch.bfloat16, seed=1), keep, k, target_width=T)
        pl = ops.plan_blocks(ops.attention_plan(csr, T_M), N, H, T)
        assert int(pl[:, :, :2].min()) == 1                              # rows 0..31: dense causal triangle
        shares[name] = pl[:, :, 3 * T // 64:].float().mean().item()      # last quarter: K_t is down to the strong pixels
    assert shares["structured"] > 0.5 > shares["random"], shares
