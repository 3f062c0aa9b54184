"""-m gpu: backward of the sparse attention against torch autograd through the DENSE masked formulation of
the same operator (the reference's dense branch, attention.py:1061-1133: softmax(q k^T + mask) v with the mask the CSR
densifies to).  fp32: gradients to 1e-4 relative; 16-bit inputs: against the fp32 evaluation of the same rounded
inputs (the kernels accumulate in fp32), bar = one rounding step of the 16-bit gradient they are cast to."""
import pytest
import torch

from oracle import sea_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from sea_attention_amd.perlin_attention import ops
    return ops


def _dense_reference(q, k, v, dense_mask, rs, avg, mix):
    """fp64 torch: masked softmax attention + row scale + mix, rows/heads without entries give 0."""
    s = torch.matmul(q, k.transpose(-1, -2)).masked_fill(~dense_mask, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)                       # empty (row, head): all -inf
    o = torch.matmul(p, v)
    if rs is not None:
        o = o * rs.unsqueeze(-1)
    if mix is not None:
        a = mix.unsqueeze(-1)
        o = o * a + (1.0 - a) * avg
    return o


@pytest.fixture(params=["gather", "atomic"])
def form(request, monkeypatch):
    """both backward forms: dK / dV gathered over the transposed pattern (default) and the first version's fp32 atomics"""
    from sea_attention_amd.perlin_attention.ops import flat_csr
    monkeypatch.setattr(flat_csr._SparseAttentionFn, "backward_form", request.param)
    return request.param


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T_dst,T_src,T_M,k,d", [(2, 4, 128, 128, 32, 8, 32), (1, 3, 100, 256, 64, 16, 64),
                                                     (1, 2, 64, 64, 16, 4, 128), (1, 5, 33, 90, 32, 8, 80),
                                                     (2, 9, 700, 700, 64, 16, 64)])     # long key lists, nine heads on eight XCDs
def test_backward_matches_dense_autograd(ops, form, dtype, N, H, T_dst, T_src, T_M, k, d):
    g = torch.Generator().manual_seed(17)
    probs = torch.softmax(torch.randn((N, H, T_dst, T_M), generator=g), -1)
    keep = O.keep_counts_module(H, T_src, T_M, k)[-T_dst:].contiguous()
    if T_dst > 40:
        keep[20:24] = 0                                     # a few rows that keep nothing
    csr, _ = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T_src)
    dense_mask = ops.flat_csr_to_dense(csr, T_src, H) > 0                                  # (N,H,T_dst,T_src)
    mk = lambda *shape, sc=1.0: (torch.randn(shape, generator=g) * sc).to(dtype).to(DEV)
    q, kk, v, avg = mk(N, H, T_dst, d, sc=d ** -0.5), mk(N, H, T_src, d), mk(N, H, T_src, d), mk(N, H, T_dst, d)
    rs = torch.sigmoid(torch.randn((N, H, T_dst), generator=g)).to(DEV)
    mix = torch.sigmoid(torch.randn((N, H, T_dst), generator=g)).to(DEV)
    w = torch.randn((N, H, T_dst, d), generator=g).to(DEV)                                 # dL/dout
    leaves = [t.clone().requires_grad_(True) for t in (q, kk, v, avg, rs, mix)]
    out = ops.sparse_attention(leaves[0], leaves[1], leaves[2], csr, row_scale=leaves[4], avg=leaves[3], mix=leaves[5])
    assert out.requires_grad and out.dtype == torch.float32
    (out * w).sum().backward()
    ref_leaves = [t.double().clone().requires_grad_(True) for t in (q, kk, v, avg, rs, mix)]
    ref = _dense_reference(ref_leaves[0], ref_leaves[1], ref_leaves[2], dense_mask, ref_leaves[4], ref_leaves[3], ref_leaves[5])
    (ref * w.double()).sum().backward()
    tol_out = 1e-5 if dtype == torch.float32 else 2e-3
    assert (out.double() - ref).abs().max().item() < tol_out
    names = ["dq", "dk", "dv", "davg", "drow_scale", "dmix"]
    for nm, a, b in zip(names, leaves, ref_leaves):
        ga, gb = a.grad.double(), b.grad
        rel = ((ga - gb).norm() / gb.norm().clamp_min(1e-30)).item()
        # fp32: accumulation order only.  16-bit: the gradient itself is rounded to the 16-bit dtype of its tensor
        bar = 2e-5 if dtype == torch.float32 else (6e-3 if dtype == torch.bfloat16 else 8e-4)
        if nm in ("drow_scale", "dmix"):
            bar = 2e-5 if dtype == torch.float32 else 2e-3   # fp32 tensors; their inputs (o, avg) carry 16-bit values
        assert rel < bar, (nm, rel)
    assert torch.all(leaves[0].grad[:, :, 20:24] == 0) if T_dst > 40 else True             # empty rows: no gradient to q


def test_gather_and_atomic_backward_agree_at_a_layer_sized_case(ops, monkeypatch):
    """One OPT-1.3B-shaped sequence cut to 1024 tokens (bf16, 32 heads, k = 64): the two forms give the same dQ / dK / dV up to
    fp32 summation order, and the gather form writes every dK / dV row itself (no zero-fill: poison the allocator first)."""
    from sea_attention_amd.perlin_attention.ops import flat_csr
    N, H, T, T_M, k, d = 1, 32, 1024, 256, 64, 64
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(torch.randn((N, H, T, T_M), generator=g), -1)
    keep = O.keep_counts_module(H, T, T_M, k).clamp_max(H * T_M)
    csr, _ = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T)
    mk = lambda sc=1.0: (torch.randn((N, H, T, d), generator=g) * sc).bfloat16().to(DEV)
    q, kk, v = mk(d ** -0.5), mk(), mk()
    w = torch.randn((N, H, T, d), generator=g).to(DEV)
    res = {}
    for form_ in ("atomic", "gather"):
        monkeypatch.setattr(flat_csr._SparseAttentionFn, "backward_form", form_)
        junk = torch.full((N, H, T, d), float("nan"), device=DEV); del junk          # what a fresh torch.empty may hand out
        a, b, c = (t.clone().requires_grad_(True) for t in (q, kk, v))
        (ops.sparse_attention_autograd(a, b, c, csr) * w).sum().backward()
        res[form_] = (a.grad.float(), b.grad.float(), c.grad.float())
        assert all(torch.isfinite(t).all() for t in res[form_])
    for nm, x, y in zip(("dq", "dk", "dv"), res["atomic"], res["gather"]):
        rel = ((x - y).norm() / x.norm()).item()
        assert rel < 4e-3, (nm, rel)                       # both are rounded to bf16 at the end: one rounding step apart at most


def test_backward_is_deterministic_up_to_atomic_order_and_gradcheck_small(ops, form):
    """fp32 finite-difference check on a small case (the kernels are fp32, so gradcheck runs at loose fp32 settings)."""
    N, H, T, T_M, k, d = 1, 2, 24, 8, 4, 16
    g = torch.Generator().manual_seed(5)
    probs = torch.softmax(torch.randn((N, H, T, T_M), generator=g), -1)
    keep = O.keep_counts_module(H, T, T_M, k)
    csr, _ = ops.topk_to_csr(probs.to(DEV), keep.to(torch.int32).to(DEV), k, target_width=T)
    q = (torch.randn((N, H, T, d), generator=g) * 0.5).to(DEV).requires_grad_(True)
    kk = torch.randn((N, H, T, d), generator=g).to(DEV).requires_grad_(True)
    v = torch.randn((N, H, T, d), generator=g).to(DEV).requires_grad_(True)
    fn = lambda a, b, c: ops.sparse_attention_autograd(a, b, c, csr)
    assert torch.autograd.gradcheck(fn, (q, kk, v), eps=1e-2, atol=2e-2, rtol=2e-2, nondet_tol=1e-5, fast_mode=True)
    # two backward passes agree to fp32 atomic-order noise
    outs = []
    for _ in range(2):
        for t in (q, kk, v):
            t.grad = None
        fn(q, kk, v).square().sum().backward()
        outs.append([t.grad.clone() for t in (q, kk, v)])
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() < 1e-5


def test_layer_trains_through_the_sparse_branch():
    """The SEA layer in sparse mode with autograd on: gradients reach q / k / v and the predictor's gate parameters, and
    equal dense mode's (same mask, same estimator graph; only steps J-L differ: HIP forward/backward vs torch matmuls)."""
    import sea_attention_amd as S
    from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention

    class Cfg:
        hidden_size, num_attention_heads, max_position_embeddings = 4 * 32, 4, 256
    N, H, T, d, T_M, k = 1, 4, 256, 32, 64, 16
    S.seed(42)
    pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                               k_flatten_dim='causal_batch', context_output_method='mix')
    layer = PerlinSelfAttention(Cfg(), pc).to(DEV).eval()
    S.seed(1)
    x = torch.randn((N, H, T, d), device=DEV)
    fp_min = torch.finfo(torch.float32).min / 2
    mask = ((torch.arange(T, device=DEV).view(1, T) > torch.arange(T, device=DEV).view(T, 1)) * fp_min).view(1, 1, T, T)
    w = torch.randn((N, T, H * d), device=DEV)
    grads = {}
    for mode in (False, True):
        for m in layer.modules():
            if hasattr(m, 'benchmarking'):
                m.benchmarking = mode
        layer.zero_grad()
        q = (x * d ** -0.5).clone().requires_grad_(True)
        kk, v = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        out = layer(None, None, None, query_layer=q, key_layer=kk, value_layer=v, attention_mask=mask)
        (out.context_layer.float() * w).sum().backward()
        sc = layer.attention.attention_predictor_dec_scaler[0].weight.grad
        assert sc is not None and torch.isfinite(sc).all()
        grads[mode] = (out.context_layer.detach().float(), q.grad.clone(), kk.grad.clone(), v.grad.clone(), sc.clone())
    for name, a, b in zip(("context", "dq", "dk", "dv", "d dec_scaler.weight"), grads[False], grads[True]):
        rel = ((a - b).norm() / a.norm()).item()
        assert rel < 2e-4, (name, rel)
