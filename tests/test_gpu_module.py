"""-m gpu: module-level parity, following the reference's own protocol
(src/main/tests/test_perlin_opt_consist.py:103-232): run the attention layer once in dense mode
(`benchmarking=False`, plain torch) and once in sparse mode (`benchmarking=True`, HIP kernels), capture
every named temp buffer, compare.  The reference's bar is sum-squared-error <= 1e-5 per buffer with the
mask buffers exactly equal; plus the causality canary of test_perlin_opt_causality.py:246-276."""
import math

import pytest
import torch

import sea_attention_amd as S
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Cfg:
    def __init__(self, hidden, heads, max_pos):
        self.hidden_size, self.num_attention_heads, self.max_position_embeddings = hidden, heads, max_pos


def causal_mask(N, T, dtype):
    fp_min = torch.finfo(torch.float32 if dtype == torch.float32 else torch.float16).min / 2
    m = ((torch.arange(T, device=DEV).view(1, T) > torch.arange(T, device=DEV).view(T, 1)) * fp_min).view(1, 1, T, T)
    return m.expand(N, 1, T, T).contiguous().to(dtype)


def make_layer(H, d, T_M, k, max_pos, dtype=torch.float32):
    S.seed(42)
    pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True,
                               k_flatten=True, k_flatten_dim='causal_batch', context_output_method='mix')
    return PerlinSelfAttention(Cfg(H * d, H, max_pos), pc).to(DEV).to(dtype).eval()


def run(layer, q, k, v, mask, benchmarking, capture=True):
    for m in layer.modules():
        if hasattr(m, 'benchmarking'):
            m.benchmarking = benchmarking
    bench = S.get_bench()
    bench.reset_temp_buffers()
    bench.activate_temp_buffers = capture
    try:
        with torch.no_grad():
            out = layer(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=mask)
        bufs = {n: b[-1] for n, b in bench.buffers.items()}
    finally:
        bench.activate_temp_buffers = False
        bench.reset_temp_buffers()
    return out, bufs


@pytest.mark.parametrize("N,H,T,d,T_M,k", [(1, 12, 2048, 64, 256, 64), (2, 4, 512, 32, 64, 16)])
def test_dense_vs_sparse_consistency(N, H, T, d, T_M, k):
    layer = make_layer(H, d, T_M, k, T)
    S.seed(42)
    q = torch.randn((N, H, T, d), device=DEV)
    qs = q * d ** -0.5
    mask = causal_mask(N, T, torch.float32)
    out_d, bd = run(layer, qs, q.clone(), q.clone(), mask, False)
    out_s, bs = run(layer, qs, q.clone(), q.clone(), mask, True)

    def sse(a, b):
        return (a.float() - b.float()).square().sum().item()

    same = ['q', 'k', 'v', 'v_for_atten', 'performer_context_layer', 'performer_value',
            'estimated_attention_score_dec_row', 't_attention_predictor', 'estimated_attention_score',
            'estimated_attention_probs', 'masked_estimated_attention_probs', 'estimated_scales', 'average_scale',
            'average_context_layer', 'q_for_score', 'k_for_score']
    for name in same:
        assert sse(bd[name], bs[name]) <= 1e-5, name
    # masks: exactly equal (dense mode stores 0 / FP_MIN, sparse mode 1 / 0)
    alive_d = bd['partial_attention_mask_before_interp'] > -1
    alive_s = bs['partial_attention_mask_before_interp'] > 0
    assert torch.equal(alive_d, alive_s)
    assert torch.equal(bd['partial_attention_mask'] > -1, bs['partial_attention_mask'] > 0)
    # the dense branch leaves K_t unclamped (rank < K_t is all-true either way); the kernel table clamps to H*T_M
    assert torch.equal(bd['per_item_top_k'].view(-1).float().clamp_max(H * T_M), bs['per_item_top_k'].view(-1).float())
    for name in ['partial_context_layer_1', 'partial_context_layer_2', 'partial_context_layer_sparse', 'partial_context_layer']:
        e = sse(bd[name], bs[name])
        assert e <= 1e-5 * max(1.0, N * T / 2048), (name, e)
    assert sse(out_d.context_layer, out_s.context_layer) <= 1e-5 * max(1.0, N * T / 2048)
    assert out_s.context_layer.dtype == torch.float32 and tuple(out_s.context_layer.shape) == (N, T, H * d)
    # fused production path (no probes) == probed path, bit for bit on the mask and within fp32 noise on the output
    out_f, _ = run(layer, qs, q.clone(), q.clone(), mask, True, capture=False)
    assert (out_f.context_layer - out_s.context_layer).abs().max().item() < 1e-5
    assert isinstance(out_f.partial_attention_mask, ops.FlatCSR) and out_f.partial_attention_mask.is_sparse_csr
    t = out_f.partial_attention_mask.to_sparse_csr()
    assert t.is_sparse_csr and tuple(t.shape) == (N, T, H * T)
    assert torch.equal(ops.flat_csr_to_dense(t, T, H) > 0, bd['partial_attention_mask'] > -1)


def test_sparse_mode_bf16_close_to_fp32_dense():
    """cfg-3 dtype: bf16 layer in sparse mode stays within bf16 noise of the same layer's dense mode."""
    N, H, T, d, T_M, k = 1, 8, 1024, 64, 256, 64
    layer = make_layer(H, d, T_M, k, T, torch.bfloat16)
    S.seed(1)
    q = torch.randn((N, H, T, d), device=DEV).to(torch.bfloat16)
    mask = causal_mask(N, T, torch.bfloat16)
    out_d, bd = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, False)
    out_s, bs = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    # identical estimator -> identical probability map -> identical mask
    assert torch.equal(bd['estimated_attention_probs'], bs['estimated_attention_probs'])
    assert torch.equal(bd['partial_attention_mask_before_interp'] > -1, bs['partial_attention_mask_before_interp'] > 0)
    ref = out_d.context_layer.float()
    rel = ((out_s.context_layer.float() - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel          # dense mode rounds scores/probs to bf16; the HIP path keeps them in fp32


@pytest.mark.parametrize("H,d", [(1, 32), (7, 64), (12, 64), (16, 32)])
def test_sparse_mode_is_causal(H, d):
    N, T, T_M, k = 1, 512, 64, 16
    layer = make_layer(H, d, T_M, k, T)
    S.seed(3)
    q = torch.randn((N, H, T, d), device=DEV)
    q2 = q.clone()
    q2[:, :, T // 2] = 3e5
    mask = causal_mask(N, T, torch.float32)
    a, _ = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True, capture=False)
    b, _ = run(layer, q2 * d ** -0.5, q2.clone(), q2.clone(), mask, True, capture=False)
    diff = (a.context_layer[:, :T // 2] - b.context_layer[:, :T // 2]).abs().sum().item()
    assert math.log10(diff + 1e-20) < -3, diff


def test_padded_batch_sparse_mode():
    """Left rows of a batch item masked out as padding (dst mask): dense and sparse mode agree."""
    N, H, T, d, T_M, k = 2, 4, 256, 32, 64, 16
    layer = make_layer(H, d, T_M, k, T)
    S.seed(5)
    q = torch.randn((N, H, T, d), device=DEV)
    mask = causal_mask(N, T, torch.float32).clone()
    fp_min = torch.finfo(torch.float32).min / 2
    mask[1, :, :, :16] = fp_min            # item 1: 16 left-pad keys -> its dst mask marks every row dead
    out_d, bd = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, False)
    out_s, bs = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    assert (out_d.context_layer - out_s.context_layer).abs().max().item() < 1e-4


def test_row_sharded_sparse_path_equals_the_full_one():
    """SURVEY 8e, N < G: query rows split in nnz-balanced blocks, K/V replicated; every block runs steps H..L on the HIP
    kernels as the tail of its own prefix (T_dst < T_src).  Concatenated, the blocks ARE the unsharded result."""
    from sea_attention_amd import distributed as D
    N, H, T, d, T_M, k = 1, 8, 1024, 64, 256, 64
    S.seed(11)
    probs = torch.softmax(torch.randn((N, H, T, T_M), device=DEV), -1).bfloat16()
    q = (torch.randn((N, H, T, d), device=DEV) * d ** -0.5).bfloat16()
    kk, v = torch.randn((N, H, T, d), device=DEV).bfloat16(), torch.randn((N, H, T, d), device=DEV).bfloat16()
    rs = torch.sigmoid(torch.randn((N, H, T), device=DEV)); mx = torch.sigmoid(torch.randn((N, H, T), device=DEV))
    avg = ops.cumavg(v)
    full = D.sparse_rows(ops, probs, q, kk, v, 0, T, k, rs, avg, mx)
    for world in (2, 4):
        parts = [D.sparse_rows(ops, probs, q, kk, v, *D.row_shard_bounds(T, world, r, k), k, rs, avg, mx) for r in range(world)]
        assert torch.equal(torch.cat(parts, dim=1), full)


@pytest.mark.parametrize("H,d", [(8, 64), (8, 80), (8, 128)])
def test_bf16_fast_path_equals_probing_path(H, d):
    """The shapes of BASELINE configs 3-5 in bf16: the layer's fast path (one-launch MLP where d allows it, C8 convolutions,
    tail fused with the top-k selection, sequence-parallel Performer for the few (n,h) pairs of one sequence) against
    the same layer with buffer probing on (separate tail and selection launches): same map and same CSR bit for bit, the
    context within a rounding step; and the context stays within bf16 noise of dense mode."""
    N, T, T_M, k = 1, 1024, 256, 32
    layer = make_layer(H, d, T_M, k, T, torch.bfloat16)
    layer.attention.context_layer_dtype = torch.bfloat16
    S.seed(5)
    x = torch.randn((N, H, T, d), device=DEV).to(torch.bfloat16)
    mask = causal_mask(N, T, torch.bfloat16)
    out_fast, _ = run(layer, x * d ** -0.5, x.clone(), x.clone(), mask, True, capture=False)
    out_probe, bp = run(layer, x * d ** -0.5, x.clone(), x.clone(), mask, True, capture=True)
    assert torch.equal(out_fast.estimated_attention_probs, out_probe.estimated_attention_probs)
    a, b = out_fast.partial_attention_mask, out_probe.partial_attention_mask
    assert torch.equal(a.crow, b.crow)
    for n in range(N):
        z = int(a.crow[n, -1])
        assert torch.equal(a.col[n, :z], b.col[n, :z])
    # (the probing path takes the cumulative average from its own kernel and keeps per-step buffers: same key sets, the
    # context a rounding step apart here and there)
    cf, cp = out_fast.context_layer.float(), out_probe.context_layer.float()
    assert ((cf - cp).norm() / cp.norm()).item() < 4e-3
    out_dense, _ = run(layer, x * d ** -0.5, x.clone(), x.clone(), mask, False, capture=False)
    ref = out_dense.context_layer.float()
    rel = ((out_fast.context_layer.float() - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel


@pytest.mark.parametrize("path", ["gather", "tile"])
@pytest.mark.parametrize("H,d", [(8, 64), (4, 128)])
def test_bf16_layer_sparse_half_against_fp32_evaluation_of_the_same_inputs(monkeypatch, H, d, path):
    """north_star's bar at LAYER level: the bf16 layer's sparse half (steps J-L as the layer really launches them, with
    the layer's own rounded q/k/v, gates, cumulative average and HIP-selected CSR) against an fp32 evaluation of the
    same rounded inputs on the same mask (the oracle): <= 1e-3 relative.  The 2-3e-2 comparisons with bf16 DENSE mode
    elsewhere in this file measure dense mode's own bf16 rounding of scores and probabilities, not this path."""
    from oracle import sea_oracle as O
    from sea_attention_amd.perlin_attention import attention as A
    N, T, T_M, k = 1, 1024, 256, 32
    layer = make_layer(H, d, T_M, k, T, torch.bfloat16)
    layer.attention.sparse_kernel = path
    S.seed(5)
    x = torch.randn((N, H, T, d), device=DEV).to(torch.bfloat16)
    mask = causal_mask(N, T, torch.bfloat16)
    seen = {}
    real = A.ops.sparse_attention

    def spy(q, k_, v, csr, **kw):
        seen.update(q=q, k=k_, v=v, csr=csr, kw=kw)
        return real(q, k_, v, csr, **kw)
    monkeypatch.setattr(A.ops, "sparse_attention", spy)
    out, _ = run(layer, x * d ** -0.5, x.clone(), x.clone(), mask, True, capture=False)     # fp32 context (reference default)
    kw = seen["kw"]
    assert kw["path"] == path and out.context_layer.dtype == torch.float32
    csr = seen["csr"]
    crow = csr.crow.cpu().long()
    col = csr.col[:, :int(crow[:, -1].max())].cpu().long()
    sparse = O.sparse_attention(seen["q"].float().cpu(), seen["k"].float().cpu(), seen["v"].float().cpu(), crow, col,
                                kw["row_scale"].cpu() if kw.get("row_scale") is not None else None)
    a = kw["mix"].cpu().unsqueeze(-1)
    ref = sparse * a + (1.0 - a) * kw["avg"].float().cpu()
    got = out.context_layer.view(N, T, H, d).permute(0, 2, 1, 3).cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 1e-3, rel
    assert (got - ref).abs().max().item() < 4e-3


def test_sparse_mode_returns_the_csr_probabilities_on_request():
    """a15: `partial_attention_probs` in sparse mode = the CSR of rs * softmax values (attention.py:1162-1171, :1349-1359),
    produced when a caller asks (PerlinSelfAttention.checkout_last_attention_probs): same structure as the mask, values
    equal the dense branch's `attention_matrix` (post-softmax x row scale) at the kept positions, and reference-style
    accessors work on the handle."""
    N, H, T, d, T_M, k = 1, 4, 256, 32, 64, 16
    layer = make_layer(H, d, T_M, k, T)
    S.seed(2)
    q = torch.randn((N, H, T, d), device=DEV)
    mask = causal_mask(N, T, torch.float32)
    out0, _ = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True, capture=False)
    assert out0.partial_attention_probs is None                              # nobody asked: not computed
    layer.checkout_last_attention_probs = True
    out, _ = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True, capture=False)
    p = out.partial_attention_probs
    assert p is not None and p.is_sparse_csr and tuple(p.shape) == (N, T, H * T) and layer.last_attention_probs is p
    assert torch.equal(out.context_layer, out0.context_layer)
    assert torch.equal(p.crow_indices(), out.partial_attention_mask.crow_indices())
    assert torch.equal(p.col_indices(), out.partial_attention_mask.col_indices())
    dense_p = ops.flat_csr_to_dense(p, T, H)                                 # (N,H,T,T)
    out_d, bd = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, False)
    scale = torch.sigmoid(bd['estimated_scales'][..., 0:1])
    ref = bd['attention_matrix'] * scale                                     # dense branch: probs (masked) x row scale
    assert (dense_p - ref).abs().max().item() < 2e-6
    rowsum = dense_p.sum(-1)                                                 # each non-empty (row, head): rs
    nonempty = (ops.flat_csr_to_dense(out.partial_attention_mask, T, H) > 0).any(-1)
    assert ((rowsum - scale.squeeze(-1)) * nonempty).abs().max().item() < 1e-5 and torch.all(rowsum[~nonempty] == 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_compressed_predictor_method(dtype):
    """`attention_predictor_method='comp'` (attention.py:293-312, 649-661): the map is softmax-over-codebook patches; its width
    is patch_count * patch_size, not attention_predictor_length.  Dense mode == sparse mode (the reference's consistency
    protocol), the map == the formula written out here, the CSR == the oracle's top-k + interpolation on that map."""
    from oracle import sea_oracle as O
    N, H, T, d, k = 2, 4, 256, 32, 8
    S.seed(42)
    pc = PerlinAttentionConfig(k=k, attention_predictor_length=64, performer_nb_factor=8, causal=True, k_flatten=True,
                               k_flatten_dim='causal_batch', context_output_method='mix', attention_predictor_method='comp',
                               attention_predictor_comp_book_size=8, attention_predictor_comp_patch_size=4,
                               attention_predictor_comp_patch_count=8)
    layer = PerlinSelfAttention(Cfg(H * d, H, T), pc).to(DEV).to(dtype).eval()
    att = layer.attention
    assert att.attention_predictor_comp_length == 32 and tuple(att.attention_predictor_comp_codebook.shape) == (8, 4)
    assert {"attention_predictor_comp_codebook", "attention_predictor_comp_enc.1.weight", "attention_predictor_comp_enc.2.bias",
            "attention_predictor_comp_dec_row.0.weight"} <= set(att.state_dict())            # the reference's parameter names
    S.seed(3)
    x = torch.randn((N, H, T, d), device=DEV)
    q, kk, v = (x * d ** -0.5).to(dtype), torch.randn_like(x).to(dtype), torch.randn_like(x).to(dtype)
    mask = causal_mask(N, T, dtype)
    out_d, bd = run(layer, q, kk, v, mask, False)
    out_s, bs = run(layer, q, kk, v, mask, True)
    T_M = 32
    probs = bs['estimated_attention_probs']
    assert tuple(probs.shape) == (N, H, T, T_M)
    # the formula, from the captured Performer output
    pv = bs['performer_value']
    t_pred = att.attention_predictor_comp_enc(pv)
    sc = att.attention_predictor_comp_dec_row(t_pred).view(N, H, T, 8, 8)
    ref = torch.softmax(torch.matmul(torch.softmax(sc, -1).view(-1, 8), att.attention_predictor_comp_codebook).view(N, H, T, -1), -1)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (probs.float() - ref.float()).abs().max().item() <= tol
    assert (bd['estimated_attention_probs'].float() - probs.float()).abs().max().item() <= tol
    # selection + interpolation on the layer's own map == oracle, bit for bit
    out_f, _ = run(layer, q, kk, v, mask, True, capture=False)
    csr = out_f.partial_attention_mask
    pm = out_f.estimated_attention_probs_m.float().cpu()
    keep = O.keep_counts_module(H, T, T_M, k)
    crow, col = O.resize_m_to_t_csr(O.grouped_topk_mask(pm, keep), k, T, True)
    assert torch.equal(csr.crow.cpu().long(), crow)
    for n in range(N):
        z = int(crow[n, -1])
        assert torch.equal(csr.col[n, :z].cpu().long(), col[n, :z])
    # dense mode and sparse mode agree on the context (fp32: the reference's 1e-5 SSE bar)
    if dtype == torch.float32:
        assert (out_d.context_layer.float() - out_s.context_layer.float()).square().sum().item() <= 1e-5
        assert torch.equal(bd['partial_attention_mask'] > -1, bs['partial_attention_mask'] > 0)
    else:
        assert torch.isfinite(out_s.context_layer).all()
    # no cached form (attention.py:650)
    from sea_attention_amd.perlin_attention.attention_state import PerlinAttentionState
    with pytest.raises(AssertionError):
        with torch.no_grad():
            att(q[:, :, -1:], kk, v, q[:, :, -1:], kk, v, q[:, :, -1:], kk, mask[:, :, -1:], None, None,
                last_state=PerlinAttentionState(att))
