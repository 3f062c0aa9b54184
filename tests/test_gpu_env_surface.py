"""-m gpu: the reference's run-time environment switches on the HIP path.

  DYNAMIC_K=<k>    overrides pconfig.k at every forward (attention.py:348-351)
  QUERY_SKIPS=<s>  the predictor runs on every s-th query row and its map rows are repeated s times
                   (attention.py:598,617-619,640-644)

Both are read by `PerlinAttention.forward`; these tests set them and check, in sparse mode (HIP kernels), that the mask equals
dense mode's, the context stays inside the usual bars, and the fused fast paths (one-launch MLP, tail + selection) are either
taken or cleanly declined."""
import pytest
import torch

import sea_attention_amd as S
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention, ops
from sea_attention_amd.perlin_attention import attention as A

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


def _spy(monkeypatch, names):
    calls = {n: 0 for n in names}
    for n in names:
        real = getattr(A.ops, n)

        def wrap(*a, _real=real, _n=n, **kw):
            calls[_n] += 1
            return _real(*a, **kw)
        monkeypatch.setattr(A.ops, n, wrap)
    return calls


def _keep_counts(csr, H, T_M):
    """kept pixels per (n, t) from the selection's bit masks"""
    bits = csr.bits.view(torch.int32)
    x = bits.to(torch.int64) & 0xffffffff
    cnt = torch.zeros_like(x)
    for s in range(32):
        cnt += (x >> s) & 1
    return cnt.sum(-1)


# ---------------------------------------------------------------------------------------------------------------- DYNAMIC_K
def test_dynamic_k_fp32_dense_and_sparse_agree(monkeypatch):
    monkeypatch.setenv("DYNAMIC_K", "8")
    N, H, T, d, T_M, k = 2, 4, 512, 32, 64, 16
    layer = make_layer(H, d, T_M, k, T)
    S.seed(4)
    q = torch.randn((N, H, T, d), device=DEV)
    mask = causal_mask(N, T, torch.float32)
    with pytest.warns(UserWarning, match="dynamic k 8"):
        out_d, bd = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, False)
    assert layer.pconfig.k == 8                                            # the override sticks, as in the reference
    out_s, bs = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    assert torch.equal(bd['partial_attention_mask_before_interp'] > -1, bs['partial_attention_mask_before_interp'] > 0)
    assert torch.equal(bd['partial_attention_mask'] > -1, bs['partial_attention_mask'] > 0)
    want = ops.keep_table_causal(H, T, T_M, 8).float()
    assert torch.equal(bs['per_item_top_k'].view(-1).cpu(), want)
    assert (out_d.context_layer - out_s.context_layer).square().sum().item() <= 1e-5
    # and it really is another mask than k = 16 gives
    monkeypatch.delenv("DYNAMIC_K")
    layer.pconfig.k = 16
    out_16, b16 = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    assert (b16['partial_attention_mask'] > 0).sum() > (bs['partial_attention_mask'] > 0).sum()


@pytest.mark.parametrize("d", [64, 128])
def test_dynamic_k_on_the_fused_bf16_fast_path(monkeypatch, d):
    """16-bit inference shapes: the one-launch MLP and the tail fused with the selection are TAKEN under DYNAMIC_K, the keep
    table follows the new k, and the CSR equals the oracle's top-k + interpolation on the layer's own map."""
    from oracle import sea_oracle as O
    monkeypatch.setenv("DYNAMIC_K", "32")
    N, H, T, T_M, k = 1, 8, 1024, 256, 64
    layer = make_layer(H, d, T_M, k, T, torch.bfloat16)
    layer.attention.context_layer_dtype = torch.bfloat16
    calls = _spy(monkeypatch, ["predictor_mlp", "predictor_tail_select", "predictor_tail", "topk_to_csr"])
    S.seed(6)
    x = torch.randn((N, H, T, d), device=DEV).to(torch.bfloat16)
    mask = causal_mask(N, T, torch.bfloat16)
    with pytest.warns(UserWarning, match="dynamic k 32"):
        out, _ = run(layer, x * d ** -0.5, x.clone(), x.clone(), mask, True, capture=False)
    assert calls == {"predictor_mlp": 1, "predictor_tail_select": 1, "predictor_tail": 0, "topk_to_csr": 0}
    csr = out.partial_attention_mask
    keep = O.keep_counts_module(H, T, T_M, 32)
    assert torch.equal(_keep_counts(csr, H, T_M).cpu().view(-1), keep.view(-1).long().clamp_max(H * T_M))   # the table is unclamped
    probs = out.estimated_attention_probs_m.float().cpu()
    crow, col = O.resize_m_to_t_csr(O.grouped_topk_mask(probs, keep), 32, T, True)
    assert torch.equal(csr.crow.cpu().long(), crow)
    z = int(crow[0, -1])
    assert torch.equal(csr.col[0, :z].cpu().long(), col[0, :z])
    assert torch.isfinite(out.context_layer.float()).all()


# -------------------------------------------------------------------------------------------------------------- QUERY_SKIPS
def test_query_skips_fp32_dense_and_sparse_agree(monkeypatch):
    monkeypatch.setenv("QUERY_SKIPS", "2")
    N, H, T, d, T_M, k = 1, 4, 512, 32, 64, 16
    layer = make_layer(H, d, T_M, k, T)
    S.seed(8)
    q = torch.randn((N, H, T, d), device=DEV)
    mask = causal_mask(N, T, torch.float32)
    out_d, bd = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, False)
    out_s, bs = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    p = bs['estimated_attention_probs']
    assert p.shape[-2] == T and torch.equal(p[:, :, 0::2], p[:, :, 1::2])           # every predictor row serves two queries
    assert (bd['estimated_attention_probs'].float() - p.float()).square().sum().item() <= 1e-5
    assert torch.equal(bd['partial_attention_mask_before_interp'] > -1, bs['partial_attention_mask_before_interp'] > 0)
    assert torch.equal(bd['partial_attention_mask'] > -1, bs['partial_attention_mask'] > 0)
    assert (out_d.context_layer - out_s.context_layer).square().sum().item() <= 1e-5
    # the rows differ from the unskipped layer's (the switch does something)
    monkeypatch.setenv("QUERY_SKIPS", "1")
    out_1, b1 = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    assert not torch.equal(b1['estimated_attention_probs'], p)


def test_query_skips_declines_the_fused_paths_cleanly_on_bf16(monkeypatch):
    """16-bit inference shapes under QUERY_SKIPS=2: the one-launch MLP and the fused tail + selection do not apply (they
    produce one map row per query row); the layer takes the unfused HIP estimator on the subsampled rows, repeats the map,
    selects with the stand-alone top-k launch -- and dense mode, which takes the same estimator, sees the same map and mask."""
    monkeypatch.setenv("QUERY_SKIPS", "2")
    N, H, T, d, T_M, k = 1, 8, 1024, 64, 256, 32
    layer = make_layer(H, d, T_M, k, T, torch.bfloat16)
    calls = _spy(monkeypatch, ["predictor_mlp", "predictor_tail_select", "predictor_tail", "predictor_tail_z", "topk_to_csr",
                               "causal_conv_c8", "causal_conv_c8_z"])
    S.seed(10)
    x = torch.randn((N, H, T, d), device=DEV).to(torch.bfloat16)
    mask = causal_mask(N, T, torch.bfloat16)
    out_s, _ = run(layer, x * d ** -0.5, x.clone(), x.clone(), mask, True, capture=False)
    assert calls["predictor_mlp"] == 0 and calls["predictor_tail_select"] == 0
    assert calls["predictor_tail"] + calls["predictor_tail_z"] == 1 and calls["topk_to_csr"] == 1
    assert calls["causal_conv_c8"] + calls["causal_conv_c8_z"] == 2
    p = out_s.estimated_attention_probs_m
    assert p.shape == (N, H, T, T_M) and torch.equal(p[:, :, 0::2], p[:, :, 1::2])
    out_d, bd = run(layer, x * d ** -0.5, x.clone(), x.clone(), mask, False)
    assert torch.equal(bd['estimated_attention_probs'], p)
    dense_mask = bd['partial_attention_mask'] > -1
    assert torch.equal(ops.flat_csr_to_dense(out_s.partial_attention_mask, T, H) > 0, dense_mask)
    ref = out_d.context_layer.float()
    rel = ((out_s.context_layer.float() - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel                                                       # bf16 dense mode rounds scores / probs
