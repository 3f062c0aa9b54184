"""-m gpu: the reference's OWN benchmark grid on the fast path.

`src/main/benchmark_opt_ablation.py:160-186` sweeps k in {32, 64, 128} x predictor length w in {64, 128, 256, 384} on one
OPT-125m layer at T = 2048, batch 1; `src/main/exp_long_context.py:152` quotes T_M = 96, k = 128.  For every point the
LAYER runs here in sparse mode on 16-bit data and
  * the estimator stays on the hand-written kernels: one-launch predictor MLP, both C8 MFMA convolutions, the fused
    tail + top-k selection, the interpolation inside the attention launch -- asserted by spying on the operator table; no
    `nn.Linear` / `nn.Conv2d` forward of the predictor runs (library GEMM / MIOpen would), and the (N,H,T,T_M) map is left
    on chip (a `LazyTensor`);
  * the layer's flat CSR -- every row -- equals the oracle's grouped top-k + interpolation on the layer's own map, bit for
    bit, order included;
  * sampled context rows equal the oracle's sparse attention on the layer's own rounded inputs to north_star's 1e-3."""
import pytest
import torch

import sea_attention_amd as S
from oracle import sea_oracle as O
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
from sea_attention_amd.perlin_attention import attention as A

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRID = [(2048, k, w) for k in (32, 64, 128) for w in (64, 128, 256, 384)] + [(4096, 128, 96)]


class Cfg:
    def __init__(self, hidden, heads, max_pos):
        self.hidden_size, self.num_attention_heads, self.max_position_embeddings = hidden, heads, max_pos


def build_layer(H, d, T, T_M, K, dtype):
    S.seed(42)
    pc = PerlinAttentionConfig(k=K, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                               k_flatten_dim='causal_batch', context_output_method='mix')
    layer = PerlinSelfAttention(Cfg(H * d, H, T), pc).to(DEV).to(dtype).eval()
    for m in layer.modules():
        if hasattr(m, 'benchmarking'):
            m.benchmarking = True
    layer.attention.assume_not_padded = True
    return layer


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,K,T_M", GRID)
def test_reference_grid_runs_on_the_fused_kernels(monkeypatch, T, K, T_M, dtype):
    if dtype == torch.float16 and (K, T_M) not in ((64, 64), (128, 384), (128, 96), (32, 128)):
        pytest.skip("the fp16 twin runs on four points of the grid (time)")
    N, H, d = 1, 12, 64
    layer = build_layer(H, d, T, T_M, K, dtype)
    S.seed(7)
    x = torch.randn((N, H, T, d), device=DEV)
    q, k, v = (x * d ** -0.5).to(dtype), torch.randn_like(x).to(dtype), torch.randn_like(x).to(dtype)
    fp_min = torch.finfo(torch.float16).min / 2
    mask = ((torch.arange(T, device=DEV).view(1, T) > torch.arange(T, device=DEV).view(T, 1)) * fp_min).view(1, 1, T, T).to(dtype)

    calls = {}
    seen = {}

    def count(name):
        real = getattr(A.ops, name)

        def f(*a, **kw):
            calls[name] = calls.get(name, 0) + 1
            if name == "sparse_attention":
                seen.update(q=a[0], k=a[1], v=a[2], csr=a[3], kw=kw, pending=a[3].col_is_pending)
            return real(*a, **kw)
        monkeypatch.setattr(A.ops, name, f)
    for nm in ("performer_value", "predictor_mlp", "causal_conv_c8", "causal_conv_c8_z", "predictor_tail_select", "predictor_tail",
               "predictor_tail_z", "topk_to_csr",
               "split_layernorm", "split_layernorm_c8", "sparse_attention", "cumavg"):
        count(nm)
    # any predictor module that runs its torch forward (library GEMM / MIOpen) is a fallback: none may
    hooks = []
    ran = []
    att = layer.attention
    for name, mod in list(att.attention_predictor_enc.named_modules()) + list(att.attention_predictor_dec_row.named_modules()) + \
            list(att.attention_predictor_cnn.named_modules()) + list(att.attention_predictor_dec_scaler.named_modules()):
        if isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d, torch.nn.LayerNorm)):
            hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: ran.append(name)))
    with torch.no_grad():
        out = layer(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=mask)
    torch.cuda.synchronize()
    for h_ in hooks:
        h_.remove()
    assert ran == [], f"torch predictor modules ran: {ran}"
    # two (conv, ReLU) launches (the second one through sea_causal_conv_c8_z when the module's `conv_z_epilogue` is on)
    assert calls.get("performer_value") == 1 and calls.get("predictor_mlp") == 1
    assert calls.get("causal_conv_c8", 0) + calls.get("causal_conv_c8_z", 0) == 2
    assert calls.get("predictor_tail_select") == 1 and calls.get("sparse_attention") == 1
    for nm in ("predictor_tail", "predictor_tail_z", "topk_to_csr", "split_layernorm", "split_layernorm_c8", "cumavg"):
        assert nm not in calls, f"{nm} ran: the fused estimator was declined"
    assert seen["pending"], "the interpolation was not left to the attention launch"
    assert isinstance(out.estimated_attention_probs_m, A.ops.LazyTensor) and not out.estimated_attention_probs_m.is_materialized

    # ---- the layer's CSR == oracle top-k + interpolation on the layer's own map (all rows, bit for bit) ---------------------
    probs = out.estimated_attention_probs_m.float().cpu()           # (computed now, from the kept conv output)
    assert tuple(probs.shape) == (N, H, T, T_M)
    keep = O.keep_counts_module(H, T, T_M, K)
    crow, col = O.resize_m_to_t_csr(O.grouped_topk_mask(probs, keep), K, T, True)
    csr = out.partial_attention_mask
    assert torch.equal(csr.crow.cpu().long(), crow)
    z = int(crow[0, -1])
    assert torch.equal(csr.col[0, :z].cpu().long(), col[0, :z])

    # ---- sampled context rows == oracle on the layer's own rounded inputs ------------------------------------------------------
    g = torch.Generator().manual_seed(3)
    rows = torch.tensor(sorted(set(range(48)) | set(range(T - 48, T)) | set(torch.randint(0, T, (64,), generator=g).tolist())))
    lens = crow[0, rows + 1] - crow[0, rows]
    sub_crow = torch.zeros((1, rows.numel() + 1), dtype=torch.long)
    sub_crow[0, 1:] = lens.cumsum(0)
    sub_col = torch.cat([col[0, crow[0, r]:crow[0, r + 1]] for r in rows.tolist()]).view(1, -1)
    kw = seen["kw"]
    qh, kh, vh = (seen[n_].float().cpu() for n_ in ("q", "k", "v"))
    rs = kw["row_scale"].cpu()[:, :, rows] if kw.get("row_scale") is not None else None
    sparse = O.sparse_attention(qh[:, :, rows], kh, vh, sub_crow, sub_col, rs)
    a = kw["mix"].cpu()[:, :, rows].unsqueeze(-1)
    ref = sparse * a + (1.0 - a) * kw["avg"].float().cpu()[:, :, rows]
    got = out.context_layer.view(N, T, H, d).permute(0, 2, 1, 3).cpu()[:, :, rows]
    assert torch.isfinite(got).all()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 1e-3, rel


def test_eager_map_switch_writes_the_same_map():
    """`lazy_attention_probs = False` restores the reference's eager tensor; its values are the lazy handle's, bit for bit."""
    H, d, T, T_M, K = 12, 64, 1024, 128, 64
    dtype = torch.bfloat16
    layer = build_layer(H, d, T, T_M, K, dtype)
    S.seed(11)
    x = torch.randn((2, H, T, d), device=DEV)
    q, k, v = (x * d ** -0.5).to(dtype), torch.randn_like(x).to(dtype), torch.randn_like(x).to(dtype)
    fp_min = torch.finfo(torch.float16).min / 2
    mask = ((torch.arange(T, device=DEV).view(1, T) > torch.arange(T, device=DEV).view(T, 1)) * fp_min).view(1, 1, T, T).to(dtype).expand(2, 1, T, T)
    with torch.no_grad():
        a = layer(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=mask)
        layer.attention.lazy_attention_probs = False
        b = layer(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=mask)
    assert isinstance(a.estimated_attention_probs, A.ops.LazyTensor) and not isinstance(b.estimated_attention_probs, A.ops.LazyTensor)
    assert torch.equal(a.context_layer, b.context_layer)
    assert torch.equal(a.estimated_attention_probs_m, b.estimated_attention_probs_m)
    ca, cb = a.partial_attention_mask, b.partial_attention_mask
    assert torch.equal(ca.crow, cb.crow)
    for i in range(2):                                   # (entries past a row's last one are capacity, never written)
        z = int(ca.crow[i, -1])
        assert torch.equal(ca.col[i, :z], cb.col[i, :z])


@pytest.mark.parametrize("H,d,T,T_M,K", [(12, 64, 1024, 256, 64), (32, 64, 512, 256, 64), (12, 64, 512, 96, 32)])
def test_conv_z_epilogue_switch_changes_no_bit(H, d, T, T_M, K):
    """`conv_z_epilogue = True` (round 5: the tail's 1x1 convolution evaluated in the last conv launch's epilogue, z handed to
    the tail instead of the activation) is an optimisation that measured slower and stays off by default; on, the layer's
    outputs -- context, map, CSR -- are bit for bit the default's."""
    dtype = torch.bfloat16
    layer = build_layer(H, d, T, T_M, K, dtype)
    S.seed(13)
    x = torch.randn((2, H, T, d), device=DEV)
    q, k, v = (x * d ** -0.5).to(dtype), torch.randn_like(x).to(dtype), torch.randn_like(x).to(dtype)
    fp_min = torch.finfo(torch.float16).min / 2
    mask = ((torch.arange(T, device=DEV).view(1, T) > torch.arange(T, device=DEV).view(T, 1)) * fp_min).view(1, 1, T, T).to(dtype).expand(2, 1, T, T)
    assert layer.attention.conv_z_epilogue is False
    with torch.no_grad():
        a = layer(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=mask)
        layer.attention.conv_z_epilogue = True
        b = layer(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=mask)
    assert torch.equal(a.context_layer, b.context_layer)
    assert torch.equal(a.estimated_attention_probs_m, b.estimated_attention_probs_m)
    ca, cb = a.partial_attention_mask, b.partial_attention_mask
    assert torch.equal(ca.crow, cb.crow)
    for i in range(2):
        z = int(ca.crow[i, -1])
        assert torch.equal(ca.col[i, :z], cb.col[i, :z])
