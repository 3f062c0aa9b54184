"""The reference's own callers on this package (north_star: "src.trainer.perlin_trainer and src/models drop onto it
unchanged"): `tests/golden/callers.npz` holds what the reference's UNCHANGED `OPTAttention` / `OPTDecoderLayer`
(src/models/perlin_opt/perlin_opt.py:137-239,434-477,534-800) computed -- imported in place, with `src.models.perlin_attention`
aliased to this package as INTEGRATION.md section 1 shows -- on seeded inputs (`tests/golden/make_golden_callers.py`, build
container only).

CPU: `opt_plumbing.SeaOPTAttention` / `SeaOPTDecoderLayer`, the restated callers, loaded with the same state dict give the
same outputs EXACTLY (dense mode).  GPU: the sparse mode (`benchmarking=True`, HIP kernels) of the same block lands within the
dense-vs-sparse bar of the reference's consistency test (sum of squared errors <= 1e-5 on the context,
src/main/tests/test_perlin_opt_consist.py:198-232), stateless and decoded piecewise through `past_key_value = (k, v, state)`;
so does the benchmark's surgery (src/main/benchmark_bert.py:162-203).
"""
import os

import numpy as np
import pytest
import torch

from sea_attention_amd import opt_plumbing as P
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, get_default_config, register_default_config

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "callers.npz")
CASES = ["tiny", "opt125m"]


def state_dict_recipe(keys_shapes, seed):
    """tests/golden/make_golden_callers.py::state_dict_recipe, the same function (the fixture carries names, shapes, seed)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in sorted(keys_shapes):
        t = torch.randn(shape, generator=g) * 0.05
        if k.endswith("weight") and len(shape) == 1:
            t = t + 1.0
        sd[k] = t
    return sd


def exam_mask(N, T_dst, T_src, device="cpu"):
    t = torch.arange(T_dst, device=device).view(-1, 1) + (T_src - T_dst)
    s = torch.arange(T_src, device=device).view(1, -1)
    return ((s > t) * -32000.0).view(1, 1, T_dst, T_src).expand(N, 1, T_dst, T_src).contiguous()


@pytest.fixture
def restore_default():
    old = get_default_config()
    yield
    register_default_config(old)


def _load(case, use_cache=False):
    z = np.load(GOLD)
    hidden, heads, T, N, k, w, seed = (int(v) for v in z[f"{case}.meta"])
    register_default_config(PerlinAttentionConfig(
        k=k, attention_predictor_length=w, performer_nb_factor=8, k_flatten=True, k_flatten_dim="causal_batch", causal=True,
        context_output_method="mix", use_cache=use_cache))
    layer = P.SeaOPTDecoderLayer(hidden, heads, ffn_dim=hidden, max_position_embeddings=T).eval()
    keys_shapes = [(str(k_), tuple(int(i) for i in str(s_).split(",") if i)) for k_, s_ in zip(z[f"{case}.keys"], z[f"{case}.shapes"])]
    # the reference layer's state dict, key for key: nothing missing, nothing unexpected
    missing, unexpected = layer.load_state_dict(state_dict_recipe(keys_shapes, seed), strict=True)
    assert not missing and not unexpected
    return z, layer, torch.from_numpy(z[f"{case}.x"]), (hidden, heads, T, N)


@pytest.mark.parametrize("case", CASES)
def test_restated_callers_equal_the_reference_callers_exactly(case, restore_default):
    z, layer, x, (hidden, heads, T, N) = _load(case)
    mask = exam_mask(N, T, T)
    with torch.no_grad():
        y_attn, _, present = layer.self_attn(hidden_states=layer.self_attn_layer_norm(x), attention_mask=mask)
        y_layer = layer(hidden_states=x, attention_mask=mask)[0]
        y_exam = layer.exam_surgery(benchmarking=False)(hidden_states=x, attention_mask=mask)[0]
    assert torch.equal(y_attn, torch.from_numpy(z[f"{case}.y_attn"]))
    assert torch.equal(present[0], torch.from_numpy(z[f"{case}.present_k"]))
    assert torch.equal(y_layer, torch.from_numpy(z[f"{case}.y_layer"]))
    assert torch.equal(y_exam, torch.from_numpy(z[f"{case}.y_exam"]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_sparse_mode_on_the_gpu_matches_the_reference_callers_dense_output(case, restore_default):
    z, layer, x, (hidden, heads, T, N) = _load(case)
    layer = layer.cuda()
    x, mask = x.cuda(), exam_mask(N, T, T, "cuda")
    for m in layer.modules():                                   # benchmark_bert.py:172-173
        if hasattr(m, "benchmarking"):
            m.benchmarking = True
    with torch.no_grad():
        y_attn, _, _ = layer.self_attn(hidden_states=layer.self_attn_layer_norm(x), attention_mask=mask)
        y_layer = layer(hidden_states=x, attention_mask=mask)[0]
        y_exam = layer.exam_surgery()(hidden_states=x, attention_mask=mask)[0]
    for got, name in ((y_attn, "y_attn"), (y_layer, "y_layer"), (y_exam, "y_exam")):
        ref = torch.from_numpy(z[f"{case}.{name}"]).cuda()
        sse = float(((got.float() - ref) ** 2).sum())
        assert sse <= 1e-5, (name, sse)                          # test_perlin_opt_consist.py's bar on output.context_layer


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_cached_decoding_on_the_gpu_matches_the_reference_callers_stateless_output(case, restore_default):
    """`past_key_value = (k, v, state)` (perlin_opt.py:575-581,627-628): prefill + two short continuations reproduce the rows
    the reference's caller computed in one stateless dense pass."""
    z, layer, x, (hidden, heads, T, N) = _load(case, use_cache=True)
    layer = layer.cuda()
    x = x.cuda()
    layer.self_attn.benchmarking = True
    h = layer.self_attn_layer_norm(x)
    L = T - 8
    with torch.no_grad():
        y0, _, pkv = layer.self_attn(hidden_states=h[:, :L], attention_mask=exam_mask(N, L, L, "cuda"), use_cache=True)
        assert len(pkv) == 3
        y1, _, pkv = layer.self_attn(hidden_states=h[:, L:L + 4], attention_mask=exam_mask(N, 4, L + 4, "cuda"), past_key_value=pkv, use_cache=True)
        y2, _, pkv = layer.self_attn(hidden_states=h[:, L + 4:], attention_mask=exam_mask(N, 4, T, "cuda"), past_key_value=pkv, use_cache=True)
    got = torch.cat([y0, y1, y2], 1).float()
    ref = torch.from_numpy(z[f"{case}.y_attn"]).cuda()
    assert float(((got - ref) ** 2).sum()) <= 1e-5
    assert torch.equal(pkv[0].cpu(), torch.from_numpy(z[f"{case}.present_k"])) or \
        float((pkv[0].cpu() - torch.from_numpy(z[f"{case}.present_k"])).abs().max()) < 1e-5
