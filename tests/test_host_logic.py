"""Host-side logic of the package on CPU: module plumbing in dense mode (the reference's own
CPU-runnable configuration), keep tables, capacity bound, predictor blocks, Performer restatement."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sea_attention_amd as S
from sea_attention_amd.perlin_attention import (PerlinAttentionConfig, PerlinSelfAttention, PerlinAttention,
                                                PerlinAttentionOutput, register_default_config, get_default_config, ops)
from sea_attention_amd.perlin_attention import modules, performer
from oracle import sea_oracle as O


class Cfg:
    def __init__(self, hidden, heads, max_pos=256):
        self.hidden_size, self.num_attention_heads, self.max_position_embeddings = hidden, heads, max_pos


def causal_mask(N, T, dtype=torch.float32):
    fp_min = torch.finfo(dtype).min / 2
    m = ((torch.arange(T).view(1, T) > torch.arange(T).view(T, 1)) * fp_min).view(1, 1, T, T)
    return m.expand(N, 1, T, T).contiguous()


def make_layer(H=4, d=16, T_M=32, k=8, max_pos=256, seed=0):
    torch.manual_seed(seed)
    pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True,
                               k_flatten=True, k_flatten_dim='causal_batch', context_output_method='mix')
    return PerlinSelfAttention(Cfg(H * d, H, max_pos), pc).eval()


def test_keep_tables_equal_oracle():
    for H, T, T_M, k in [(12, 2048, 256, 64), (32, 4096, 256, 64), (40, 4096, 256, 64), (4, 100, 32, 8)]:
        a = ops.keep_table_causal(H, T, T_M, k)
        b = torch.clamp_max(O.keep_counts_module(H, T, T_M, k), H * T_M).to(torch.int32)
        assert torch.equal(a, b)
        assert torch.equal(ops.keep_table_kernel_test(H, T, T_M, k), O.keep_counts_kernel_test(H, T, T_M, k).to(torch.int32))


def test_z_capacity_bounds_real_nnz(golden):
    for case in ["tiny", "ragged", "short", "big", "clamp"]:
        g = golden(case)
        N, H, T_DST, T_SRC, T_M, k, d, causal = [int(x) for x in g["meta"]]
        keep = ops.keep_table_kernel_test(H, T_DST, T_M, k)
        cap = ops.z_capacity(keep, H, T_DST, T_SRC, T_M, k, True)
        assert cap >= int(g["crow"][:, -1].max())


def test_dense_resize_op_equals_golden(golden):
    for case in ["tiny", "mid", "ragged", "short", "big", "clamp"]:
        g = golden(case)
        N, H, T_DST, T_SRC, T_M, k, d, causal = [int(x) for x in g["meta"]]
        cm = causal_mask(N, T_SRC, torch.float16).float()
        out = ops.resize_from_m_to_t(torch.from_numpy(g["mask_m"]), 0, cm, target_width=T_SRC, is_causal=True,
                                     k=k, oversampled=1.0).masked_fill(cm < -1, 0)
        assert np.array_equal(out.numpy(), g["mask_dense"])


def test_causal_conv_equals_masked_full_kernel():
    """modules.py:96-192 semantics: (2k-1) x k kernel, lower rows masked, symmetric time padding."""
    torch.manual_seed(1)
    conv = modules.CausalConv2d(6, 5, 3, padding=2, dilation=2, causal=True)
    with torch.no_grad():
        conv.weight.add_(torch.randn_like(conv.weight))        # make the dead half non-zero: it must be ignored
    x = torch.randn(2, 6, 20, 9)
    w = conv.weight.masked_fill(conv.weight_mask == 0, 0)
    ref = F.conv2d(x, w, conv.bias, 1, ((3 - 1) * 2, 2), 2)
    assert torch.allclose(conv(x), ref, atol=1e-5)
    assert conv(x).shape == x.shape[:1] + (5,) + x.shape[2:]
    c1 = modules.CausalConv2d(6, 3, 1, padding=1, causal=True)
    assert c1(x).shape == (2, 3, 20, 11)
    # causality: future rows do not influence earlier outputs
    x2 = x.clone(); x2[:, :, 12:] += 100
    assert torch.allclose(conv(x)[:, :, :12], conv(x2)[:, :, :12], atol=1e-5)


def test_performer_chunked_equals_naive_prefix_sums():
    torch.manual_seed(2)
    B, H, T, nb, e = 2, 3, 200, 9, 10
    qp, kp = torch.rand(B, H, T, nb) + 1e-3, torch.rand(B, H, T, nb) + 1e-3
    v = torch.randn(B, H, T, e)
    out = performer.causal_linear_attention(qp, kp, v, chunk=64)
    ksum = kp.cumsum(-2) + 1e-6
    ctx = torch.einsum('...nd,...ne->...nde', kp, v).cumsum(-3)
    ref = torch.einsum('...nde,...nd,...n->...ne', ctx, qp, 1.0 / torch.einsum('...nd,...nd->...n', qp, ksum))
    assert torch.allclose(out, ref, atol=1e-4, rtol=1e-4)
    fa = performer.FastAttention(16, nb_features=int(16 * math.log(16) / 8), causal=True, generalized_attention=True)
    assert fa.projection_matrix.shape == (5, 16)
    assert fa(torch.randn(1, 2, 30, 16), torch.randn(1, 2, 30, 16), torch.randn(1, 2, 30, 32)).shape == (1, 2, 30, 32)


def test_state_dict_names_follow_reference_layout():
    m = make_layer()
    keys = set(m.state_dict().keys())
    expected = {
        'query_lora.lora_a', 'key_lora.lora_b', 'value_lora.lora_a',
        'attention.performer.projection_matrix',
        'attention.performer_proj_updater.instance.projection_matrix',
        'attention.performer_proj_updater.calls_since_last_redraw',
        'attention.attention_predictor_enc_head_embd',
        'attention.attention_predictor_enc_per_layer.0.weight',
        'attention.attention_predictor_enc.0.weight', 'attention.attention_predictor_enc.1.bias',
        'attention.attention_predictor_dec_row.0.weight',
        'attention.attention_predictor_cnn.0.module.weight',
        'attention.attention_predictor_cnn.1.module.net.0.module.weight',
        'attention.attention_predictor_cnn.1.module.net.0.module.weight_mask',
        'attention.attention_predictor_cnn.1.module.net.2.module.bias',
        'attention.attention_predictor_cnn.1.module.net.5.module.weight',
        'attention.attention_predictor_cnn.2.module.bias',
        'attention.attention_predictor_dec_scaler.0.weight',
        'attention.norm_performer.weight', 'attention.norm_partial.weight', 'attention.norm_random.weight',
        'attention.norm.weight', 'attention.v_eye_learned', 'attention.v_eye_learned_causal',
    }
    assert expected <= keys, expected - keys
    a = m.attention
    H, d, T_M = 4, 16, 32
    assert a.attention_predictor_cnn[1].module.net[0].module.weight.shape == (2 * H, 2 * H, 5, 3)
    assert a.attention_predictor_cnn[1].module.net[5].module.weight.shape == (H, 2 * H, 1, 1)
    assert a.attention_predictor_dec_row[0].weight.shape == ((T_M // 4) * 2, 2 * d)
    assert a.v_eye_learned_causal.shape == (1, 1, 256, d)
    assert a.performer_nb_features == int(d * math.log(d) / 8)


def test_default_config_registry():
    old = get_default_config()
    try:
        pc = PerlinAttentionConfig(k=64, attention_predictor_length=256, performer_nb_factor=8, causal=True)
        register_default_config(pc)
        assert get_default_config() is pc
        att = PerlinAttention(Cfg(64, 4))
        assert att.pconfig is pc and att.benchmarking is False
        pc.check_validity()
        assert "PerlinAttentionConfig(" in repr(pc)
    finally:
        register_default_config(old)


def test_dense_mode_forward_cpu_and_buffers():
    """cfg-1 style plumbing: the dense (training/eval) mode of the module on CPU."""
    m = make_layer()
    N, H, T, d = 2, 4, 96, 16
    torch.manual_seed(3)
    q, k, v = torch.randn(N, H, T, d) * d ** -0.5, torch.randn(N, H, T, d), torch.randn(N, H, T, d)
    bench = S.get_bench()
    bench.activate_temp_buffers = True
    bench.reset_temp_buffers()
    try:
        with torch.no_grad():
            out = m(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=causal_mask(N, T))
    finally:
        bench.activate_temp_buffers = False
    assert isinstance(out, PerlinAttentionOutput)
    assert out.context_layer.shape == (N, T, H * d)
    assert out.partial_attention_mask.shape == (N, H, T, T)
    assert out.estimated_attention_probs_m.shape == (N, H, T, 32)
    for name in ['q', 'k', 'v', 'attention_mask', 'v_for_atten', 'performer_context_layer', 'performer_value',
                 'estimated_attention_score_dec_row', 't_attention_predictor', 'estimated_attention_score',
                 'estimated_attention_probs', 'masked_estimated_attention_probs', 'per_item_top_k', 't_dead_mask',
                 'partial_attention_mask_before_interp', 'partial_attention_mask', 'q_for_score', 'k_for_score',
                 'partial_attention_scores', 'attention_matrix', 'estimated_scales', 'average_scale',
                 'average_context_layer', 'partial_context_layer_1', 'partial_context_layer_2',
                 'partial_context_layer_sparse', 'partial_context_layer']:
        assert name in bench.buffers, name
    # the module's dense top-k agrees with the oracle's grouped top-k on the captured probabilities
    probs = bench.get_temp_buffer('masked_estimated_attention_probs')
    keep = O.keep_counts_module(H, T, 32, 8)
    alive = bench.get_temp_buffer('partial_attention_mask_before_interp') > -1
    assert torch.equal(alive, O.grouped_topk_mask(probs, keep) > 0)
    # ... and its dense attention equals the oracle's sparse composition on that mask
    crow, col = O.resize_m_to_t_csr(alive.float(), 8, T, True)
    scales = bench.get_temp_buffer('estimated_scales')
    ref = O.sparse_attention(q, k, v, crow, col, torch.sigmoid(scales[..., 0]))
    assert torch.allclose(bench.get_temp_buffer('partial_context_layer_1'), ref, atol=2e-5, rtol=1e-4)
    ref2 = O.mix(ref, v, scales[..., 1]).permute(0, 2, 1, 3).reshape(N, T, H * d)
    assert torch.allclose(out.context_layer, ref2, atol=2e-5, rtol=1e-4)
    bench.reset_temp_buffers()


def test_dense_mode_is_causal():
    """Causality canary of src/main/tests/test_perlin_opt_causality.py:246-276 on the CPU path."""
    m = make_layer(seed=5)
    N, H, T, d = 1, 4, 64, 16
    torch.manual_seed(4)
    q, k, v = torch.randn(N, H, T, d) * d ** -0.5, torch.randn(N, H, T, d), torch.randn(N, H, T, d)
    q2, k2, v2 = q.clone(), k.clone(), v.clone()
    for t_ in (q2, k2, v2):
        t_[:, :, T // 2] = 3e5
    with torch.no_grad():
        a = m(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=causal_mask(N, T)).context_layer
        b = m(None, None, None, query_layer=q2, key_layer=k2, value_layer=v2, attention_mask=causal_mask(N, T)).context_layer
    assert math.log10((a[:, :T // 2] - b[:, :T // 2]).abs().sum().item() + 1e-20) < -3


def test_kd_losses_in_dense_training_mode():
    m = make_layer().train()
    N, H, T, d = 1, 4, 32, 16
    torch.manual_seed(6)
    q, k, v = torch.randn(N, H, T, d) * d ** -0.5, torch.randn(N, H, T, d), torch.randn(N, H, T, d)
    cm = causal_mask(N, T)
    truth = torch.matmul(q, k.transpose(-1, -2)) + cm
    ctx_truth = torch.matmul(torch.softmax(truth, -1), v).permute(0, 2, 1, 3).reshape(N, T, H * d)
    out = m(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=cm,
            attention_scores_truth=truth, context_layer_truth=ctx_truth)
    assert out.loss.requires_grad and torch.isfinite(out.loss)
    out.loss.backward()
    assert m.attention.attention_predictor_enc[0].weight.grad is not None


@pytest.mark.parametrize("chunk", [1, 3, 4])
def test_dense_mode_head_chunked_equals_the_unchunked_form(chunk):
    """`dense_head_chunk` bounds the memory of dense (training) mode -- the T x T scores, masks and
    probabilities exist for `chunk` heads at a time -- without changing context, loss or gradients (the KD terms are means
    over heads; training mode with the reference's 10 % resize jitter off so that both runs see the same mask)."""
    m = make_layer().eval()
    N, H, T, d = 2, 4, 48, 16
    torch.manual_seed(8)
    q, k, v = torch.randn(N, H, T, d) * d ** -0.5, torch.randn(N, H, T, d), torch.randn(N, H, T, d)
    cm = causal_mask(N, T)
    truth = torch.matmul(q, k.transpose(-1, -2)) + cm
    ctx_truth = torch.matmul(torch.softmax(truth, -1), v).permute(0, 2, 1, 3).reshape(N, T, H * d)
    res = {}
    for hc in (None, chunk):
        m.attention.dense_head_chunk = hc
        m.zero_grad()
        out = m(None, None, None, query_layer=q, key_layer=k, value_layer=v, attention_mask=cm,
                attention_scores_truth=truth, context_layer_truth=ctx_truth)
        out.loss.backward()
        res[hc] = (out, m.attention.attention_predictor_enc[0].weight.grad.clone(),
                   m.attention.attention_predictor_dec_scaler[0].weight.grad.clone())
    a, b = res[None][0], res[chunk][0]
    assert torch.allclose(a.context_layer, b.context_layer, atol=1e-6, rtol=1e-5)
    assert abs(a.loss.item() - b.loss.item()) < 1e-5 * max(1.0, abs(a.loss.item()))
    for ga, gb in zip(res[None][1:], res[chunk][1:]):
        assert torch.allclose(ga, gb, atol=1e-6, rtol=1e-4)
    # the reference's form returns the three (N,H,T,T) tensors; the bounded form does not build them
    assert a.partial_attention_mask.shape == (N, H, T, T) and a.partial_attention_probs.shape == (N, H, T, T)
    assert a.dense_attention_probs.shape == (N, H, T, T) and a.estimated_attention_probs.shape == (N, H, T, T)
    assert b.partial_attention_mask is None and b.partial_attention_probs is None and b.dense_attention_probs is None
    assert b.estimated_attention_probs.shape == (N, H, T, 32)
    m.attention.dense_head_chunk = None


def test_sparse_mode_requires_gpu():
    m = make_layer()
    for mod in m.modules():
        if hasattr(mod, 'benchmarking'):
            mod.benchmarking = True
    N, H, T, d = 1, 4, 32, 16
    q = torch.randn(N, H, T, d)
    with pytest.raises(RuntimeError, match="no CPU fallback"), torch.no_grad():
        m(None, None, None, query_layer=q, key_layer=q, value_layer=q, attention_mask=causal_mask(N, T))


def test_bench_regions_and_tracetree():
    b = S.utils.Benchmark()
    b.disabled = False
    with b.region("outer"):
        with b.region("inner"):
            pass
    assert set(b.todict()) == {"outer", "inner"}
    assert "> outer" in b.format_tracetree() and "inner" in b.format_tracetree()


def test_weight_prep_cache_is_keyed_on_tensor_identity():
    """A freed weight whose address is reused by a new tensor of the same shape must not hit the re-layout cache;
    views of one live parameter (fresh view objects per call) must."""
    from sea_attention_amd.perlin_attention.ops import predictor as P
    P._prep_cache.clear()
    builds = []
    def prep(t):
        return P._cached("t", (t,), torch.float32, lambda: (builds.append(1), t.clone())[1])
    for i in range(8):                         # same shape, same allocator bucket: addresses get recycled
        t = torch.full((64, 64), float(i))
        assert torch.equal(prep(t), t)
        del t
    w = torch.nn.Parameter(torch.randn(4, 6, 1, 1))
    n0 = len(builds)
    a = prep(w[:, :, 0, 0]); b = prep(w[:, :, 0, 0])
    assert a is b and len(builds) == n0 + 1
    with torch.no_grad():
        w.mul_(2)                              # in-place update bumps the version: rebuilt
    c = prep(w[:, :, 0, 0])
    assert len(builds) == n0 + 2 and torch.equal(c, w[:, :, 0, 0])


def test_decoding_sub_states_match_one_shot_computation():
    """attention_state.py on CPU: feeding the rows in chunks through the carried states equals the one-shot formulas
    (causal linear attention, causal CNN, cumulative average)."""
    from sea_attention_amd.perlin_attention.attention_state import PerformerState, CnnWindowState, CumAvgState
    from sea_attention_amd.perlin_attention.performer import causal_linear_attention
    from sea_attention_amd.perlin_attention.modules import CausalConv2d
    g = torch.Generator().manual_seed(0)
    N, H, T, nb, e = 2, 3, 37, 11, 6
    qp, kp = torch.rand((N, H, T, nb), generator=g) + 1e-3, torch.rand((N, H, T, nb), generator=g) + 1e-3
    v = torch.randn((N, H, T, e), generator=g)
    ref = causal_linear_attention(qp.double(), kp.double(), v.double(), chunk=T).float()
    st, outs, pos = PerformerState(), [], 0
    for step in (5, 1, 1, 17, 13):
        st, o = st.step(qp[..., pos:pos + step, :], kp[..., pos:pos + step, :], v[..., pos:pos + step, :])
        outs.append(o); pos += step
    assert st.seq_index == T
    torch.testing.assert_close(torch.cat(outs, -2), ref, atol=1e-5, rtol=1e-5)
    # cumulative average
    cs, outs, pos = CumAvgState(), [], 0
    for step in (9, 1, 27):
        cs, o = cs.step(v[..., pos:pos + step, :]); outs.append(o); pos += step
    torch.testing.assert_close(torch.cat(outs, -2), v.cumsum(-2) / torch.arange(1, T + 1).view(1, 1, -1, 1), atol=1e-5, rtol=1e-5)
    # two dilated causal convs look back 8 rows: the windowed state equals the full pass
    cnn = torch.nn.Sequential(CausalConv2d(4, 4, 3, padding=2, dilation=2, causal=True), torch.nn.ReLU(),
                              CausalConv2d(4, 4, 3, padding=2, dilation=2, causal=True))
    x = torch.randn((1, 4, 30, 8), generator=g)
    with torch.no_grad():
        full = cnn(x)
        ws, outs, pos = CnnWindowState(), [], 0
        for step in (12, 1, 1, 16):
            ws, o = ws.step(cnn, x[..., pos:pos + step, :]); outs.append(o); pos += step
    torch.testing.assert_close(torch.cat(outs, -2), full, atol=1e-5, rtol=1e-5)
    assert ws.rows.shape[-2] == CnnWindowState.LOOKBACK


def test_cnn_lookback_follows_the_conv_stack(monkeypatch):
    """The kv-cache CNN window is derived from the predictor's causal convolutions: 8 rows for the standard
    two dilated 3-tap convs, 12 with PERLIN_HOTFIX_OPT_DEEPER=1."""
    from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
    from sea_attention_amd.perlin_attention.attention_state import cnn_lookback

    class Cfg:
        hidden_size, num_attention_heads, max_position_embeddings = 64, 4, 128
    pc = PerlinAttentionConfig(k=8, attention_predictor_length=32, performer_nb_factor=8, causal=True, k_flatten=True,
                               k_flatten_dim='causal_batch', context_output_method='mix')
    assert cnn_lookback(PerlinSelfAttention(Cfg(), pc).attention.attention_predictor_cnn) == 8
    monkeypatch.setenv("PERLIN_HOTFIX_OPT_DEEPER", "1")
    assert cnn_lookback(PerlinSelfAttention(Cfg(), pc).attention.attention_predictor_cnn) == 12


def test_prep_cache_key_sees_storage_swaps_and_device_tags():
    """ops.predictor._cached: a parameter whose storage was swapped (`param.data = other`) or that moved must miss; an
    untouched one must hit; clear_prep_cache() empties it."""
    import torch
    from sea_attention_amd.perlin_attention.ops import predictor as P
    P.clear_prep_cache()
    w = torch.nn.Parameter(torch.randn(4, 4))
    calls = []
    build = lambda: calls.append(1) or w.detach().clone()
    a = P._cached("t", (w,), torch.float32, build)
    b = P._cached("t", (w,), torch.float32, build)
    assert a is b and len(calls) == 1
    w.data = torch.randn(4, 4)                                   # same Parameter object, same _version, new storage
    c = P._cached("t", (w,), torch.float32, build)
    assert c is not a and len(calls) == 2
    with torch.no_grad():
        w.add_(1.0)                                              # autograd-visible in-place edit: version bump
    P._cached("t", (w,), torch.float32, build)
    assert len(calls) == 3
    P.clear_prep_cache()
    P._cached("t", (w,), torch.float32, build)
    assert len(calls) == 4


def test_fused_interpolation_is_for_launches_of_many_rows():
    """`fused_interp_supported`: the head shapes of the fused form, and -- when the caller says how many (n, h, t) rows the
    launch has -- only above `attention_few_rows()` (a decoding step emits its columns and lets the plain kernel pre-touch its
    K / V rows; include/sea_hip.h, sea_attention_few_rows).  The library answers without a GPU."""
    import torch
    from sea_attention_amd.perlin_attention import ops
    few = ops.attention_few_rows()
    assert few >= 256                                                       # at least a batch-8, 32-head decoding step
    for dt, d in ((torch.bfloat16, 64), (torch.float16, 80), (torch.bfloat16, 128), (torch.float32, 32), (torch.float32, 64)):
        assert ops.fused_interp_supported(dt, d, 256)
        assert ops.fused_interp_supported(dt, d, 256, rows=few + 1)
        assert not ops.fused_interp_supported(dt, d, 256, rows=few)
    assert not ops.fused_interp_supported(torch.bfloat16, 32, 256)          # rows of 4 lanes
    assert not ops.fused_interp_supported(torch.float32, 128, 256)          # rows wider than 16 lanes
    assert not ops.fused_interp_supported(torch.bfloat16, 64, 48)           # T_m not a multiple of 32


# ---- round 4 host logic ----------------------------------------------------------------------------------------------
def test_lazy_tensor_computes_once_and_only_when_touched():
    calls = []

    def thunk():
        calls.append(1)
        return torch.arange(24.0).view(2, 3, 4)
    t = ops.LazyTensor((2, 3, 4), torch.float32, torch.device("cpu"), thunk)
    # metadata is served by the wrapper: nothing is computed
    assert isinstance(t, torch.Tensor) and tuple(t.shape) == (2, 3, 4) and t.dtype == torch.float32 and t.stride() == (12, 4, 1)
    assert t.device.type == "cpu" and t.is_contiguous() and not t.is_materialized and calls == []
    assert "materialized=False" in repr(t)
    # any operation on the values computes them, once
    assert float(t[1, 2, 3]) == 23.0 and calls == [1] and t.is_materialized
    assert torch.equal(t, torch.arange(24.0).view(2, 3, 4)) and torch.equal(t.transpose(1, 2), torch.arange(24.0).view(2, 3, 4).transpose(1, 2))
    assert float((t * 2).sum()) == 552.0 and t.float().cpu().shape == (2, 3, 4) and calls == [1]
    assert ops.realize(t) is t.materialize() and ops.realize(None) is None
    x = torch.ones(3)
    assert ops.realize(x) is x


def test_lazy_tensor_serves_c_level_consumers():
    """a wrapper subclass has no storage -- `data_ptr()` must not hand 0 to the C ABI (an optional pointer would be
    skipped silently), and numpy / pickling / deepcopy must see the real values."""
    import copy
    import io
    import pickle
    from sea_attention_amd import _lib
    from sea_attention_amd.perlin_attention.ops import predictor as PR, flat_csr as FC
    mk = lambda: ops.LazyTensor((2, 3), torch.float32, torch.device("cpu"), lambda: torch.arange(6.0).view(2, 3))
    t = mk()
    assert not t.is_materialized and t.data_ptr() != 0 and t.is_materialized and t.data_ptr() == t.materialize().data_ptr()
    assert PR._p(mk()).value and FC._p(mk()).value
    assert mk().numpy().sum() == 15.0 and mk().tolist() == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]
    assert torch.equal(copy.deepcopy(mk()), torch.arange(6.0).view(2, 3))
    assert torch.equal(pickle.loads(pickle.dumps(mk())), torch.arange(6.0).view(2, 3))
    buf = io.BytesIO()
    torch.save(mk(), buf)
    buf.seek(0)
    assert torch.equal(torch.load(buf), torch.arange(6.0).view(2, 3))
    u = mk()
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # the device check looks at the REAL tensor
        _lib.require_gpu(u)
    assert u.is_materialized


def test_fused_estimator_predicates_cover_the_reference_grid():
    y16 = torch.empty((1, 1, 1, 1, 8), dtype=torch.bfloat16)
    # src/main/benchmark_opt_ablation.py:160-186 (w in 64 / 128 / 256 / 384 at H = 12) and exp_long_context.py:152 (T_M = 96)
    for T_M in (64, 96, 128, 256, 384):
        assert ops.predictor_tail_select_supported(y16, 12, T_M)
        assert ops.predictor_mlp_supported(128, T_M // 2, 12, 192)
    assert ops.predictor_tail_select_supported(y16, 32, 512) and ops.predictor_tail_select_supported(y16, 40, 256)
    assert not ops.predictor_tail_select_supported(y16, 64, 512)                       # H * T_M > 16384
    assert ops.predictor_tail_select_supported(y16.float(), 12, 256)                   # round 5: fp32 data, T_M = 256, H <= 32
    assert not ops.predictor_tail_select_supported(y16.float(), 12, 128)
    assert not ops.predictor_tail_select_supported(y16.float(), 40, 256)
    assert not ops.predictor_tail_select_supported(y16.float(), 12, 256, decode=True)
    assert not ops.predictor_tail_select_supported(y16.double(), 12, 256)
    assert not ops.predictor_tail_select_supported(y16, 12, 128, decode=True)          # the graph-replayed decode form: T_M = 256
    assert ops.predictor_tail_select_supported(y16, 32, 256, decode=True)
    assert not ops.predictor_mlp_supported(128, 24, 12, 192)                            # T_M / 4 must be a multiple of 8
    assert ops.predictor_mlp_supported(256, 128, 40, 384) and not ops.predictor_mlp_supported(256, 96, 40, 384)
    layer = make_layer(H=12, d=64, T_M=96, k=16, max_pos=64)
    body = list(layer.attention.attention_predictor_cnn[1].module.net.children())
    assert layer.attention._c8_cnn_ok(torch.empty(1, dtype=torch.bfloat16), body)       # W = 24: no power-of-two requirement
    assert layer.attention._c8_cnn_ok(torch.empty(1, dtype=torch.float32), body)        # round 5: fp32 data on the fp32-MFMA conv
    assert not layer.attention._c8_cnn_ok(torch.empty(1, dtype=torch.float64), body)
    assert ops.conv_c8_f32_supported(24, 24, 3) and ops.conv_c8_f32_supported(64, 64, 3) and not ops.conv_c8_f32_supported(80, 80, 3)
    # a shape the hand-written kernels decline is declined LOUDLY (once): the layer still runs, on the framework's convolutions
    big = make_layer(H=40, d=128, T_M=256, k=16, max_pos=64)
    body40 = list(big.attention.attention_predictor_cnn[1].module.net.children())
    assert big.attention._c8_cnn_ok(torch.empty(1, dtype=torch.bfloat16), body40)
    with pytest.warns(UserWarning, match="does not fit the 160 KB LDS"):
        assert not big.attention._c8_cnn_ok(torch.empty(1, dtype=torch.float32), body40)


def test_flat_csr_items_share_storage_and_pending_state():
    N, T, H, z = 3, 4, 2, 10
    crow = torch.arange(N * (T + 1), dtype=torch.int32).view(N, T + 1)
    col = torch.zeros((N, z), dtype=torch.int32)
    ho = torch.zeros((N, T, H + 1), dtype=torch.int32)
    bits = torch.zeros((N, T, 1), dtype=torch.int32)
    csr = ops.FlatCSR(crow, col, ho, H, 8, bits=bits, row_nnz=torch.zeros((N, T), dtype=torch.int32))
    fired = []
    csr._pending = (32, 4, True, lambda: fired.append(1))
    sub = csr.items(1, 3)
    assert (sub.N, sub.T_dst, sub.H, sub.T_src) == (2, T, H, 8) and sub.col_is_pending
    assert sub.crow.data_ptr() == crow[1:].data_ptr() and sub._col.data_ptr() == col[1:].data_ptr() and sub.bits.data_ptr() == bits[1:].data_ptr()
    sub._col[0, 0] = 7                                       # a launch over the part writes the shared array
    assert int(col[1, 0]) == 7
    _ = sub.col                                              # a non-fused reader of the part runs the PARENT's emit launch
    assert fired == [1] and not csr.col_is_pending and not sub.col_is_pending
    assert not csr.items(0, 1).col_is_pending


def test_compressed_predictor_parameters_and_unknown_method():
    """`attention_predictor_method='comp'` builds the reference's parameters under the reference's names (attention.py:293-312);
    anything else than 'mlp' / 'comp' raises like the reference's `raise Exception()` (attention.py:663)."""
    from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention

    class Cfg:
        hidden_size, num_attention_heads, max_position_embeddings = 128, 4, 256

    pc = PerlinAttentionConfig(k=8, attention_predictor_length=64, causal=True, attention_predictor_method='comp',
                               attention_predictor_comp_book_size=8, attention_predictor_comp_patch_size=4,
                               attention_predictor_comp_patch_count=8)
    att = PerlinSelfAttention(Cfg(), pc).attention
    sd = att.state_dict()
    assert tuple(sd["attention_predictor_comp_codebook"].shape) == (8, 4)
    assert tuple(sd["attention_predictor_comp_enc.1.weight"].shape) == (64, 96)          # Linear(3 d -> 2 d) behind the Dropout
    assert tuple(sd["attention_predictor_comp_dec_row.0.weight"].shape) == (8 * 8, 64)   # book_size * patch_count
    assert att.attention_predictor_comp_length == 32
    with pytest.raises(Exception):
        PerlinSelfAttention(Cfg(), PerlinAttentionConfig(causal=True, attention_predictor_method='vqvae'))


@pytest.mark.parametrize("kw", [dict(k_flatten=False), dict(k_flatten_dim='batch'), dict(k_flatten_dim='head')])
def test_causal_model_refuses_other_poolings(kw):
    """The reference's per-query / per-batch / per-head poolings assert `not causal` (attention.py:834, 839, 851): the causal module
    refuses them before any kernel runs instead of selecting with the wrong pooling."""
    from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention

    class Cfg:
        hidden_size, num_attention_heads, max_position_embeddings = 64, 2, 64

    pc = PerlinAttentionConfig(k=4, attention_predictor_length=32, causal=True)
    layer = PerlinSelfAttention(Cfg(), pc).eval()
    for name, val in kw.items():
        setattr(layer.attention.pconfig, name, val)
    x = torch.randn((1, 2, 16, 32))
    mask = torch.zeros((1, 1, 16, 16))
    with pytest.raises(AssertionError, match="causal SEA selects per row"):
        layer(None, None, None, query_layer=x, key_layer=x, value_layer=x, attention_mask=mask)


def test_dead_reference_options_raise_instead_of_being_ignored():
    """`attention_predictor_backend != 'performer'` and `random_lookup` are not served: both raise (the reference raises for
    `random_lookup` itself, attention.py:1254-1255; its cosformer backend needs a module this build does not carry)."""
    from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention

    class Cfg:
        hidden_size, num_attention_heads, max_position_embeddings = 64, 2, 64

    with pytest.raises(Exception, match="attention_predictor_backend"):
        PerlinSelfAttention(Cfg(), PerlinAttentionConfig(causal=True, attention_predictor_backend='cosformer'))
