"""SURVEY §8a row P / BASELINE config 0: the reference's command line reaches the same SEA layer.

CPU part: flag names -> trainer keywords -> PerlinAttentionConfig (src/trainer/perlin_trainer.py:41-155) and the
construction of an OPT-125m-shaped attention block (src/models/perlin_opt/perlin_opt.py:175-239).  GPU part: one
forward of that block over 2048 tokens, and decoding through its `(k, v, state)` cache tuple.
"""
import argparse
import copy
import warnings

import pytest
import torch

from sea_attention_amd import opt_plumbing as P
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, get_default_config, register_default_config


@pytest.fixture
def restore_default():
    old = get_default_config()
    yield
    register_default_config(old)


def _parse(argv, **defaults):
    return P.add_perlin_model_options(argparse.ArgumentParser(), **defaults).parse_args(argv)


def test_flag_defaults_are_the_reference_defaults():
    a = _parse([])
    assert (a.method, a.k, a.k_flatten_dim, a.attention_predictor_method) == ("perlin", 7, "batch", "mlp")
    assert (a.performer_nb_feature_factor, a.predictor_length, a.predictor_backend) == (1.0, 128, "performer")
    assert (a.n_hashs, a.context_output_method, a.k_oversample) == (8, "norm", 1)
    assert not (a.layerwise or a.enable_lora or a.k_colwise or a.random_lookup or a.token_merging or a.enc_per_layer)
    assert (a.random_lookup_count, a.token_merging_preserve, a.token_merging_ratio) == (3, 0.2, 0.5)
    # the OPT entry point overrides the defaults through the function's arguments
    b = _parse([], context_output_method="mix", predictor_length=256, k=64, nbf=8, epl=True)
    assert (b.context_output_method, b.predictor_length, b.k, b.performer_nb_feature_factor, b.enc_per_layer) == \
        ("mix", 256, 64, 8, True)


def test_baseline_command_line_to_config(restore_default):
    argv = "--k 64 --predictor-length 256 --performer-nb-feature-factor 8 --context-output-method mix".split()
    kw = P.parse_perlin_model_options(_parse(argv))
    assert set(kw) == {
        'perlin_k', 'attention_method', 'perlin_k_flatten', 'perlin_k_flatten_dim', 'perlin_layerwise', 'perlin_lora',
        'perlin_attention_predictor_method', 'perlin_performer_nb_feature_factor', 'perlin_random_lookup',
        'perlin_random_lookup_count', 'perlin_token_merging', 'perlin_token_merging_preserve',
        'perlin_token_merging_ratio', 'perlin_predictor_length', 'perlin_predictor_backend', 'perlin_n_hashs',
        'perlin_enc_per_layer', 'perlin_context_output_method', 'perlin_k_oversample'}
    cfg = P.perlin_config_from_options(**kw)
    assert get_default_config() is cfg
    assert (cfg.k, cfg.attention_predictor_length, cfg.performer_nb_factor, cfg.context_output_method) == (64, 256, 8.0, "mix")
    assert cfg.k_flatten and cfg.k_flatten_dim == "batch" and not cfg.causal and cfg.k_oversample == 1
    # the causal attention block then forces the causal pooling (perlin_opt.py:224-228)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        blk = P.SeaOPTAttention(768, 12)
    assert any("causal_batch" in str(x.message) for x in w)
    assert blk.pconfig is cfg and cfg.causal and cfg.k_flatten_dim == "causal_batch"
    assert P.parse_perlin_model_options(_parse(["--k-colwise"]))["perlin_k_flatten"] is False
    with pytest.raises(NotImplementedError):
        P.perlin_config_from_options(attention_method="reformer", register=False)


def test_opt125m_block_parameters(restore_default):
    P.perlin_config_from_options(perlin_k=64, perlin_predictor_length=256, perlin_performer_nb_feature_factor=8,
                                 perlin_k_flatten_dim="causal_batch")
    blk = P.SeaOPTAttention(768, 12, max_position_embeddings=2048)
    sd = blk.state_dict()
    for name in ("q_proj.weight", "k_proj.bias", "v_proj.weight", "out_proj.bias",
                 "perlin_self_attention.attention.performer.projection_matrix",
                 "perlin_self_attention.attention.attention_predictor_enc.0.weight",
                 "perlin_self_attention.attention.attention_predictor_cnn.1.module.net.0.module.weight",
                 "perlin_self_attention.query_lora.lora_a"):
        assert name in sd, name
    att = blk.perlin_self_attention.attention
    assert sd["q_proj.weight"].shape == (768, 768) and blk.head_dim == 64 and abs(blk.scaling - 0.125) < 1e-12
    assert att.performer_nb_features == 33                                    # int(64 ln 64 / 8)
    assert "perlin_out_lora.lora_a" not in sd
    blk.benchmarking = True
    assert att.benchmarking is True
    # with LoRA the block grows the output adapter the reference checkpoints carry
    P.perlin_config_from_options(perlin_lora=True, perlin_k_flatten_dim="causal_batch")
    assert "perlin_out_lora.lora_b" in P.SeaOPTAttention(64, 4).state_dict()


def test_causal_additive_mask_with_prefix():
    m = P.causal_additive_mask(2, 3, 5, torch.float32, "cpu")
    assert m.shape == (2, 1, 3, 5)
    assert ((m[0, 0] > -1) == torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 0], [1, 1, 1, 1, 1]], dtype=torch.bool)).all()


def test_main_without_gpu_stops_before_the_forward(restore_default):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(SystemExit, match="no CPU path"):
        P.main(["--k", "64", "--predictor-length", "256", "--performer-nb-feature-factor", "8"])


# ------------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_opt125m_block_forward_2048_tokens(restore_default, dtype):
    """BASELINE config 0's layer: OPT-125m shape, k=64, T_M=256, nbf=8, one 2048-token batch.  The block's output must
    be out_proj of the SEA layer called directly with the projected, head-split tensors (the boundary of §8b)."""
    dev = "cuda"
    P.perlin_config_from_options(perlin_k=64, perlin_predictor_length=256, perlin_performer_nb_feature_factor=8,
                                 perlin_context_output_method="mix")
    torch.manual_seed(42)
    blk = P.SeaOPTAttention(768, 12, max_position_embeddings=2048).to(dev, dtype).eval()
    blk.benchmarking = True
    N, T = 2, 2048
    x = torch.randn(N, T, 768, device=dev, dtype=dtype)
    mask = P.causal_additive_mask(N, T, T, dtype, dev)
    with torch.no_grad():
        y, probs, present = blk(x, attention_mask=mask)
        q = blk._heads(blk.q_proj(x) * blk.scaling)
        k, v = blk._heads(blk.k_proj(x)), blk._heads(blk.v_proj(x))
        ref = blk.perlin_self_attention(blk.q_proj, blk.k_proj, blk.v_proj, None, q, k, v, mask, None, None)
        y_ref = blk.out_proj(ref.context_layer.to(dtype))
    assert y.shape == (N, T, 768) and y.dtype == dtype and probs is None and torch.isfinite(y).all()
    assert present[0].shape == (N, 12, T, 64) and torch.equal(present[1], v)
    assert torch.equal(y, y_ref)
    # causality canary (test_perlin_opt_causality.py:246-276): changing the last 100 tokens leaves the earlier rows alone
    x2 = x.clone()
    x2[:, -100:] = torch.randn_like(x2[:, -100:])
    with torch.no_grad():
        y2, _, _ = blk(x2, attention_mask=mask)
    assert torch.equal(y[:, :-100], y2[:, :-100])
    assert not torch.equal(y[:, -100:], y2[:, -100:])


@pytest.mark.gpu
def test_block_decodes_through_its_cache_tuple(restore_default):
    """`past_key_value = (k, v, state)` as the OPT decoder threads it (perlin_opt.py:519-525,584-585): prefill plus
    token-by-token decoding reproduces the rows of one full forward (fp32, the protocol of test_perlin_opt_cache.py)."""
    dev, dtype = "cuda", torch.float32
    cfg = P.perlin_config_from_options(perlin_k=8, perlin_predictor_length=32, perlin_performer_nb_feature_factor=8,
                                       perlin_context_output_method="mix")
    torch.manual_seed(3)
    blk = P.SeaOPTAttention(64, 4, max_position_embeddings=128).to(dev, dtype).eval()
    blk.benchmarking = True
    N, T, T0 = 2, 96, 80
    x = torch.randn(N, T, 64, device=dev, dtype=dtype)
    with torch.no_grad():
        full, _, kv = blk(x, attention_mask=P.causal_additive_mask(N, T, T, dtype, dev))
    assert len(kv) == 2                                              # no cache requested -> no state handed back
    cfg.use_cache = True
    with torch.no_grad():
        y0, _, past = blk(x[:, :T0], attention_mask=P.causal_additive_mask(N, T0, T0, dtype, dev))
        assert len(past) == 3 and past[0].shape[2] == T0
        rows = [y0]
        for t in range(T0, T):
            yt, _, past = blk(x[:, t:t + 1], past_key_value=past, attention_mask=P.causal_additive_mask(N, 1, t + 1, dtype, dev))
            rows.append(yt)
    got = torch.cat(rows, dim=1)
    assert past[0].shape[2] == T and past[2].seq_len == T
    assert (got - full).abs().max().item() < 2e-4 * max(1.0, full.abs().max().item())


@pytest.mark.gpu
def test_block_generates_through_the_graph_replayed_session(restore_default):
    """`decode_graph_capacity` (opt-in): the one-token calls of a generation loop run as a replayed HIP graph and return the
    rows -- and a cache tuple of the same shapes -- the regular cached forward returns, bit for bit; a several-token call in
    the middle falls back (the ticket is turned into a real state) and the next single token opens a new session."""
    from sea_attention_amd.perlin_attention.decode import SessionState
    dev, dtype = "cuda", torch.bfloat16
    cfg = P.perlin_config_from_options(perlin_k=16, perlin_predictor_length=256, perlin_performer_nb_feature_factor=8,
                                       perlin_context_output_method="mix")
    cfg.use_cache = True
    torch.manual_seed(5)
    blk = P.SeaOPTAttention(4 * 64, 4, max_position_embeddings=160).to(dev, dtype).eval()
    blk.benchmarking = True
    blk.perlin_self_attention.attention.context_layer_dtype = dtype
    N, T0 = 2, 100
    plan = [1, 1, 1, 3, 1, 1]                                        # tokens per call after the prefill
    T = T0 + sum(plan)
    x = torch.randn(N, T, 4 * 64, device=dev, dtype=dtype)

    def run(capacity):
        blk.decode_graph_capacity, blk._decode_session = capacity, None
        rows, kinds = [], []
        with torch.no_grad():
            y, _, past = blk(x[:, :T0], attention_mask=P.causal_additive_mask(N, T0, T0, dtype, dev))
            rows.append(y)
            pos = T0
            for n_new in plan:
                y, _, past = blk(x[:, pos:pos + n_new], past_key_value=past,
                                 attention_mask=P.causal_additive_mask(N, n_new, pos + n_new, dtype, dev))
                pos += n_new
                rows.append(y.clone())
                kinds.append(isinstance(past[2], SessionState))
                assert past[0].shape == (N, 4, pos, 64) and past[1].shape == (N, 4, pos, 64) and past[2].seq_len == pos
        return torch.cat(rows, dim=1), kinds, past

    ref, kinds_ref, _ = run(None)
    got, kinds, past = run(T + 4)
    assert kinds_ref == [False] * len(plan)
    assert kinds == [True, True, True, False, True, True]             # the 3-token call falls back, the next token re-opens
    assert torch.equal(got, ref)
    with torch.no_grad():                                            # a stale ticket (the session has moved on) is refused
        stale = past
        _, _, past = blk(x[:, :1], past_key_value=past, attention_mask=P.causal_additive_mask(N, 1, T + 1, dtype, dev))
        with pytest.raises(AssertionError, match="stale"):
            blk(x[:, :1], past_key_value=stale, attention_mask=P.causal_additive_mask(N, 1, T + 1, dtype, dev))


@pytest.mark.gpu
@pytest.mark.parametrize("cached", [False, True])
def test_output_attentions_returns_the_sparse_probabilities(restore_default, cached):
    """in sparse mode `output_attentions=True` must hand back the reference's `partial_attention_probs`
    (attention.py:1162-1171: rs * softmax on the mask's CSR) -- not None -- in the stateless and in the cached path, for that
    call only (the per-entry store stays off for calls that do not ask); values equal to the probing path's."""
    import sea_attention_amd as S
    dev, dtype = "cuda", torch.bfloat16
    cfg = P.perlin_config_from_options(perlin_k=16, perlin_predictor_length=256, perlin_performer_nb_feature_factor=8,
                                       perlin_context_output_method="mix")
    cfg.use_cache = cached
    torch.manual_seed(11)
    blk = P.SeaOPTAttention(4 * 64, 4, max_position_embeddings=256).to(dev, dtype).eval()
    blk.benchmarking = True
    att = blk.perlin_self_attention.attention
    N, T = 2, 192
    x = torch.randn(N, T, 4 * 64, device=dev, dtype=dtype)
    mask = P.causal_additive_mask(N, T, T, dtype, dev)
    with torch.no_grad():
        y0, p0, _ = blk(x, attention_mask=mask)
        assert p0 is None and att.return_attention_probs is False
        y1, p1, _ = blk(x, attention_mask=mask, output_attentions=True)
        assert att.return_attention_probs is False                         # restored after the call
    assert p1 is not None and p1.vals is not None
    assert torch.equal(y0, y1) or (y0.float() - y1.float()).abs().max().item() < 2e-2   # auto may run the tile kernel without probs
    # the same values through the operator with want_probs (what probing registers as partial_attention_probs)
    from sea_attention_amd.perlin_attention import ops
    with torch.no_grad():
        q = blk._heads(blk.q_proj(x) * blk.scaling)
        k, v = blk._heads(blk.k_proj(x)), blk._heads(blk.v_proj(x))
        out = blk.perlin_self_attention(blk.q_proj, blk.k_proj, blk.v_proj, None, q, k, v, mask, None, None)
        assert out.partial_attention_probs is None
        csr = p1
        z = int(csr.crow[0, -1].item())
        # rows of the probabilities sum to the row scale per (row, head): softmax mass 1 x sigmoid gate in (0, 1)
        dense = ops.flat_csr_to_dense(csr, T, 4)                              # (N, H, T, T)
        s = dense.sum(-1)
        assert z > 0 and torch.isfinite(dense).all() and (s > 0).all() and (s < 1.0 + 1e-3).all()
        # and they are the values of the fused kernel's own probs output on the same CSR
        rs = None
        S.get_bench().activate_temp_buffers = True
        try:
            S.get_bench().reset_temp_buffers()
            att.benchmarking = True
            if not cached:
                blk.perlin_self_attention(blk.q_proj, blk.k_proj, blk.v_proj, None, q, k, v, mask, None, None)
                probe = S.get_bench().get_temp_buffer('partial_attention_probs')
                assert (probe.float() - dense.float()).abs().max().item() < 1e-6
        finally:
            S.get_bench().activate_temp_buffers = False
            S.get_bench().reset_temp_buffers()
