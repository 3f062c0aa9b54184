"""-m gpu: the graph-replayed decode step (perlin_attention/decode.py; SURVEY 8f-3, the generation loop of
src/main/opt_generate.py:131).  `DecodeSession.step` must give, position by position, bitwise the context rows of the
cached forward (`_forward_cached`, itself held to the stateless forward by test_kv_cache.py) -- eagerly launched and as a
replayed HIP graph -- while nothing position-dependent travels in kernel arguments."""
import pytest
import torch

import sea_attention_amd as S
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
from sea_attention_amd.perlin_attention.decode import DecodeSession

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Cfg:
    def __init__(self, hidden, heads, max_pos):
        self.hidden_size, self.num_attention_heads, self.max_position_embeddings = hidden, heads, max_pos


def _mask(N, T_dst, T_src, dtype):
    fp_min = torch.finfo(torch.float16).min / 2
    rows = torch.arange(T_src - T_dst, T_src, device=DEV).view(T_dst, 1)
    m = ((torch.arange(T_src, device=DEV).view(1, T_src) > rows) * fp_min).view(1, 1, T_dst, T_src)
    return m.expand(N, 1, T_dst, T_src).contiguous().to(dtype)


def _layer(H, d, T_M, k, T, dtype):
    S.seed(42)
    pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                               k_flatten_dim='causal_batch', context_output_method='mix', use_cache=True)
    layer = PerlinSelfAttention(Cfg(H * d, H, T), pc).to(DEV).to(dtype).eval()
    for m in layer.modules():
        if hasattr(m, 'benchmarking'):
            m.benchmarking = True
    layer.attention.context_layer_dtype = dtype
    return layer


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("dtype,N,H,d,T0,steps", [(torch.bfloat16, 2, 8, 64, 250, 14),     # crosses T_M = 256: pixel widths 1 -> 2
                                                  (torch.float16, 1, 4, 64, 40, 6),
                                                  (torch.bfloat16, 1, 8, 128, 300, 5),
                                                  (torch.bfloat16, 1, 4, 80, 63, 4),        # crosses a Performer chunk boundary
                                                  # round 5: every instantiation of the fused CNN + tail + selection launch
                                                  (torch.bfloat16, 1, 12, 64, 70, 4),       # 24 channels: 2 tiles, 1 k-chunk
                                                  (torch.float16, 1, 20, 64, 50, 3),        # 40 channels: 3 tiles (odd), 2 k-chunks
                                                  (torch.bfloat16, 2, 32, 64, 130, 3),      # 64 channels: the OPT-1.3B form
                                                  (torch.bfloat16, 1, 40, 64, 90, 3),       # 80 channels: 5 tiles, emit as its own launch
                                                  (torch.bfloat16, 1, 16, 64, 40, 3)])      # 32 channels
def test_session_steps_equal_cached_forward(dtype, N, H, d, T0, steps, use_graph):
    T_M, k = 256, 16
    T = T0 + steps
    layer = _layer(H, d, T_M, k, T + 3, dtype)
    S.seed(9)
    x = torch.randn((N, H, T, d), device=DEV).to(dtype)
    q = (x.float() * d ** -0.5).to(dtype)
    with torch.no_grad():
        out = layer(None, None, None, query_layer=q[:, :, :T0], key_layer=x[:, :, :T0], value_layer=x[:, :, :T0],
                    attention_mask=_mask(N, T0, T0, dtype))
        state = out.state
        sess = DecodeSession(layer.attention, state, x[:, :, :T0], x[:, :, :T0], capacity=T + 3, use_graph=use_graph)
        assert (sess.graph is not None) == use_graph
        for i in range(steps):
            hi = T0 + i + 1
            ref = layer(None, None, None, query_layer=q[:, :, hi - 1:hi], key_layer=x[:, :, :hi], value_layer=x[:, :, :hi],
                        attention_mask=_mask(N, 1, hi, dtype), last_state=state)
            state = ref.state
            got = sess.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi])
            assert got.shape == ref.context_layer.shape == (N, 1, H * d)
            assert torch.isfinite(got.float()).all()
            assert torch.equal(got, ref.context_layer), (i, (got.float() - ref.context_layer.float()).abs().max().item())
            assert torch.equal(sess.probs, ref.estimated_attention_probs_m), i
        assert sess.length == T and int(sess.seen32.item()) == T and int(sess.tsrc32.item()) == T + 1
        # the in-place state is the state the cached forward carries
        from sea_attention_amd.perlin_attention.attention_state import PerlinAttentionState as PS
        assert torch.equal(sess.image, state.states[PS.PERFORMER].image)
        assert sess.fused_cnn                                  # round 5: conv1 + conv2 + tail + selection + counters in one launch
        assert torch.equal(sess.win, state.states[PS.CNN].rows_c8)                                    # the ring, read out by age
        assert torch.equal(sess.export_state().states[PS.CNN].rows_c8, state.states[PS.CNN].rows_c8)
        assert torch.equal(sess.k_cache[:, :, :T], x) and torch.equal(sess.v_cache[:, :, :T], x)


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("dtype,N,H,d,T0", [(torch.bfloat16, 2, 8, 64, 250), (torch.bfloat16, 1, 4, 80, 90), (torch.float16, 1, 8, 128, 300),
                                            (torch.bfloat16, 1, 40, 64, 70)])
def test_session_attention_forms_agree(dtype, N, H, d, T0, use_graph):
    """Round 5: the attention launch of a position expands the kept pixels itself (sea_sparse_attention_fused_at) instead of
    reading the columns an emit phase / launch wrote.  Both forms give the same bits, and the pending handle's columns -- emitted
    on first read -- are the ones the unfused form walked."""
    T_M, k, steps = 256, 16, 5
    T = T0 + steps
    layer = _layer(H, d, T_M, k, T + 3, dtype)
    S.seed(11)
    x = torch.randn((N, H, T, d), device=DEV).to(dtype)
    q = (x.float() * d ** -0.5).to(dtype)
    with torch.no_grad():
        out = layer(None, None, None, query_layer=q[:, :, :T0], key_layer=x[:, :, :T0], value_layer=x[:, :, :T0],
                    attention_mask=_mask(N, T0, T0, dtype))
        a = DecodeSession(layer.attention, out.state, x[:, :, :T0], x[:, :, :T0], capacity=T + 3, use_graph=use_graph)
        b = DecodeSession(layer.attention, out.state, x[:, :, :T0], x[:, :, :T0], capacity=T + 3, use_graph=use_graph, fused_attention=False)
        assert a.fused_attention and not b.fused_attention
        for i in range(steps):
            hi = T0 + i + 1
            ga = a.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi]).clone()
            gb = b.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi]).clone()
            assert torch.equal(ga, gb), (i, (ga.float() - gb.float()).abs().max().item())
            assert a.csr.t_src_dev is not None and a.csr.col_is_pending           # nobody has read the columns: none were written
            assert torch.equal(a.csr.crow, b.csr.crow) and torch.equal(a.csr.head_off, b.csr.head_off)
            if i == steps - 1:
                ca, cb = a.csr.col, b.csr.col                                       # first read: the emit launch runs now
                for n in range(N):
                    nnz = int(a.csr.crow[n, 1].item())
                    assert nnz > 0 and torch.equal(ca[n, :nnz], cb[n, :nnz])


@pytest.mark.parametrize("dtype,H,d,T_src,k,heavy", [(torch.bfloat16, 4, 64, 3000, 64, True),    # one head takes ~3000 entries: 12 chunks of 256
                                                     (torch.float16, 4, 128, 3000, 64, True),    # 16-lane rows: chunks of 128
                                                     (torch.bfloat16, 4, 64, 3000, 4, True),     # every pixel thinned to max_k = 4
                                                     (torch.bfloat16, 8, 64, 517, 16, False),    # ragged step boundary (entries % 4 != 0)
                                                     (torch.bfloat16, 4, 80, 3000, 64, True),    # d = 80: 8 lanes x (8 + 2) elements per row
                                                     (torch.bfloat16, 4, 64, 1, 64, False)])     # the very first position: one key
def test_decode_attention_operator_forms_agree(dtype, H, d, T_src, k, heavy):
    """`sea_sparse_attention_fused_at` (one new row per sequence: sparse_attn_decode1_kernel -- the whole workgroup serves the
    row) against `sea_csr_emit_at` + the unfused launch on one selection: bitwise, for rows far longer than a chunk, thinned
    pixels, ragged ends and empty heads."""
    from sea_attention_amd.perlin_attention import ops
    N, T_m, T_cap = 2, 256, T_src + 40
    S.seed(5)
    probs = torch.rand((N, H, 1, T_m), device=DEV) * 0.1
    if heavy:
        probs[:, 0] += 1.0                                                  # head 0 wins the pooled top-k
        probs[1, 1] = 0.0                                                   # ... and item 1's head 1 keeps nothing
    probs = probs.to(dtype)
    keep = torch.full((1,), (H * T_m) // 3 if heavy else 37, dtype=torch.int32, device=DEV)
    lib_bits = ops.topk_to_csr(probs, keep, k, target_width=T_src, is_causal=True)[0]
    bits, row_nnz, head_off = lib_bits.bits, lib_bits.row_nnz, lib_bits.head_off
    crow = torch.stack([torch.zeros_like(row_nnz[:, 0]), row_nnz[:, 0]], 1).contiguous()
    z_cap = max(int(row_nnz.max().item()), 1)
    ts = torch.tensor([T_src], dtype=torch.int32, device=DEV)
    q = torch.randn((N, H, 1, d), device=DEV).to(dtype)
    kk = torch.randn((N, H, T_cap, d), device=DEV).to(dtype)
    vv = torch.randn((N, H, T_cap, d), device=DEV).to(dtype)
    rs = torch.rand((N, H, 1), device=DEV)
    avg = torch.randn((N, H, 1, d), device=DEV).to(dtype)
    mix = torch.rand((N, H, 1), device=DEV)
    outs = []
    for defer in (True, False):
        csr = ops.csr_from_selection(bits, row_nnz, head_off, H, T_m, T_cap, k, True, z_cap, t_src_dev=ts, crow=crow, defer_emit=defer)
        assert csr.col_is_pending == defer
        o = ops.sparse_attention(q, kk, vv, csr, row_scale=rs, avg=avg, mix=mix, path="gather", keep_columns_pending=True,
                                 out_dtype=dtype if defer else None, out=None)
        outs.append((o, csr))
    (oa, ca), (ob, cb) = outs
    assert ca.col_is_pending                                                 # the decode form wrote no columns
    ob16 = ops.sparse_attention(q, kk, vv, cb, row_scale=rs, avg=avg, mix=mix, path="gather", out_dtype=dtype)
    assert torch.isfinite(oa.float()).all() and torch.equal(oa, ob16), (oa.float() - ob16.float()).abs().max().item()
    o32 = ops.sparse_attention(q, kk, vv, ca, row_scale=rs, avg=avg, mix=mix, path="gather", keep_columns_pending=True)   # fp32 context
    assert torch.equal(o32, ob)
    if heavy:
        assert int((head_off[0, 0, 1] - head_off[0, 0, 0]).item()) > (1000 if k > 4 else 300)
        assert int((head_off[1, 0, 2] - head_off[1, 0, 1]).item()) == 0
        assert torch.equal(oa[1, 1].float(), ((1.0 - mix[1, 1]).view(1, 1) * avg[1, 1].float()).to(dtype).float())     # empty head: the average alone
    for n in range(N):                                                      # the pending handle's columns, emitted on first read
        assert torch.equal(ca.col[n, :int(row_nnz[n, 0])], cb.col[n, :int(row_nnz[n, 0])])
    # ... and written by the launch itself when the caller does not keep them pending (write_columns = 1)
    cw = ops.csr_from_selection(bits, row_nnz, head_off, H, T_m, T_cap, k, True, z_cap, t_src_dev=ts, crow=crow, defer_emit=True)
    ow = ops.sparse_attention(q, kk, vv, cw, row_scale=rs, avg=avg, mix=mix, path="gather", out_dtype=dtype)
    assert not cw.col_is_pending and torch.equal(ow, oa)
    for n in range(N):
        assert torch.equal(cw.col[n, :int(row_nnz[n, 0])], cb.col[n, :int(row_nnz[n, 0])])


@pytest.mark.parametrize("dtype,H,d,T_dst", [(torch.bfloat16, 4, 64, 3), (torch.float16, 4, 128, 8), (torch.bfloat16, 4, 80, 2)])
def test_decode_attention_operator_few_rows(dtype, H, d, T_dst):
    """T_dst = 2 .. 8 new rows per sequence through `sea_sparse_attention_fused_at`: the lane-group decode form (`DEC`
    instantiations: expansion inside the launch, row widths from the device counter, lists warmed by the whole block) against
    `sea_csr_emit_at` + the unfused launch -- bitwise."""
    from sea_attention_amd.perlin_attention import ops
    N, T_m, T_src, k = 2, 256, 1500, 16
    T_cap = T_src + 24
    S.seed(6)
    probs = torch.rand((N, H, T_dst, T_m), device=DEV).to(dtype)
    keep = torch.full((T_dst,), 61, dtype=torch.int32, device=DEV)
    sel = ops.topk_to_csr(probs, keep, k, target_width=T_src, is_causal=True)[0]
    ts = torch.tensor([T_src], dtype=torch.int32, device=DEV)
    z_cap = max(int(sel.crow[:, -1].max().item()), 1)
    q = torch.randn((N, H, T_dst, d), device=DEV).to(dtype)
    kk = torch.randn((N, H, T_cap, d), device=DEV).to(dtype)
    vv = torch.randn((N, H, T_cap, d), device=DEV).to(dtype)
    rs = torch.rand((N, H, T_dst), device=DEV)
    outs = []
    for defer in (True, False):
        csr = ops.csr_from_selection(sel.bits, sel.row_nnz, sel.head_off, H, T_m, T_cap, k, True, z_cap, t_src_dev=ts, defer_emit=defer)
        outs.append((ops.sparse_attention(q, kk, vv, csr, row_scale=rs, path="gather", keep_columns_pending=True), csr))
    (oa, ca), (ob, cb) = outs
    assert ca.col_is_pending and not cb.col_is_pending
    assert torch.isfinite(oa).all() and torch.equal(oa, ob), (oa - ob).abs().max().item()
    for n in range(N):
        z = int(sel.crow[n, -1])
        assert z > 0 and torch.equal(ca.col[n, :z], cb.col[n, :z])


def test_session_refuses_what_it_cannot_continue():
    dtype, N, H, d, T_M, k = torch.bfloat16, 1, 4, 64, 256, 16
    layer = _layer(H, d, T_M, k, 64, dtype)
    x = torch.randn((N, H, 40, d), device=DEV).to(dtype)
    with torch.no_grad():
        out = layer(None, None, None, query_layer=x[:, :, :4], key_layer=x[:, :, :4], value_layer=x[:, :, :4], attention_mask=_mask(N, 4, 4, dtype))
        with pytest.raises(AssertionError, match="reach"):                # prefix shorter than the CNN's lookback
            DecodeSession(layer.attention, out.state, x[:, :, :4], x[:, :, :4], capacity=32, use_graph=False)
        out = layer(None, None, None, query_layer=x, key_layer=x, value_layer=x, attention_mask=_mask(N, 40, 40, dtype))
        with pytest.raises(AssertionError, match="capacity"):
            DecodeSession(layer.attention, out.state, x, x, capacity=40, use_graph=False)
        with pytest.raises(AssertionError, match="exactly the prefix"):
            DecodeSession(layer.attention, out.state, x[:, :, :30], x[:, :, :30], capacity=64, use_graph=False)
        sess = DecodeSession(layer.attention, out.state, x, x, capacity=41, use_graph=False)
        sess.step(x[:, :, :1], x[:, :, :1], x[:, :, :1])
        with pytest.raises(AssertionError, match="capacity reached"):
            sess.step(x[:, :, :1], x[:, :, :1], x[:, :, :1])


def test_session_with_the_deeper_predictor_cnn(monkeypatch):
    """PERLIN_HOTFIX_OPT_DEEPER=1 builds three dilated convolutions (12-row reach): the session's window follows."""
    monkeypatch.setenv("PERLIN_HOTFIX_OPT_DEEPER", "1")
    dtype, N, H, d, T_M, k, T0, steps = torch.bfloat16, 1, 4, 64, 256, 16, 30, 5
    layer = _layer(H, d, T_M, k, T0 + steps + 2, dtype)
    S.seed(3)
    x = torch.randn((N, H, T0 + steps, d), device=DEV).to(dtype)
    q = (x.float() * d ** -0.5).to(dtype)
    with torch.no_grad():
        out = layer(None, None, None, query_layer=q[:, :, :T0], key_layer=x[:, :, :T0], value_layer=x[:, :, :T0],
                    attention_mask=_mask(N, T0, T0, dtype))
        state = out.state
        sess = layer.attention.decode_session(state, x[:, :, :T0], x[:, :, :T0], capacity=T0 + steps + 1)
        assert sess.win.shape[1] == 12 and not sess.fused_cnn       # three convolutions: the round-4 launches
        for i in range(steps):
            hi = T0 + i + 1
            ref = layer(None, None, None, query_layer=q[:, :, hi - 1:hi], key_layer=x[:, :, :hi], value_layer=x[:, :, :hi],
                        attention_mask=_mask(N, 1, hi, dtype), last_state=state)
            state = ref.state
            got = sess.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi])
            assert torch.equal(got, ref.context_layer), i


def test_session_long_run_stays_bitwise():
    """300 replayed positions from a 20-token prefix: past two multiples of T_M (the pixel widths change twice), five
    Performer chunks and the point where K_t stops being clamped -- every position equals the cached forward."""
    dtype, N, H, d, T_M, k, T0, steps = torch.bfloat16, 1, 4, 64, 256, 8, 20, 300
    layer = _layer(H, d, T_M, k, T0 + steps + 1, dtype)
    S.seed(17)
    x = torch.randn((N, H, T0 + steps, d), device=DEV).to(dtype)
    q = (x.float() * d ** -0.5).to(dtype)
    with torch.no_grad():
        out = layer(None, None, None, query_layer=q[:, :, :T0], key_layer=x[:, :, :T0], value_layer=x[:, :, :T0],
                    attention_mask=_mask(N, T0, T0, dtype))
        state = out.state
        sess = layer.attention.decode_session(state, x[:, :, :T0], x[:, :, :T0], capacity=T0 + steps)
        bad = []
        for i in range(steps):
            hi = T0 + i + 1
            ref = layer(None, None, None, query_layer=q[:, :, hi - 1:hi], key_layer=x[:, :, :hi], value_layer=x[:, :, :hi],
                        attention_mask=_mask(N, 1, hi, dtype), last_state=state)
            state = ref.state
            got = sess.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi])
            if not torch.equal(got, ref.context_layer):
                bad.append(hi)
        assert not bad, bad[:10]
        assert sess.length == T0 + steps


def test_session_survives_a_cleared_weight_cache_and_follows_edited_weights():
    """the captured graph holds raw pointers into the re-laid-out predictor weights of the process-wide prep cache.
    (1) the session pins the packs its launches used, so `clear_prep_cache()` (another layer's `.to()` / `load_state_dict`)
    followed by allocations that would recycle their memory cannot corrupt a replay; (2) a cleared cache means the weights
    may have been edited behind autograd's back: the session re-captures and the next position uses the NEW weights, exactly
    like the eager cached forward."""
    from sea_attention_amd.perlin_attention import ops
    dtype, N, H, d, T_M, k, T0, steps = torch.bfloat16, 1, 4, 64, 256, 16, 40, 6
    layer = _layer(H, d, T_M, k, T0 + steps + 1, dtype)
    S.seed(23)
    x = torch.randn((N, H, T0 + steps, d), device=DEV).to(dtype)
    q = (x.float() * d ** -0.5).to(dtype)
    with torch.no_grad():
        out = layer(None, None, None, query_layer=q[:, :, :T0], key_layer=x[:, :, :T0], value_layer=x[:, :, :T0],
                    attention_mask=_mask(N, T0, T0, dtype))
        state = out.state
        sess = layer.attention.decode_session(state, x[:, :, :T0], x[:, :, :T0], capacity=T0 + steps)
        assert sess.graph is not None and sess._pinned, "the capture pinned the packs it points into"
        gen0 = sess._prep_generation
        for i in range(steps):
            hi = T0 + i + 1
            if i == 2:
                ops.clear_prep_cache()                                      # e.g. model.to(...) on another layer
                junk = [torch.randn(1 << 16, device=DEV) for _ in range(64)]   # would land in the freed packs' memory
            if i == 4:
                # an edit behind autograd's back (no version bump) + the documented hook
                layer.attention.attention_predictor_dec_row[0].weight.data.mul_(1.5)
                ops.clear_prep_cache()
            ref = layer(None, None, None, query_layer=q[:, :, hi - 1:hi], key_layer=x[:, :, :hi], value_layer=x[:, :, :hi],
                        attention_mask=_mask(N, 1, hi, dtype), last_state=state)
            state = ref.state
            got = sess.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi])
            assert torch.equal(got, ref.context_layer), i
            assert torch.equal(sess.probs, ref.estimated_attention_probs_m), i
        assert sess._prep_generation == ops.prep_generation() and sess._prep_generation >= gen0 + 2
        del junk
