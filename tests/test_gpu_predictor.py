"""-m gpu: the estimator/epilogue HIP kernels (csrc/sea_predictor.hip) against plain fp32 torch references
of the same ops (oracle.split_layernorm / predictor_tail / cumavg).  Floating point: fp32 inputs within 1e-5
relative; 16-bit inputs are compared against the fp32 reference evaluated on the same rounded inputs, so
the only differences are the final rounding of the outputs (bf16: 2^-8 relative)."""
import pytest
import torch

from oracle import sea_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from sea_attention_amd.perlin_attention import ops
    return ops


def _tol(dtype):
    return {torch.float32: (1e-5, 1e-5), torch.bfloat16: (1e-2, 1e-2), torch.float16: (2e-3, 2e-3)}[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,T,S,W", [(2, 12, 128, 2, 16), (1, 32, 512, 2, 64), (1, 5, 70, 2, 32), (1, 4, 33, 4, 24),
                                       (1, 8, 64, 2, 128), (1, 12, 100, 1, 128), (1, 3, 50, 1, 256)])
def test_split_layernorm(ops, dtype, N, C, T, S, W):
    if dtype != torch.float32 and W % 8:
        pytest.skip("16-bit rows need W % 8 == 0")
    g = torch.Generator().manual_seed(0)
    x = (torch.randn((N, C, T, S * W), generator=g) * 2 + 0.3).to(dtype)
    w = (torch.rand(W, generator=g) + 0.5).to(dtype)
    b = torch.randn(W, generator=g).to(dtype)
    ref = O.split_layernorm(x.float(), S, w.float(), b.float(), 1e-5)
    out = ops.split_layernorm(x.to(DEV), S, w.to(DEV), b.to(DEV), 1e-5)
    assert out.dtype == dtype and tuple(out.shape) == (N, C * S, T, W)
    atol, rtol = _tol(dtype)
    torch.testing.assert_close(out.float().cpu(), ref, atol=atol * 4, rtol=rtol)
    if S == 1 or True:
        out_g = ops.split_layernorm(x.to(DEV), S, w.to(DEV), b.to(DEV), 1e-5, gelu=True)
        torch.testing.assert_close(out_g.float().cpu(), torch.nn.functional.gelu(ref), atol=atol * 4, rtol=rtol)


@pytest.mark.parametrize("layout", ["nchw", "nhwc", "c8"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T,T_M", [(2, 12, 128, 64), (1, 32, 300, 256), (1, 40, 64, 256), (1, 4, 50, 128),
                                       (1, 6, 40, 512), (1, 3, 20, 32)])
def test_predictor_tail(ops, dtype, N, H, T, T_M, layout):
    C, W4 = 2 * H, T_M // 4
    if dtype != torch.float32 and W4 % 8:
        pytest.skip("16-bit rows need W4 % 8 == 0")
    g = torch.Generator().manual_seed(1)
    y = torch.relu(torch.randn((N, C, T, W4), generator=g)).to(dtype)
    cw = (torch.randn((H, C, 1, 1), generator=g) * C ** -0.5).to(dtype)
    cb = (torch.randn(H, generator=g) * 0.1).to(dtype)
    lw = (torch.rand(T_M, generator=g) + 0.5).to(dtype)
    lb = (torch.randn(T_M, generator=g) * 0.1).to(dtype)
    p_ref, s_ref = O.predictor_tail(y.float(), cw.float(), cb.float(), lw.float(), lb.float(), 4, T_M)
    yd = y.to(DEV)
    if layout == "nhwc":                     # 16-bit channels-last input takes the MFMA variant of the kernel
        if C % (4 if dtype == torch.float32 else 8):
            pytest.skip("channels-last rows need C % vec == 0")
        yd = yd.contiguous(memory_format=torch.channels_last)
    if layout == "c8":                       # the conv kernels' channel-blocked layout (MFMA variant for 16-bit data)
        if C % 8:
            pytest.skip("C8 needs C % 8 == 0")
        yd = ops.to_c8(yd)
    probs, scores = ops.predictor_tail(yd, cw[:, :, 0, 0].to(DEV), cb.to(DEV), lw.to(DEV), lb.to(DEV),
                                       up=4, T_m=T_M, want_scores=True)
    assert probs.dtype == dtype and tuple(probs.shape) == (N, H, T, T_M)
    if dtype == torch.float32:
        torch.testing.assert_close(scores.cpu(), s_ref, atol=2e-5, rtol=1e-5)
        torch.testing.assert_close(probs.cpu(), p_ref, atol=1e-7, rtol=1e-4)
    elif dtype == torch.bfloat16:
        torch.testing.assert_close(scores.float().cpu(), s_ref, atol=3e-2, rtol=1e-2)
        torch.testing.assert_close(probs.float().cpu(), p_ref, atol=1e-5, rtol=4e-2)
    else:
        torch.testing.assert_close(scores.float().cpu(), s_ref, atol=4e-3, rtol=2e-3)
        torch.testing.assert_close(probs.float().cpu(), p_ref, atol=1e-6, rtol=6e-3)
    assert torch.allclose(probs.float().sum(-1), torch.ones((), device=DEV), atol=2e-2 if dtype != torch.float32 else 1e-5)
    p2, s2 = ops.predictor_tail(yd, cw[:, :, 0, 0].to(DEV), cb.to(DEV), lw.to(DEV), lb.to(DEV), up=4, T_m=T_M)
    assert s2 is None and torch.equal(p2, probs)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T,D", [(2, 3, 100, 64), (1, 4, 4096, 64), (1, 2, 33, 80), (2, 3, 1500, 80), (1, 2, 257, 128), (1, 1, 7, 16), (1, 2, 50, 32)])
def test_cumavg(ops, dtype, N, H, T, D):
    g = torch.Generator().manual_seed(2)
    v = torch.randn((N, H, T, D), generator=g).to(dtype)
    ref = v.float().cumsum(-2) / torch.arange(1, T + 1).view(1, 1, -1, 1)
    out = ops.cumavg(v.to(DEV))
    assert out.dtype == dtype
    atol, rtol = _tol(dtype)
    torch.testing.assert_close(out.float().cpu(), ref, atol=atol, rtol=rtol)
    # strided input view (N,T,H,D) -> (N,H,T,D)
    vt = v.permute(0, 2, 1, 3).contiguous().to(DEV).permute(0, 2, 1, 3)
    assert torch.equal(ops.cumavg(vt), out)


def test_module_hip_estimator_matches_torch_estimator():
    """Same layer, same inputs: estimator through the HIP kernels vs through the torch modules."""
    import sea_attention_amd as S
    from test_gpu_module import make_layer, run, causal_mask
    N, H, T, d, T_M, k = 1, 12, 512, 64, 256, 64
    layer = make_layer(H, d, T_M, k, T)
    S.seed(9)
    q = torch.randn((N, H, T, d), device=DEV)
    mask = causal_mask(N, T, torch.float32)
    _, b_hip = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    layer.attention.force_torch_estimator = True
    try:
        _, b_torch = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    finally:
        layer.attention.force_torch_estimator = False
    for name, atol in [('t_attention_predictor', 1e-5), ('estimated_attention_score_dec_row', 2e-5), ('estimated_attention_score', 2e-4),
                       ('estimated_attention_probs', 1e-6), ('average_context_layer', 1e-5)]:
        err = (b_hip[name].float() - b_torch[name].float()).abs().max().item()
        assert err <= atol, (name, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T,D,nbf", [(1, 2, 256, 64, 8), (2, 3, 200, 64, 8), (1, 2, 130, 80, 8), (1, 2, 96, 128, 8),
                                         (1, 1, 64, 64, 4), (1, 4, 1024, 64, 8), (1, 2, 333, 64, 8), (2, 2, 1000, 128, 8),
                                         (1, 3, 161, 128, 8), (1, 1, 31, 128, 8), (2, 3, 1000, 80, 8), (1, 2, 95, 80, 6),
                                         (1, 2, 300, 64, 4)])      # 66 features at d = 64: one phi image set, several chunks
def test_performer_value(ops, dtype, N, H, T, D, nbf):
    """Fused Performer kernel vs the torch restatement (perlin_attention/performer.py) evaluated in fp32, and vs
    the naive prefix-sum formula of the published algorithm."""
    import math
    from sea_attention_amd.perlin_attention.performer import FastAttention
    torch.manual_seed(5)
    nb = int(D * math.log(D) / nbf)
    fa = FastAttention(D, nb_features=nb, causal=True, generalized_attention=True)
    q = (torch.randn(N, H, T, D) * D ** -0.5).to(dtype)
    k = torch.randn(N, H, T, D).to(dtype)
    v = torch.randn(N, H, T, D).to(dtype)
    pos = torch.randn(T + 7, D).to(dtype)
    W = fa.projection_matrix.to(dtype).float()
    vaug = torch.cat([pos[:T].float().expand(N, H, T, D), v.float()], -1)
    qp = torch.relu(D ** -0.25 * q.float() @ W.t()) + 1e-3
    kp = torch.relu(D ** -0.25 * k.float() @ W.t()) + 1e-3
    ksum = kp.double().cumsum(-2) + 1e-6
    ctx_cum = torch.einsum('...nd,...ne->...nde', kp.double(), vaug.double()).cumsum(-3)
    ref = torch.einsum('...nde,...nd,...n->...ne', ctx_cum, qp.double(), 1.0 / torch.einsum('...nd,...nd->...n', qp.double(), ksum)).float()
    out = ops.performer_value(q.to(DEV), k.to(DEV), v.to(DEV), pos.to(DEV), fa.projection_matrix.to(DEV))
    assert out.dtype == dtype and tuple(out.shape) == (N, H, T, 3 * D)
    assert torch.equal(out[..., 2 * D:].cpu(), v)                      # the concatenated copy of v is exact
    ctx = out[..., :2 * D].float().cpu()
    if dtype == torch.float32:
        torch.testing.assert_close(ctx, ref, atol=2e-4, rtol=2e-4)
        mine = fa(q, k, vaug)                                           # the package's torch path (chunked GEMMs)
        torch.testing.assert_close(ctx, mine, atol=2e-4, rtol=2e-4)
    elif dtype == torch.float16:
        torch.testing.assert_close(ctx, ref, atol=4e-3, rtol=4e-3)
        refh = ref.to(dtype).float()                      # split-fp16 MFMA kernels, 2 x 11 significand bits
        if D in (64, 80, 128):
            assert ((ctx - refh).abs() <= refh.abs() * 2.0 ** -10 + 2.0 ** -14).all()
    else:
        torch.testing.assert_close(ctx, ref, atol=3e-2, rtol=2e-2)
        # bf16 data runs the split-bf16 MFMA kernels (D = 64, 80, 128): their result is the fp32 formula to ~2^-16 of the row's
        # magnitude (outputs are averages of O(1) values; an element that cancels to ~0 keeps that ABSOLUTE error),
        # i.e. after the final rounding almost every element equals bf16(ref) and none is further than one bf16
        # step plus 2^-13
        refb = ref.to(dtype).float()
        tol = refb.abs() * 2.0 ** -7 + 2.0 ** -13
        assert ((ctx - refb).abs() <= tol).all(), ((ctx - refb).abs() - tol).max()
        assert (ctx != refb).float().mean().item() < 0.02


def _causal_conv_ref(x, weight, bias, k, dil, pad_w, relu):
    """fp32 reference: CausalConv2d semantics (modules.py:96-192): live rows :k of the (2k-1) x k kernel, top padding."""
    F = torch.nn.functional
    y = F.conv2d(F.pad(x.float(), (0, 0, (k - 1) * dil, 0)), weight[:, :, :k, :].float(), bias.float(), 1, (0, pad_w), dil)
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("dil", [2, 1, 3])
@pytest.mark.parametrize("N,Cin,Cout,T,W", [(2, 64, 64, 40, 64), (1, 24, 24, 33, 16), (1, 80, 80, 20, 64), (1, 64, 64, 9, 32),
                                            (1, 32, 48, 17, 128), (1, 16, 32, 12, 40), (1, 24, 24, 11, 24), (1, 24, 24, 9, 96)])
def test_causal_conv_c8(ops, dtype, N, Cin, Cout, T, W, dil):
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, Cin, T, W), generator=g).to(dtype)
    wt = (torch.randn((Cout, Cin, 5, 3), generator=g) * (Cin * 9) ** -0.5).to(dtype)
    b = (torch.randn(Cout, generator=g) * 0.1).to(dtype)
    for relu in (True, False):
        ref = _causal_conv_ref(x, wt, b, 3, dil, dil, relu)
        xd = ops.to_c8(x.to(DEV))
        y = ops.causal_conv_c8(xd, wt.to(DEV), b.to(DEV), 3, dil, dil, relu=relu)
        assert tuple(y.shape) == (N, T, Cout // 8, W, 8) and y.is_contiguous()
        torch.testing.assert_close(ops.from_c8(y).float().cpu(), ref, atol=2e-2, rtol=2e-2)
    # a one-hot image: every output pixel is a single weight (+ bias), exact in 16 bits -> tap positions checked bitwise
    xo = torch.zeros_like(x); xo[0, Cin // 2, T // 2, W // 3] = 1.0
    ref = _causal_conv_ref(xo, wt, b.float().to(dtype), 3, dil, dil, False).to(dtype)
    y = ops.from_c8(ops.causal_conv_c8(ops.to_c8(xo.to(DEV)), wt.to(DEV), b.to(DEV), 3, dil, dil, relu=False)).cpu()
    assert torch.equal(y, ref)
    xo = torch.zeros_like(x); xo[0, 0, T - 1, 0] = 1.0; xo[0, Cin - 1, 0, W - 1] = -2.0           # the row ends
    ref = _causal_conv_ref(xo, wt, b.float().to(dtype), 3, dil, dil, False).to(dtype)
    y = ops.from_c8(ops.causal_conv_c8(ops.to_c8(xo.to(DEV)), wt.to(DEV), b.to(DEV), 3, dil, dil, relu=False)).cpu()
    assert torch.equal(y, ref)
    # causality along T: future rows do not leak
    x2 = x.clone(); x2[:, :, T // 2:] += 50
    y1 = ops.causal_conv_c8(ops.to_c8(x.to(DEV)), wt.to(DEV), b.to(DEV), 3, dil, dil)
    y2 = ops.causal_conv_c8(ops.to_c8(x2.to(DEV)), wt.to(DEV), b.to(DEV), 3, dil, dil)
    assert torch.equal(y1[:, :T // 2], y2[:, :T // 2])
    # 1x1 kernel (no taps to shift): plain channel GEMM per pixel
    w1 = (torch.randn((Cout, Cin, 1, 1), generator=g) * Cin ** -0.5).to(dtype)
    y = ops.causal_conv_c8(ops.to_c8(x.to(DEV)), w1.to(DEV), b.to(DEV), 1, 1, 0, relu=False)
    ref1 = torch.nn.functional.conv2d(x.float(), w1.float(), b.float())
    torch.testing.assert_close(ops.from_c8(y).float().cpu(), ref1, atol=2e-2, rtol=2e-2)


def test_causal_conv_c8_launch_geometries_agree(ops):
    """Round 5: the convolution picks 8-wave workgroups for narrow layers / launches of few rows and 6-wave ones for big
    64 -> 64-channel launches (and one workgroup per CU for weight images beyond 80 KB).  Same rows, bit for bit, whichever
    geometry a launch gets: a 5-sequence batch (20480 rows: the 6-wave form) against its sequences run one by one (8-wave)."""
    g = torch.Generator().manual_seed(5)
    N, C, T, W = 5, 64, 4096, 64
    x = ops.to_c8(torch.randn((N, C, T, W), generator=g).to(torch.bfloat16).to(DEV))
    wt = (torch.randn((C, C, 5, 3), generator=g) * (C * 9) ** -0.5).to(torch.bfloat16).to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    y = ops.causal_conv_c8(x, wt, b, 3, 2, 2)
    for n in range(N):
        assert torch.equal(y[n:n + 1], ops.causal_conv_c8(x[n:n + 1].contiguous(), wt, b, 3, 2, 2)), n
    # 80 -> 80 channels (138 KB image: one workgroup per CU), many rows against few
    C = 80
    x = ops.to_c8(torch.randn((2, C, 3000, W), generator=g).to(torch.bfloat16).to(DEV))
    wt = (torch.randn((C, C, 5, 3), generator=g) * (C * 9) ** -0.5).to(torch.bfloat16).to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    y = ops.causal_conv_c8(x, wt, b, 3, 2, 2)
    assert torch.equal(y[1:, :200], ops.causal_conv_c8(x[1:, :200].contiguous(), wt, b, 3, 2, 2))


@pytest.mark.parametrize("dil", [2, 1, 3])
@pytest.mark.parametrize("N,Cin,Cout,T,W", [(2, 24, 24, 40, 64), (1, 64, 64, 21, 64), (1, 24, 24, 33, 16), (1, 32, 48, 17, 128),
                                            (1, 16, 32, 12, 40), (1, 24, 24, 9, 96), (1, 8, 8, 7, 24), (1, 40, 40, 13, 64)])
def test_causal_conv_c8_fp32(ops, N, Cin, Cout, T, W, dil):
    """`sea_causal_conv_c8_f32` (round 5): fp32 data on the fp32 MFMA -- exact fp32 products, so the bar is fp32 rounding noise
    (the 16-bit kernels' 2e-2 becomes 1e-5), tap positions and causality as for the 16-bit kernel."""
    assert ops.conv_c8_f32_supported(Cin, Cout, 3) and not ops.conv_c8_f32_supported(80, 80, 3)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, Cin, T, W), generator=g)
    wt = torch.randn((Cout, Cin, 5, 3), generator=g) * (Cin * 9) ** -0.5
    b = torch.randn(Cout, generator=g) * 0.1
    for relu in (True, False):
        ref = _causal_conv_ref(x.double(), wt.double(), b.double(), 3, dil, dil, relu) if False else \
            torch.nn.functional.conv2d(torch.nn.functional.pad(x.double(), (0, 0, 2 * dil, 0)), wt[:, :, :3, :].double(), b.double(), 1, (0, dil), dil)
        ref = torch.relu(ref) if relu else ref
        y = ops.causal_conv_c8(ops.to_c8(x.to(DEV)), wt.to(DEV), b.to(DEV), 3, dil, dil, relu=relu)
        assert y.dtype == torch.float32 and tuple(y.shape) == (N, T, Cout // 8, W, 8) and y.is_contiguous()
        torch.testing.assert_close(ops.from_c8(y).double().cpu(), ref, atol=1e-5, rtol=1e-5)
    xo = torch.zeros_like(x); xo[0, Cin // 2, T // 2, W // 3] = 1.0; xo[0, 0, T - 1, 0] = 1.0; xo[0, Cin - 1, 0, W - 1] = -2.0
    ref = _causal_conv_ref(xo, wt, b, 3, dil, dil, False)
    y = ops.from_c8(ops.causal_conv_c8(ops.to_c8(xo.to(DEV)), wt.to(DEV), b.to(DEV), 3, dil, dil, relu=False)).cpu()
    torch.testing.assert_close(y, ref, atol=1e-6, rtol=1e-6)                      # one weight (+ bias) per lit output pixel
    assert torch.equal(y == b.view(1, -1, 1, 1), ref == b.view(1, -1, 1, 1))      # and the same pixels stay at the bare bias
    x2 = x.clone(); x2[:, :, T // 2:] += 50                                       # causality along T
    y1 = ops.causal_conv_c8(ops.to_c8(x.to(DEV)), wt.to(DEV), b.to(DEV), 3, dil, dil)
    y2 = ops.causal_conv_c8(ops.to_c8(x2.to(DEV)), wt.to(DEV), b.to(DEV), 3, dil, dil)
    assert torch.equal(y1[:, :T // 2], y2[:, :T // 2])
    w1 = torch.randn((Cout, Cin, 1, 1), generator=g) * Cin ** -0.5
    y = ops.causal_conv_c8(ops.to_c8(x.to(DEV)), w1.to(DEV), b.to(DEV), 1, 1, 0, relu=False)
    torch.testing.assert_close(ops.from_c8(y).cpu(), torch.nn.functional.conv2d(x, w1, b), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("N,C,T,S,W", [(2, 12, 40, 2, 16), (1, 32, 70, 2, 64), (1, 40, 9, 2, 64), (1, 4, 5, 2, 128),
                                       (1, 12, 21, 2, 24), (1, 12, 13, 2, 96), (1, 4, 7, 2, 40)])    # lanes per row not a power of two
def test_split_layernorm_c8(ops, dtype, N, C, T, S, W):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn((N, C, T, S * W), generator=g) * 2 + 0.3).to(dtype)
    w = (torch.rand(W, generator=g) + 0.5).to(dtype)
    b = torch.randn(W, generator=g).to(dtype)
    a = ops.split_layernorm(x.to(DEV), S, w.to(DEV), b.to(DEV), 1e-5)
    c = ops.split_layernorm_c8(x.to(DEV), S, w.to(DEV), b.to(DEV), 1e-5)
    assert tuple(c.shape) == (N, T, C * S // 8, W, 8)
    c = ops.from_c8(c)
    # same arithmetic, different layout; the two kernels may contract one FMA differently: <= 1 ulp of the 16-bit type
    if dtype == torch.float32:                                       # (round 5) fp32 rows: fp32 rounding noise only
        torch.testing.assert_close(a, c, atol=2e-6, rtol=2e-6)
        return
    torch.testing.assert_close(a.float(), c.float(), atol=1e-3, rtol=8e-3)
    assert (a != c).float().mean().item() < 1e-3


def test_predictor_tail_accepts_channels_last(ops):
    g = torch.Generator().manual_seed(1)
    N, H, T, T_M = 1, 32, 50, 256
    C, W4 = 2 * H, T_M // 4
    y = torch.relu(torch.randn((N, C, T, W4), generator=g)).bfloat16().to(DEV)
    cw = (torch.randn((H, C), generator=g) * C ** -0.5).bfloat16().to(DEV)
    cb = (torch.randn(H, generator=g) * 0.1).bfloat16().to(DEV)
    lw = (torch.rand(T_M, generator=g) + 0.5).bfloat16().to(DEV); lb = torch.zeros(T_M).bfloat16().to(DEV)
    p1, s1 = ops.predictor_tail(y, cw, cb, lw, lb, up=4, T_m=T_M, want_scores=True)              # NCHW: VALU variant
    p2, s2 = ops.predictor_tail(y.contiguous(memory_format=torch.channels_last), cw, cb, lw, lb, up=4, T_m=T_M,
                                want_scores=True)                                                # NHWC: MFMA variant
    p_ref, s_ref = O.predictor_tail(y.float().cpu(), cw.float().cpu().view(H, C, 1, 1), cb.float().cpu(), lw.float().cpu(),
                                    lb.float().cpu(), 4, T_M)
    for pp, ss in ((p1, s1), (p2, s2)):
        torch.testing.assert_close(ss.float().cpu(), s_ref, atol=3e-2, rtol=1e-2)
        torch.testing.assert_close(pp.float().cpu(), p_ref, atol=1e-5, rtol=4e-2)
    assert (p1.float() - p2.float()).abs().max().item() < 2e-4


def test_module_bf16_hip_estimator_close_to_torch_estimator():
    """bf16 layer: channels-last MFMA CNN + fused kernels vs the torch modules (which round every intermediate to bf16)."""
    import sea_attention_amd as S
    from test_gpu_module import make_layer, run, causal_mask
    N, H, T, d, T_M, k = 1, 12, 512, 64, 256, 64
    layer = make_layer(H, d, T_M, k, T, torch.bfloat16)
    S.seed(9)
    q = torch.randn((N, H, T, d), device=DEV).bfloat16()
    mask = causal_mask(N, T, torch.bfloat16)
    _, b_hip = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    layer.attention.force_torch_estimator = True
    try:
        _, b_torch = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, True)
    finally:
        layer.attention.force_torch_estimator = False
    for name, tol in [('performer_context_layer', 3e-2), ('t_attention_predictor', 6e-2), ('estimated_attention_score', 0.25)]:
        a, b = b_hip[name].float(), b_torch[name].float()
        rel = ((a - b).norm() / b.norm()).item()
        assert rel < tol, (name, rel)


def test_estimator_kernels_are_bitwise_reproducible(ops):
    """Two launches on the same inputs give identical bits (no float atomics on the estimator path): the module-level
    dense-vs-sparse consistency test relies on it."""
    import math
    from sea_attention_amd.perlin_attention.performer import FastAttention
    torch.manual_seed(0)
    N, H, T, D, T_M = 2, 8, 700, 64, 256
    fa = FastAttention(D, nb_features=int(D * math.log(D) / 8), causal=True, generalized_attention=True).to(DEV)
    q = (torch.randn(N, H, T, D, device=DEV) * D ** -0.5).bfloat16(); k = torch.randn(N, H, T, D, device=DEV).bfloat16()
    v = torch.randn(N, H, T, D, device=DEV).bfloat16(); pos = torch.randn(T, D, device=DEV).bfloat16()
    a = ops.performer_value(q, k, v, pos, fa.projection_matrix)
    for _ in range(3):
        assert torch.equal(a, ops.performer_value(q, k, v, pos, fa.projection_matrix))
    y = ops.to_c8(torch.relu(torch.randn(N, 2 * H, T, T_M // 4, device=DEV)).bfloat16())
    cw = torch.randn(H, 2 * H, device=DEV).bfloat16(); cb = torch.zeros(H, device=DEV).bfloat16()
    lw = torch.ones(T_M, device=DEV).bfloat16(); lb = torch.zeros(T_M, device=DEV).bfloat16()
    p0, _ = ops.predictor_tail(y, cw, cb, lw, lb, up=4, T_m=T_M)
    wt = torch.randn(2 * H, 2 * H, 5, 3, device=DEV).bfloat16() * 0.05; b = torch.zeros(2 * H, device=DEV).bfloat16()
    c0 = ops.causal_conv_c8(y, wt, b, 3, 2, 2)
    for _ in range(3):
        assert torch.equal(p0, ops.predictor_tail(y, cw, cb, lw, lb, up=4, T_m=T_M)[0])
        assert torch.equal(c0, ops.causal_conv_c8(y, wt, b, 3, 2, 2))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T,d,T_M", [(2, 32, 40, 64, 256), (1, 12, 33, 64, 256), (1, 4, 70, 64, 128), (1, 8, 20, 64, 512),
                                         # the reference's own grid (benchmark_opt_ablation.py:160-186, exp_long_context.py:152) and
                                         # the padded split widths (T_M / 4 not a multiple of 16: 96 -> 24, 160 -> 40, 32 -> 8)
                                         (1, 12, 50, 64, 64), (1, 12, 37, 64, 96), (2, 12, 40, 64, 384), (1, 8, 19, 64, 160),
                                         (1, 4, 9, 64, 32), (1, 16, 21, 64, 448), (1, 4, 30, 64, 480),
                                         (1, 20, 17, 80, 256),
                                         (1, 40, 50, 128, 256), (2, 8, 300, 128, 256)])   # d = 128: encoder weights streamed
def test_predictor_mlp(ops, dtype, N, H, T, d, T_M):
    """Fused enc Linear+LN+GELU -> dec_row Linear+ChannelSplit+LN (C8) + gate Linear+sigmoid vs the module chain in fp32
    with the 16-bit rounding of every module output."""
    nn = torch.nn
    g = torch.Generator().manual_seed(5)
    Din, D1, Wd = 3 * d, 2 * d, T_M // 4
    D2 = 2 * Wd
    mods = [nn.Linear(Din, D1), nn.LayerNorm(D1), nn.Linear(D1, D2), nn.LayerNorm(Wd), nn.Linear(D1, 2)]
    with torch.no_grad():
        for m in mods:
            for prm in m.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g) * (0.3 if prm.dim() == 1 else prm.shape[-1] ** -0.5))
        mods[1].weight.add_(1.0); mods[3].weight.add_(1.0)
    import copy
    enc_lin, enc_ln, dec_lin, ln1, sc = [copy.deepcopy(m).to(DEV).to(dtype) for m in mods]
    x = torch.randn((N, H, T, Din), generator=g).to(dtype)
    r = lambda t: t.to(dtype).float()
    F = torch.nn.functional
    xf = x.float()
    e = r(xf @ r(mods[0].weight).T + r(mods[0].bias))
    e = r(F.gelu(F.layer_norm(e, (D1,), r(mods[1].weight), r(mods[1].bias), mods[1].eps)))
    dd = r(e @ r(mods[2].weight).T + r(mods[2].bias)).view(N, H, T, 2, Wd)
    y = r(F.layer_norm(dd, (Wd,), r(mods[3].weight), r(mods[3].bias), mods[3].eps))          # (N,H,T,2,Wd)
    y_ref = y.permute(0, 1, 3, 2, 4).reshape(N, H * 2, T, Wd)                                  # channel = 2h + split
    gate = torch.sigmoid(r(e @ r(mods[4].weight).T + r(mods[4].bias)))
    x_c8, tp, rs, av = ops.predictor_mlp(x.to(DEV), enc_lin, enc_ln, dec_lin, ln1, sc, want_tpred=True)
    assert tuple(x_c8.shape) == (N, T, H * 2 // 8, Wd, 8) and tp.dtype == dtype
    atol, rtol = (6e-2, 3e-2) if dtype == torch.bfloat16 else (8e-3, 4e-3)
    torch.testing.assert_close(tp.float().cpu(), e, atol=atol, rtol=rtol)
    got = ops.from_c8(x_c8).float().cpu()
    torch.testing.assert_close(got, y_ref, atol=atol * 2, rtol=rtol)
    assert (got - y_ref).abs().mean().item() < atol / 8
    torch.testing.assert_close(rs.cpu(), gate[..., 0], atol=atol / 2, rtol=rtol)
    torch.testing.assert_close(av.cpu(), gate[..., 1], atol=atol / 2, rtol=rtol)
    x2, tp2, _, _ = ops.predictor_mlp(x.to(DEV), enc_lin, enc_ln, dec_lin, ln1, sc, want_tpred=False)
    assert tp2 is None and torch.equal(x2, x_c8)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T,k,T_M", [(2, 32, 300, 64, 256), (1, 12, 257, 16, 256), (1, 4, 64, 8, 256), (1, 40, 100, 64, 256),
                                         (1, 64, 40, 32, 256),
                                         # any predictor length (round 4): the flat-LDS-image form of the fused kernel
                                         (1, 12, 300, 32, 64), (2, 12, 130, 128, 96), (1, 12, 200, 64, 128), (1, 12, 150, 32, 384),
                                         (1, 32, 70, 64, 512), (1, 6, 90, 16, 256), (1, 32, 50, 64, 128), (1, 3, 40, 8, 36)])
def test_predictor_tail_select_bit_identical(ops, dtype, N, H, T, k, T_M):
    """One-launch tail + top-k selection == predictor_tail followed by topk_to_csr, bit for bit (map, CSR, offsets)."""
    C, W4 = 2 * H, T_M // 4
    C = (C + 7) // 8 * 8
    assert ops.predictor_tail_select_supported(torch.empty((1, 1, 1, 1, 8), dtype=dtype), H, T_M)
    g = torch.Generator().manual_seed(9)
    y = ops.to_c8(torch.relu(torch.randn((N, C, T, W4), generator=g)).to(dtype).to(DEV))
    cw = (torch.randn((H, C), generator=g) * C ** -0.5).to(dtype).to(DEV)
    cb = (torch.randn(H, generator=g) * 0.1).to(dtype).to(DEV)
    lw = (torch.rand(T_M, generator=g) + 0.5).to(dtype).to(DEV)
    lb = (torch.randn(T_M, generator=g) * 0.1).to(dtype).to(DEV)
    keep = ops.keep_table_causal(H, T, T_M, k, device=DEV)
    p0, s0 = ops.predictor_tail(y, cw, cb, lw, lb, up=4, T_m=T_M, want_scores=True)
    c0, _ = ops.topk_to_csr(p0, keep, k, target_width=T)
    p1, s1, sel = ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T, want_scores=True)
    c1 = ops.csr_from_selection(*sel, H, T_M, T, k, True, None, keep)
    assert torch.equal(p0, p1) and torch.equal(s0, s1)
    assert torch.equal(c0.bits, c1.bits) and torch.equal(c0.crow, c1.crow) and torch.equal(c0.head_off, c1.head_off)
    for i in range(N):
        Z = int(c0.crow[i, -1])
        assert torch.equal(c0.col[i, :Z], c1.col[i, :Z])
    # the map left on chip (round 4): same selection, and the lazy handle computes the very same map when somebody asks
    pl, _, sel_l = ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T, lazy_probs=True)
    assert isinstance(pl, ops.LazyTensor) and not pl.is_materialized and tuple(pl.shape) == tuple(p0.shape) and pl.dtype == dtype
    cl = ops.csr_from_selection(*sel_l, H, T_M, T, k, True, None, keep)
    assert torch.equal(c0.bits, cl.bits) and torch.equal(c0.crow, cl.crow) and torch.equal(c0.head_off, cl.head_off)
    assert not pl.is_materialized
    assert torch.equal(pl, p0) and pl.is_materialized and torch.equal(pl[:, :, -1].float(), p0[:, :, -1].float())
    # all-equal map (massive ties): the selection's slow path re-reads the map the same launch has just written
    y0 = torch.zeros_like(y)
    p2, _, sel2 = ops.predictor_tail_select(y0, cw * 0, cb * 0, lw, lb * 0, up=4, T_m=T_M, keep=keep, k=k, T_src=T)
    c2 = ops.csr_from_selection(*sel2, H, T_M, T, k, True, None, keep)
    c3, _ = ops.topk_to_csr(ops.predictor_tail(y0, cw * 0, cb * 0, lw, lb * 0, up=4, T_m=T_M)[0], keep, k, target_width=T)
    assert torch.equal(c2.bits, c3.bits) and torch.equal(c2.crow, c3.crow)
    # a bump on a flat map: a few distinct levels, the K-th key inside a tie of thousands -- the threshold bin overflows the
    # candidate list AFTER the histogram pass (keys differ), and the fallback gets its key range from the packed 16-bit keys
    y1 = torch.zeros_like(y)
    y1[:, :, :, 5:7 if W4 > 8 else 5:6, :] = 1.0                           # (N, T, C/8, W4, 8): two pixel columns lit
    args1 = (y1, torch.full_like(cw, 0.05), cb * 0, torch.ones_like(lw), lb * 0)   # every head, every flat pixel alike
    p4, _, sel4 = ops.predictor_tail_select(*args1, up=4, T_m=T_M, keep=keep, k=k, T_src=T)
    c4 = ops.csr_from_selection(*sel4, H, T_M, T, k, True, None, keep)
    _, _, sel4l = ops.predictor_tail_select(*args1, up=4, T_m=T_M, keep=keep, k=k, T_src=T, lazy_probs=True)   # slow path, no map in memory
    assert torch.equal(sel4l[0], sel4[0]) and torch.equal(sel4l[1], sel4[1]) and torch.equal(sel4l[2], sel4[2])
    p5 = ops.predictor_tail(*args1, up=4, T_m=T_M)[0]
    c5, _ = ops.topk_to_csr(p5, keep, k, target_width=T)
    assert torch.equal(p4, p5) and p4.float().unique().numel() < 64
    assert torch.equal(c4.bits, c5.bits) and torch.equal(c4.crow, c5.crow)


@pytest.mark.parametrize("N,H,T,k", [(2, 12, 150, 64), (1, 32, 133, 64), (1, 4, 64, 8), (1, 20, 70, 16), (1, 8, 300, 32)])
def test_predictor_tail_select_fp32_bit_identical(ops, N, H, T, k):
    """fp32 data (round 5): the fused tail + selection on the fp32 MFMA == `predictor_tail` (fp32, C8 input: the same device
    code) followed by `topk_to_csr`, bit for bit; and the map agrees with the torch evaluation of the tail to fp32 noise."""
    T_M, C, W4 = 256, 2 * H, 64
    assert ops.predictor_tail_select_supported(torch.empty((1, 1, 1, 1, 8), dtype=torch.float32), H, T_M)
    assert not ops.predictor_tail_select_supported(torch.empty((1, 1, 1, 1, 8), dtype=torch.float32), 40, T_M)
    g = torch.Generator().manual_seed(17)
    yn = torch.relu(torch.randn((N, C, T, W4), generator=g))
    y = ops.to_c8(yn.to(DEV))
    cw = (torch.randn((H, C), generator=g) * C ** -0.5).to(DEV)
    cb = (torch.randn(H, generator=g) * 0.1).to(DEV)
    lw = (torch.rand(T_M, generator=g) + 0.5).to(DEV)
    lb = (torch.randn(T_M, generator=g) * 0.1).to(DEV)
    keep = ops.keep_table_causal(H, T, T_M, k, device=DEV)
    p0, s0 = ops.predictor_tail(y, cw, cb, lw, lb, up=4, T_m=T_M, want_scores=True)
    c0, _ = ops.topk_to_csr(p0, keep, k, target_width=T)
    p1, s1, sel = ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T, want_scores=True, lazy_probs=True)
    assert p1.dtype == torch.float32 and not isinstance(p1, ops.LazyTensor)          # the fp32 map is always written
    c1 = ops.csr_from_selection(*sel, H, T_M, T, k, True, None, keep)
    assert torch.equal(p0, p1) and torch.equal(s0, s1)
    assert torch.equal(c0.bits, c1.bits) and torch.equal(c0.crow, c1.crow) and torch.equal(c0.head_off, c1.head_off)
    # the map against torch: 1x1 conv (pad 1) on the x4-upsampled rows, area resize to T_M, LayerNorm, softmax
    z = torch.einsum("hc,nctw->nhtw", cw.cpu().double(), yn.double()) + cb.cpu().double().view(1, H, 1, 1)
    up = z.repeat_interleave(4, dim=-1)
    padded = torch.cat([cb.cpu().double().view(1, H, 1, 1).expand(N, H, T, 1), up, cb.cpu().double().view(1, H, 1, 1).expand(N, H, T, 1)], -1)
    a = torch.nn.functional.adaptive_avg_pool2d(padded, (T, T_M))
    ref = torch.softmax(torch.nn.functional.layer_norm(a, (T_M,), lw.cpu().double(), lb.cpu().double(), 1e-5), -1)
    assert (p1.cpu().double() - ref).abs().max().item() < 2e-6
    # ties: an all-equal map goes through the unpacked selection's slow path, which re-reads the row this launch has stored
    y0 = torch.zeros_like(y)
    p2, _, sel2 = ops.predictor_tail_select(y0, cw * 0, cb * 0, lw, lb * 0, up=4, T_m=T_M, keep=keep, k=k, T_src=T)
    c2 = ops.csr_from_selection(*sel2, H, T_M, T, k, True, None, keep)
    c3, _ = ops.topk_to_csr(ops.predictor_tail(y0, cw * 0, cb * 0, lw, lb * 0, up=4, T_m=T_M)[0], keep, k, target_width=T)
    assert torch.equal(c2.bits, c3.bits) and torch.equal(c2.crow, c3.crow)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T,k,T_M", [(2, 32, 150, 64, 256), (1, 12, 133, 16, 256), (1, 40, 100, 64, 256), (1, 4, 40, 8, 256),
                                         (1, 20, 50, 16, 256), (1, 12, 90, 32, 96), (1, 12, 70, 32, 384), (1, 32, 40, 64, 128),
                                         (1, 36, 30, 32, 256)])
def test_conv_z_epilogue_bit_identical(ops, dtype, N, H, T, k, T_M):
    """`sea_causal_conv_c8_z` (round 5): the last (conv, ReLU) launch with the tail's 1x1 convolution in its epilogue.  Its
    activation is bitwise `causal_conv_c8`'s, z is what the tail's own z stage computes from that activation (checked through
    the maps: tail(z) == tail(y) bit for bit, tail + selection likewise) and equals the fp32 product of the rounded operands."""
    C, W4 = 2 * H, T_M // 4
    if C % 8:
        pytest.skip("C8 needs whole blocks of 8 channels")
    assert ops.conv_z_supported(C, H, 3, W4)
    g = torch.Generator().manual_seed(21)
    x = ops.to_c8(torch.randn((N, C, T, W4), generator=g).to(dtype).to(DEV))
    w = (torch.randn((C, C, 5, 3), generator=g) * (9 * C) ** -0.5).to(dtype).to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).to(dtype).to(DEV)
    cw = (torch.randn((H, C), generator=g) * C ** -0.5).to(dtype).to(DEV)
    cb = (torch.randn(H, generator=g) * 0.1).to(dtype).to(DEV)
    lw = (torch.rand(T_M, generator=g) + 0.5).to(dtype).to(DEV)
    lb = (torch.randn(T_M, generator=g) * 0.1).to(dtype).to(DEV)
    y0 = ops.causal_conv_c8(x, w, b, 3, 2, 2, relu=True)
    y1, z = ops.causal_conv_c8_z(x, w, b, 3, 2, 2, cw, cb, lw, lb, relu=True, want_y=True)
    yn, z2 = ops.causal_conv_c8_z(x, w, b, 3, 2, 2, cw, cb, lw, lb, relu=True)
    assert yn is None and torch.equal(z, z2) and torch.equal(y0, y1)
    assert z.shape == (N, T, H, W4) and z.dtype == torch.float32
    ref = torch.einsum("hc,nctw->nthw", cw.float(), ops.from_c8(y0).float()) + cb.float().view(1, 1, H, 1)
    assert (z - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    p0, s0 = ops.predictor_tail(y0, cw, cb, lw, lb, up=4, T_m=T_M, want_scores=True)
    p1, s1 = ops.predictor_tail_z(z, cw, cb, lw, lb, up=4, T_m=T_M, dtype=dtype, want_scores=True)
    assert torch.equal(p0, p1) and torch.equal(s0, s1)
    if ops.predictor_tail_select_supported(y0, H, T_M):
        keep = ops.keep_table_causal(H, T, T_M, k, device=DEV)
        pa, _, sa = ops.predictor_tail_select(y0, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T)
        pb, _, sb = ops.predictor_tail_select(None, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T, z=z, map_dtype=dtype)
        pl, _, sl = ops.predictor_tail_select(None, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T, z=z, map_dtype=dtype,
                                              lazy_probs=True)
        assert torch.equal(pa, p0) and torch.equal(pb, p0)
        for u, v_, w_ in zip(sa, sb, sl):
            assert torch.equal(u, v_) and torch.equal(u, w_)
        assert isinstance(pl, ops.LazyTensor) and not pl.is_materialized and torch.equal(pl, p0)


@pytest.mark.parametrize("D", [64, 80, 128])
@pytest.mark.parametrize("N,H,T", [(1, 2, 256), (2, 3, 200), (1, 4, 1000)])
def test_performer_emits_cumulative_average(ops, N, H, T, D):
    """The bf16 Performer launch can also write cumsum(v)/(t+1) (step K's input): equals the cumavg kernel's values
    (fp32 accumulation on both sides, at most a bf16 rounding step apart) and leaves the main output untouched."""
    import math
    from sea_attention_amd.perlin_attention.performer import FastAttention
    torch.manual_seed(11)
    fa = FastAttention(D, nb_features=int(D * math.log(D) / 8), causal=True, generalized_attention=True).to(DEV)
    q = (torch.randn(N, H, T, D, device=DEV) * D ** -0.5).bfloat16(); k = torch.randn(N, H, T, D, device=DEV).bfloat16()
    v = torch.randn(N, H, T, D, device=DEV).bfloat16(); pos = torch.randn(T, D, device=DEV).bfloat16()
    out0 = ops.performer_value(q, k, v, pos, fa.projection_matrix)
    out1, avg = ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=True)
    assert torch.equal(out0, out1)
    ref = v.float().cumsum(-2) / torch.arange(1, T + 1, device=DEV).view(1, 1, -1, 1)
    torch.testing.assert_close(avg.float(), ref, atol=1e-2, rtol=1e-2)
    ca = ops.cumavg(v).float()
    assert ((avg.float() - ca).abs() <= ca.abs() * 2.0 ** -7 + 1e-6).all()
    assert (avg.float() != ca).float().mean().item() < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,N,H,T,D,nbf", [
    (torch.bfloat16, 1, 4, 1024, 64, 8),     # split-bf16 kernel, plan cuts (4 pairs)
    (torch.float16, 1, 3, 777, 64, 8),       # ragged last segment / last chunk
    (torch.bfloat16, 1, 4, 1024, 128, 8),    # wide split-bf16 kernel, 32-row chunks
    (torch.float16, 1, 2, 1111, 128, 8),
    (torch.float32, 1, 2, 600, 128, 8),      # fp32-MFMA kernel, 32-row chunks
    (torch.bfloat16, 2, 2, 640, 80, 8),      # wide kernel, padded head dimension, ragged column blocks
    (torch.float16, 1, 3, 1500, 80, 8),
    (torch.float32, 1, 2, 900, 64, 8),
])
def test_performer_sequence_parallel_equals_sequential(ops, dtype, N, H, T, D, nbf):
    """`sea_performer_causal_segmented`: cutting the rows into segments (two launches, carried state) gives the rows of
    the one-pass kernel up to fp32 summation order -- i.e. at most one rounding step of the output dtype apart on a
    small fraction of the elements -- for the plan's choice and for forced 2 / 4 segments; the copy of v stays exact."""
    import math
    from sea_attention_amd.perlin_attention.performer import FastAttention
    from sea_attention_amd.perlin_attention.ops import predictor as PR
    torch.manual_seed(21)
    nb = int(D * math.log(D) / nbf)
    fa = FastAttention(D, nb_features=nb, causal=True, generalized_attention=True).to(DEV)
    q = (torch.randn(N, H, T, D, device=DEV) * D ** -0.5).to(dtype); k = torch.randn(N, H, T, D, device=DEV).to(dtype)
    v = torch.randn(N, H, T, D, device=DEV).to(dtype); pos = torch.randn(T, D, device=DEV).to(dtype)
    plan = PR.performer_plan(N, H, T, D, nb, dtype)
    assert plan[0] > 1 and plan[1] > 0                               # few pairs: the plan cuts
    assert PR.performer_plan(64, 8, T, D, nb, dtype) == (1, 0)       # many pairs: it does not
    want_avg = PR.performer_avg_supported(q, nb)
    seq = ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=want_avg, n_segments=1)
    eps = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10, torch.float32: 2.0 ** -18}[dtype]
    for nseg in (None, 2, 4):
        got = ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=want_avg, n_segments=nseg)
        pairs = [(got[0], seq[0]), (got[1], seq[1])] if want_avg else [(got, seq)]
        for g, s in pairs:
            g, s = g.float(), s.float()
            assert torch.isfinite(g).all()
            assert ((g - s).abs() <= s.abs() * eps + 1e-6).all(), (nseg, (g - s).abs().max().item())
            if dtype != torch.float32:
                assert (g != s).float().mean().item() < 0.02, nseg
        main = got[0] if want_avg else got
        assert torch.equal(main[..., 2 * D:], v)
    # reproducible run to run (the carried increments are added in segment order)
    a = ops.performer_value(q, k, v, pos, fa.projection_matrix, n_segments=4)
    b = ops.performer_value(q, k, v, pos, fa.projection_matrix, n_segments=4)
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="empty"):                 # a cut that would leave a segment without rows
        ops.performer_value(q, k, v, pos, fa.projection_matrix, n_segments=(T + 63) // 64 + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,N,H,T,D", [(torch.bfloat16, 2, 4, 448, 64), (torch.float16, 1, 4, 300, 64),
                                           (torch.bfloat16, 1, 3, 448, 128), (torch.float16, 2, 2, 300, 128),
                                           (torch.bfloat16, 1, 3, 448, 80), (torch.float16, 1, 2, 300, 80)])
def test_performer_step_continues_the_sequence(ops, dtype, N, H, T, D):
    """`sea_performer_causal_step` (kv-cache decoding), chunk aligned: the state image is the state at the last chunk
    boundary and every call walks the open chunk again from the kv-cache, so feeding the rows in ANY pieces -- chunk
    aligned, ragged, single rows -- gives the rows and the cumulative average of v of the one-pass kernel BIT FOR BIT
    (reference protocol test_perlin_opt_cache.py:7-32)."""
    import math
    from sea_attention_amd.perlin_attention.performer import FastAttention
    torch.manual_seed(31)
    nb = int(D * math.log(D) / 8)
    fa = FastAttention(D, nb_features=nb, causal=True, generalized_attention=True).to(DEV)
    q = (torch.randn(N, H, T, D, device=DEV) * D ** -0.5).to(dtype); k = torch.randn(N, H, T, D, device=DEV).to(dtype)
    v = torch.randn(N, H, T, D, device=DEV).to(dtype); pos = torch.randn(T + 8, D, device=DEV).to(dtype)
    ref, ref_avg = ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=True, n_segments=1)
    C = ops.performer_chunk_rows(D, nb, dtype)
    assert C == (64 if D == 64 else 32)

    def feed(cuts, nseg_first=1):
        state, t0, outs, avgs, states = None, 0, [], [], []
        for i, t1 in enumerate(cuts):
            # the kv-cache up to the new rows, from row 0; the call reads it from the last chunk boundary on
            o, a, state = ops.performer_step(q[:, :, t0:t1], k[:, :, :t1], v[:, :, :t1], pos, fa.projection_matrix,
                                             state_in=state, t_base=t0, n_segments=nseg_first if i == 0 else 1)
            outs.append(o); avgs.append(a); states.append(state); t0 = t1
        return torch.cat(outs, 2), torch.cat(avgs, 2), states

    out, avg, st_aligned = feed((128, 320, T) if T > 320 else (64, 256, T))   # chunk-aligned pieces
    assert torch.equal(out, ref) and torch.equal(avg, ref_avg)
    cuts = [200] + list(range(201, 216)) + [T - 37, T]                   # a ragged prefill, 15 single rows, two longer pieces
    out, avg, st_ragged = feed(cuts)
    assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
    assert torch.equal(avg, ref_avg)
    assert torch.equal(out[..., 2 * D:], v)
    # every image is the state at a chunk boundary: two feeds that end at the same row count hold the same image, and a
    # single-row step inside a chunk leaves it untouched
    assert torch.equal(st_aligned[-1], st_ragged[-1])
    for a, b, t_prev, t_now in zip(st_ragged[1:], st_ragged[:-1], cuts[:-1], cuts[1:]):
        if t_now // C == t_prev // C:
            assert torch.equal(a, b), (t_prev, t_now)
    # one token at a time across two chunk boundaries, from a one-row prefill
    lo = 2 * C - 3
    out1, avg1, _ = feed([1] + list(range(2, lo)) + list(range(lo, lo + C + 6)) + [T])
    assert torch.equal(out1, ref) and torch.equal(avg1, ref_avg)
    # the prefill piece may itself be cut into segments (another fp32 summation order of the carried sums, as in the
    # sequence-parallel stateless kernel): equal to a rounding step of the output dtype
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    out2, avg2, _ = feed(cuts, 2)
    for g, s in ((out2, ref), (avg2, ref_avg)):
        g, s = g.float(), s.float()
        assert ((g - s).abs() <= s.abs() * eps + 1e-6).all(), (g - s).abs().max().item()
        assert (g != s).float().mean().item() < 0.02
    with pytest.raises(AssertionError):                                  # an image without its row count
        ops.performer_step(q[:, :, 5:13], k[:, :, :13], v[:, :, :13], pos, fa.projection_matrix, state_in=None, t_base=5)
    with pytest.raises(AssertionError):                                  # fp32 data: no chunk-aligned step (torch-side state serves it)
        ops.performer_step(q[:, :, :8].float(), k[:, :, :8].float(), v[:, :, :8].float(), pos.float(), fa.projection_matrix)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T,D", [(1, 4, 5000, 80), (1, 2, 4096, 128), (2, 2, 2500, 64)])
def test_cumavg_sliced(ops, dtype, N, H, T, D):
    """`sea_cumavg_sliced`: the rows cut into slices with carried column totals (two launches) -- the averages of the
    one-launch kernel up to fp32 summation order; the default picks slices by itself when N*H is small."""
    g = torch.Generator().manual_seed(4)
    v = torch.randn((N, H, T, D), generator=g).to(dtype).to(DEV)
    one = ops.cumavg(v, n_slices=1).float()
    ref = v.float().cumsum(-2) / torch.arange(1, T + 1, device=DEV).view(1, 1, -1, 1)
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    for ns in (None, 3, 16):
        got = ops.cumavg(v, n_slices=ns).float()
        assert ((got - one).abs() <= one.abs() * eps + 1e-6).all(), (ns, (got - one).abs().max().item())
        assert (got != one).float().mean().item() < 0.02
        torch.testing.assert_close(got, ref, atol=1e-2, rtol=1e-2)
    with pytest.raises(RuntimeError, match="workspace"):
        from sea_attention_amd import _lib
        out = torch.empty_like(v)
        _lib.check(_lib.load().sea_cumavg_sliced(v.data_ptr(), _lib.dtype_code(dtype), N, H, T, D, _lib.strides3(v), out.data_ptr(),
                                                 4, None, 0, None), "sea_cumavg_sliced")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,T,D", [(2, 3, 300, 64), (1, 4, 257, 80), (2, 2, 190, 128)])
def test_performer_takes_head_interleaved_views(ops, dtype, N, H, T, D):
    """q / k / v as the model hands them over -- (N, T, H, D) projections viewed as (N, H, T, D), i.e. a row stride of H*D --
    give bitwise the result of contiguous copies (the kernels address through the three strides), average output included."""
    import math
    from sea_attention_amd.perlin_attention.performer import FastAttention
    torch.manual_seed(13)
    nb = int(D * math.log(D) / 8)
    fa = FastAttention(D, nb_features=nb, causal=True, generalized_attention=True).to(DEV)
    base = [torch.randn(N, T, H, D, device=DEV).to(dtype) for _ in range(3)]
    base[0] = (base[0].float() * D ** -0.5).to(dtype)
    qv, kv, vv = (b.permute(0, 2, 1, 3) for b in base)                 # views: strides (T*H*D, D, H*D, 1)
    assert not qv.is_contiguous()
    pos = torch.randn(T, D, device=DEV).to(dtype)
    a, a_avg = ops.performer_value(qv, kv, vv, pos, fa.projection_matrix, want_avg=True)
    b, b_avg = ops.performer_value(qv.contiguous(), kv.contiguous(), vv.contiguous(), pos, fa.projection_matrix, want_avg=True)
    assert torch.equal(a, b) and torch.equal(a_avg, b_avg)
    assert torch.equal(a[..., 2 * D:], vv)
