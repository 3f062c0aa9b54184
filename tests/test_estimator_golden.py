"""The predictor CNN pieces (steps E-F) against the REFERENCE's own module classes: tests/golden/estimator.npz holds
seeded inputs, the parameters the reference's constructors produced and the outputs of its forwards
(make_golden_estimator.py imports src/models/perlin_attention/modules.py of the reference in place).

CPU half (not gpu): this package's torch modules -- the drop-in `perlin_attention.modules` -- loaded with those parameters
reproduce the reference's outputs.  GPU half: the HIP kernels (C8 convolutions, predictor tail) on the same parameters,
bf16 data, against the reference's fp32 outputs.  Together with test_gpu_predictor.py (HIP vs this package's modules) this
pins steps E-F to the reference's code, not only to a restatement of it."""
import os

import numpy as np
import pytest
import torch
from torch import nn

from sea_attention_amd.perlin_attention import modules as M

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "estimator.npz"))


def _t(key):
    return torch.from_numpy(G[key])


def _load(module, prefix):
    sd = {k[len(prefix) + 7:]: _t(k) for k in G.files if k.startswith(prefix + "/param/")}
    assert sd, prefix
    module.load_state_dict(sd, strict=True)          # same parameter names and shapes as the reference's classes
    return module.eval()


@pytest.mark.parametrize("name,args", [("conv_k3_d2", (8, 8, 3, 2, 2)), ("conv_k3_d1", (4, 12, 3, 1, 1)), ("conv_k1_p1", (8, 4, 1, 1, 1))])
def test_causal_conv2d_equals_reference(name, args):
    cin, cout, ks, dil, pad = args
    m = _load(M.CausalConv2d(cin, cout, ks, padding=pad, dilation=dil, stride=(1, 1) if ks == 3 else 1, causal=True), name)
    with torch.no_grad():
        y = m(_t(name + "/x"))
    torch.testing.assert_close(y, _t(name + "/y"), atol=1e-6, rtol=1e-6)


def test_upsample_and_interpolate_equal_reference():
    with torch.no_grad():
        assert torch.equal(M.UpsampleFP32((1, 4), torch.float16)(_t("upsample/x")), _t("upsample/y"))
        torch.testing.assert_close(M.interpolate(_t("interp_area/x"), (6, 64)), _t("interp_area/y"), atol=1e-7, rtol=1e-6)
        torch.testing.assert_close(M.interpolate(_t("interp_bilinear/x"), (6, 64)), _t("interp_bilinear/y"), atol=1e-7, rtol=1e-6)


def _cnn(name):
    H, inner, T, T_M, nconv = (int(v) for v in G[name + "/meta"])
    C = inner * H
    body = []
    for _ in range(nconv):
        body += [M.CausalConv2d(C, C, 3, padding=2, dilation=2, stride=(1, 1), causal=True), nn.ReLU()]
    body += [M.UpsampleFP32((1, 4), torch.float16), M.CausalConv2d(C, H, 1, padding=1, causal=True)]
    keepres = _load(M.KeepRes(*body, output_width=T_M), name)
    ln2 = nn.LayerNorm(T_M).eval()
    with torch.no_grad():
        ln2.weight.copy_(_t(name + "/ln2_weight")); ln2.bias.copy_(_t(name + "/ln2_bias"))
    return keepres, ln2, (H, C, T, T_M, nconv)


@pytest.mark.parametrize("name", ["cnn_std", "cnn_deeper", "cnn_tm256"])
def test_predictor_cnn_stack_equals_reference(name):
    keepres, ln2, _ = _cnn(name)
    with torch.no_grad():
        y = keepres(_t(name + "/x"))
        torch.testing.assert_close(y, _t(name + "/y"), atol=2e-6, rtol=1e-6)
        torch.testing.assert_close(torch.softmax(ln2(y), -1), _t(name + "/probs"), atol=1e-6, rtol=1e-5)


def test_resblock_equals_reference():
    rb = _load(M.ResBlock(6, causal=True), "resblock")
    with torch.no_grad():
        torch.testing.assert_close(rb(_t("resblock/x")), _t("resblock/y"), atol=1e-6, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", ["cnn_std", "cnn_deeper", "cnn_tm256"])
def test_hip_predictor_cnn_against_reference_outputs(name, dtype):
    """C8 convolutions + predictor tail (HIP, 16-bit data) on the reference's parameters vs the reference's fp32 outputs: the
    inputs and weights are rounded to the data type first (what the kernels see), the reference's module stack is run on those
    rounded values in fp32 as the bar's centre, and its own un-rounded output bounds how far rounding alone moves it."""
    from sea_attention_amd.perlin_attention import ops
    dev = "cuda:0"
    keepres, ln2, (H, C, T, T_M, nconv) = _cnn(name)
    x = _t(name + "/x").to(dtype)
    with torch.no_grad():
        for p in keepres.parameters():
            p.copy_(p.to(dtype).float())
        ln2.weight.copy_(ln2.weight.to(dtype).float()); ln2.bias.copy_(ln2.bias.to(dtype).float())
        ref_scores = ln2(keepres(x.float()))
        ref_probs = torch.softmax(ref_scores, -1)
        # rounding the data alone keeps the reference's module close to its golden output (sanity of the bar below)
        assert (ref_probs - _t(name + "/probs")).abs().max() < (2e-2 if dtype == torch.bfloat16 else 3e-3)
        y = ops.to_c8(x.to(dev))
        convs = [m for m in keepres.net if isinstance(m, M.CausalConv2d)]
        for conv in convs[:-1]:
            y = ops.causal_conv_c8(y, conv.weight.to(dev), conv.bias.to(dev), conv.kernel_size, conv.dilation, conv.padding[1], relu=True)
        c4 = convs[-1]
        probs, scores = ops.predictor_tail(y, c4.weight[:, :, 0, 0].to(dev), c4.bias.to(dev), ln2.weight.to(dev), ln2.bias.to(dev),
                                           up=4, T_m=T_M, eps=ln2.eps, want_scores=True)
    assert probs.shape == (2, H, T, T_M)
    tol = dict(atol=6e-2, rtol=3e-2) if dtype == torch.bfloat16 else dict(atol=8e-3, rtol=4e-3)   # scores are O(1) LayerNorm outputs
    torch.testing.assert_close(scores.float().cpu(), ref_scores, **tol)
    ptol = 4e-3 if dtype == torch.bfloat16 else 6e-4                                               # probabilities <= 1 / a few
    assert (probs.float().cpu() - ref_probs).abs().max().item() < ptol
