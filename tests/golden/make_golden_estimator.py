#!/usr/bin/env python3
"""Golden vectors for the predictor CNN pieces (steps E-F of the estimator) from the reference's own module classes.

Run ONLY in the build container (where /root/reference exists):

    python tests/golden/make_golden_estimator.py

`src/models/perlin_attention/modules.py` of the reference imports nothing but torch, so it is loaded IN PLACE as a
stand-alone module (its package `__init__` -- which needs performer_pytorch / numba / transformers -- never runs) and
its classes `CausalConv2d`, `KeepRes`, `UpsampleFP32`, `ResBlock`, `Residual` and `interpolate` are instantiated and
run on the CPU in fp32.  Nothing of the reference is copied here: what is written is `estimator.npz` -- seeded inputs,
the (seeded) parameters the reference's constructors produced, and the outputs its `forward`s gave for them.

The stacks built are the ones `PerlinAttention.__init__` builds for the causal configuration
(attention.py:253-281 of the reference: two / three dilated `CausalConv2d` + ReLU, `UpsampleFP32((1, 4))`, the 1x1
`CausalConv2d(.., padding=1)`, all inside `KeepRes(output_width=T_M)`), without the `ModuleBenchmark` timer wrappers
and `ChannelSplit` (those live in attention.py, which cannot be imported without performer_pytorch).
"""
import importlib.util
import os

import numpy as np
import torch
from torch import nn

REF = "/root/reference/src/models/perlin_attention/modules.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_modules():
    spec = importlib.util.spec_from_file_location("ref_perlin_modules", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    R = load_reference_modules()
    out = {}

    def put(prefix, module, x, y):
        for k, v in module.state_dict().items():
            out[f"{prefix}/param/{k}"] = v.detach().numpy().copy()
        out[f"{prefix}/x"] = x.numpy().copy()
        out[f"{prefix}/y"] = y.detach().numpy().copy()

    # -- single CausalConv2d: (kernel, dilation, padding) as the predictor uses them + the plain-dilation case
    for name, (cin, cout, ks, dil, pad, T, W) in {
            "conv_k3_d2": (8, 8, 3, 2, 2, 11, 16), "conv_k3_d1": (4, 12, 3, 1, 1, 9, 8), "conv_k1_p1": (8, 4, 1, 1, 1, 7, 16)}.items():
        torch.manual_seed(100 + len(out))
        m = R.CausalConv2d(cin, cout, ks, padding=pad, dilation=dil, stride=(1, 1) if ks == 3 else 1, causal=True).eval()
        x = torch.randn(2, cin, T, W)
        with torch.no_grad():
            put(name, m, x, m(x))

    # -- UpsampleFP32 and the area / bilinear `interpolate`
    torch.manual_seed(7)
    x = torch.randn(1, 3, 5, 16)
    with torch.no_grad():
        out["upsample/x"] = x.numpy().copy()
        out["upsample/y"] = R.UpsampleFP32((1, 4), torch.float16)(x).numpy().copy()
        xi = torch.randn(1, 2, 6, 66)
        out["interp_area/x"] = xi.numpy().copy()
        out["interp_area/y"] = R.interpolate(xi, (6, 64)).numpy().copy()              # T_M + 2 -> T_M: 'area'
        xb = torch.randn(1, 2, 6, 16)
        out["interp_bilinear/x"] = xb.numpy().copy()
        out["interp_bilinear/y"] = R.interpolate(xb, (6, 64)).numpy().copy()          # wider: 'bilinear'

    # -- the causal predictor CNN body (KeepRes stack), standard and PERLIN_HOTFIX_OPT_DEEPER depth, + cnn.lnorm2
    for name, (H, inner, T, T_M, nconv) in {"cnn_std": (4, 2, 13, 64, 2), "cnn_deeper": (4, 2, 17, 32, 3), "cnn_tm256": (4, 2, 6, 256, 2)}.items():
        torch.manual_seed(200 + nconv + T)
        C, W = inner * H, T_M // 4
        body = []
        for _ in range(nconv):
            body += [R.CausalConv2d(C, C, 3, padding=2, dilation=2, stride=(1, 1), causal=True), nn.ReLU()]
        body += [R.UpsampleFP32((1, 4), torch.float16), R.CausalConv2d(C, H, 1, padding=1, causal=True)]
        keepres = R.KeepRes(*body, output_width=T_M).eval()
        ln2 = nn.LayerNorm(T_M).eval()
        with torch.no_grad():
            ln2.weight.copy_(1.0 + 0.1 * torch.randn(T_M)); ln2.bias.copy_(0.1 * torch.randn(T_M))
        x = torch.randn(2, C, T, W)
        with torch.no_grad():
            y = keepres(x)
            put(name, keepres, x, y)
            out[f"{name}/ln2_weight"], out[f"{name}/ln2_bias"] = ln2.weight.numpy().copy(), ln2.bias.numpy().copy()
            out[f"{name}/scores"] = ln2(y).numpy().copy()
            out[f"{name}/probs"] = torch.softmax(ln2(y), -1).numpy().copy()
        out[f"{name}/meta"] = np.array([H, inner, T, T_M, nconv], dtype=np.int64)

    # -- ResBlock / Residual (not on the causal predictor's path; kept so the drop-in modules stay pinned too)
    torch.manual_seed(11)
    rb = R.ResBlock(6, causal=True).eval()
    x = torch.randn(1, 6, 8, 12)
    with torch.no_grad():
        put("resblock", rb, x, rb(x))

    np.savez_compressed(os.path.join(HERE, "estimator.npz"), **out)
    print("wrote", os.path.join(HERE, "estimator.npz"), len(out), "arrays,",
          os.path.getsize(os.path.join(HERE, "estimator.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
