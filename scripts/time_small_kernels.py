#!/usr/bin/env python3
"""Fixed cost of the estimator kernels: the same launches over 1 ... 4096 rows of one sequence, timed inside a HIP graph of 50
launches (so that what is measured is the device time per launch, not the host's enqueue rate)."""
import json, math, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sea_attention_amd.perlin_attention import ops
dev = "cuda:0"; dt = torch.bfloat16
torch.manual_seed(0)
C, W = 64, 64


def graph_time(fn, reps=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / reps * 1e3


res = {}
wt = (torch.randn((C, C, 3, 3), device=dev) * 0.04).to(dt); b = torch.zeros(C, device=dev, dtype=dt)
for T in (1, 8, 64, 512, 4096):
    x = ops.to_c8(torch.relu(torch.randn((1, C, T, W), device=dev)).to(dt))
    res[f"conv_T{T}"] = round(graph_time(lambda: ops.causal_conv_c8(x, wt, b, 3, 2, 2)), 2)
e = torch.empty(1, device=dev)
res["empty_kernel(fill_1)"] = round(graph_time(lambda: e.fill_(1.0)), 2)
print(json.dumps(res))
