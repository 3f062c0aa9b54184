#!/usr/bin/env python3
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sea_attention_amd.perlin_attention import ops
N, C, T, W = 8, 64, 4096, 64; dev = "cuda:0"; dt = torch.bfloat16
torch.manual_seed(0)
x = ops.to_c8(torch.relu(torch.randn((N, C, T, W), device=dev)).to(dt))
wt = (torch.randn((C, C, 5, 3), device=dev) * 0.04).to(dt); b = torch.zeros(C, device=dev, dtype=dt)
def run(): return ops.causal_conv_c8(x, wt, b, 3, 2, 2)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y = run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"conv_us": round(ms * 1e3, 1), "TFLOPs": round(2 * N * T * W * C * C * 9 / ms / 1e9, 1), "checksum": float(y.float().abs().mean())}))
