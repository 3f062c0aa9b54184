#!/usr/bin/env python3
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sea_attention_amd.perlin_attention import ops
N, H, T, D = 8, 32, 4096, 64; dev = "cuda:0"
torch.manual_seed(0)
v = torch.randn((N, T, H, D), device=dev).bfloat16().permute(0, 2, 1, 3)     # the layer's (N,T,H*d) projection viewed per head
for _ in range(3): ops.cumavg(v)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): o = ops.cumavg(v)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"cumavg_us": round(ms * 1e3, 1), "GBs": round(3 * v.numel() * 2 / ms / 1e6, 1), "checksum": float(o.float().abs().mean())}))
