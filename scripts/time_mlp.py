#!/usr/bin/env python3
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sea_attention_amd.perlin_attention import ops
N, H, T, d, T_M = 8, 32, 4096, 64, 256; dev = "cuda:0"; dt = torch.bfloat16
nn = torch.nn; torch.manual_seed(0)
mods = [nn.Linear(3 * d, 2 * d), nn.LayerNorm(2 * d), nn.Linear(2 * d, T_M // 2), nn.LayerNorm(T_M // 4), nn.Linear(2 * d, 2)]
mods = [m.to(dev).to(dt) for m in mods]
x = torch.randn((N, H, T, 3 * d), device=dev).to(dt)
def run(tp=False): return ops.predictor_mlp(x, *mods, want_tpred=tp)
res = {}
for name, tp in (("mlp_us", False), ("mlp_tpred_us", True)):
    for _ in range(3): run(tp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = run(tp)
    e1.record(); torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / 10 * 1e3, 1)
res["checksum"] = float(y[0].float().abs().mean())
print(json.dumps(res))
