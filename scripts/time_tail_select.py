#!/usr/bin/env python3
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sea_attention_amd.perlin_attention import ops
N, H, T, T_M, k = 8, 32, 4096, 256, 64; dev = "cuda:0"; dt = torch.bfloat16
torch.manual_seed(0)
y = ops.to_c8(torch.relu(torch.randn((N, 2 * H, T, 64), device=dev)).to(dt))
cw = (torch.randn((H, 2 * H), device=dev) * 0.125).to(dt); cb = torch.zeros(H, device=dev, dtype=dt)
lw = torch.ones(T_M, device=dev, dtype=dt); lb = torch.zeros(T_M, device=dev, dtype=dt)
keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
def sep():
    p, _ = ops.predictor_tail(y, cw, cb, lw, lb, up=4, T_m=T_M)
    return ops.topk_to_csr(p, keep, k, target_width=T, z_cap=10_100_000)
def fused():
    p, _, sel = ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T)
    return ops.csr_from_selection(*sel, H, T_M, T, k, True, 10_100_000)
def tail_only():
    return ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T)
def tail_lazy():
    return ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T, lazy_probs=True)
res = {}
for name, fn in (("separate_us", sep), ("fused_us", fused), ("tail_select_launch_us", tail_only), ("tail_select_no_map_us", tail_lazy)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / 10 * 1e3, 1)
print(json.dumps(res))
