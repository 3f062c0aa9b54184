#!/usr/bin/env python3
import os, sys, json, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sea_attention_amd import _lib
from sea_attention_amd.perlin_attention import ops
N, H, T, T_M = 8, 32, 4096, 256; C, W4 = 2 * H, T_M // 4; dev = "cuda:0"; dt = torch.bfloat16
torch.manual_seed(0)
y = torch.relu(torch.randn((N, C, T, W4), device=dev)).to(dt)
if os.environ.get("NHWC", "1") == "1": y = y.contiguous(memory_format=torch.channels_last)
cw = (torch.randn((H, C), device=dev) * C ** -0.5).to(dt); cb = torch.zeros(H, device=dev, dtype=dt)
lw = torch.ones(T_M, device=dev, dtype=dt); lb = torch.zeros(T_M, device=dev, dtype=dt)
def run(): return ops.predictor_tail(y, cw, cb, lw, lb, up=4, T_m=T_M)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"tail_us": round(e0.elapsed_time(e1) / 10 * 1e3, 1)}))
if os.environ.get("STAMPS"):
    lib = _lib.load(); buf = (ctypes.c_ulonglong * 8)()
    lib.sea_debug_tail_stamps(buf); run(); torch.cuda.synchronize(); lib.sea_debug_tail_stamps(buf)
    tot = sum(buf[i] for i in range(3)); print({n: round(buf[i] / tot, 3) for i, n in enumerate(["stage", "gemm", "phaseB"])}, "cycles/row", tot / (N * T))
