#!/usr/bin/env python3
import os, sys, json, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sea_attention_amd.perlin_attention import ops
from sea_attention_amd.perlin_attention.performer import FastAttention
N, H, T, D = 8, 32, 4096, 64; dev = "cuda:0"; dt = torch.bfloat16
torch.manual_seed(0)
fa = FastAttention(D, nb_features=int(D * math.log(D) / 8), causal=True, generalized_attention=True).to(dev)
q = (torch.randn((N, H, T, D), device=dev) * D ** -0.5).to(dt); k = torch.randn((N, H, T, D), device=dev).to(dt); v = torch.randn((N, H, T, D), device=dev).to(dt)
pos = torch.randn((T, D), device=dev).to(dt)
WA = bool(os.environ.get('WANT_AVG'))
def run():
    r = ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=WA)
    return r[0] if WA else r
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
IT = int(os.environ.get('ITERS', 10))
for _ in range(IT): out = run()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"performer_us": round(e0.elapsed_time(e1) / IT * 1e3, 1), "checksum": float(out.float().abs().mean())}))
if os.environ.get("STAMPS"):
    import ctypes
    from sea_attention_amd import _lib
    lib = _lib.load(); buf = (ctypes.c_ulonglong * 16)()
    lib.sea_debug_perf_stamps(buf); run(); torch.cuda.synchronize(); lib.sea_debug_perf_stamps(buf)
    tot = sum(buf[i] for i in range(5)); print({n: round(buf[i] / tot, 3) for i, n in enumerate(["stage", "features", "A+den", "O+S", "ksum"])}, "cycles/chunk", tot / (N * H * T / 64))
