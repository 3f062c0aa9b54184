import os, sys, json, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_attention_amd.perlin_attention import ops
from sea_attention_amd import _lib
N, H, T, T_M, k = 8, 32, 4096, 256, 64; dev = "cuda:0"; dt = torch.bfloat16
torch.manual_seed(0)
y = ops.to_c8(torch.relu(torch.randn((N, 2 * H, T, 64), device=dev)).to(dt))
cw = (torch.randn((H, 2 * H), device=dev) * 0.125).to(dt); cb = torch.zeros(H, device=dev, dtype=dt)
lw = torch.ones(T_M, device=dev, dtype=dt); lb = torch.zeros(T_M, device=dev, dtype=dt)
keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
lib = _lib.load(); buf = (ctypes.c_ulonglong * 16)()
f = lambda: ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T, lazy_probs=True)
f(); torch.cuda.synchronize(); lib.sea_debug_stamps(buf); f(); torch.cuda.synchronize(); lib.sea_debug_stamps(buf)
names = {8: "z tile", 9: "head loop", 1: "minmax", 2: "hist+bin", 3: "select flags", 4: "bits+widths"}
tot = sum(buf[i] for i in names)
print({names[i]: round(buf[i] / tot, 3) for i in names}, "ticks/row", tot / (N * T))
