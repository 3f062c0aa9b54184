#!/usr/bin/env python3
"""Where does the tile kernel stop paying along the sequence?  Plans that give rows t < X to the tile kernel and the rest to
the gather kernels (and the reverse), swept over X, on random and structured maps (opt-1.3b shape)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sea_attention_amd import synthetic
from sea_attention_amd.perlin_attention import ops
from bench import WORKLOADS
dev = torch.device("cuda:0")
w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "opt-1.3b"]
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
dt = torch.bfloat16
torch.manual_seed(0)
q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(dt)
kk = torch.randn((NB, H, T, d), device=dev).to(dt); v = torch.randn((NB, H, T, d), device=dev).to(dt)
keep = ops.keep_table_causal(H, T, T_M, k, device=dev); z_cap = ops.z_capacity(keep.cpu(), H, T, T, T_M, k, True)
out = torch.empty((NB, H, T, d), dtype=dt, device=dev)
TB = (T + 15) // 16


def timeit(csr, plan):
    for _ in range(2):
        ops.sparse_attention(q, kk, v, csr, out=out, path="auto", plan=plan)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        ops.sparse_attention(q, kk, v, csr, out=out, path="auto", plan=plan)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 8


for name, gen in (("random", synthetic.random_probs), ("structured", synthetic.structured_probs)):
    csr, _ = ops.topk_to_csr(gen(NB, H, T, T_M, dev, dt, seed=1), keep, k, target_width=T, z_cap=z_cap)
    t_idx = torch.arange(TB, device=dev).view(1, 1, TB) * 16
    for X in (0, 128, 256, 512, 768, 1024, 1536, 2048, 3072, T):
        lo = ops.make_plan((t_idx < X).expand(NB, H, TB).to(torch.uint8).contiguous())     # tile kernel owns rows < X
        hi = ops.make_plan((t_idx >= X).expand(NB, H, TB).to(torch.uint8).contiguous())    # tile kernel owns rows >= X
        lo[-4:].zero_(); hi[-4:].zero_()                                          # keep the per-block split (no all-tile rule)
        print(json.dumps({"map": name, "X": X, "tile_below_X_ms": round(timeit(csr, lo), 4),
                          "tile_from_X_ms": round(timeit(csr, hi), 4)}), flush=True)
