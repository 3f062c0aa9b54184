#!/usr/bin/env python3
"""Per-block dispatch threshold (entries per staged tile at which a 16-row block goes to the MFMA tile kernel): time `auto`
over a sweep of thresholds, next to gather-only and tile-only, on the layer's own selection, softmax(randn) and the
structured map of a BASELINE shape.  Round 3: the gather kernels got ~15 % faster (rows dealt by length), the cut moves up."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import LayerBench
from sea_attention_amd import synthetic
from sea_attention_amd.perlin_attention import ops
wl = sys.argv[1] if len(sys.argv) > 1 else "opt-1.3b"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
lb = LayerBench(wl, NB, "bf16", dev)
lb.layer.attention.sparse_kernel = "gather"
out0 = lb.forward()
w = lb.w; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
q, kk, v = lb.q, lb.k, lb.v
keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
maps = {"layer": out0.partial_attention_mask}
maps["random"], _ = ops.topk_to_csr(synthetic.random_probs(NB, H, T, T_M, dev, torch.bfloat16, seed=1), keep, k, target_width=T)
maps["structured"], _ = ops.topk_to_csr(synthetic.structured_probs(NB, H, T, T_M, dev, torch.bfloat16, seed=1), keep, k, target_width=T)
rs = torch.sigmoid(torch.randn((NB, H, T), device=dev)); mx = torch.sigmoid(torch.randn((NB, H, T), device=dev))
avg = ops.cumavg(v)
ctx = torch.empty((NB, T, H * d), dtype=torch.bfloat16, device=dev)


def timeit(csr, path, thr=None):
    def run():
        plan = ops.attention_plan(csr, T_M, entries_per_tile=thr) if thr is not None else None
        ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, out=ctx.view(NB, T, H, d).permute(0, 2, 1, 3), path=path, plan=plan)
        return plan
    for _ in range(3):
        plan = run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8)
    share = None
    if plan is not None:
        nb = NB * H * ((T + 15) // 16)
        share = round(plan[:nb].float().mean().item(), 3)
    return round(best, 4), share


for name, csr in maps.items():
    row = {"map": name, "gather": timeit(csr, "gather")[0], "tile": timeit(csr, "tile")[0]}
    for thr in (24, 30, 36, 42, 48, 56, 64, 80):
        row[f"auto@{thr}"] = timeit(csr, "auto", float(thr))
    print(json.dumps(row), flush=True)
