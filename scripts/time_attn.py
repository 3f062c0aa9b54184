#!/usr/bin/env python3
"""Time the fused sparse-attention kernel alone (HIP events) for A/B variants (SEA_ATTN_VARIANT / SEA_HIP_LIB)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import WORKLOADS
from sea_attention_amd.perlin_attention import ops
w = WORKLOADS[os.environ.get("WL", "opt-1.3b")]; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
NB = int(os.environ.get("NB", 8)); dev = "cuda:0"; dt = torch.bfloat16
torch.manual_seed(42)
probs = torch.softmax(torch.randn((NB, H, T, T_M), device=dev), -1).to(dt)
q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(dt); kk = torch.randn((NB, H, T, d), device=dev).to(dt); v = torch.randn((NB, H, T, d), device=dev).to(dt)
rs = torch.sigmoid(torch.randn((NB, H, T), device=dev)); mx = torch.sigmoid(torch.randn((NB, H, T), device=dev))
avg = ops.cumavg(v)
keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
csr, _ = ops.topk_to_csr(probs, keep, k, target_width=T)
ctx = torch.empty((NB, T, H * d), dtype=dt, device=dev)
def run(): ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, out=ctx.view(NB, T, H, d).permute(0, 2, 1, 3))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
Z = int(csr.crow[:, -1].sum()); b = ops.sparse_attention_bytes(Z, NB, H, T, d, 2)
print(json.dumps({"variant": os.environ.get("SEA_ATTN_VARIANT", "default"), "ms": round(ms, 4), "GBs": round(b / ms / 1e6, 1), "checksum": float(ctx.float().abs().mean())}))
