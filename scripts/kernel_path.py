#!/usr/bin/env python3
"""Kernel-level path only (HIP top-k -> CSR -> fused sparse attention) at a BASELINE workload shape.
Used under rocprofv3 (kernel trace / PMC passes): few launches, no torch model around them."""
import argparse, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="opt-1.3b"); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=5); ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
from sea_attention_amd.perlin_attention import ops
w = WORKLOADS[a.workload]; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.dtype]
dev = "cuda:0"; NB = a.batch
torch.manual_seed(42)
probs = torch.softmax(torch.randn((NB, H, T, T_M), device=dev), -1).to(dt)
q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(dt)
kk = torch.randn((NB, H, T, d), device=dev).to(dt); v = torch.randn((NB, H, T, d), device=dev).to(dt)
rs = torch.sigmoid(torch.randn((NB, H, T), device=dev)); mx = torch.sigmoid(torch.randn((NB, H, T), device=dev))
avg = (v.float().cumsum(-2) / torch.arange(1, T + 1, device=dev).view(1, 1, -1, 1)).to(dt)
keep = ops.keep_table_causal(H, T, T_M, k, device=dev); z_cap = ops.z_capacity(keep.cpu(), H, T, T, T_M, k, True)
ctx = torch.empty((NB, T, H * d), dtype=dt, device=dev)
for _ in range(a.iters):
    c, _ = ops.topk_to_csr(probs, keep, k, target_width=T, z_cap=z_cap)
    ops.sparse_attention(q, kk, v, c, row_scale=rs, avg=avg, mix=mx, out=ctx.view(NB, T, H, d).permute(0, 2, 1, 3))
torch.cuda.synchronize()
Z = int(c.crow[:, -1].sum().item())
print(json.dumps({"nnz": Z, "alg_bytes_attn": ops.sparse_attention_bytes(Z, NB, H, T, d, q.element_size()),
                  "alg_bytes_topk": probs.numel() * probs.element_size() + Z * 4 + NB * T * (H + 2) * 4}))
