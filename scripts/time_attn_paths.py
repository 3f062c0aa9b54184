#!/usr/bin/env python3
"""Fused sparse attention: gather path vs MFMA tile path (and the tile path's knobs) on three kinds of map --
the layer's own (random-init predictor), softmax(randn), structured -- at a BASELINE workload shape."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sea_attention_amd as S
from sea_attention_amd import synthetic
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention, ops
from bench import WORKLOADS, _Cfg

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="opt-1.3b")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--maps", default="layer,random,structured")
ap.add_argument("--variants", default="gather,tile:1:0,plan")
ap.add_argument("--T", type=int, default=0, help="override the workload's sequence length (e.g. 512: only the dense low rows)")
a = ap.parse_args()
dev = torch.device("cuda:0")
w = WORKLOADS[a.workload]
H, d, T, T_M, k = w["H"], w["d"], a.T or w["T"], w["T_M"], w["k"]
NB = a.batch
dtype = torch.bfloat16
torch.manual_seed(42)
q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(dtype)
kk = torch.randn((NB, H, T, d), device=dev).to(dtype)
v = torch.randn((NB, H, T, d), device=dev).to(dtype)
rs = torch.sigmoid(torch.randn((NB, H, T), device=dev))
mx = torch.sigmoid(torch.randn((NB, H, T), device=dev))
avg = (v.float().cumsum(-2) / torch.arange(1, T + 1, device=dev).view(1, 1, -1, 1)).to(dtype)
keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
z_cap = ops.z_capacity(keep.cpu(), H, T, T, T_M, k, True)
ctx = torch.empty((NB, T, H * d), dtype=dtype, device=dev)
ctx_ref = torch.empty_like(ctx)


def layer_csr():
    S.seed(42)
    pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=w["nbf"], causal=True,
                               k_flatten=True, k_flatten_dim='causal_batch', context_output_method='mix')
    layer = PerlinSelfAttention(_Cfg(H * d, H, T), pc).to(dev).to(dtype).eval()
    for m in layer.modules():
        if hasattr(m, 'benchmarking'):
            m.benchmarking = True
    layer.attention.assume_not_padded = True
    fp_min = torch.finfo(torch.float16).min / 2
    mask = ((torch.arange(T, device=dev).view(1, T) > torch.arange(T, device=dev).view(T, 1)) * fp_min)
    mask = mask.view(1, 1, T, T).to(dtype).expand(NB, 1, T, T)
    with torch.no_grad():
        out = layer(None, None, None, query_layer=q, key_layer=kk, value_layer=v, attention_mask=mask)
    return out.partial_attention_mask


def csr_of(name):
    if name == "layer":
        return layer_csr()
    gen = synthetic.random_probs if name == "random" else synthetic.structured_probs
    c, _ = ops.topk_to_csr(gen(NB, H, T, T_M, dev, dtype, seed=1), keep, k, target_width=T, z_cap=z_cap)
    return c


for name in a.maps.split(","):
    csr = csr_of(name)
    Z = int(csr.crow[:, -1].sum().item())
    alg = ops.sparse_attention_bytes(Z, NB, H, T, d, 2)
    ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, out=ctx_ref.view(NB, T, H, d).permute(0, 2, 1, 3), path="gather")
    for var in a.variants.split(","):
        parts = var.split(":")
        kw = dict(path=parts[0])
        if parts[0] == "plan":                                # per-block dispatch, optional entries-per-tile threshold
            pl = ops.attention_plan(csr, T_M, entries_per_tile=float(parts[1]) if len(parts) > 1 else 0.0)
            kw = dict(path="auto", plan=pl)
            share = float(ops.plan_blocks(pl, NB, H, T).float().mean().item())
        if parts[0] == "tile":
            kw.update(row_tiles=int(parts[1]), key_window=int(parts[2]))
        try:
            for _ in range(2):
                ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, out=ctx.view(NB, T, H, d).permute(0, 2, 1, 3), **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, out=ctx.view(NB, T, H, d).permute(0, 2, 1, 3), **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            err = (ctx.float() - ctx_ref.float()).abs().max().item()
            rec = {"map": name, "variant": var, "ms": round(ms, 4), "alg_TBs": round(alg / ms / 1e9, 2),
                   "nnz": Z, "max_abs_diff_vs_gather": round(err, 5)}
            if parts[0] == "plan":
                rec["blocks_on_tile_kernel"] = round(share, 4)
            print(json.dumps(rec), flush=True)
        except Exception as ex:  # noqa: BLE001
            print(json.dumps({"map": name, "variant": var, "error": str(ex)[:200]}), flush=True)
