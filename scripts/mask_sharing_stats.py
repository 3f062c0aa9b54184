#!/usr/bin/env python3
"""How much do neighbouring query rows share their kept keys?  (drives the tiled attention kernel's design)

For a FlatCSR of one (n) item: per (head, R-row block) the number of 16-key tiles that hold at least one kept key,
against the entries in them.  `density` = entries / (tiles x R x 16) is what an MFMA tile path computes usefully;
`share` = entries / (tiles x 16) is how many of the R rows use a staged key on average.
Maps: the layer's own (random-init predictor, as bench.py times it), softmax(randn), and the structured map."""
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

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "opt-1.3b"
w = WORKLOADS[wl]
H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
dtype = torch.bfloat16


def stats(csr, name):
    dense = ops.flat_csr_to_dense(csr, T, H)[0] > 0                       # (H, T, T) bool, item 0
    out = {"map": name, "nnz": int(dense.sum().item())}
    for R in (16, 32, 64):
        for KT in (16, 32):
            blk = dense.view(H, T // R, R, T // KT, KT)
            tiles = blk.any(-1).any(2)                                    # (H, T/R, T/KT)
            nt = int(tiles.sum().item())
            out[f"R{R}_K{KT}"] = {"tiles": nt, "density": round(out["nnz"] / (nt * R * KT), 4),
                                  "share": round(out["nnz"] / (nt * KT), 3),
                                  "key_rows_staged_vs_gathered": round(nt * KT / out["nnz"], 4)}
    print(json.dumps(out), flush=True)


keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
z_cap = ops.z_capacity(keep.cpu(), H, T, T, T_M, k, True)

# 1. the layer's own map (random-init predictor, seed 42: what bench.py times)
S.seed(42)
pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=w["nbf"], causal=True,
                           k_flatten=True, k_flatten_dim='causal_batch', context_output_method='mix')
layer = PerlinSelfAttention(_Cfg(H * d, H, T), pc).to(dev).to(dtype).eval()
for m in layer.modules():
    if hasattr(m, 'benchmarking'):
        m.benchmarking = True
layer.attention.assume_not_padded = True
torch.manual_seed(42)
q = (torch.randn((1, H, T, d), device=dev) * d ** -0.5).to(dtype)
kk = torch.randn((1, H, T, d), device=dev).to(dtype)
v = torch.randn((1, H, T, d), device=dev).to(dtype)
fp_min = torch.finfo(torch.float16).min / 2
mask = ((torch.arange(T, device=dev).view(1, T) > torch.arange(T, device=dev).view(T, 1)) * fp_min)
mask = mask.view(1, 1, T, T).to(dtype)
with torch.no_grad():
    out = layer(None, None, None, query_layer=q, key_layer=kk, value_layer=v, attention_mask=mask)
stats(out.partial_attention_mask, "layer(random-init predictor)")
pm = out.estimated_attention_probs_m.float()
print(json.dumps({"layer_probs_row_to_row_corr": round(float(torch.corrcoef(
    torch.stack([pm[0, 0, 2000], pm[0, 0, 2001]]))[0, 1]), 4)}), flush=True)
del out, layer

# 2. softmax(randn)
c, _ = ops.topk_to_csr(synthetic.random_probs(1, H, T, T_M, dev, dtype, seed=1), keep, k, target_width=T, z_cap=z_cap)
stats(c, "softmax(randn)")
# 3. structured
c, _ = ops.topk_to_csr(synthetic.structured_probs(1, H, T, T_M, dev, dtype, seed=1), keep, k, target_width=T, z_cap=z_cap)
stats(c, "structured")
