#!/usr/bin/env python3
"""csr_emit on the LAYER's own selection (random-init predictor, what bench.py times) vs on a softmax(randn) map, back to back:
is the in-layer 190 us (vs 143 us in scripts/time_topk.py) the data or the neighbourhood of the other launches?"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sea_attention_amd as S
from sea_attention_amd import synthetic
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention, ops
from bench import WORKLOADS, _Cfg
dev = torch.device("cuda:0"); dtype = torch.bfloat16
w = WORKLOADS["opt-1.3b"]; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]; NB = 8
S.seed(42)
pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=w["nbf"], causal=True, k_flatten=True, k_flatten_dim='causal_batch', context_output_method='mix')
layer = PerlinSelfAttention(_Cfg(H * d, H, T), pc).to(dev).to(dtype).eval()
for m in layer.modules():
    if hasattr(m, 'benchmarking'): m.benchmarking = True
layer.attention.context_layer_dtype = dtype; layer.attention.assume_not_padded = True
torch.manual_seed(42)
q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(dtype); kk = torch.randn((NB, H, T, d), device=dev).to(dtype); v = torch.randn((NB, H, T, d), device=dev).to(dtype)
fp_min = torch.finfo(torch.float16).min / 2
mask = ((torch.arange(T, device=dev).view(1, T) > torch.arange(T, device=dev).view(T, 1)) * fp_min).view(1, 1, T, T).to(dtype).expand(NB, 1, T, T).contiguous()
with torch.no_grad():
    out = layer(None, None, None, query_layer=q, key_layer=kk, value_layer=v, attention_mask=mask)
csr_l = out.partial_attention_mask
keep = ops.keep_table_causal(H, T, T_M, k, device=dev); z_cap = ops.z_capacity(keep.cpu(), H, T, T, T_M, k, True)
csr_r, _ = ops.topk_to_csr(synthetic.random_probs(NB, H, T, T_M, dev, dtype, seed=1), keep, k, target_width=T, z_cap=z_cap)
res = {}
for name, c in (("layer_map", csr_l), ("random_map", csr_r)):
    f = lambda: ops.csr_from_selection(c.bits, c.row_nnz, c.head_off, H, T_M, T, k, True, z_cap)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    res[name + "_scan_plus_emit_us"] = round(e0.elapsed_time(e1) * 100, 1)
    kept = torch.tensor([bin(int(x) & 0xffffffff).count("1") for x in c.bits[0, T - 1].tolist()]).sum().item()
    res[name + "_kept_pixels_last_row"] = int(kept); res[name + "_nnz"] = int(c.crow[:, -1].sum().item())
print(json.dumps(res))
