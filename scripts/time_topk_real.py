#!/usr/bin/env python3
"""select/emit timings on the probability map the actual (random-init) estimator produces."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sea_attention_amd as S
from bench import WORKLOADS, _Cfg
from sea_attention_amd import _lib
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention, ops
from sea_attention_amd.perlin_attention.ops import flat_csr as F
w = WORKLOADS["opt-1.3b"]; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
NB = int(os.environ.get("NB", 2)); dev = "cuda:0"; dt = torch.bfloat16
S.seed(42)
pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten_dim='causal_batch')
layer = PerlinSelfAttention(_Cfg(H * d, H, T), pc).to(dev).to(dt).eval()
for m in layer.modules():
    if hasattr(m, 'benchmarking'): m.benchmarking = True
layer.attention.assume_not_padded = True
q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(dt); kk = torch.randn((NB, H, T, d), device=dev).to(dt); v = torch.randn((NB, H, T, d), device=dev).to(dt)
fp_min = torch.finfo(torch.float16).min / 2
mask = (((torch.arange(T, device=dev).view(1, T) > torch.arange(T, device=dev).view(T, 1)) * fp_min).view(1, 1, T, T).to(dt)).expand(NB, 1, T, T).contiguous()
with torch.no_grad():
    out = layer(None, None, None, query_layer=q, key_layer=kk, value_layer=v, attention_mask=mask)
probs = out.estimated_attention_probs_m
print("probs", tuple(probs.shape), probs.dtype, "distinct values per (n,t) row (pooled over heads), mean:",
      float(torch.tensor([torch.unique(probs[0, :, t].float()).numel() for t in (100, 1000, 2000, 4000)]).float().mean()))
keep = ops.keep_table_causal(H, T, T_M, k, device=dev); z_cap = ops.z_capacity(keep.cpu(), H, T, T, T_M, k, True)
lib = _lib.load(); st = _lib.stream_ptr(); P = F._p
W = (H * T_M + 31) // 32
bits = torch.empty((NB, T, W), dtype=torch.int32, device=dev); row_nnz = torch.empty((NB, T), dtype=torch.int32, device=dev)
head_off = torch.empty((NB, T, H + 1), dtype=torch.int32, device=dev); crow = torch.empty((NB, T + 1), dtype=torch.int32, device=dev)
col = torch.empty((NB, z_cap), dtype=torch.int32, device=dev)
def sel(p): _lib.check(lib.sea_topk_select(P(p), 2, NB, H, T, T_M, *p.stride()[:3], P(keep), 0, T, 1, k, P(bits), None, P(row_nnz), P(head_off), st), "sel")
def scan(p): _lib.check(lib.sea_csr_row_scan(P(row_nnz), NB, T, P(crow), 4, st), "scan")
def emit(p): _lib.check(lib.sea_csr_emit(P(bits), P(crow), P(head_off), NB, H, T, T_M, T, 1, k, P(col), 4, col.stride(0), z_cap, None, st), "emit")
synth = torch.softmax(torch.randn((NB, H, T, T_M), device=dev), -1).to(dt)
for tag, p in (("real", probs), ("synthetic", synth)):
    res = {}
    for name, fn in [("select", sel), ("scan", scan), ("emit", emit)]:
        for _ in range(2): fn(p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn(p)
        e1.record(); torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / 5 * 1e3, 1)
    hh = (head_off[0, :, 1:] - head_off[0, :, :-1]).float()
    print(tag, json.dumps(res), "nnz/item", int(crow[0, -1]), "max entries of one head in a row", int(hh.max()), "mean", float(hh.mean()))
