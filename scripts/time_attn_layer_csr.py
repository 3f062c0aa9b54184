#!/usr/bin/env python3
"""Attention kernel back-to-back on (a) the CSR the layer's own estimator selects and (b) a softmax(randn) map's CSR:
separates "the pattern" from "the context" in the in-layer vs kernel-path difference.  Also per-(row, head) nnz spread."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sea_attention_amd as S
from bench import WORKLOADS, _Cfg
from sea_attention_amd.perlin_attention import ops, PerlinAttentionConfig, PerlinSelfAttention
w = WORKLOADS["opt-1.3b"]; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
NB, dev, dt = 8, "cuda:0", torch.bfloat16
S.seed(42)
pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                           k_flatten_dim='causal_batch', context_output_method='mix')
layer = PerlinSelfAttention(_Cfg(H * d, H, T), pc).to(dev).to(dt).eval()
for m in layer.modules():
    if hasattr(m, 'benchmarking'): m.benchmarking = True
layer.attention.context_layer_dtype = dt
layer.attention.assume_not_padded = True
torch.manual_seed(42)
q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(dt); kk = torch.randn((NB, H, T, d), device=dev).to(dt); v = torch.randn((NB, H, T, d), device=dev).to(dt)
fp_min = torch.finfo(torch.float16).min / 2
mask = ((torch.arange(T, device=dev).view(1, T) > torch.arange(T, device=dev).view(T, 1)) * fp_min).view(1, 1, T, T).to(dt).expand(NB, 1, T, T).contiguous()
with torch.no_grad():
    out = layer(None, None, None, query_layer=q, key_layer=kk, value_layer=v, attention_mask=mask)
csr_layer = out.partial_attention_mask
probs = torch.softmax(torch.randn((NB, H, T, T_M), device=dev), -1).to(dt)
keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
csr_rand, _ = ops.topk_to_csr(probs, keep, k, target_width=T)
rs = torch.sigmoid(torch.randn((NB, H, T), device=dev)); mx = torch.sigmoid(torch.randn((NB, H, T), device=dev))
avg = ops.cumavg(v)
ctx = torch.empty((NB, T, H * d), dtype=dt, device=dev)
res = {}
for name, csr in (("layer_csr", csr_layer), ("randn_csr", csr_rand), ("layer_csr_again", csr_layer)):
    def run(): ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, out=ctx.view(NB, T, H, d).permute(0, 2, 1, 3))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ho = csr.head_off.view(NB, T, H + 1).long()
    per = (ho[..., 1:] - ho[..., :-1])[:, 1024:].float()               # nnz per (row, head), rows past the dense start
    res[name] = {"ms": round(e0.elapsed_time(e1) / 10, 4), "nnz": int(csr.crow[:, -1].sum()),
                 "per_head_mean": round(per.mean().item(), 1), "per_head_std": round(per.std().item(), 1), "per_head_max": int(per.max().item())}
print(json.dumps(res))
