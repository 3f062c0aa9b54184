#!/usr/bin/env python3
"""The fused attention launch (interpolation + attention) back to back on the layer's own selection at the headline shape;
for A/B and ablation builds selected with SEA_HIP_LIB."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sea_attention_amd as S
from bench import WORKLOADS, _Cfg
from sea_attention_amd.perlin_attention import ops, PerlinAttentionConfig, PerlinSelfAttention
from sea_attention_amd.perlin_attention import attention as A
w = WORKLOADS[os.environ.get("WL", "opt-1.3b")]; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
NB, dev, dt = int(os.environ.get("NB", 8)), "cuda:0", torch.bfloat16
S.seed(42)
pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=w["nbf"], causal=True, k_flatten=True,
                           k_flatten_dim='causal_batch', context_output_method='mix')
layer = PerlinSelfAttention(_Cfg(H * d, H, T), pc).to(dev).to(dt).eval()
for m in layer.modules():
    if hasattr(m, 'benchmarking'): m.benchmarking = True
layer.attention.context_layer_dtype = torch.float32 if os.environ.get('CTX', 'bf16') == 'fp32' else dt
layer.attention.assume_not_padded = True
S.seed(7)
x = torch.randn((NB, H, T, d), device=dev)
q, kk, v = (x * d ** -0.5).to(dt), torch.randn_like(x).to(dt), torch.randn_like(x).to(dt)
fp_min = torch.finfo(torch.float16).min / 2
mask = ((torch.arange(T, device=dev).view(1, T) > torch.arange(T, device=dev).view(T, 1)) * fp_min).view(1, 1, T, T).to(dt).expand(NB, 1, T, T)
seen = {}
real = A.ops.sparse_attention
def spy(q_, k_, v_, csr, **kw):
    seen.update(q=q_, k=k_, v=v_, csr=csr, kw=dict(kw), pending=csr._pending)
    return real(q_, k_, v_, csr, **kw)
A.ops.sparse_attention = spy
with torch.no_grad():
    layer(None, None, None, query_layer=q, key_layer=kk, value_layer=v, attention_mask=mask)
A.ops.sparse_attention = real
csr, kw = seen["csr"], seen["kw"]
def run():
    csr._pending = seen["pending"]
    return real(seen["q"], seen["k"], seen["v"], csr, **kw)
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ts.append(round(e0.elapsed_time(e1) / 10, 4))
print(json.dumps({"ctx": os.environ.get("CTX", "bf16"), "lib": os.path.basename(os.environ.get("SEA_HIP_LIB", "libsea_hip.so")), "fused_attention_ms": ts, "nnz": int(csr.crow[:, -1].sum())}))
