#!/usr/bin/env python3
"""KV-cache decoding latency of one SEA layer: prefill T0 tokens, then single-token steps.
Reports ms per decoded token (all sequences of the batch advance together) at OPT-1.3B shape."""
import os, sys, json, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sea_attention_amd as S
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
N, H, d, T0, steps, T_M, k = int(os.environ.get("NB", 8)), 32, 64, int(os.environ.get("T0", 4000)), 32, 256, 64
dev, dt = "cuda:0", torch.bfloat16
class Cfg:
    hidden_size, num_attention_heads, max_position_embeddings = H * d, H, T0 + steps
S.seed(42)
pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                           k_flatten_dim='causal_batch', context_output_method='mix', use_cache=True)
layer = PerlinSelfAttention(Cfg(), pc).to(dev).to(dt).eval()
for m in layer.modules():
    if hasattr(m, 'benchmarking'): m.benchmarking = True
layer.attention.context_layer_dtype = dt
x = torch.randn((N, H, T0 + steps, d), device=dev).to(dt); q = (x.float() * d ** -0.5).to(dt)
fp_min = torch.finfo(torch.float16).min / 2
def mask(T_dst, T_src):
    rows = torch.arange(T_src - T_dst, T_src, device=dev).view(T_dst, 1)
    return ((torch.arange(T_src, device=dev).view(1, T_src) > rows) * fp_min).view(1, 1, T_dst, T_src).expand(N, 1, T_dst, T_src).to(dt)
with torch.no_grad():
    t0 = time.perf_counter()
    out = layer(None, None, None, query_layer=q[:, :, :T0], key_layer=x[:, :, :T0], value_layer=x[:, :, :T0], attention_mask=mask(T0, T0))
    torch.cuda.synchronize(); t_prefill = time.perf_counter() - t0
    st = out.state
    for i in range(4):    # warm-up decode steps
        hi = T0 + i + 1
        st = layer(None, None, None, query_layer=q[:, :, hi - 1:hi], key_layer=x[:, :, :hi], value_layer=x[:, :, :hi], attention_mask=mask(1, hi), last_state=st).state
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(4, steps):
        hi = T0 + i + 1
        st = layer(None, None, None, query_layer=q[:, :, hi - 1:hi], key_layer=x[:, :, :hi], value_layer=x[:, :, :hi], attention_mask=mask(1, hi), last_state=st).state
    torch.cuda.synchronize(); t_dec = (time.perf_counter() - t0) / (steps - 4)
    # the same positions through the graph-replayed session (perlin_attention/decode.py)
    from sea_attention_amd.perlin_attention.decode import DecodeSession
    res_s = {}
    for label, use_graph in (("session_eager", False), ("session_graph", True)):
        sess = DecodeSession(layer.attention, out.state, x[:, :, :T0], x[:, :, :T0], capacity=T0 + steps, use_graph=use_graph)
        for i in range(4):
            hi = T0 + i + 1
            sess.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(4, steps):
            hi = T0 + i + 1
            sess.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi])
        torch.cuda.synchronize()
        res_s[label + "_ms_per_step"] = round((time.perf_counter() - t0) / (steps - 4) * 1e3, 3)
print(json.dumps({**res_s, "batch": N, "prefill_tokens": T0, "prefill_ms": round(t_prefill * 1e3, 2), "decode_ms_per_token_step": round(t_dec * 1e3, 3),
                  "decode_tokens_per_s": round(N / t_dec, 1)}))
