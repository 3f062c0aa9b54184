#!/usr/bin/env python3
"""Host cost of one DecodeSession position: the stage launch (the one launch whose arguments change) and the graph replay,
timed as CPU enqueue time (no synchronisation inside the loops), next to the whole step."""
import os, sys, json, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sea_attention_amd as S
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
from sea_attention_amd.perlin_attention.decode import DecodeSession
N, H, d, T0, steps, T_M, k = int(os.environ.get("NB", 1)), 32, 64, 2000, 1200, 256, 64
dev, dt = "cuda:0", torch.bfloat16
class Cfg:
    hidden_size, num_attention_heads, max_position_embeddings = H * d, H, T0 + steps + 8
S.seed(42)
pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                           k_flatten_dim='causal_batch', context_output_method='mix', use_cache=True)
layer = PerlinSelfAttention(Cfg(), pc).to(dev).to(dt).eval()
for m in layer.modules():
    if hasattr(m, 'benchmarking'): m.benchmarking = True
layer.attention.context_layer_dtype = dt
x = torch.randn((N, H, T0 + steps, d), device=dev).to(dt); q = (x.float() * d ** -0.5).to(dt)
fp_min = torch.finfo(torch.float16).min / 2
rows = torch.arange(T0, device=dev).view(T0, 1)
mask = ((torch.arange(T0, device=dev).view(1, T0) > rows) * fp_min).view(1, 1, T0, T0).expand(N, 1, T0, T0).to(dt)
res = {}
with torch.no_grad():
    out = layer(None, None, None, query_layer=q[:, :, :T0], key_layer=x[:, :, :T0], value_layer=x[:, :, :T0], attention_mask=mask)
    sess = DecodeSession(layer.attention, out.state, x[:, :, :T0], x[:, :, :T0], capacity=T0 + steps + 8, use_graph=True)
    rows_q = [q[:, :, T0 + i:T0 + i + 1] for i in range(steps)]; rows_x = [x[:, :, T0 + i:T0 + i + 1] for i in range(steps)]
    for i in range(20): sess.step(rows_q[i], rows_x[i], rows_x[i])
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for i in range(n): sess._stage(rows_q[i], rows_x[i], rows_x[i])
    res["stage_enqueue_us"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    torch.cuda.synchronize()
    sess.ctr32.copy_(torch.tensor([T0 + 20, T0 + 21, T0 + 21], dtype=torch.int32))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): sess.graph.replay()
    res["replay_enqueue_us"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): sess.step(rows_q[400 + i], rows_x[400 + i], rows_x[400 + i])
    res["step_enqueue_us"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    torch.cuda.synchronize()
    res["step_wall_us"] = round((time.perf_counter() - t0) / n * 1e6, 2)
print(json.dumps(res))
