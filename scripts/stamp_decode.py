#!/usr/bin/env python3
"""Where a decode position's fused CNN + tail + selection launch spends its time: a -DSEA_STAMP build of the library
(s_memtime stamps of thread 0 of every workgroup, summed per phase) under the graph-replayed DecodeSession.
`--build` here (hipcc cross-compiles), then on the GPU box:  NB=1 python scripts/stamp_decode.py"""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "sea-attention_amd", "build", "libsea_hip_stamp.so")
if "--build" in sys.argv:
    from sea_attention_amd import _build
    print(_build.build_library(extra_flags=("-DSEA_STAMP",), out=LIB))
    sys.exit(0)
if os.environ.get("SEA_HIP_LIB") != LIB:                  # the library path is read at import: run the measurement as a child
    sys.exit(subprocess.run([sys.executable, __file__], env=dict(os.environ, SEA_HIP_LIB=LIB)).returncode)
import torch
import sea_attention_amd as S
from sea_attention_amd import _lib
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
from sea_attention_amd.perlin_attention.decode import DecodeSession
N, H, d, T0, steps, T_M, k = int(os.environ.get("NB", 8)), 32, 64, 4000, 36, 256, 64
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
rows = torch.arange(T0, device=dev).view(T0, 1)
mask = ((torch.arange(T0, device=dev).view(1, T0) > rows) * fp_min).view(1, 1, T0, T0).expand(N, 1, T0, T0).to(dt)
lib = _lib.load(); buf = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    out = layer(None, None, None, query_layer=q[:, :, :T0], key_layer=x[:, :, :T0], value_layer=x[:, :, :T0], attention_mask=mask)
    sess = DecodeSession(layer.attention, out.state, x[:, :, :T0], x[:, :, :T0], capacity=T0 + steps, use_graph=True)
    for i in range(4):
        hi = T0 + i + 1
        sess.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi])
    torch.cuda.synchronize(); lib.sea_debug_stamps(buf)
    for i in range(4, steps):
        hi = T0 + i + 1
        sess.step(q[:, :, hi - 1:hi], x[:, :, hi - 1:hi], x[:, :, hi - 1:hi])
    torch.cuda.synchronize(); lib.sea_debug_stamps(buf)
names = {5: "start -> conv1 operands + weights in LDS", 10: "conv1 row (MFMA + store)", 11: "conv2 row", 12: "ring copy", 8: "z tile", 9: "head loop", 1: "minmax", 2: "hist+bin",
         3: "select flags", 4: "bits+widths", 13: "emit"}
per = (steps - 4) * N                                       # workgroup runs in the stamped window
print(json.dumps({"batch": N, "us_per_workgroup (100 MHz ticks / 100)": {names[i]: round(buf[i] / per / 100, 2) for i in names},
                  "sum_us": round(sum(buf[i] for i in names) / per / 100, 2)}))
