"""Performer (with the cumulative-average output) at a BASELINE shape for forced segment counts."""
import json, math, sys, torch
sys.path.insert(0, ".")
import sea_attention_amd as S
from sea_attention_amd.perlin_attention import ops
from sea_attention_amd.perlin_attention.performer import FastAttention
dev = "cuda"
N, H, T, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (8, 32, 4096, 64)))
nb = int(D * math.log(D) / 8)
fa = FastAttention(D, nb_features=nb, causal=True, generalized_attention=True).to(dev)
q = (torch.randn(N, H, T, D, device=dev) * D ** -0.5).bfloat16(); k = torch.randn(N, H, T, D, device=dev).bfloat16()
v = torch.randn(N, H, T, D, device=dev).bfloat16(); pos = torch.randn(T, D, device=dev).bfloat16()
res = {}
ref = None
for nseg in (1, 2, 3, 4):
    for _ in range(3):
        out = ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=True, n_segments=nseg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=True, n_segments=nseg)
    e1.record(); torch.cuda.synchronize()
    res[f"nseg{nseg}_us"] = round(e0.elapsed_time(e1) * 100, 1)
    if ref is None: ref = out
    res[f"nseg{nseg}_maxdiff"] = float((out[0].float() - ref[0].float()).abs().max())
print(json.dumps(res))
