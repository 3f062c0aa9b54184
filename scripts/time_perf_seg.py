"""Sequence-parallel Performer vs the one-pass kernel at the one-sequence-per-GPU shapes of BASELINE configs 4-5."""
import json, math, sys, torch
sys.path.insert(0, ".")
import sea_attention_amd as S
from sea_attention_amd.perlin_attention import ops
from sea_attention_amd.perlin_attention.ops import predictor as PR
from sea_attention_amd.perlin_attention.performer import FastAttention
dev = "cuda"
res = {}
for name, N, H, T, D in (("opt-2.7b x1", 1, 32, 8192, 80), ("llama-13b x1", 1, 40, 4096, 128), ("llama-13b x2", 2, 40, 4096, 128),
                         ("opt-1.3b x1", 1, 32, 4096, 64), ("opt-1.3b x8", 8, 32, 4096, 64), ("opt-125m x8", 8, 12, 2048, 64)):
    nb = int(D * math.log(D) / 8)
    fa = FastAttention(D, nb_features=nb, causal=True, generalized_attention=True).to(dev)
    q = (torch.randn(N, H, T, D, device=dev) * D ** -0.5).bfloat16(); k = torch.randn(N, H, T, D, device=dev).bfloat16()
    v = torch.randn(N, H, T, D, device=dev).bfloat16(); pos = torch.randn(T, D, device=dev).bfloat16()
    plan = PR.performer_plan(N, H, T, D, nb, torch.bfloat16)
    row = {"plan": plan[0]}
    for label, nseg in (("one_pass_us", 1), ("planned_us", None)):
        for _ in range(3):
            ops.performer_value(q, k, v, pos, fa.projection_matrix, n_segments=nseg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.performer_value(q, k, v, pos, fa.projection_matrix, n_segments=nseg)
        e1.record(); torch.cuda.synchronize()
        row[label] = round(e0.elapsed_time(e1) * 100, 1)
    res[name] = row
print(json.dumps(res))
