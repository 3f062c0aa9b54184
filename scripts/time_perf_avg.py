import os, sys, json, math, torch
sys.path.insert(0, '/root/repo')
from sea_attention_amd.perlin_attention import ops
from sea_attention_amd.perlin_attention.performer import FastAttention
N, H, T, D = 8, 32, 4096, 64; dev = "cuda:0"; dt = torch.bfloat16
torch.manual_seed(0)
fa = FastAttention(D, nb_features=int(D * math.log(D) / 8), causal=True, generalized_attention=True).to(dev)
q = (torch.randn((N, H, T, D), device=dev) * D ** -0.5).to(dt); k = torch.randn((N, H, T, D), device=dev).to(dt); v = torch.randn((N, H, T, D), device=dev).to(dt)
pos = torch.randn((T, D), device=dev).to(dt)
res = {}
for name, wa in (("plain_us", False), ("with_avg_us", True)):
    run = lambda: ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=wa)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / 10 * 1e3, 1)
print(json.dumps(res))
