import os, sys, json, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sea_attention_amd as S
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
from sea_attention_amd.perlin_attention.decode import DecodeSession
N, H, d, T, T_M, k = int(os.environ.get("NB", 8)), 32, 64, 4096, 256, 64
nd = 32; T0 = T - nd - 8
dev, dt = "cuda:0", torch.bfloat16
class Cfg:
    hidden_size, num_attention_heads, max_position_embeddings = H * d, H, T
S.seed(42)
pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                           k_flatten_dim='causal_batch', context_output_method='mix', use_cache=True)
layer = PerlinSelfAttention(Cfg(), pc).to(dev).to(dt).eval()
for m in layer.modules():
    if hasattr(m, 'benchmarking'): m.benchmarking = True
if os.environ.get("CTX16"): layer.attention.context_layer_dtype = dt
x = torch.randn((N, H, T, d), device=dev).to(dt); q = (x.float() * d ** -0.5).to(dt)
fp_min = torch.finfo(torch.float16).min / 2
rows = torch.arange(T0, device=dev).view(T0, 1)
mask = ((torch.arange(T0, device=dev).view(1, T0) > rows) * fp_min).view(1, 1, T0, T0).expand(N, 1, T0, T0).to(dt)
with torch.no_grad():
    out = layer(None, None, None, query_layer=q[:, :, :T0], key_layer=x[:, :, :T0], value_layer=x[:, :, :T0], attention_mask=mask)
    sess = DecodeSession(layer.attention, out.state, x[:, :, :T0], x[:, :, :T0], capacity=int(os.environ.get("CAP", T)), use_graph=True)
    ts = []
    for i in range(4 + nd):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sess.step(q[:, :, T0 + i:T0 + i + 1], x[:, :, T0 + i:T0 + i + 1], x[:, :, T0 + i:T0 + i + 1])
        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 3))
print(ts)
