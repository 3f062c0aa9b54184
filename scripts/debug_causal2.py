import sys, os, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sea_attention_amd as S
from test_gpu_module import make_layer, run, causal_mask, DEV
H, d = int(sys.argv[1]), int(sys.argv[2])
N, T, T_M, k = 1, 512, 64, 16
layer = make_layer(H, d, T_M, k, T)
cnn = layer.attention.attention_predictor_cnn
S.seed(3)
x = torch.randn((N, 2 * H, T, T_M // 4), device=DEV)
x2 = x.clone(); x2[:, :, T // 2] = 30.0
caps = {}
def hook(name):
    def f(mod, inp, out):
        caps.setdefault(name, []).append(out.detach().clone())
    return f
hs = []
for name, mod in cnn.named_modules():
    if name and not list(mod.children()):
        hs.append(mod.register_forward_hook(hook(name + ":" + type(mod).__name__)))
from sea_attention_amd.perlin_attention import modules
orig = modules.interpolate
def interp(xx, size, mode=None):
    y = orig(xx, size, mode); caps.setdefault("keepres.interpolate", []).append(y.detach().clone()); return y
modules.interpolate = interp
with torch.no_grad():
    cnn(x); cnn(x2)
for name, (a, b) in caps.items():
    dd = (a[..., :T // 2, :].double() - b[..., :T // 2, :].double()).abs()
    rows = dd.sum(-1).reshape(-1, T // 2).sum(0)
    nz = (rows > 0).nonzero().view(-1).tolist()
    print(f"{name:50s} shape={tuple(a.shape)} sum={dd.sum().item():.3e} rows_with_diff={nz[:8]}")
