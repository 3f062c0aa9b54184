import sys, os, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sea_attention_amd as S
from test_gpu_module import make_layer, run, causal_mask, DEV
H, d = int(sys.argv[1]), int(sys.argv[2])
N, T, T_M, k = 1, 512, 64, 16
layer = make_layer(H, d, T_M, k, T)
S.seed(3)
q = torch.randn((N, H, T, d), device=DEV); q2 = q.clone(); q2[:, :, T // 2] = 3e5
mask = causal_mask(N, T, torch.float32)
for mode in (False, True):
    a, ba = run(layer, q * d ** -0.5, q.clone(), q.clone(), mask, mode)
    b, bb = run(layer, q2 * d ** -0.5, q2.clone(), q2.clone(), mask, mode)
    print("mode benchmarking=", mode)
    for name in ba:
        x, y = ba[name], bb[name]
        if not isinstance(x, torch.Tensor) or x.dim() < 2 or x.shape[-2] != T: continue
        dd = (x[..., :T // 2, :].double() - y[..., :T // 2, :].double()).abs()
        rows = dd.sum(-1).reshape(-1, T // 2).sum(0)
        worst = int(rows.argmax())
        print(f"  {name:40s} sum={dd.sum().item():.3e} worst_row={worst} rowsum={rows[worst].item():.3e}")
