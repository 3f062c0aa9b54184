import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sea_attention_amd.perlin_attention import ops
DEV = "cuda:0"
for dtype in (torch.bfloat16, torch.float16):
  for (N, H, T, k, T_M) in [(1, 12, 300, 32, 64), (2, 12, 130, 128, 96), (1, 12, 200, 64, 128), (1, 12, 150, 32, 384), (1, 32, 70, 64, 512), (1, 6, 90, 16, 256), (1, 3, 40, 8, 36), (1, 12, 77, 16, 192), (2, 32, 300, 64, 256)]:
    C, W4 = (2 * H + 7) // 8 * 8, T_M // 4
    g = torch.Generator().manual_seed(9)
    y = ops.to_c8(torch.relu(torch.randn((N, C, T, W4), generator=g)).to(dtype).to(DEV))
    cw = (torch.randn((H, C), generator=g) * C ** -0.5).to(dtype).to(DEV)
    cb = (torch.randn(H, generator=g) * 0.1).to(dtype).to(DEV)
    lw = (torch.rand(T_M, generator=g) + 0.5).to(dtype).to(DEV)
    lb = (torch.randn(T_M, generator=g) * 0.1).to(dtype).to(DEV)
    keep = ops.keep_table_causal(H, T, T_M, k, device=DEV)
    p0, s0 = ops.predictor_tail(y, cw, cb, lw, lb, up=4, T_m=T_M, want_scores=True)
    p0b, _ = ops.predictor_tail(y, cw, cb, lw, lb, up=4, T_m=T_M, want_scores=False)
    p1, s1, sel = ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T, want_scores=True)
    p2, _, sel2 = ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T)
    torch.cuda.synchronize()
    dp = (p0 != p1); ds = (s0 != s1)
    print(dtype, (N, H, T, k, T_M), "p mismatches", int(dp.sum()), "s mismatches", int(ds.sum()), "p0 vs p0b", int((p0 != p0b).sum()), "p1 vs p2", int((p1 != p2).sum()),
          "max ulp-ish", float((p0.float() - p1.float()).abs().max() / p0.float().abs().max()))
    if dp.any():
        idx = dp.nonzero()[:6]
        print(idx.tolist(), p0[dp][:6].tolist(), p1[dp][:6].tolist())
    c0, _ = ops.topk_to_csr(p1, keep, k, target_width=T)
    print("  bits equal given p1:", bool(torch.equal(c0.bits, sel[0])), bool(torch.equal(c0.head_off, sel[2])))
