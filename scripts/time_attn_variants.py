#!/usr/bin/env python3
"""The gather attention kernel back to back on the layer's own CSR, a softmax(randn) CSR and the structured map at a BASELINE
shape (default OPT-1.3B x 8, bf16), with the statistics that explain its time: entries per head / per XCD and the lane
efficiency of the 8-rows-per-wave lockstep walk for natural and length-sorted row assignments.  (Round 3 used it with
experiment kernels behind flag bits -- cache policies of the gathers, a software-pipelined walk, 32 / 64 / 128 rows per
block; results in DESIGN.md section 9 -- the surviving form is the product kernel.)"""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sea_attention_amd as S
from bench import LayerBench
from sea_attention_amd import synthetic
from sea_attention_amd.perlin_attention import ops
wl = sys.argv[1] if len(sys.argv) > 1 else "opt-1.3b"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
variants = [0]
dev = torch.device("cuda:0")
lb = LayerBench(wl, NB, "bf16", dev)
lb.layer.attention.sparse_kernel = "gather"
out = lb.forward()
w = lb.w; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
q, kk, v = lb.q, lb.k, lb.v
keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
maps = {"layer": out.partial_attention_mask}
probs = torch.softmax(torch.randn((NB, H, T, T_M), device=dev), -1).to(torch.bfloat16)
maps["random"], _ = ops.topk_to_csr(probs, keep, k, target_width=T); del probs
sp = synthetic.structured_probs(NB, H, T, T_M, dev, torch.bfloat16, seed=1)
maps["structured"], _ = ops.topk_to_csr(sp, keep, k, target_width=T); del sp
for name, csr in maps.items():                            # entries per head and per XCD under the kernel's (n, h) -> XCD map
    ho = csr.head_off.long()
    per_nh = (ho[..., 1:] - ho[..., :-1]).sum(1)                                       # (N, H)
    xcd_now = torch.zeros(8, dtype=torch.long, device=dev); xcd_rot = torch.zeros(8, dtype=torch.long, device=dev)
    for n in range(NB):
        for h in range(H):
            xcd_now[(n * H + h) % 8] += per_nh[n, h]
            xcd_rot[(n * H + ((h - n) % H)) % 8] += per_nh[n, h]
    cnt = (ho[..., 1:] - ho[..., :-1]).permute(0, 2, 1).contiguous().float()           # (N, H, T) entries per (row, head)
    tot = cnt.sum().item()
    def eff(c, g=8):                                                                   # useful lane-steps / issued lane-steps with 8 rows per wave
        c4 = ((c + 3) // 4 * 4).clamp_min(0)                                           # the walk advances 4 entries at a time
        return tot / (c4.view(NB, H, -1, g).amax(-1).sum().item() * g)
    effs = {"as_is": eff(cnt)}
    for blk in (32, 64, 128, 256):
        effs[f"sorted_in_{blk}"] = eff(cnt.view(NB, H, -1, blk).sort(-1).values.reshape(NB, H, T))
    print(name, "lane efficiency of the 8-rows-per-wave walk:", {k_: round(v_, 3) for k_, v_ in effs.items()}, flush=True)
    ph = per_nh.sum(0).float()
    print(name, "per-head nnz / mean:", [round(x, 2) for x in (ph / ph.mean()).tolist()], flush=True)
    print(name, "per-XCD load / mean now:", [round(x, 3) for x in (xcd_now.float() / xcd_now.float().mean()).tolist()],
          "rotated:", [round(x, 3) for x in (xcd_rot.float() / xcd_rot.float().mean()).tolist()], flush=True)
rs = torch.sigmoid(torch.randn((NB, H, T), device=dev)); mx = torch.sigmoid(torch.randn((NB, H, T), device=dev))
avg = ops.cumavg(v)
ctx = torch.empty((NB, T, H * d), dtype=torch.bfloat16, device=dev)
ref = torch.empty_like(ctx)
res = {}
for name, csr in maps.items():
    res[name] = {}
    for var in variants:
        def run(dst): ops.sparse_attention(q, kk, v, csr, row_scale=rs, avg=avg, mix=mx, out=dst.view(NB, T, H, d).permute(0, 2, 1, 3), path="gather")
        for _ in range(3): run(ctx)
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): run(ctx)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 8)
        if var == variants[0]:
            ref.copy_(ctx)
        res[name][var] = {"ms": round(min(ts), 4), "ms_max": round(max(ts), 4), "equal_ref": bool(torch.equal(ctx, ref))}
    print(name, json.dumps(res[name]), flush=True)
print(json.dumps(res))
