#!/usr/bin/env python3
"""A/B: register cap (waves per SIMD) of the fused tail + selection kernel after the constants-table change.
`--build` here, then run on the GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
VARIANTS = {"occ7": [], "occ6": ["-DSEA_TSEL_OCC32=6", "-DSEA_TSEL_OCC16=6"], "occ8_16": ["-DSEA_TSEL_OCC16=8"], "occ5": ["-DSEA_TSEL_OCC32=5"]}
def lib(v): return os.path.join(ROOT, "sea-attention_amd", "build", f"libsea_hip_occ_{v}.so")
if "--build" in sys.argv:
    from sea_attention_amd import _build
    for v, fl in VARIANTS.items():
        print(_build.build_library(extra_flags=tuple(fl) or ("-DSEA_AB_BASE",), out=lib(v)))
elif "--one" in sys.argv:
    import torch
    from sea_attention_amd.perlin_attention import ops
    res = {}
    for name, (N, H, T) in {"opt13b_x8": (8, 32, 4096), "opt125m_32k": (1, 12, 32768), "opt125m_x8": (8, 12, 2048), "opt27b_x1": (1, 32, 8192)}.items():
        torch.manual_seed(0)
        T_M, k, dt = 256, 64, torch.bfloat16
        y = ops.to_c8(torch.relu(torch.randn((N, 2 * H, T, 64), device="cuda")).to(dt))
        cw = (torch.randn((H, 2 * H), device="cuda") * 0.125).to(dt); cb = torch.zeros(H, device="cuda", dtype=dt)
        lw = (torch.rand(T_M, device="cuda") + 0.5).to(dt); lb = (torch.randn(T_M, device="cuda") * 0.1).to(dt)
        keep = ops.keep_table_causal(H, T, T_M, k, device="cuda")
        f = lambda: ops.predictor_tail_select(y, cw, cb, lw, lb, up=4, T_m=T_M, keep=keep, k=k, T_src=T, lazy_probs=True)
        for _ in range(3): out = f()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): out = f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        res[name] = {"us": round(best, 1), "nnz": int(out[2][1].sum())}
    print(json.dumps(res))
else:
    for rnd in range(2):
        for v in VARIANTS:
            env = dict(os.environ, SEA_HIP_LIB=lib(v))
            out = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True)
            print(v, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
