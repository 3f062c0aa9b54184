#!/usr/bin/env python3
"""What bounds causal_conv_c8_kernel: timing-only builds (WRONG results on purpose) that drop one resource each --
   onetap: one pixel load per (tap row, channel chunk) instead of three column taps (L1 traffic / 3)
   noload: no pixel loads at all (MFMA + LDS weight reads + stores)
   nostore: no output stores
`--build` here, then run on the GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
VARIANTS = {"base": [], "stagger1": ["-DSEA_CONV_STAGGER=1"], "stagger2": ["-DSEA_CONV_STAGGER=2"], "stagger3": ["-DSEA_CONV_STAGGER=3"]}
def lib(v): return os.path.join(ROOT, "sea-attention_amd", "build", f"libsea_hip_convb_{v}.so")
if "--build" in sys.argv:
    from sea_attention_amd import _build
    for v, fl in VARIANTS.items():
        print(_build.build_library(extra_flags=tuple(fl) or ("-DSEA_AB_BASE",), out=lib(v)))
elif "--one" in sys.argv:
    import torch
    from sea_attention_amd.perlin_attention import ops
    res = {}
    for name, (N, C, T) in {"opt13b_x8": (8, 64, 4096), "llama13b_x1": (1, 80, 4096), "opt27b_x1": (1, 64, 8192), "opt125m_x8": (8, 24, 2048)}.items():
        torch.manual_seed(0)
        x = ops.to_c8(torch.relu(torch.randn((N, C, T, 64), device="cuda")).to(torch.bfloat16))
        wt = (torch.randn((C, C, 5, 3), device="cuda") * 0.04).to(torch.bfloat16); b = torch.zeros(C, device="cuda", dtype=torch.bfloat16)
        for _ in range(5): y = ops.causal_conv_c8(x, wt, b, 3, 2, 2)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): y = ops.causal_conv_c8(x, wt, b, 3, 2, 2)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        res[name] = [round(best, 1), float(y.float().abs().sum())]
    print(json.dumps(res))
else:
    for rnd in range(2):
        for v in VARIANTS:
            env = dict(os.environ, SEA_HIP_LIB=lib(v))
            out = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True)
            print(v, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
