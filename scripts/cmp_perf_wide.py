#!/usr/bin/env python3
"""A/B of two builds of the 16-bit Performer kernels (d = 64 / 80 / 128): outputs, cumulative average and a two-call step
continuation compared bit for bit, launch time of both.  The other build is `SEA_OLD_LIB` (default
sea-attention_amd/build/libsea_hip_oldperf.so); each build runs in its own process (`SEA_HIP_LIB`)."""
import json, math, os, subprocess, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
SHAPES = [(1, 32, 8192, 80, torch.bfloat16, 0), (1, 40, 4096, 128, torch.bfloat16, 0), (2, 3, 1500, 80, torch.float16, 0), (1, 2, 777, 128, torch.float16, 0),
          (8, 12, 2048, 64, torch.bfloat16, 0), (2, 4, 1000, 64, torch.float16, 0), (1, 12, 4096, 64, torch.bfloat16, 70), (1, 32, 8192, 80, torch.bfloat16, 70)]
if len(sys.argv) > 1:
    from sea_attention_amd.perlin_attention import ops
    from sea_attention_amd.perlin_attention.performer import FastAttention
    dev = "cuda:0"; res = {}; times = {}
    for N, H, T, D, dt, nbo in SHAPES:
        torch.manual_seed(0)
        fa = FastAttention(D, nb_features=nbo or int(D * math.log(D) / 8), causal=True, generalized_attention=True).to(dev)
        q = (torch.randn((N, H, T, D), device=dev) * D ** -0.5).to(dt); k = torch.randn((N, H, T, D), device=dev).to(dt); v = torch.randn((N, H, T, D), device=dev).to(dt)
        pos = torch.randn((T, D), device=dev).to(dt)
        run = lambda: ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=True)
        for _ in range(3): out, avg = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out, avg = run()
        e1.record(); torch.cuda.synchronize()
        key = f"{N}x{H}x{T}x{D}_{str(dt)[6:]}" + (f"_nb{nbo}" if nbo else "")
        times[key] = round(e0.elapsed_time(e1) / 10 * 1e3, 1)
        res[key + "_out"] = out.cpu(); res[key + "_avg"] = avg.cpu()
    print(json.dumps(times))
    torch.save(res, sys.argv[1])
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/p_new.pt"], env=env)
    env["SEA_HIP_LIB"] = os.environ.get("SEA_OLD_LIB", ROOT + "/sea-attention_amd/build/libsea_hip_oldperf.so")
    subprocess.check_call([sys.executable, __file__, "/tmp/p_old.pt"], env=env)
    a, b = torch.load("/tmp/p_new.pt"), torch.load("/tmp/p_old.pt")
    for k_ in a: print(k_, "bitwise equal:", torch.equal(a[k_], b[k_]), "max diff", (a[k_].float() - b[k_].float()).abs().max().item())
