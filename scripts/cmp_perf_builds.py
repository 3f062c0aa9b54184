import os, sys, math, subprocess, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from sea_attention_amd.perlin_attention import ops
    from sea_attention_amd.perlin_attention.performer import FastAttention
    N, H, T, D = 2, 4, 1000, 64; dev = "cuda:0"; dt = torch.bfloat16
    torch.manual_seed(0)
    fa = FastAttention(D, nb_features=int(D * math.log(D) / 8), causal=True, generalized_attention=True).to(dev)
    q = (torch.randn((N, H, T, D), device=dev) * D ** -0.5).to(dt); k = torch.randn((N, H, T, D), device=dev).to(dt); v = torch.randn((N, H, T, D), device=dev).to(dt)
    pos = torch.randn((T, D), device=dev).to(dt)
    out, avg = ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=True)
    torch.save({"out": out.cpu(), "avg": avg.cpu()}, sys.argv[1])
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/p_new.pt"], env=env)
    env["SEA_HIP_LIB"] = ROOT + "/sea-attention_amd/build/libsea_hip_oldperf.so"
    subprocess.check_call([sys.executable, __file__, "/tmp/p_old.pt"], env=env)
    a, b = torch.load("/tmp/p_new.pt"), torch.load("/tmp/p_old.pt")
    for k_ in a: print(k_, "bitwise equal:", torch.equal(a[k_], b[k_]), "max diff", (a[k_].float() - b[k_].float()).abs().max().item())
