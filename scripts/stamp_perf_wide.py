#!/usr/bin/env python3
"""Per-phase cycles of the wide-head Performer kernel's chunk walk from a -DSEA_STAMP build (SEA_HIP_LIB): threads 0 (wave 0)
and 448 (wave 7) of every workgroup of the output pass add their s_memtime deltas to eight counters."""
import ctypes, json, math, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sea_attention_amd.perlin_attention import ops
from sea_attention_amd.perlin_attention.performer import FastAttention
from sea_attention_amd import _lib
N, H, T, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (1, 32, 8192, 80)))
dev = "cuda:0"; dt = torch.bfloat16
torch.manual_seed(0)
fa = FastAttention(D, nb_features=int(D * math.log(D) / 8), causal=True, generalized_attention=True).to(dev)
q = (torch.randn((N, H, T, D), device=dev) * D ** -0.5).to(dt); k = torch.randn((N, H, T, D), device=dev).to(dt); v = torch.randn((N, H, T, D), device=dev).to(dt)
pos = torch.randn((T, D), device=dev).to(dt)
run = lambda: ops.performer_value(q, k, v, pos, fa.projection_matrix, want_avg=True)
for _ in range(3): run()
torch.cuda.synchronize()
lib = _lib.load(); buf = (ctypes.c_ulonglong * 16)()
lib.sea_debug_perf_stamps(buf); run(); torch.cuda.synchronize(); lib.sea_debug_perf_stamps(buf)
chunks = N * H * T / 32
nseg = ops.performer_plan(N, H, T, D, fa.projection_matrix.shape[0], dt)[0]
print(json.dumps({"shape": [N, H, T, D], "wave0_cycles_per_chunk": [round(buf[i] / chunks) for i in range(4)],
                  "wave7_cycles_per_chunk": [round(buf[i] / chunks) for i in range(4, 8)],
                  "state_pass_wave0_cycles_per_chunk": [round(buf[10 + i] / max(N * H * (nseg - 1) * math.ceil(T / nseg / 32), 1)) for i in range(4)],
                  "nseg": nseg, "state_pass_cycles_per_workgroup": round(buf[8] / max(N * H * (nseg - 1), 1)),
                  "output_pass_cycles_per_workgroup": round(buf[9] / (N * H * nseg))}))

if hasattr(lib, "sea_debug_perf_wg"):
    wg = (ctypes.c_ulonglong * 2048)(); lib.sea_debug_perf_wg(wg)
    n_wg = N * H * nseg
    st = torch.tensor([wg[2 * i] for i in range(n_wg)], dtype=torch.float64); en = torch.tensor([wg[2 * i + 1] for i in range(n_wg)], dtype=torch.float64)
    t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0        # microseconds (100 MHz)
    dur = en - st
    print(json.dumps({"workgroups": n_wg, "start_us_min_med_max": [round(float(x), 1) for x in (st.min(), st.median(), st.max())],
                      "duration_us_min_med_max": [round(float(x), 1) for x in (dur.min(), dur.median(), dur.max())], "end_us_max": round(float(en.max()), 1),
                      "duration_by_segment_us": [round(float(dur[i * N * H:(i + 1) * N * H].mean()), 1) for i in range(nseg)]}))
    if os.environ.get("DUMP_WG"):
        for sgi in (0, nseg - 2, nseg - 1):
            print("seg", sgi, "start", [round(float(x), 1) for x in st[sgi * N * H:(sgi + 1) * N * H]][:16], "dur", [round(float(x), 1) for x in dur[sgi * N * H:(sgi + 1) * N * H]][:16])
