#!/usr/bin/env python3
"""Upper bound of what overlapping the estimator (MFMA / vector bound) with the sparse attention launch (L2-gather bound) on two
HIP streams could give: the layer's estimator graph and its attention launch, which normally run back to back, launched
CONCURRENTLY on two streams (the attention reads the selection the graph is rewriting with the same values -- a benign race for
timing purposes) against the sequential step."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as B
wl, nb = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("opt-1.3b", 8)
dev = torch.device("cuda:0")
lb = B.LayerBench(wl, nb, "bf16", dev)
for _ in range(4): lb.forward()
assert lb.capture("gather")
for _ in range(5): lb.step()
torch.cuda.synchronize()
def timed(fn, n=20):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return round(best, 4)
seq = timed(lambda: lb.step())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
# PRIVATE copies of everything the attention launch reads or writes: the graph on the other stream rewrites the originals
# (a first version let the launch read the row pointers while the row scan was rewriting them: out-of-range reads)
from sea_attention_amd.perlin_attention import ops
q_, k_, v_, csr0 = lb.rec["a"][:4]
rest = lb.rec["a"][4:]
kw0 = lb.rec["kw"]
torch.cuda.synchronize()
cl = lambda t: t.clone() if torch.is_tensor(t) else t
csr = ops.FlatCSR(csr0.crow.clone(), csr0._col.clone(), csr0.head_off.clone(), csr0.H, csr0.T_src, bits=csr0.bits.clone(), row_nnz=cl(csr0.row_nnz))
pend = lb._pending
kw = {k2: (cl(v2) if k2 in ("row_scale", "avg", "mix") else v2) for k2, v2 in kw0.items()}
if torch.is_tensor(kw.get("out")): kw["out"] = torch.empty_strided(kw0["out"].shape, kw0["out"].stride(), dtype=kw0["out"].dtype, device=dev)
torch.cuda.synchronize()
def attn_private():
    if pend is not None: csr._pending = pend
    lb._real_attn(q_, k_, v_, csr, *rest, **kw)
def conc():
    with torch.cuda.stream(s1):
        lb.graph.replay()
    with torch.cuda.stream(s2):
        attn_private()
def only_graph():
    lb.graph.replay()
def only_attn():
    attn_private()
res = {"workload": f"{wl} x{nb}", "sequential_ms": seq, "estimator_graph_alone_ms": timed(only_graph), "attention_alone_ms": timed(only_attn),
       "concurrent_two_streams_ms": timed(conc)}
print(json.dumps(res))
