#!/usr/bin/env python3
"""Does the sparse attention launch (L2-gather bound) need all 256 CUs?  The launch on streams created with a CU mask
(hipExtStreamCreateWithCUMask) of all / half / a quarter of the compute units, and the estimator graph likewise.  If the
attention kept most of its speed on a fraction of the CUs, estimator and attention of two half-batches could run side by
side on disjoint CU sets."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as B
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
dev = torch.device("cuda:0")
lb = B.LayerBench("opt-1.3b", 8, "bf16", dev)
for _ in range(4): lb.forward()
assert lb.capture("gather")
for _ in range(5): lb.step()
torch.cuda.synchronize()
from sea_attention_amd.perlin_attention import ops
q_, k_, v_, csr0 = lb.rec["a"][:4]; rest = lb.rec["a"][4:]; kw0 = lb.rec["kw"]
cl = lambda t: t.clone() if torch.is_tensor(t) else t
csr = ops.FlatCSR(csr0.crow.clone(), csr0._col.clone(), csr0.head_off.clone(), csr0.H, csr0.T_src, bits=csr0.bits.clone(), row_nnz=cl(csr0.row_nnz))
pend = lb._pending
kw = {k2: (cl(v2) if k2 in ("row_scale", "avg", "mix") else v2) for k2, v2 in kw0.items()}
if torch.is_tensor(kw.get("out")): kw["out"] = torch.empty_strided(kw0["out"].shape, kw0["out"].stride(), dtype=kw0["out"].dtype, device=dev)
torch.cuda.synchronize()
def attn():
    if pend is not None: csr._pending = pend
    lb._real_attn(q_, k_, v_, csr, *rest, **kw)
def timed(fn, stream, n=10):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(n): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return round(best, 4)
res = {}
masks = {"all": [0xFFFFFFFF] * 8, "half_of_every_word": [0x0000FFFF] * 8, "quarter_of_every_word": [0x000000FF] * 8,
         "first_half_of_words": [0xFFFFFFFF] * 4 + [0] * 4}
for name, m in masks.items():
    try:
        st = masked_stream(m)
        res[name] = {"attention_ms": timed(attn, st), "estimator_graph_ms": timed(lambda: lb.graph.replay(), st)}
    except Exception as e:
        res[name] = f"{type(e).__name__}: {e}"[:120]
    print(name, res[name], flush=True)
print(json.dumps(res))
