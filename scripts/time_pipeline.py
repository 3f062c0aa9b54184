#!/usr/bin/env python3
"""Does the layer step gain from running two half-batches on two HIP streams (the L2-request-bound attention launch of one
half beside the vector-issue / matrix-pipe bound estimator of the other)?  Same layer and inputs as bench.py."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sea_attention_amd as S
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
from bench import WORKLOADS, _Cfg

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="opt-1.3b"); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=10); ap.add_argument("--chunks", default="1,2,4")
a = ap.parse_args()
dev = torch.device("cuda:0"); dtype = torch.bfloat16
w = WORKLOADS[a.workload]; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]; NB = a.batch
S.seed(42)
pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=w["nbf"], causal=True, k_flatten=True,
                           k_flatten_dim='causal_batch', context_output_method='mix')
layer = PerlinSelfAttention(_Cfg(H * d, H, T), pc).to(dev).to(dtype).eval()
for m in layer.modules():
    if hasattr(m, 'benchmarking'): m.benchmarking = True
layer.attention.context_layer_dtype = dtype; layer.attention.assume_not_padded = True
torch.manual_seed(42)
q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(dtype); kk = torch.randn((NB, H, T, d), device=dev).to(dtype)
v = torch.randn((NB, H, T, d), device=dev).to(dtype)
fp_min = torch.finfo(torch.float16).min / 2
mask = ((torch.arange(T, device=dev).view(1, T) > torch.arange(T, device=dev).view(T, 1)) * fp_min).view(1, 1, T, T).to(dtype).expand(NB, 1, T, T).contiguous()
ctx_full = torch.empty((NB, T, H * d), dtype=dtype, device=dev)

def run(nchunk, streams):
    per = NB // nchunk
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for c in range(nchunk):
        sl = slice(c * per, (c + 1) * per)
        st = streams[c % len(streams)] if streams else cur
        with torch.cuda.stream(st), torch.no_grad():
            out = layer(None, None, None, query_layer=q[sl], key_layer=kk[sl], value_layer=v[sl], attention_mask=mask[sl])
            ctx_full[sl].copy_(out.context_layer)
    for s in streams: cur.wait_stream(s)

# staggered form: the estimator of one half starts when the other half ENTERS its attention launch
from sea_attention_amd.perlin_attention import attention as _A
_real = _A.ops.sparse_attention
_hook = {"ev": None}
def _patched(*args, **kw):
    if _hook["ev"] is not None:
        _hook["ev"].record(torch.cuda.current_stream())
    return _real(*args, **kw)

def run_staggered(streams, evs, state):
    cur = torch.cuda.current_stream()
    per = NB // 2
    _A.ops.sparse_attention = _patched
    try:
        for c in range(2):
            sl = slice(c * per, (c + 1) * per)
            st = streams[c]
            if state["last"] is not None:
                st.wait_event(state["last"])            # the previous half's estimator is done (it is in its attention launch)
            else:
                st.wait_stream(cur)
            _hook["ev"] = evs[c]
            with torch.cuda.stream(st), torch.no_grad():
                out = layer(None, None, None, query_layer=q[sl], key_layer=kk[sl], value_layer=v[sl], attention_mask=mask[sl])
                ctx_full[sl].copy_(out.context_layer)
            state["last"] = evs[c]
    finally:
        _A.ops.sparse_attention = _real
        _hook["ev"] = None

res = {}
ref = None
if True:
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    state = {"last": None}
    for _ in range(3): run_staggered(streams, [torch.cuda.Event(), torch.cuda.Event()], state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters): run_staggered(streams, [torch.cuda.Event(), torch.cuda.Event()], state)
    torch.cuda.synchronize()
    res["staggered_2halves_ms"] = round((time.perf_counter() - t0) / a.iters * 1e3, 4)
    state["last"] = None
for nchunk in [int(x) for x in a.chunks.split(",")]:
    for nstream in sorted({1, min(2, nchunk), nchunk}):
        if nchunk == 1 and nstream > 1: continue
        streams = [torch.cuda.Stream() for _ in range(nstream)] if nstream > 1 else []
        for _ in range(3): run(nchunk, streams)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): run(nchunk, streams)
        e1.record(); torch.cuda.synchronize()
        res[f"chunks{nchunk}_streams{nstream}_ms"] = round(e0.elapsed_time(e1) / a.iters, 4)
        if ref is None: ref = ctx_full.clone()
        res[f"chunks{nchunk}_streams{nstream}_equal"] = bool(torch.equal(ref, ctx_full))
print(json.dumps(res))
