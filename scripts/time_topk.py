#!/usr/bin/env python3
"""Time the top-k -> CSR kernels in isolation (HIP events), for A/B builds selected with SEA_HIP_LIB."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import WORKLOADS
from sea_attention_amd import _lib
from sea_attention_amd.perlin_attention import ops
from sea_attention_amd.perlin_attention.ops import flat_csr as F
w = WORKLOADS[os.environ.get("WL", "opt-1.3b")]; H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
NB = int(os.environ.get("NB", 8)); dev = "cuda:0"
torch.manual_seed(42)
probs = torch.softmax(torch.randn((NB, H, T, T_M), device=dev), -1).to(torch.bfloat16)
if os.environ.get('LAYOUT') == 'nthm':
    probs = probs.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)   # logical (N,H,T,T_M), row-contiguous storage
keep = ops.keep_table_causal(H, T, T_M, k, device=dev); z_cap = ops.z_capacity(keep.cpu(), H, T, T, T_M, k, True)
lib = _lib.load(); st = _lib.stream_ptr(); P = F._p
W = (H * T_M + 31) // 32
bits = torch.empty((NB, T, W), dtype=torch.int32, device=dev); row_nnz = torch.empty((NB, T), dtype=torch.int32, device=dev)
head_off = torch.empty((NB, T, H + 1), dtype=torch.int32, device=dev); crow = torch.empty((NB, T + 1), dtype=torch.int32, device=dev)
col = torch.empty((NB, z_cap), dtype=torch.int32, device=dev)
def sel(): _lib.check(lib.sea_topk_select(P(probs), 2, NB, H, T, T_M, *probs.stride()[:3], P(keep), 0, T, 1, k, P(bits), None, P(row_nnz), P(head_off), st), "sel")
def scan(): _lib.check(lib.sea_csr_row_scan(P(row_nnz), NB, T, P(crow), 4, st), "scan")
def emit(): _lib.check(lib.sea_csr_emit(P(bits), P(crow), P(head_off), NB, H, T, T_M, T, 1, k, P(col), 4, col.stride(0), z_cap, None, st), "emit")
res = {}
for name, fn in [("select", sel), ("scan", scan), ("emit", emit)]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / 10 * 1e3, 1)
print(json.dumps({"lib": os.path.basename(os.environ.get("SEA_HIP_LIB", "libsea_hip.so")), "us": res, "nnz": int(crow[:, -1].sum())}))
if os.environ.get("STAMPS"):
    import ctypes
    buf = (ctypes.c_ulonglong * 16)()
    lib.sea_debug_stamps(buf)
    sel(); torch.cuda.synchronize()
    lib.sea_debug_stamps(buf)
    names = ["load+keys", "minmax", "radix passes", "select flags", "bits+widths+heads"]
    tot = sum(buf[i] for i in range(5))
    print(json.dumps({names[i]: round(buf[i] / tot, 3) for i in range(5)}), "cycles/row", tot / (NB * T))
