#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of scripts/gpu_pmc.sh (taken on bench.py's own in-layer launches) into
profiles/<tag>_pmc_traffic.json + profiles/traffic_<workload>_x<batch>.json (and traffic_latest.json for the headline
workload), stamped with the attention kernel's source hash: bench.py's roofline() looks through profiles/traffic_*.json and
reports `roofline.traffic` only from a record whose stamp, entry count, kernel path and context dtype match the launch it timed.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 for 16 B/lane loads (MI355X_MICROARCH.md, HBM section).
usage: pmc_to_traffic.py <tag> [build note]"""
import csv, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import _attn_source_sha
tag = sys.argv[1]; note = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum_TCC_REQ_sum"):
    f = os.path.join(src, sub, "pmc_counter_collection.csv")
    keep = []
    rd = csv.DictReader(open(f))
    for r in rd:
        if r["Kernel_Name"].startswith(("void sea::", "sea::")):
            keep.append(r)
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{sub}.csv"), "w", newline="") as fo:   # the library's kernels only
        w = csv.DictWriter(fo, fieldnames=rd.fieldnames); w.writeheader(); w.writerows(keep)
log = [l for l in open(os.path.join(src, "FETCH_SIZE.log")).read().splitlines() if l.startswith("{")]
bench_line = json.loads(log[-1]) if log else {}
cmd_file = os.path.join(src, "command.txt")                 # written by scripts/gpu_pmc.sh: the command the passes really ran
ab = bench_line.get("attention_path_ab") or {}
wl = (bench_line.get("config") or {}).get("workload", "")
out = {"round": 5, "build": note, "kernel_source_sha256": _attn_source_sha(),
       "command": open(cmd_file).read().strip() if os.path.exists(cmd_file) else "(scripts/gpu_pmc.sh; command file missing)",
       "attention_path": ab.get("chosen") or ab.get("requested"),
       "context_dtype": "torch.float32" if "context_layer fp32" in wl else ("torch.bfloat16" if "context_layer bf16" in wl else None),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the bench command itself (in-layer launches), gfx950 correction applied",
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16 B/lane loads -> hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md, HBM)",
       "workload": bench_line.get("config", {}).get("workload"), "nnz": (bench_line.get("roofline") or {}).get("nnz"), "kernels": {}}
for name, c in acc.items():
    mean = lambda k: sum(c[k]) / len(c[k]) if c[k] else 0.0
    fs, ws, hit, miss, req = mean("FETCH_SIZE"), mean("WRITE_SIZE"), mean("TCC_HIT_sum"), mean("TCC_MISS_sum"), mean("TCC_REQ_sum")
    out["kernels"][name] = {"launches": len(c["FETCH_SIZE"]), "FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws,
                            "hbm_bytes_per_launch_corrected": int((2 * fs + ws) * 1024),
                            "l2_requests_per_launch": int(req), "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
    if "sparse_attn" in name:          # per-block dispatch launches the gather AND the tile kernel: the step's traffic is their sum
        out["sea_sparse_attention_hbm_bytes_per_launch"] = out.get("sea_sparse_attention_hbm_bytes_per_launch", 0) + int((2 * fs + ws) * 1024)
        out["sea_sparse_attention_l2_requests_per_launch"] = out.get("sea_sparse_attention_l2_requests_per_launch", 0) + int(req)
import re
m = re.match(r"([\w.\-]+) SEA attention layer.*?batch (\d+) sequences/GPU", wl or "")
slug = (m.group(1).replace(".", "").replace("-", "") + "_x" + m.group(2)) if m else "unknown"
dsts = [f"{tag}_pmc_traffic.json", f"traffic_{slug}.json"] + (["traffic_latest.json"] if slug == "opt13b_x8" else [])
for dst in dsts:
    json.dump(out, open(os.path.join(ROOT, "profiles", dst), "w"), indent=1)
print("wrote", dsts)
print(json.dumps({k: (v["hbm_bytes_per_launch_corrected"], v["l2_requests_per_launch"], v["l2_hit_rate"]) for k, v in out["kernels"].items()}, indent=1))
