#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel in one HIP source, from hipcc's own resource remarks (no GPU needed).

    python scripts/kernel_resources.py sea_attn.hip [substring ...] [-- -DFLAG ...]

prints one line per kernel whose (mangled) name contains every substring."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sea-attention_amd", "csrc")


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    src, subs = argv[0], argv[1:]
    path = src if os.path.exists(src) else os.path.join(CSRC, src)
    cmd = ["/opt/rocm/bin/hipcc", "-c", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Rpass-analysis=kernel-resource-usage",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, path, "-o", "/dev/null", *extra]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for ln in err.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = rows.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+?):\s+(\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for name, r in rows.items():
        if all(s in name for s in subs):
            print(f"{name:110s} vgpr={r.get('VGPRs', -1):3d} agpr={r.get('AGPRs', 0):3d} spill={r.get('VGPRs Spill', 0):3d} "
                  f"scratch={r.get('ScratchSize [bytes/lane]', 0):4d} occ={r.get('Occupancy [waves/SIMD]', -1)} "
                  f"lds={r.get('LDS Size [bytes/block]', 0)}")


if __name__ == "__main__":
    main()
