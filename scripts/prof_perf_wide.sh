#!/bin/bash
# rocprofv3 per-kernel durations of scripts/cmp_perf_wide.py's shapes under one library build ($1 = tag, SEA_HIP_LIB = build)
set -u
TAG=${1:-new}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_perfw_$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 scripts/cmp_perf_wide.py /tmp/x_$TAG.pt > "$OUT/run.log" 2>&1
echo "rocprof exit=$?"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
grep -i "performer" "$f" | cut -c1-260
