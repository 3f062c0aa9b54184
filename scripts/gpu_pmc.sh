#!/bin/bash
# HBM traffic of the hot kernels from PMC counters: separate --pmc passes with kernel-trace only
# (MI355X_MICROARCH.md, HBM / rocprofv3 sections).  Results under gpurun_out/pmc_<tag>/.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$N" -o pmc -- \
    python3 scripts/kernel_path.py --iters 3 ${KP_EXTRA:-} > "$OUT/$N.log" 2>&1
  echo "pmc $N exit=$?"; tail -1 "$OUT/$N.log"
done
find "$OUT" -name "*counter_collection.csv" | head
