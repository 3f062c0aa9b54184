#!/bin/bash
# HBM traffic of the layer's kernels from PMC counters, on the IN-LAYER launches of the bench command itself: separate
# --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 sections; the program directly after `--`).
# Results under gpurun_out/pmc_<tag>/; scripts/pmc_to_traffic.py <tag> turns them into profiles/<tag>_pmc_traffic.json and
# profiles/traffic_latest.json (stamped with the attention kernel's source hash, which bench.py checks).
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
BENCH_ARGS="--steps 3 --warmup 1 --prewarm 2 --no-cpu-baseline --kernel-iters 0 --no-output-check --no-other-workloads --no-train-step --decode-steps 0 --repeats 0 --sparse-kernel ${ATTN_PATH:-gather} ${BENCH_EXTRA:-}"
echo "rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python3 bench.py $BENCH_ARGS   (one pass per counter set: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; the layer's in-step launches only)" > "$OUT/command.txt"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$N" -o pmc -- \
    python3 bench.py $BENCH_ARGS > "$OUT/$N.log" 2>&1
  echo "pmc $N exit=$?"; tail -c 400 "$OUT/$N.log"; echo
done
find "$OUT" -name "*counter_collection.csv" | head
