#!/bin/bash
# rocprofv3 kernel stats (+ optionally the three PMC passes) of bench.py at the per-GPU shapes of BASELINE cfg 4 / cfg 5
# (ONE sequence per GPU): opt-2.7b x1, llama-13b x1.   usage: scripts/gpu_profile_shapes.sh <tag> [pmc]
set -u
TAG=${1:-r05}; PMC=${2:-}
for WL in llama-13b opt-2.7b; do
  S=$(echo $WL | tr -d '.-')
  BENCH_EXTRA="--workload $WL --batch 1" BENCH_STEPS=20 bash scripts/gpu_profile.sh ${TAG}_${S}_x1 || exit 1
  if [ -n "$PMC" ]; then
    BENCH_EXTRA="--workload $WL --batch 1" bash scripts/gpu_pmc.sh ${TAG}_${S}_x1 || exit 1
  fi
done
