#!/bin/bash
# One call: the round's rocprofv3 evidence.  Kernel stats + the three PMC passes of bench.py at the headline (opt-1.3b x 8) and
# at the per-GPU shapes of BASELINE cfg 4 / cfg 5 (opt-2.7b x 1, llama-13b x 1); kernel stats of the fp32-data protocol (cfg 2).
#   usage: scripts/gpu_profile_round.sh <tag>      (then, here: scripts/pmc_to_traffic.py <tag>_<shape> for each PMC set)
set -u
TAG=${1:-r05}
BENCH_STEPS=20 bash scripts/gpu_profile.sh ${TAG}_opt13b_x8 || exit 1
bash scripts/gpu_pmc.sh ${TAG}_opt13b_x8 || exit 1
for WL in llama-13b opt-2.7b; do
  S=$(echo $WL | tr -d '.-')
  BENCH_EXTRA="--workload $WL --batch 1" BENCH_STEPS=20 bash scripts/gpu_profile.sh ${TAG}_${S}_x1 || exit 1
  BENCH_EXTRA="--workload $WL --batch 1" bash scripts/gpu_pmc.sh ${TAG}_${S}_x1 || exit 1
done
BENCH_EXTRA="--workload opt-125m --batch 8" BENCH_STEPS=20 bash scripts/gpu_profile.sh ${TAG}_opt125m_x8 || exit 1
BENCH_EXTRA="--workload opt-125m --batch 8" bash scripts/gpu_pmc.sh ${TAG}_opt125m_x8 || exit 1
BENCH_EXTRA="--workload opt-125m --batch 1 --seq-len 32768" BENCH_STEPS=20 bash scripts/gpu_profile.sh ${TAG}_long32k || exit 1
BENCH_EXTRA="--workload opt-125m --batch 8 --dtype fp32" BENCH_STEPS=20 bash scripts/gpu_profile.sh ${TAG}_fp32_opt125m_x8 || exit 1
