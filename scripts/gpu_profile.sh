#!/bin/bash
# rocprofv3 kernel trace of the bench command (N=1 GPU; without the extra kernel-only leg and the lone-item output check, so that the CSV row of the
# attention kernel averages in-layer launches only and can be compared with roofline.avg_launch_ms).  Output under gpurun_out/prof_<tag>/; copy the
# *_kernel_stats.csv you want judged into profiles/.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- \
  python3 bench.py --steps ${BENCH_STEPS:-10} --warmup 3 --no-cpu-baseline --kernel-iters 0 --no-output-check --no-other-workloads --no-train-step --decode-steps 0 --repeats 0 --sparse-kernel ${ATTN_PATH:-gather} ${BENCH_EXTRA:-} > "$OUT/bench.log" 2>&1
echo "rocprof exit=$?"
tail -2 "$OUT/bench.log"
find "$OUT" -name "*stats*" | head
