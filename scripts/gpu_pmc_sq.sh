#!/bin/bash
# Where the waves of each kernel spend their cycles (SQ counters, one --pmc pass with --kernel-trace only; program directly
# after `--`): parked on s_waitcnt / barriers (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY; _LDS a sub-bucket), issuing
# (SQ_ACTIVE_INST_ANY), and how busy the matrix pipe is (SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES).
#   usage: scripts/gpu_pmc_sq.sh <tag> ; BENCH_EXTRA as for gpu_pmc.sh
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$TAG
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
BENCH_ARGS="--steps 3 --warmup 1 --prewarm 2 --no-cpu-baseline --kernel-iters 0 --no-output-check --no-other-workloads --no-train-step --decode-steps 0 --repeats 0 --sparse-kernel gather ${BENCH_EXTRA:-}"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT" -o pmc -- \
  python3 bench.py $BENCH_ARGS > "$OUT/run.log" 2>&1
echo "pmc exit=$?"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/pmc_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "sea::" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].replace("void ", "").split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':64s} {'parked':>7s} {'stall':>6s} {'(lds)':>6s} {'issue':>6s} {'valu':>6s} {'mfma_busy':>9s}")
for k, c in acc.items():
    m = lambda n: sum(c[n]) / max(len(c[n]), 1)
    wc = m("SQ_WAVE_CYCLES") or 1.0
    print(f"{k:64s} {m('SQ_WAIT_ANY')/wc:7.2f} {m('SQ_WAIT_INST_ANY')/wc:6.2f} {m('SQ_WAIT_INST_LDS')/wc:6.2f} {m('SQ_ACTIVE_INST_ANY')/wc:6.2f} {m('SQ_ACTIVE_INST_VALU')/wc:6.2f} {m('SQ_VALU_MFMA_BUSY_CYCLES')/max(m('SQ_BUSY_CYCLES'),1):9.3f}")
PY
