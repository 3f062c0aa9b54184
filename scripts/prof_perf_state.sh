#!/bin/bash
# rocprofv3 kernel stats of the two Performer launches (state pass / output pass) at one-sequence shapes.
set -u
cd /tmp && export TMPDIR=/tmp
for w in "opt-2.7b" "llama-13b"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_state_$w; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --kernel-iters 0 --no-output-check --no-other-workloads --no-train-step --decode-steps 0 --repeats 0 --sparse-kernel gather --workload $w --batch 1 > "$OUT/run.log" 2>&1
  echo "== $w rc=$?"
  python3 - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "performer" in r["Name"]: print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
