#!/bin/bash
# rocprofv3 kernel trace of the graph-replayed decode session (scripts/time_decode.py): which launches a position is made of
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_decode
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 scripts/time_decode.py > "$OUT/run.log" 2>&1
echo "rc=$?"; tail -1 "$OUT/run.log"
python3 - <<'PY'
import csv, os, collections
f = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/prof_decode/trace_kernel_trace.csv")
rows = list(csv.DictReader(open(f)))
# the last 20 replayed positions: take kernels by start time from the tail of the trace
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-13 * 28 - 3 * 28:]
acc = collections.defaultdict(list)
for r in tail:
    acc[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/28:8.2f} us/pos  n={len(v):4d}  avg={sum(v)/len(v):7.2f}  {k}")
PY
