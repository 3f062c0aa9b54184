#!/bin/bash
# rocprofv3 kernel trace of the graph-replayed decode session with TIMESTAMPS: per launch of one position its duration and the
# gap to the previous launch's end (what a fused launch would remove).   NB=1|8 scripts/prof_decode_gaps.sh
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_decode_nb${NB:-8}
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o trace -- python3 scripts/time_decode.py > "$OUT/run.log" 2>&1
echo "rc=$?"; tail -1 "$OUT/run.log"
OUT="$OUT" python3 - <<'PY'
import csv, os, collections
f = os.path.join(os.environ["OUT"], "trace_kernel_trace.csv")
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# positions of the graph session = the tail of the trace; a position starts at decode_stage_kernel
idx = [i for i, r in enumerate(rows) if "decode_stage_kernel" in r["Kernel_Name"]]
starts = idx[-21:]                      # last 20 complete positions
per = collections.OrderedDict()
tot_busy = tot_span = 0
for a, b in zip(starts[:-1], starts[1:]):
    seq = rows[a:b]
    prev_end = None
    for j, r in enumerate(seq):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        key = (j, r["Kernel_Name"].replace("void ", "").split("(")[0][:70])
        d = per.setdefault(key, [0.0, 0.0, 0])
        d[0] += (e - s) / 1e3
        d[1] += ((s - prev_end) / 1e3) if prev_end is not None else 0.0
        d[2] += 1
        prev_end = e
    tot_busy += sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq) / 1e3
    tot_span += (int(rows[b]["Start_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3
n = len(starts) - 1
print(f"positions {n}: span {tot_span/n:.1f} us per position, kernels busy {tot_busy/n:.1f} us, idle {100*(1-tot_busy/tot_span):.0f} %")
for (j, name), (dur, gap, c) in per.items():
    print(f"{j:2d} dur {dur/c:7.2f} us  gap-before {gap/c:6.2f} us  n={c:3d}  {name}")
PY
