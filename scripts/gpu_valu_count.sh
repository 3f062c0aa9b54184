#!/bin/bash
# vector / scalar / LDS instructions per wave of the kernels matching $KERNEL_RE for a python command (one --pmc pass):
#   KERNEL_RE=tail_select scripts/gpu_valu_count.sh scripts/time_tail_select.py
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/valu_count; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT" -o pmc -- python3 "$@" > "$OUT/run.log" 2>&1
echo "exit=$?"
KERNEL_RE="${KERNEL_RE:-.}" OUTDIR="$OUT" python3 - <<'PY'
import csv, glob, collections, os, re
rx=re.compile(os.environ["KERNEL_RE"]); agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.environ["OUTDIR"]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if rx.search(r['Kernel_Name']): agg[r['Kernel_Name'].split('(')[0][-70:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    w=sorted(d['SQ_WAVES'])[len(d['SQ_WAVES'])//2]
    print(k, "waves", int(w), {c.replace('SQ_INSTS_',''): round(sorted(v)[len(v)//2]/w,1) for c,v in d.items() if c!='SQ_WAVES'})
PY
