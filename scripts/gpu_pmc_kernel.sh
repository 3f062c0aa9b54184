#!/bin/bash
# SQ / cache counters of one kernel of one timing script (one rocprofv3 --pmc pass per counter group).
#   usage: scripts/gpu_pmc_kernel.sh <script.py> <kernel-name-substring> [tag]
set -u
SCRIPT=$1; export KSUB=$2; TAG=${3:-pmc_kernel}
cd /tmp && export TMPDIR=/tmp
export OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/g$i" -o pmc -- python3 "$SCRIPT" > "$OUT/g$i.log" 2>&1
  echo "group $i exit=$?"
done
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ["OUT"]; ks=os.environ["KSUB"]
agg=collections.defaultdict(list)
for f in glob.glob(out+"/g*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if ks in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+"/summary.txt","w") as fh:
    for k,v in sorted(agg.items()):
        line=f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):.4g}"; print(line); fh.write(line+"\n")
PY
