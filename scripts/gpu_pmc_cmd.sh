#!/bin/bash
# SQ / cache counters of kernels matching $KERNEL_RE for an arbitrary python command (one rocprofv3 --pmc pass per group).
#   KERNEL_RE=sparse_attn_tile TAG=tile1 scripts/gpu_pmc_cmd.sh scripts/time_attn_paths.py --maps layer --variants tile:1:0 --iters 3
set -u
cd /tmp && export TMPDIR=/tmp
TAG=${TAG:-pmc}; KERNEL_RE=${KERNEL_RE:-sparse_attn}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/g$i" -o pmc -- python3 "$@" > "$OUT/g$i.log" 2>&1
  echo "group $i exit=$?"
done
KERNEL_RE="$KERNEL_RE" python3 - <<'PY'
import csv, glob, collections, os, re
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_"+os.environ.get("TAG","pmc")
rx=re.compile(os.environ["KERNEL_RE"])
agg=collections.defaultdict(list)
for f in glob.glob(out+"/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if rx.search(r['Kernel_Name']): agg[r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+"/summary.txt","w") as fo:
    for k,v in sorted(agg.items()):
        line=f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):.5g}"
        print(line); fo.write(line+"\n")
PY
