#!/bin/bash
# SQ / TA / cache counters of EVERY kernel of the bench step (one rocprofv3 --pmc pass per counter group, --kernel-trace
# only), tabulated per kernel: gpurun_out/pmc_<tag>/per_kernel.txt.   scripts/gpu_pmc_all.sh <tag> [bench args...]
set -u
TAG=${1:-all}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/g$i" -o pmc -- \
    python3 bench.py --steps 2 --warmup 1 --prewarm 1 --no-cpu-baseline --kernel-iters 0 --no-output-check --no-other-workloads --no-train-step --sparse-kernel ${ATTN_PATH:-auto} "$@" > "$OUT/g$i.log" 2>&1
  echo "group $i exit=$?"
done
OUTDIR="$OUT" python3 - <<'PY'
import csv, glob, collections, os, re
out=os.environ["OUTDIR"]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name=re.sub(r"^void ","",r['Kernel_Name']); name=re.sub(r"^sea::","",name); name=name.split("(")[0][:70]
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+"/per_kernel.txt","w") as fo:
    for k in sorted(agg, key=lambda k: -sum(agg[k].get('GRBM_GUI_ACTIVE',[0]))/max(1,len(agg[k].get('GRBM_GUI_ACTIVE',[0])))):
        fo.write(f"== {k}\n")
        for c,v in sorted(agg[k].items()):
            v=sorted(v); fo.write(f"   {c:34s} n={len(v):3d} median={v[len(v)//2]:.6g}\n")
print(open(out+"/per_kernel.txt").read()[:200])
PY
