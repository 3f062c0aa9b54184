#!/bin/bash
# round 4, first GPU call: this round's starting numbers + three ablations (same box, alternating builds)
set -u
mkdir -p gpurun_out
L=$PWD/build_abl
(
echo "== tail+select: map store ablation"
for r in 1 2; do
  python scripts/time_tail_select.py
  SEA_HIP_LIB=$L/libsea_nomap.so python scripts/time_tail_select.py
done
echo "== fused attention, ctx bf16 vs fp32, store policies"
for r in 1 2; do
  CTX=bf16 python scripts/time_attn_fused.py
  CTX=fp32 python scripts/time_attn_fused.py
  CTX=fp32 SEA_HIP_LIB=$L/libsea_outnt.so python scripts/time_attn_fused.py
  CTX=fp32 SEA_HIP_LIB=$L/libsea_outsc1.so python scripts/time_attn_fused.py
done
) > gpurun_out/r04a_abl.log 2>&1
echo "== bench" >> gpurun_out/r04a_abl.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --decode-steps 32 > gpurun_out/r04a_bench.log 2>&1
tail -c 3000 gpurun_out/r04a_abl.log
