#!/bin/bash
# SQ counter passes of the bench workload (tools/pmc_target.py), summarised per kernel.  Usage (GPU box): tools/pmc_sq.sh OUTDIR [EPOCH]
# Counters are collected on their own (--kernel-trace only), one rocprofv3 run per group.
OUT=${1:-gpurun_out/pmc_sq}; export DBW_EPOCH=${2:-0}
mkdir -p $OUT; export TMPDIR=/tmp
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
B="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU"
C="FETCH_SIZE"
D="WRITE_SIZE"
i=0
for grp in "$A" "$B" "$C" "$D"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p --output-format csv -- python tools/pmc_target.py > $OUT/g$i.log 2>&1
done
python tools/pmc_sq_summary.py $OUT
