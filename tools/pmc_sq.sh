#!/bin/bash
# SQ counter passes of the bench workload (tools/pmc_target.py), summarised per kernel.  Usage (GPU box): tools/pmc_sq.sh OUTDIR [EPOCH]
# Counters are collected on their own (--kernel-trace only), one rocprofv3 run per group.
OUT=${1:-gpurun_out/pmc_sq}; export DBW_EPOCH=${2:-0}
# (counter collection runs one kernel at a time: the step's streams then wait for each other through events, not through polled words)
export DBW_STEP_EVENTS=1
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
# calibration of the byte counters on 4 B/lane plane traffic of known size (tools/ubench/plane_rw.hip)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/calib_$c -o p --output-format csv -- tools/ubench/plane_rw > $OUT/calib_$c.log 2>&1
done
python tools/pmc_sq_summary.py $OUT
