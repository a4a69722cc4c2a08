#!/bin/bash
# round-3 GPU call: SQ / memory counters of the epoch-800 step (binned uv backward), one rocprofv3 --pmc pass per group
OUT=gpurun_out/pmc800; export DBW_EPOCH=${1:-800}
mkdir -p $OUT; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT" \
           "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p --output-format csv -- python tools/pmc_target.py > $OUT/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $OUT/g$i.log)"
done
python tools/pmc_sq_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A1 "render_bwd_uv\|texbin_reduce\|render_bwd_hard" $OUT/summary.txt
rm -rf $OUT/g*/
