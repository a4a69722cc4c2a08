#!/bin/bash
# VALU / SALU wave-instructions per launch of the K = 10 fused forward under the ablation switches of dbw_debug_set_flags:
# what each part of the kernel costs in instructions (49 views, config 2).  usage: r06_pmc_ablate.sh <tag> [variant.so]
O=gpurun_out/r06/$1; mkdir -p $O; export TMPDIR=/tmp
[ -n "$2" ] && export DBW_HIP_LIB=tools/variants/$2.so
for f in 0 8192 16384 32768 65536 131072; do
  DBW_DEBUG_FLAGS=$f DBW_STEP_EVENTS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $O/pmc_$f -o p --output-format csv -- python tools/pmc_target.py > $O/pmc_$f.log 2>&1
  echo "== flags $f"; python tools/diag/pmc_fwd_insts.py $O/pmc_$f
  rm -rf $O/pmc_$f
done
DBW_PMC_EMPTY=1 DBW_STEP_EVENTS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $O/pmc_empty -o p --output-format csv -- python tools/pmc_target.py > $O/pmc_empty.log 2>&1
echo "== all tiles empty"; python tools/diag/pmc_fwd_insts.py $O/pmc_empty; rm -rf $O/pmc_empty
