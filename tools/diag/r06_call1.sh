#!/bin/bash
# round 6, call 1: GPU suite on the new forward, A/B against HEAD's build, VALU instruction counts of the fg forward
O=gpurun_out/r06/c1; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in base new base new; do
  if [ $v = base ]; then export DBW_HIP_LIB=tools/variants/base.so; else unset DBW_HIP_LIB; fi
  timeout 600 python tools/ablate.py >> $O/ab.log 2>&1
done
unset DBW_HIP_LIB
cat $O/ab.log | grep -v Warning
for v in base new; do
  if [ $v = base ]; then export DBW_HIP_LIB=tools/variants/base.so; else unset DBW_HIP_LIB; fi
  DBW_STEP_EVENTS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $O/pmc_$v -o p --output-format csv -- python tools/pmc_target.py > $O/pmc_$v.log 2>&1
  echo "== $v"; python tools/diag/pmc_fwd_insts.py $O/pmc_$v
  rm -rf $O/pmc_$v
done
