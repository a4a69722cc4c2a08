#!/bin/bash
# round-3 GPU call: same-box A/B (kernels alone, whole step) of tools/variants/<name>.so builds against the in-tree build ("new")
#   tools/r03_call_ab.sh <epoch> <variant> ...      (env TESTS=1: the parity suites first)
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
if [ -n "$TESTS" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_tiled_images.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/ab/pytest.log 2>&1
  tail -3 gpurun_out/ab/pytest.log
fi
EP=${1:-0}; shift
VARS="$* new $* new"
for v in $VARS; do
  if [ $v = new ]; then unset DBW_HIP_LIB; else export DBW_HIP_LIB=tools/variants/$v.so; fi
  echo "== $v"; timeout 300 python tools/diag/ab_kernels.py $EP 0:0 2>&1 | grep flags | tail -1
done | tee gpurun_out/ab/ab_kernels.log
for v in $VARS; do
  if [ $v = new ]; then unset DBW_HIP_LIB; else export DBW_HIP_LIB=tools/variants/$v.so; fi
  echo "== $v"; timeout 300 python tools/diag/ab_step.py $EP 0:0 2>&1 | tail -1
done | tee gpurun_out/ab/ab_step.log
