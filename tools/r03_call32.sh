#!/bin/bash
mkdir -p gpurun_out/c32
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_model.py -x -q > gpurun_out/c32/pytest.txt 2>&1
tail -4 gpurun_out/c32/pytest.txt
for i in 1 2; do
for v in product norow; do
  if [ $v = product ]; then unset DBW_HIP_LIB; else export DBW_HIP_LIB=tools/variants/$v.so; fi
  timeout 300 python tools/diag/ab_kernels.py 0 0:0 2>/dev/null | tail -1 | sed "s/^/$v /" | cut -c1-200
  timeout 300 python tools/diag/ab_step.py 0 0:0 2>/dev/null | tail -1 | sed "s/^/$v /"
done; done
unset DBW_HIP_LIB
for e in 800 1600; do timeout 300 python tools/diag/ab_step.py $e 0:0 2>/dev/null | tail -1 | sed "s/^/epoch $e /"; done
