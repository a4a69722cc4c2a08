#!/bin/bash
mkdir -p gpurun_out/c24
for i in 1 2; do
for v in product nodefer; do
  if [ $v = product ]; then unset DBW_HIP_LIB; else export DBW_HIP_LIB=tools/variants/$v.so; fi
  timeout 300 python tools/diag/ab_kernels.py 0 0:0 2>/dev/null | tail -1 | sed "s/^/$v /" >> gpurun_out/c24/abk.txt
done; done
unset DBW_HIP_LIB
cat gpurun_out/c24/abk.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
