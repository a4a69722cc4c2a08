#!/bin/bash
mkdir -p gpurun_out/c35
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_model.py -x -q > gpurun_out/c35/pytest.txt 2>&1
tail -3 gpurun_out/c35/pytest.txt
for i in 1 2; do
  timeout 300 python tools/diag/ab_kernels.py 0 0:0 2>/dev/null | tail -1 | cut -c1-200
  timeout 300 python tools/diag/ab_step.py 0 0:0 2>/dev/null | tail -1
done
timeout 300 python tools/diag/ab_step.py 800 0:0 2>/dev/null | tail -1
