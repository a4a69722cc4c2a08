#!/bin/bash
mkdir -p gpurun_out/c27
timeout 1200 python -m pytest tests/test_tiled_images.py tests/test_gpu_model.py tests/test_gpu_parallel.py -x -q > gpurun_out/c27/pytest.txt 2>&1
tail -5 gpurun_out/c27/pytest.txt
timeout 600 python tools/diag/ab_kernels.py 0 0:0 0:0 2>&1 | grep -v amdgpu | tail -2 | cut -c1-200
timeout 600 python tools/diag/ab_step.py 0 0:0 2>&1 | tail -1
