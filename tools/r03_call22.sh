#!/bin/bash
mkdir -p gpurun_out/c22
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_lpips.py -x -q > gpurun_out/c22/pytest.txt 2>&1
tail -25 gpurun_out/c22/pytest.txt
