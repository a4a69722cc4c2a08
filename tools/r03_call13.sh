#!/bin/bash
mkdir -p gpurun_out/c13
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -x -q > gpurun_out/c13/pytest.txt 2>&1
tail -30 gpurun_out/c13/pytest.txt
