#!/bin/bash
mkdir -p gpurun_out/c14
timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -s > gpurun_out/c14/pytest.txt 2>&1
grep -E "config 1|passed|failed|Error|off" gpurun_out/c14/pytest.txt | tail -20
