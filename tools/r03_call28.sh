#!/bin/bash
mkdir -p gpurun_out/c28
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c28/pytest.txt 2>&1
tail -4 gpurun_out/c28/pytest.txt
timeout 600 python tools/diag/ab_step.py 0 0:0 2>&1 | tail -1
