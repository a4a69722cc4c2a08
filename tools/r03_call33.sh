#!/bin/bash
timeout 300 python tools/diag/ab_kernels.py 0 0:0 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python tools/diag/ab_step.py 0 0:0 2>/dev/null | tail -1
timeout 300 python tools/diag/ab_kernels.py 800 0:0 2>/dev/null | tail -1 | cut -c1-260
for e in 800 1600; do timeout 300 python tools/diag/ab_step.py $e 0:0 2>/dev/null | tail -1 | sed "s/^/epoch $e /"; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py tests/test_tiled_images.py -x -q 2>&1 | tail -2
