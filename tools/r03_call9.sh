#!/bin/bash
mkdir -p gpurun_out/c9
timeout 600 python tools/diag/ab_kernels.py 0 0:0 4096:0 0:1 > gpurun_out/c9/abk.txt 2>&1; tail -6 gpurun_out/c9/abk.txt
