#!/bin/bash
mkdir -p gpurun_out/c10
timeout 600 python tools/diag/ab_kernels.py 0 0:0 8192:0 16384:0 32768:0 49152:0 4096:0 > gpurun_out/c10/abk.txt 2>&1; tail -12 gpurun_out/c10/abk.txt
