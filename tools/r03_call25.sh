#!/bin/bash
mkdir -p gpurun_out/c25
timeout 600 python tools/diag/ab_kernels.py 0 0:0 8192:0 16384:0 32768:0 0:256 0:512 0:0 4096:0 4096:512 2>&1 | grep -v amdgpu > gpurun_out/c25/abk.txt; tail -9 gpurun_out/c25/abk.txt | cut -c1-120
