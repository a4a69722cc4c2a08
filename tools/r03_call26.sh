#!/bin/bash
mkdir -p gpurun_out/c26
timeout 600 python tools/diag/ab_kernels.py 0 0:0 0:1024 0:512 0:1536 2>&1 | grep -v amdgpu > gpurun_out/c26/abk.txt; tail -4 gpurun_out/c26/abk.txt | cut -c1-130
