#!/bin/bash
mkdir -p gpurun_out/c18
timeout 900 python tools/diag/cell_stats.py 25 1080 1920 50 16 512 > gpurun_out/c18/c5.txt 2>&1; grep -v amdgpu gpurun_out/c18/c5.txt | tail -6
timeout 900 python tools/diag/cell_stats.py 8 576 768 20 16 256 > gpurun_out/c18/c4.txt 2>&1; grep -v amdgpu gpurun_out/c18/c4.txt | tail -6
