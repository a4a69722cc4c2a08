#!/bin/bash
mkdir -p gpurun_out/c17
timeout 900 python tools/diag/degrade.py > gpurun_out/c17/degrade.txt 2>&1; cat gpurun_out/c17/degrade.txt | grep -v amdgpu
