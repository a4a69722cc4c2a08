#!/bin/bash
timeout 600 python tools/diag/ab_kernels.py 0 0:0 1:0 2:0 16:0 128:0 3:0 19:0 147:0 2>&1 | grep -v amdgpu | tail -8 | cut -c1-200
