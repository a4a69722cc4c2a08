#!/bin/bash
mkdir -p gpurun_out/c4
DBW_HIP_LIB=tools/variants/fprof.so timeout 300 python tools/fwd_timeline.py 0 > gpurun_out/c4/timeline_cells.txt 2>&1
DBW_DEBUG_FLAGS=4096 DBW_HIP_LIB=tools/variants/fprof.so timeout 300 python tools/fwd_timeline.py 0 > gpurun_out/c4/timeline_legacy.txt 2>&1
cat gpurun_out/c4/timeline_cells.txt
