#!/bin/bash
# round-3 GPU call 1: kernel times of the insert / clamp / exp variants, cycle accounting of the new build, GPU test suite
mkdir -p gpurun_out/c1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for v in base ins med3 fexp; do
  DBW_HIP_LIB=tools/variants/$v.so timeout 300 python tools/diag/kernel_times.py 0 10 2>/dev/null | tail -1 >> gpurun_out/c1/kernel_times.txt
done
timeout 300 python tools/diag/kernel_times.py 0 10 2>/dev/null | tail -1 >> gpurun_out/c1/kernel_times.txt
DBW_HIP_LIB=tools/variants/base.so timeout 300 python tools/diag/kernel_times.py 0 10 2>/dev/null | tail -1 >> gpurun_out/c1/kernel_times.txt
DBW_HIP_LIB=tools/variants/fprof.so timeout 300 python tools/fwd_cycles.py 0 > gpurun_out/c1/fwd_cycles.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest.txt 2>&1
tail -3 gpurun_out/c1/pytest.txt
cat gpurun_out/c1/kernel_times.txt
tail -4 gpurun_out/c1/fwd_cycles.txt
