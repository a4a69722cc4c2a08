#!/bin/bash
# round-3 GPU call 2: per-tile face lists -- parity first, then kernel times with / without them, hard pass on 8x8 tiles, cycle accounting
mkdir -p gpurun_out/c2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/c2/pytest_parity.txt 2>&1
tail -5 gpurun_out/c2/pytest_parity.txt
KT="timeout 300 python tools/diag/kernel_times.py 0 10"
$KT 2>/dev/null | tail -1 >> gpurun_out/c2/kernel_times.txt
DBW_DEBUG_FLAGS=4096 $KT 2>/dev/null | tail -1 >> gpurun_out/c2/kernel_times.txt
DBW_RENDER_VARIANT=1 $KT 2>/dev/null | tail -1 >> gpurun_out/c2/kernel_times.txt
$KT 2>/dev/null | tail -1 >> gpurun_out/c2/kernel_times.txt
DBW_HIP_LIB=tools/variants/fprof.so timeout 300 python tools/fwd_cycles.py 0 > gpurun_out/c2/fwd_cycles.txt 2>&1
cat gpurun_out/c2/kernel_times.txt
tail -4 gpurun_out/c2/fwd_cycles.txt
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py > gpurun_out/c2/pytest_rest.txt 2>&1
tail -5 gpurun_out/c2/pytest_rest.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err
python -c "
import json; d=json.load(open('gpurun_out/c2/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('all_kernels_ms'))"
