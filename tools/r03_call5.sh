#!/bin/bash
mkdir -p gpurun_out/c5; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/c5/pytest_parity.txt 2>&1
tail -3 gpurun_out/c5/pytest_parity.txt
KT="timeout 300 python tools/diag/kernel_times.py 0 10"
$KT 2>/dev/null | tail -1 >> gpurun_out/c5/kernel_times.txt
DBW_DEBUG_FLAGS=4096 $KT 2>/dev/null | tail -1 >> gpurun_out/c5/kernel_times.txt
cat gpurun_out/c5/kernel_times.txt
DBW_HIP_LIB=tools/variants/fprof.so timeout 300 python tools/fwd_timeline.py 0 > gpurun_out/c5/timeline_cells.txt 2>&1
head -18 gpurun_out/c5/timeline_cells.txt; tail -9 gpurun_out/c5/timeline_cells.txt
for f in 0; do
  DBW_DEBUG_FLAGS=$f DBW_STEPS=6 rocprofv3 --kernel-trace -d gpurun_out/c5/t$f -o p -- python tools/pmc_target.py > gpurun_out/c5/t$f.log 2>&1
  db=$(find gpurun_out/c5/t$f -name "*.db" | head -1)
  python tools/rocprof_summary.py $db gpurun_out/c5/stats_flags$f.txt "flags $f" | head -16
  rm -rf gpurun_out/c5/t$f
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.err
python -c "
import json; d=json.load(open('gpurun_out/c5/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('all_kernels_ms'))"
