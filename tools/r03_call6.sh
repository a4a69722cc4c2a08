#!/bin/bash
mkdir -p gpurun_out/c6; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/c6/pytest_parity.txt 2>&1
tail -3 gpurun_out/c6/pytest_parity.txt
DBW_STEPS=6 rocprofv3 --kernel-trace -d gpurun_out/c6/t0 -o p -- python tools/pmc_target.py > gpurun_out/c6/t0.log 2>&1
db=$(find gpurun_out/c6/t0 -name "*.db" | head -1)
python tools/rocprof_summary.py $db gpurun_out/c6/stats.txt "cells + LPT" | head -22
python tools/step_sequence.py $db > gpurun_out/c6/step_sequence.txt 2>&1
rm -rf gpurun_out/c6/t0
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.err
python -c "
import json; d=json.load(open('gpurun_out/c6/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('all_kernels_ms'))"
bash tools/pmc_sq.sh gpurun_out/c6/pmc 0 > gpurun_out/c6/pmc.txt 2>&1
rm -rf gpurun_out/c6/pmc/g1 gpurun_out/c6/pmc/g2 gpurun_out/c6/pmc/g3 gpurun_out/c6/pmc/g4
grep -A1 "render_fwd_kernel<10" gpurun_out/c6/pmc.txt | head -4
cat gpurun_out/c6/step_sequence.txt | head -50
