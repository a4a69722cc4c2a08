#!/bin/bash
mkdir -p gpurun_out/c15
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py -x -q > gpurun_out/c15/pytest.txt 2>&1
tail -15 gpurun_out/c15/pytest.txt
( time timeout 900 python bench.py --no-cpu-baseline > gpurun_out/c15/bench.json 2> gpurun_out/c15/bench.err ) 2>&1 | tail -3
tail -5 gpurun_out/c15/bench.err
python -c "
import json; d=json.load(open('gpurun_out/c15/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
for k in ('batch4','batch4_no_reads','sustained'): print(k, d.get(k))
print(d.get('configs'))"
