#!/bin/bash
mkdir -p gpurun_out/c12
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c12/pytest.txt 2>&1
tail -4 gpurun_out/c12/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/c12/bench.json 2> gpurun_out/c12/bench.err
python -c "
import json; d=json.load(open('gpurun_out/c12/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('all_kernels_ms')); print(d['phases'])"
tail -3 gpurun_out/c12/bench.err
