#!/bin/bash
mkdir -p gpurun_out/c7; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/c7/pytest_parity.txt 2>&1
tail -3 gpurun_out/c7/pytest_parity.txt
rocprofv3 --kernel-trace -d gpurun_out/c7/t0 -o p --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases > gpurun_out/c7/t0.log 2>&1
csv=$(find gpurun_out/c7/t0 -name "*kernel_trace.csv" | head -1)
python tools/step_sequence.py $csv > gpurun_out/c7/step_sequence.txt 2>&1
python - $csv <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print('%-62s %5d calls  avg %8.1f us' % (k, len(v), sum(v)/len(v)))
PY
rm -rf gpurun_out/c7/t0
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c7/bench.json 2> gpurun_out/c7/bench.err
python -c "
import json; d=json.load(open('gpurun_out/c7/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('all_kernels_ms'))"
cat gpurun_out/c7/step_sequence.txt | head -50
