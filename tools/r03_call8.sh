#!/bin/bash
mkdir -p gpurun_out/c8; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/c8/pytest_parity.txt 2>&1
tail -3 gpurun_out/c8/pytest_parity.txt
timeout 600 python tools/diag/ab_step.py 0 0:0 4096:0 0:1 > gpurun_out/c8/ab.txt 2>&1; tail -3 gpurun_out/c8/ab.txt
for v in 0 1; do
DBW_RENDER_VARIANT=$v DBW_STEPS=8 rocprofv3 --kernel-trace -d gpurun_out/c8/t$v -o p --output-format csv -- python tools/pmc_target.py > gpurun_out/c8/t$v.log 2>&1
csv=$(find gpurun_out/c8/t$v -name "*kernel_trace.csv" | head -1)
python - $csv <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print('%-62s %5d calls  avg %8.1f us' % (k, len(v), sum(v)/len(v)))
PY
rm -rf gpurun_out/c8/t$v
done
