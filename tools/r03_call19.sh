#!/bin/bash
mkdir -p gpurun_out/c19; export TMPDIR=/tmp
for c in "25 1080 1920 50 16 512" "8 576 768 20 16 256"; do
n=$(echo $c | cut -d' ' -f1)
rocprofv3 --kernel-trace -d gpurun_out/c19/t$n -o p --output-format csv -- python tools/diag/trace_cfg.py $c 5 > gpurun_out/c19/t$n.log 2>&1
csv=$(find gpurun_out/c19/t$n -name "*kernel_trace.csv" | head -1)
python - $csv <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:70]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print('%-72s %5d calls  avg %9.1f us  total %9.1f' % (k, len(v), sum(v)/len(v), sum(v)))
PY
rm -rf gpurun_out/c19/t$n
done
