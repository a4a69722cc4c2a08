#!/bin/bash
# VALU / SALU wave-instructions per launch of every kernel of the step whose name matches $2 (49 views, config 2, epoch $3)
O=gpurun_out/r06/$1; mkdir -p $O; export TMPDIR=/tmp
DBW_EPOCH=${3:-0} DBW_STEP_EVENTS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $O/pmc -o p --output-format csv -- python tools/pmc_target.py > $O/pmc.log 2>&1
python - $O/pmc "$2" <<'PY'
import csv, glob, sys
vals = {}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r['Kernel_Name']: continue
        vals.setdefault((r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name']), {}).setdefault(r['Dispatch_Id'], 0.0)
        vals[(r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
for (k, c), d in sorted(vals.items()):
    xs = sorted(d.values()); print(k, c, 'median %.3f M' % (xs[len(xs) // 2] / 1e6))
PY
rm -rf $O/pmc
