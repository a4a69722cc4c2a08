#!/bin/bash
# SQ counters of the fg forward in a fast and a slow scene state.  usage: r06_spike3.sh <tag> slow_step fast_step
O=gpurun_out/r06/$1; mkdir -p $O; export TMPDIR=/tmp
export DBW_STEP_EVENTS=1      # (a counter pass runs one kernel at a time: the step's streams must wait through events, not polled words)
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -o p --output-format csv -- python tools/diag/r06_spike3.py $2 $3 > $O/g$i.log 2>&1
done
python - $O <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for g in ('g1', 'g2'):
    per = {}
    for f in glob.glob(os.path.join(out, g, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'render_fwd_kernel<10' in r['Kernel_Name']:
                per.setdefault(int(r['Dispatch_Id']), {}).setdefault(r['Counter_Name'], 0.0)
                per[int(r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    ids = sorted(per)[-6:]
    for name in sorted(per[ids[0]]):
        print('%-24s fast %s | slow %s' % (name, ' '.join('%.3f' % (per[i][name] / 1e6) for i in ids[:3]), ' '.join('%.3f' % (per[i][name] / 1e6) for i in ids[3:])))
    dur = {}
    for f in glob.glob(os.path.join(out, g, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'render_fwd_kernel<10' in r['Kernel_Name']: dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print('us under counters        fast %s | slow %s' % (' '.join('%.0f' % dur.get(i, 0) for i in ids[:3]), ' '.join('%.0f' % dur.get(i, 0) for i in ids[3:])))
PY
rm -rf $O/g1 $O/g2
