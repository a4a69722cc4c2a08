"""From a rocprofv3 --pmc counter_collection CSV: SQ_INSTS_VALU / SQ_WAVES per launch of the K = 10 fused forward (median over dispatches)."""
import csv, glob, sys
vals = {}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'render_fwd_kernel<10' not in r['Kernel_Name']:
            continue
        vals.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
        vals[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for c, d in vals.items():
    xs = sorted(d.values())
    print(c, 'median per launch %.3f M over %d launches' % (xs[len(xs) // 2] / 1e6, len(xs)))
