"""From a rocprofv3 --kernel-trace CSV: the duration (us) of every launch, in launch order, of each kernel whose name contains one of the
given substrings.  usage: per_launch.py t_kernel_trace.csv substr..."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1], encoding='utf-8', errors='replace')), key=lambda r: int(r['Start_Timestamp']))
for sub in sys.argv[2:]:
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if sub in r['Kernel_Name']]
    print(sub, ' '.join('%.0f' % x for x in d))
