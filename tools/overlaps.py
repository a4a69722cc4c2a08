"""From a rocprofv3 --kernel-trace CSV: for every launch of the kernel whose name contains <substr> and that ran longer than <min_us>, the
other kernels whose execution overlapped it.  usage: overlaps.py t_kernel_trace.csv substr min_us"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1], encoding='utf-8', errors='replace')), key=lambda r: int(r['Start_Timestamp']))
sub, min_us = sys.argv[2], float(sys.argv[3])
for i, r in enumerate(rows):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if sub in r['Kernel_Name'] and (e - s) / 1e3 >= min_us:
        print('%s: %.1f us (queue %s)' % (r['Kernel_Name'][:60], (e - s) / 1e3, r.get('Queue_Id')))
        for q in rows:
            s2, e2 = int(q['Start_Timestamp']), int(q['End_Timestamp'])
            if q is not r and s2 < e and e2 > s:
                print('    overlaps %s: from %.1f to %.1f us of it (queue %s, grid %s, workgroup %s)' % (q['Kernel_Name'][:70], (s2 - s) / 1e3, (e2 - s) / 1e3, q.get('Queue_Id'),
                                                                                                  q.get('Grid_Size', q.get('Grid_Size_X')), q.get('Workgroup_Size', q.get('Workgroup_Size_X'))))
