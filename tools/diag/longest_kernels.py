"""From a rocprofv3 --kernel-trace CSV: the longest kernels with their start times, and the gaps > 50 ms between consecutive kernel activity."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1], encoding='utf-8', errors='replace')), key=lambda r: int(r['Start_Timestamp']))
t00 = int(rows[0]['Start_Timestamp'])
for r in sorted(rows, key=lambda r: int(r['Start_Timestamp']) - int(r['End_Timestamp']))[:8]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%10.3f ms  +%9.3f ms  q%s s%s %s' % ((s - t00) / 1e6, (e - s) / 1e6, r.get('Queue_Id'), r.get('Stream_Id'), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:80]))
end = 0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if end and s - end > 50e6:
        print('GAP of %.1f ms before %.3f ms: next kernel %s (q%s s%s)' % ((s - end) / 1e6, (s - t00) / 1e6, r['Kernel_Name'][:60], r.get('Queue_Id'), r.get('Stream_Id')))
    end = max(end, e)
