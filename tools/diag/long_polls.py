"""From a rocprofv3 --kernel-trace CSV: every poll (sync_wait_kernel) longer than 0.2 s, and what ran on which queue from 50 ms before it to its end."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1], encoding='utf-8', errors='replace')), key=lambda r: int(r['Start_Timestamp']))
t00 = int(rows[0]['Start_Timestamp'])
longw = [r for r in rows if 'sync_wait' in r['Kernel_Name'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 200e6]
print(len(rows), 'kernels,', len(longw), 'polls longer than 0.2 s')
for r in longw[:2]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('POLL %.3f -> %.3f ms  queue %s stream %s' % ((s - t00) / 1e6, (e - t00) / 1e6, r.get('Queue_Id'), r.get('Stream_Id')))
    for q in rows:
        qs, qe = int(q['Start_Timestamp']), int(q['End_Timestamp'])
        if qe > s - 50e6 and qs < e + 20e6 and q is not r:
            print('   %10.3f -> %10.3f ms (%9.3f)  q%s s%s  %s' % ((qs - t00) / 1e6, (qe - t00) / 1e6, (qe - qs) / 1e6, q.get('Queue_Id'), q.get('Stream_Id'), q['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]))
