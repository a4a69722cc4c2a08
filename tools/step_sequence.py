"""From a rocprofv3 --kernel-trace CSV of bench.py: the kernel sequence of ONE steady-state optimisation step (between two Adam launches),
with start offsets, the idle gap before each kernel (negative: it overlaps a kernel of the other stream) and durations.  usage: step_sequence.py t_kernel_trace.csv"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1], encoding='utf-8', errors='replace')), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
adam = [i for i, n in enumerate(names) if 'adam_' in n]
# a step ends with the Adam launch(es): one (dbw_adam_step_groups) or two (one per learning-rate group)
ends = adam if any('adam_groups' in names[i] for i in adam) else adam[1::2]
a, b = ends[9] + 1, ends[10] + 1
prev_end, tot, gaps = None, 0.0, 0.0
print(f'{b - a} kernels in the step')
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f'{(s - int(rows[a]["Start_Timestamp"])) / 1e3:8.1f} us  {gap:7.1f} us gap  {(e - s) / 1e3:8.1f} us  {n[:100]}')
    if 'sync_wait_kernel' in n:          # a one-thread poll (a stream waiting for another; the NEXT step's polls start early): waiting, not work
        continue
    tot += (e - s) / 1e3; gaps += max(gap, 0.0)
    prev_end = max(e, prev_end or 0)
print(f'kernel time {tot:.1f} us, gaps {gaps:.1f} us, span {(int(rows[b-1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3:.1f} us')
