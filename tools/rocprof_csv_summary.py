"""Turn a rocprofv3 --kernel-trace CSV into the per-kernel stats table kept under profiles/.  usage: rocprof_csv_summary.py trace.csv out.txt title"""
import collections, csv, sys
src, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
d = collections.defaultdict(list)
t0, t1 = None, None
for r in csv.DictReader(open(src, encoding='utf-8', errors='replace')):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d[r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')].append((e - s) / 1e3)
    t0, t1 = (s if t0 is None else min(t0, s)), (e if t1 is None else max(t1, e))
# (sync_wait_kernel: the one-thread polls through which the step's streams wait for each other -- their duration is waiting, not work:
# listed, but left out of the sum and of the percentages)
polls = {n: v for n, v in d.items() if 'sync_wait_kernel' in n}
tot = sum(sum(v) for n, v in d.items() if n not in polls)
pct = lambda n, v: '-' if n in polls else f'{100 * sum(v) / tot:.2f}'
lines = [f'# {title}', f'# durations in microseconds; sum of kernel time {tot / 1e3:.2f} ms over a {(t1 - t0) / 1e6:.2f} ms trace window'
         + (' (the one-thread polls between the streams, sync_wait_kernel, wait rather than work: not in the sum)' if polls else ''),
         f"{'kernel':72s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
for n, v in sorted(d.items(), key=lambda kv: (kv[0] in polls, -sum(kv[1])))[:45]:
    lines.append(f'{n[:72]:72s} {len(v):6d} {sum(v):12.1f} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {pct(n, v):>6s}')
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:30]))
