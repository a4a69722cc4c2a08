"""Turn a rocprofv3 results .db (kernel trace) into the per-kernel stats table kept under profiles/."""
import sqlite3, sys
db, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
t0, t1 = cur.execute("select min(start), max(end) from kernels").fetchone()
tot = sum(r[2] for r in rows)
lines = [f"# {title}", f"# durations in microseconds; sum of kernel time {tot/1e3:.2f} ms over a {(t1-t0)/1e6:.2f} ms trace window",
         f"{'kernel':72s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
for n, c, t, a, mi, ma in rows[:45]:
    n = n.replace('(anonymous namespace)::', '')
    lines.append(f"{n[:72]:72s} {c:6d} {t:12.1f} {a:10.1f} {mi:10.1f} {ma:10.1f} {100*t/tot:6.2f}")
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:34]))
