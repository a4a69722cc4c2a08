"""Profiling helper: from a rocprofv3 kernel-trace .db, the GPU busy time (union of kernel intervals) vs wall time over the
last N launches of a marker kernel (one per training step) -- shows how much of a step the GPU sits idle (launch-bound)."""
import sqlite3, sys
db = sys.argv[1]; marker = sys.argv[2] if len(sys.argv) > 2 else 'adam_kernel'; nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
marks = [r[2] for r in rows if marker in r[0]]
per = len(marks) // max(1, len(set(marks)) and 1)
# adam runs twice per step (two lr groups share one launch?) -> take every launch, find period from count / steps
ends = marks[-(nsteps * 2 + 1)::2] if len(marks) >= nsteps * 2 + 1 else marks
t0, t1 = ends[0], ends[-1]
iv = [(max(s, t0), min(e, t1)) for _, s, e in rows if e > t0 and s < t1]
iv.sort()
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
n = len(ends) - 1
print(f'{n} steps: wall {(t1 - t0) / n / 1e6:.3f} ms/step, GPU busy (union) {busy / n / 1e6:.3f} ms/step, idle {(t1 - t0 - busy) / n / 1e6:.3f} ms/step, '
      f'kernel-time sum {sum(e - s for s, e in iv) / n / 1e6:.3f} ms/step, launches {len(iv) / n:.0f}/step')
