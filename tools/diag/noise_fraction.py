"""GPU helper: run-to-run spread of the 'share of parameters further apart than 1e-4' statistic of the step-equivalence tests."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ('tests', 'differentiable-blocksworld_amd', 'oracle'):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import test_gpu_c_step as T
rows = []
def cmp(a, b, names):
    (_, va, ga, pa), (_, vb, gb, pb) = a, b
    diff = (pa - pb).abs()
    frac = float((diff > 1e-4).float().mean())
    worst_sig = 0.0
    for n, off, k in names:
        y = gb[off:off + k]
        sig = y.abs() > 1e-2 * y.abs().max()
        lr = 5e-2 if 'texture' in n else 5e-3
        if sig.any():
            worst_sig = max(worst_sig, float(diff[off:off + k][sig].max()) / lr)
    rows.append((frac, float(diff.max()), worst_sig))
T._compare = cmp
for name, args in (('test_c_step_streams_wait_through_memory_words_as_through_events', (0,)), ('test_c_step_streams_wait_through_memory_words_as_through_events', (800,))):
    rows.clear()
    for _ in range(8):
        getattr(T, name)(*args)
    print(name, args, ' '.join('%.4f/%.4f/%.3f' % r for r in rows), flush=True)
