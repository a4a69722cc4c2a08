"""Time line of ONE launch of the uv backward of the fg pass (bench workload, operator level) from a -DDBW_TILE_CLOCK build: start / end stamps
of every workgroup -> workgroups in flight per XCD over time, run times by the number of layers.  usage: DBW_HIP_LIB=tools/variants/tclk.so r06_bwd_clock.py [epoch]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(a, dev)
model.set_cur_epoch(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
model(inp, None)
kb = bench.kernel_breakdown(model, inp, reps=2)
print({k.replace('render_', '').replace('_fused', ''): round(v[0], 4) for k, v in kb.items()})
lib = _lib.load()
NB = 1 << 16
buf = (ctypes.c_uint * (NB * 4))()
lib.dbw_debug_read_bwd_clock(buf, NB)
t = np.frombuffer(buf, dtype=np.uint32).reshape(NB, 4)
total = a.views * ((a.H + 15) // 16) * ((a.W + 15) // 16)
grid = 8 * ((total + 7) // 8)
t = t[:grid]
ok = t[:, 2] != 0
t0 = int(t[ok, 3].astype(np.int64).min())
st, en = (t[:, 3].astype(np.int64) - t0) / 100.0, (t[:, 2].astype(np.int64) - t0) / 100.0
lay = t[:, 1].astype(np.int32)
print('%d workgroups (%d with stamps); last end %.1f us; left at the first barrier: %d' % (grid, int(ok.sum()), en[ok].max(), int((lay[ok] == -1).sum())))
for x in (0, 3, 7):
    m = ok & (np.arange(grid) % 8 == x)
    print('XCD %d: workgroups in flight every 10 us:' % x, ' '.join(str(int(((st[m] <= a0) & (en[m] > a0)).sum())) for a0 in range(0, int(en[m].max()) + 10, 10)))
m = ok
for lo, hi in ((-1, -1), (0, 0), (1, 2), (3, 4), (5, 6), (7, 8), (9, 10)):
    mm = m & (lay >= lo) & (lay <= hi)
    if mm.any():
        d = (en - st)[mm]
        print('   layers (first wave) %d-%d: %d workgroups, run time p50 %.1f p90 %.1f max %.1f us' % (lo, hi, int(mm.sum()), np.percentile(d, 50), np.percentile(d, 90), d.max()))
