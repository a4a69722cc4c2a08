"""GPU helper: time line of ONE launch of the fused forward (fg pass of the bench config) from a -DDBW_PROFILE_FWD build: when every
wave started and ended (100 MHz wall clock), per XCD -- load balance between the XCDs, the tail, resident waves over time.
usage: DBW_HIP_LIB=tools/variants/fprof.so python tools/fwd_timeline.py [epoch]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib

class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = int(os.environ.get('DBW_VIEWS', '49')), 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
lib = _lib.load()
if os.environ.get('DBW_DEBUG_FLAGS'):
    lib.dbw_debug_set_flags(int(os.environ['DBW_DEBUG_FLAGS']))
NB = 1 << 17
buf = np.zeros((NB, 16), dtype=np.uint64)
p = buf.ctypes.data_as(ctypes.c_void_p)
model(inp, None); torch.cuda.synchronize()
lib.dbw_debug_read_fwd_profile_raw(p, NB, 1)
model(inp, None); torch.cuda.synchronize()            # one soft forward (+ the hard pass, not recorded: FPROF_SEL)
lib.dbw_debug_read_fwd_profile_raw(p, NB, 1)
used = (buf[:, 13] > 0) & (buf[:, 14] > 0)
b = np.nonzero(used)[0]
t0, t1 = buf[b, 13].astype(np.int64), buf[b, 14].astype(np.int64)
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0           # microseconds
pairs, cyc = buf[b, 4].astype(np.int64), buf[b, 3].astype(np.int64)
print('waves %d, kernel span %.1f us' % (len(b), t1.max()))
for x in range(8):
    m = (b % 8) == x
    print('XCD %d: waves %6d  first start %7.1f  last start %7.1f  last end %7.1f us   staged pairs %8d   wave-cycles %.1f M   longest wave %.1f us'
          % (x, m.sum(), t0[m].min(), t0[m].max(), t1[m].max(), pairs[m].sum(), cyc[m].sum() / 1e6, (t1[m] - t0[m]).max()))
dur = t1 - t0
# the phases of a wave's life (cycle counters of the wave, scaled to its wall-clock life): prologue, list walk + staging, staged-face
# loop (evaluate + insert), shading + stores + loss epilogue
ph = buf[b][:, [12, 0, 1, 2]].astype(np.float64)
ph = ph / np.maximum(cyc, 1)[:, None] * dur[:, None]
for lo, hi in ((0, 1), (1, 4), (4, 16), (16, 64), (64, 10 ** 9)):
    m = (pairs >= lo) & (pairs < hi)
    if m.any():
        print('tiles with %3d..%-4s staged faces: %6d   mean life %6.1f us   max %6.1f us   share of the wave time %.1f %%   mean us of: prologue %.1f, list walk %.1f, '
              'evaluate + insert %.1f (%.2f per face), shading + stores + epilogue %.1f'
              % (lo, hi - 1 if hi < 10 ** 9 else '', m.sum(), dur[m].mean(), dur[m].max(), 100 * dur[m].sum() / dur.sum(), ph[m, 0].mean(), ph[m, 1].mean(),
                 ph[m, 2].mean(), ph[m, 2].sum() / max(pairs[m].sum(), 1), ph[m, 3].mean()))
# resident waves over time (whole GPU: 1024 SIMDs)
edges = np.linspace(0, t1.max(), 41)
for i in range(40):
    mid = 0.5 * (edges[i] + edges[i + 1])
    alive = ((t0 <= mid) & (t1 > mid))
    busy = alive & (pairs > 0)
    print('t = %6.1f us: %5d waves resident (%.2f per SIMD), %5d of them with faces' % (mid, alive.sum(), alive.sum() / 1024.0, busy.sum()))
