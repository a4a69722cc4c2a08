"""GPU helper: HIP-event times of the four render kernels (bench.kernel_breakdown: the kernels alone, set-up outside the timed region)
under alternating debug flags / render variants in ONE process.  usage: ab_kernels.py [epoch] "flags:variant" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
epoch = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfgs = [tuple(int(x) for x in a.split(':')) for a in sys.argv[2:]] or [(0, 0), (4096, 0)]
lib = _lib.load()
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(epoch)
model(inp, None)
for rep in range(2):
    for c in cfgs:
        lib.dbw_debug_set_flags(c[0]); lib.dbw_debug_set_render_variant(c[1])
        kb = bench.kernel_breakdown(model, inp, reps=10)
        print('flags %5d variant %d:' % c, {k.replace('render_', '').replace('_fused', ''): round(v[0], 4) for k, v in kb.items()})
lib.dbw_debug_set_flags(0); lib.dbw_debug_set_render_variant(0)
if bench.BIN_STATS:
    print('texture bins:', bench.BIN_STATS)
