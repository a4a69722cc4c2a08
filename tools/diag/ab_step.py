"""GPU helper: ms per training step of the bench configuration under alternating library debug flags / render variants in ONE process
on ONE box (box-to-box differences are +-3 %).  usage: ab_step.py [epoch] "flags:variant" "flags:variant" ...   e.g.  ab_step.py 0 0:0 4096:0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib
from dbw_amd.parallel import ShardedTrainStep
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
epoch = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfgs = [tuple(int(x) for x in a.split(':')) for a in sys.argv[2:]] or [(0, 0), (4096, 0)]
lib = _lib.load()
res = {c: [] for c in cfgs}
for rep in range(3):
    for c in cfgs:
        torch.manual_seed(0)
        model, inp = bench.build_workload(args, dev)
        model.sync_free = True
        model.set_cur_epoch(epoch)
        step = ShardedTrainStep(model, seed=1)
        lib.dbw_debug_set_flags(c[0]); lib.dbw_debug_set_render_variant(c[1])
        for _ in range(5):
            step(inp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            step(inp)
        e1.record(); torch.cuda.synchronize()
        res[c].append(e0.elapsed_time(e1) / 20)
lib.dbw_debug_set_flags(0); lib.dbw_debug_set_render_variant(0)
for c, v in res.items():
    print('flags %5d variant %d: ms/step %s' % (c[0], c[1], ' '.join('%.4f' % x for x in v)))
