"""ms per step of the C step (49 views, config 2) under debug flags: usage r06_flags.py epoch flags..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib
from dbw_amd.parallel import ShardedTrainStep
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = int(os.environ.get('DBW_VIEWS', '49')), 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(a, dev)
model.set_cur_epoch(int(sys.argv[1])); model.sync_free = True
step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=1, **({'fuse': int(os.environ['DBW_FUSE'])} if os.environ.get('DBW_FUSE') else {}))
lib = _lib.load()
for rep in range(2):
    for f in [int(x) for x in sys.argv[2:]]:
        lib.dbw_debug_set_flags(f)
        for _ in range(10): step(inp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): step(inp)
        torch.cuda.synchronize()
        print('flags %9d: %.4f ms/step' % (f, (time.perf_counter() - t0) / 50 * 1e3))
lib.dbw_debug_set_flags(0)
