"""GPU profiling helper (not product code): step time with / without the regularisers (45 tiny launches) to price the
launch-bound sections of an iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch
import bench
from dbw_amd.parallel import ShardedTrainStep

class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.sync_free = True; model.overlap_passes = True
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391)
full = dict(model.loss_weights)
for name, w in (('all losses', full), ('rgb only', {'rgb': full['rgb']}), ('all losses', full)):
    model.loss_weights = w
    for _ in range(3): step(inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step(inp)
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) * 50, 3), 'ms/step')
