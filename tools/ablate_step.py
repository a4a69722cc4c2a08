"""GPU profiling helper (not product code): whole-step time of one training phase under the backward ablation flags
(1 = no texel gradients, 2 = no opacity gradients).  usage: ablate_step.py EPOCH [no-overlap]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch
import bench
from dbw_amd import _lib
from dbw_amd.parallel import ShardedTrainStep

class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(int(sys.argv[1])); model.sync_free = True; model.overlap_passes = len(sys.argv) < 3
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391)
lib = _lib.load()
for flags in [0, 1, 2, 3, 0]:
    lib.dbw_debug_set_flags(flags)
    for _ in range(3): step(inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(inp)
    torch.cuda.synchronize()
    print('flags', flags, round((time.perf_counter() - t0) * 100, 3), 'ms/step')
lib.dbw_debug_set_flags(0)
