"""GPU helper: is the optimisation step host-bound?  Host time to ENQUEUE a step (no synchronisation inside the loop) next to the
GPU time per step of the same loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.sync_free = True
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=1)
for _ in range(5):
    step(inp)
torch.cuda.synchronize()
for n in (20, 50):
    t0 = time.perf_counter()
    for _ in range(n):
        step(inp)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{n} steps: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, until the GPU is done {1e3 * (t2 - t0) / n:.3f} ms/step')
