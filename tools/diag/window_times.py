"""GPU helper: ms/step of consecutive 20-step windows of one uninterrupted run (events, no sync in between): separates clock ramp-up
after an idle period from a workload that drifts as the optimisation proceeds (opacities, killed blocks)."""
import os, sys
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
freeze = len(sys.argv) > 1 and sys.argv[1] == 'freeze'
snap = step.params.flat.clone()
for _ in range(3):
    step(inp)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(16)]
ev[0].record()
for w in range(15):
    for _ in range(20):
        step(inp)
        if freeze:
            step.params.flat.copy_(snap)          # same parameters every step: the workload cannot drift
    ev[w + 1].record()
torch.cuda.synchronize()
print('freeze' if freeze else 'training', [round(ev[i].elapsed_time(ev[i + 1]) / 20, 3) for i in range(15)], 'alpha', [round(float(a), 2) for a in model.get_opacities()])
