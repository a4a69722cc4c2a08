"""Workload for `rocprofv3 --pmc ...`: replays the bench workload's first steps, then runs the operator-level fg forward three times in the
fast state and three times in the slow one (r06_spike.py found them): the LAST SIX dispatches of render_fwd_kernel<10, ...> in the counter
CSV are fast, fast, fast, slow, slow, slow.  usage: r06_spike3.py slow_step fast_step"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(a, dev)
model.set_cur_epoch(0); model.sync_free = True
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391)
slow_i, fast_i = int(sys.argv[1]), int(sys.argv[2])
snaps = {}
for i in range(max(slow_i, fast_i) + 1):
    if i in (slow_i, fast_i): snaps[i] = step.params.flat.clone()
    step(inp)
torch.cuda.synchronize()
for i in (fast_i, slow_i):
    step.params.flat.copy_(snaps[i])
    bench.kernel_breakdown(model, inp, reps=1)
    torch.cuda.synchronize()
