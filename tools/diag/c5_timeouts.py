"""GPU helper: does a cross-stream poll of the C step give up at BASELINE config 5 (25 views of 1080x1920, 50 blocks, fpp 16, 512^2 textures)?
Steps enqueued back to back, a synchronisation every `group` steps (bench.py's measure_other does 5).  usage: c5_timeouts.py [groups] [group] [events]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 6
group = int(sys.argv[2]) if len(sys.argv) > 2 else 5
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = [int(x) for x in os.environ.get('DBW_CFG', '25 1080 1920 50 16 512').split()]
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(a, dev)
model.set_cur_epoch(int(os.environ.get('DBW_EPOCH', '0')))
model.sync_free = True
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391)
if len(sys.argv) > 3:
    step.cstep.sync_events = True
import warnings
warnings.simplefilter('always')
for _ in range(2):
    step(inp)
torch.cuda.synchronize()
for g in range(groups):
    t0 = time.perf_counter()
    for _ in range(group):
        step(inp)
    torch.cuda.synchronize()
    print(f'group {g}: {(time.perf_counter() - t0) / group * 1e3:8.2f} ms/step, voided runs so far {step.cstep.voided_runs()}, sync_events {step.cstep.sync_events}', flush=True)
