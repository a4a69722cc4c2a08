"""GPU helper: ms/step in chunks of 100 steps for 2000 steps of the headline workload (frozen parameters or not), reserved memory and
the GPU clock before / after.  usage: sustained.py [lr_scale]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
lr_scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
sync_every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
model, inp = bench.build_workload(args, dev)
model.sync_free = True
step = ShardedTrainStep(model, lr=5e-3 * lr_scale, lr_texture=5e-2 * lr_scale, seed=227391)
def clk():
    try:
        return subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=20).stdout.replace('\n', ' | ')[:400]
    except Exception as e:
        return str(e)
for _ in range(10):
    step(inp)
torch.cuda.synchronize()
print('clocks before:', clk())
for chunk in range(20):
    t0 = time.perf_counter()
    for i in range(100):
        step(inp)
        if (i + 1) % sync_every == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print('steps %4d..%4d: %.4f ms/step, reserved %.2f GB' % (chunk * 100, chunk * 100 + 99, (time.perf_counter() - t0) * 10, torch.cuda.memory_reserved() / 2**30))
print('clocks after:', clk())
