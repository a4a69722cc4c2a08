"""GPU helper: A/B of a debug flag (dbw_debug_set_flags) on the same box, same process: the four big kernels alone and in the step, and ms
per step.  usage: ab_flags.py views flag [epoch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib
from dbw_amd.parallel import ShardedTrainStep
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = int(sys.argv[1]), 300, 400, 10, 10, 256
flag = int(sys.argv[2]); epoch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device('cuda', 0)
for rep in range(2):
    for f in (0, flag):
        _lib.load().dbw_debug_set_flags(f)
        model, inp = bench.build_workload(a, dev)
        model.set_cur_epoch(epoch)
        model.sync_free = True
        step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=1)
        for _ in range(5):
            step(inp)
        alone = step.cstep.kernel_times(inp, reps=10, alone=True)
        inst = step.cstep.kernel_times(inp, reps=10)
        torch.cuda.synchronize()
        n = 200 if a.views <= 8 else 40
        t0 = time.perf_counter()
        for _ in range(n):
            step(inp)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f'B={a.views} flags={f:8d}  {ms:.4f} ms/step  alone: ' + ' '.join(f'{k} {v * 1e3:6.1f}' for k, v in alone.items()) + '   in step: ' + ' '.join(f'{k} {v * 1e3:6.1f}' for k, v in inst.items()), flush=True)
        del step, model, inp
_lib.load().dbw_debug_set_flags(0)
