"""ms per step of the C step at one epoch with the env backward chain next to the fg backward kernel (both) or behind it, next to the bin
reduction (seq).  usage: r06_order.py epoch [views H W blocks fpp txt]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
class A: pass
a = A()
v = [int(x) for x in sys.argv[2:]] + [49, 300, 400, 10, 10, 256][len(sys.argv) - 2:]
a.views, a.H, a.W, a.blocks, a.fpp, a.txt = v
dev = torch.device('cuda', 0)
for rep in range(2):
    for order, both in ((0, True), (1, False), (1, True), (0, False)):
        model, inp = bench.build_workload(a, dev)
        model.set_cur_epoch(int(sys.argv[1])); model.sync_free = True
        step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=1)
        step.cstep.backward_order, step.cstep.binned_concurrent = order, both
        n = 40 if a.views * a.H * a.W < 2e7 else 6
        for _ in range(6): step(inp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): step(inp)
        torch.cuda.synchronize()
        print('backward_order %d binned_concurrent %d: %.4f ms/step' % (order, both, (time.perf_counter() - t0) / n * 1e3))
        del step, model, inp; torch.cuda.empty_cache()
