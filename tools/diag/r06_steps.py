"""ms of every single step (a device synchronisation behind each) from a fresh plan on, with the bench's learning rates and with both at 0
(the scene does not move).  usage: r06_steps.py epoch nsteps [views H W blocks fpp txt]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
class A: pass
a = A()
v = [int(x) for x in sys.argv[3:]] + [49, 300, 400, 10, 10, 256][len(sys.argv) - 3:]
a.views, a.H, a.W, a.blocks, a.fpp, a.txt = v
dev = torch.device('cuda', 0)
for lr_scale in (1.0, 0.0, 1.0):
    model, inp = bench.build_workload(a, dev)
    model.set_cur_epoch(int(sys.argv[1])); model.sync_free = True
    step = ShardedTrainStep(model, lr=5e-3 * lr_scale, lr_texture=5e-2 * lr_scale, seed=227391)
    ts = []
    for _ in range(int(sys.argv[2])):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step(inp)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print('lr x %.0f:' % lr_scale, ' '.join('%.3f' % t for t in ts), flush=True)
    if os.environ.get('DBW_SHOW'):
        with torch.no_grad():
            print('   alpha', [round(float(x), 3) for x in torch.sigmoid(model.alpha_logit.flatten())][:50] if hasattr(model, 'alpha_logit') else None)
    del step, model, inp; torch.cuda.empty_cache()
