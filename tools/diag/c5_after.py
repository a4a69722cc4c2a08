import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
dev = torch.device('cuda', 0)
warnings.simplefilter('always')
def leg(cfg, n):
    class A: pass
    a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = cfg
    model, inp = bench.build_workload(a, dev)
    model.sync_free = True
    step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391)
    sync_first = os.environ.get('DBW_SYNC_FIRST', '0') != '0'
    if sync_first:
        torch.cuda.synchronize()
    if os.environ.get('DBW_SYNC_MID', '0') != '0':
        orig = step.cstep._plan_for
        def pf(*a, **k):
            r = orig(*a, **k)
            torch.cuda.synchronize()
            return r
        step.cstep._plan_for = pf
    if os.environ.get('DBW_TIME_PLAN', '0') != '0':
        orig2 = step.cstep._plan_for
        def pf2(*a, **k):
            t = time.perf_counter(); r = orig2(*a, **k); print('   _plan_for %.1f ms' % ((time.perf_counter() - t) * 1e3)); return r
        step.cstep._plan_for = pf2
    for i in range(n):
        t0 = time.perf_counter()
        step(inp)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print('   host call %.1f ms' % ((t1 - t0) * 1e3))
        print(cfg, 'step', i, '%.1f ms' % ((time.perf_counter() - t0) * 1e3), 'voided', step.cstep.voided_runs(), 'timeouts', step.cstep.sync_timeouts(), step.cstep.last_timeout(), flush=True)
    del step, model, inp
    torch.cuda.empty_cache()
leg((8, 576, 768, 20, 16, 256), 2)
leg((25, 1080, 1920, 50, 16, 512), 3)
