"""GPU helper: ms per step of the iteration with the perceptual term (bench.measure_perceptual's loop) under variants of the step's
cross-stream machinery.  usage: perc_variants.py
(round 5 also ran it with the step's launches on a stream of their own -- no difference, profiles/r05_experiments.md; the switch is gone)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.lpips_vgg import LPIPSVGG
from dbw_amd.parallel import ShardedTrainStep
dev = torch.device('cuda', 0)
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = 4, 300, 400, 10, 10, 256
def run(tag, events=False, side=True, reads=True):
    model, inp = bench.build_workload(a, dev)
    lw = {'rgb': model.loss_weights['rgb'], 'perceptual': 0.1}
    lw.update({k: v for k, v in model.loss_weights.items() if k != 'rgb'})
    model.loss_weights = lw
    torch.manual_seed(5)
    net = LPIPSVGG(allow_random_init=True).to(dev)
    model.set_perceptual(net)
    model.sync_free = True
    step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391)
    step.cstep.read_losses = reads
    step.cstep.sync_events = events
    step.cstep.use_side_stream = side
    for _ in range(5):
        o = step(inp)
        if reads: o.host()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        o = step(inp)
        if reads: o.host()
    torch.cuda.synchronize()
    print(f'{tag:40s} {(time.perf_counter() - t0) / 30 * 1e3:8.3f} ms/step', flush=True)
run('polled words, side streams, reads')
run('events, side streams, reads', events=True)
run('single stream, reads', side=False)
run('polled words, side streams, no reads', reads=False)
run('events, side streams, no reads', events=True, reads=False)
