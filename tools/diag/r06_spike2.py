"""Which parameter makes a slow state of the fg forward slow (r06_spike.py found the states): the parameters of a fast step with ONE
tensor -- then one block's entries of it -- taken from the slow step.  usage: r06_spike2.py slow_step fast_step"""
import os, sys, time
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
fwd = lambda: round(bench.kernel_breakdown(model, inp, reps=3)['render_fwd_fused K=10 (fg pass)'][0], 4)
step.params.flat.copy_(snaps[fast_i]); print('fast state:', fwd())
step.params.flat.copy_(snaps[slow_i]); print('slow state:', fwd())
hit = []
for n, off, k in step.params.names:
    step.params.flat.copy_(snaps[fast_i]); step.params.flat[off:off + k].copy_(snaps[slow_i][off:off + k])
    t = fwd(); print('  %-28s (%d values) from the slow state: %.4f' % (n, k, t))
    if k <= 4096: hit.append((n, off, k))
for n, off, k in hit:
    nb = a.blocks
    if k % nb: continue
    per = k // nb
    row = []
    for b in range(nb):
        step.params.flat.copy_(snaps[fast_i]); step.params.flat[off + b * per:off + (b + 1) * per].copy_(snaps[slow_i][off + b * per:off + (b + 1) * per])
        row.append(fwd())
    print('  %-28s block by block:' % n, row)
    d = (snaps[slow_i][off:off + k] - snaps[fast_i][off:off + k]).view(nb, per)
    print('     slow:', [round(float(x), 4) for x in snaps[slow_i][off:off + k]][:40])
    print('     fast:', [round(float(x), 4) for x in snaps[fast_i][off:off + k]][:40])
