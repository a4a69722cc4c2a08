"""Workload for a rocprofv3 kernel trace of the iteration WITH the perceptual term (LPIPS-VGG16, seeded weights) at the reference's batch
size: 14 steps of bench.measure_perceptual's loop.  usage: trace_perceptual.py [views]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.lpips_vgg import LPIPSVGG
from dbw_amd.parallel import ShardedTrainStep
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(a, dev)
lw = {'rgb': model.loss_weights['rgb'], 'perceptual': 0.1}
lw.update({k: v for k, v in model.loss_weights.items() if k != 'rgb'})
model.loss_weights = lw
torch.manual_seed(5)
model.set_perceptual(LPIPSVGG(allow_random_init=True).to(dev))
model.sync_free = True
fuse = int(os.environ.get('DBW_FUSE', '127'))
if not fuse & 1 or os.environ.get('DBW_NOISE_OVERRIDE'):
    model._noise_override = torch.zeros(10, device=dev)
if not fuse & 4:
    model._overlap_u_override = torch.rand(10, 1000, 3, device=dev)
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391, fuse=fuse)
step.cstep.read_losses = True
for _ in range(14):
    step(inp).host()
torch.cuda.synchronize()
