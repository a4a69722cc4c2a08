"""GPU helper: where the time of LPIPS-VGG16 with cached targets goes (torch profiler, 4 images of 300x400)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from dbw_amd.lpips_vgg import LPIPSVGG
dev = torch.device('cuda', 0)
torch.manual_seed(5)
net = LPIPSVGG(allow_random_init=True).to(dev)
imgs = torch.rand(4, 3, 300, 400, device=dev)
rec0 = torch.rand(4, 3, 300, 400, device=dev)
ids = torch.arange(4, device=dev)
net.cache_targets(imgs)
def base():
    rec = rec0.clone().requires_grad_(True)
    return torch.autograd.grad(net(imgs, rec, view_ids=ids), rec)[0]
for _ in range(3):
    base()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    base(); torch.cuda.synchronize()
rows = sorted(((e.key, e.count, e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total) for e in prof.key_averages()), key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print('total device time %.3f ms' % (tot / 1e3))
conv = sum(r[2] for r in rows if 'conv' in r[0].lower() or 'igemm' in r[0].lower() or 'gemm' in r[0].lower() or 'Cijk' in r[0])
print('convolution kernels %.3f ms' % (conv / 1e3))
for k, c, t in rows[:30]:
    print('%9.1f us  x%-3d %s' % (t, c, k[:150]))
