"""GPU helper: LPIPS forward + backward (cached targets) under MIOpen solver switches given in the environment; prints ms and the top conv kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from dbw_amd.lpips_vgg import LPIPSVGG
torch.backends.cudnn.benchmark = os.environ.get('BENCH', '0') != '0'
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
t0 = time.perf_counter()
for _ in range(10):
    base()
torch.cuda.synchronize()
print('%.3f ms fwd+bwd' % ((time.perf_counter() - t0) / 10 * 1e3))
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    base(); torch.cuda.synchronize()
rows = sorted(((e.key, e.count, e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total) for e in prof.key_averages()), key=lambda r: -r[2])
for k, c, t in rows[:5]:
    print('%9.1f us  x%-3d %s' % (t, c, k[:110]))
