"""GPU helper: forward + backward of LPIPS-VGG16 on 4 + 4 images of 300x400 under different settings (ms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch
from dbw_amd.lpips_vgg import LPIPSVGG
dev = torch.device('cuda', 0)
torch.manual_seed(5)
net = LPIPSVGG(allow_random_init=True).to(dev)
imgs = torch.rand(4, 3, 300, 400, device=dev)
rec0 = torch.rand(4, 3, 300, 400, device=dev)
ids = torch.arange(4, device=dev)

def run(name, fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f'{name:60s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms', flush=True)

def base(view_ids=None):
    rec = rec0.clone().requires_grad_(True)
    return torch.autograd.grad(net(imgs, rec, view_ids=view_ids), rec)[0]
g0 = base()
run('default', base)
net.cache_targets(imgs)
print([ (tuple(c.shape), c.is_contiguous(), c.is_contiguous(memory_format=torch.channels_last)) for c in net.target_cache])
run('cached targets (NCHW)', lambda: base(ids))
def fwd_rec():
    with torch.no_grad():
        net.features(rec0 * 2 - 1)
run('features of rec only, no grad', fwd_rec)
def gather():
    [c.index_select(0, ids) for c in net.target_cache]
run('the gathers alone', gather)
net.cache_targets(None)
net.nhwc()
run('nhwc', base)
net.cache_targets(imgs)
print([ (tuple(c.shape), c.is_contiguous(), c.is_contiguous(memory_format=torch.channels_last)) for c in net.target_cache])
run('nhwc + cached targets', lambda: base(ids))
g1 = base(ids)
print('max diff', float((g1 - g0).abs().max()), 'of', float(g0.abs().max()))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    base(ids); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=70))
