"""GPU measurement helper for DESIGN.md section 7: step time when the target images of the step are uploaded from pinned host
memory every iteration (the boundary handing over host buffers), same stream (serial) and on a copy stream (overlapped)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch
import bench
from dbw_amd.parallel import ShardedTrainStep

class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.sync_free = True; model.overlap_passes = True
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=1)
host = inp['imgs'].cpu().pin_memory()
bufs = [torch.empty_like(inp['imgs']) for _ in range(2)]
copy_stream = torch.cuda.Stream()

def run(mode, n=30):
    for it in range(n + 5):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == 'resident':
            pass
        elif mode == 'serial':
            inp['imgs'].copy_(host, non_blocking=True)
        else:   # double-buffered upload of the NEXT step's images on a copy stream
            nxt = bufs[(it + 1) % 2]
            copy_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(copy_stream):
                nxt.copy_(host, non_blocking=True)
            inp['imgs'] = bufs[it % 2]
        step(inp)
        if mode == 'overlapped':
            torch.cuda.current_stream().wait_stream(copy_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'{mode:10s} {dt * 1e3:.3f} ms/step  {args.views / dt:.0f} views/s')

t0 = time.perf_counter(); inp['imgs'].copy_(host); torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); inp['imgs'].copy_(host, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'H2D {host.numel() * 4 / 1e6:.1f} MB in {dt * 1e3:.2f} ms = {host.numel() * 4 / dt / 1e9:.1f} GB/s')
for mode in ('resident', 'serial', 'overlapped'):
    run(mode)
