"""GPU helper: ms per step of the launch-by-launch native step and of the C step (operator-level kernels from C, fused kernels), at
the reference's batch size (4), at the per-rank batch of config 3 (7) and at the benchmark batch (49); with the loss values read every
step and without; and the host time to ENQUEUE a step (no synchronisation inside the timed loop, one at its end).
usage: cstep_times.py [epoch] [batches...] [variants: py c0 c15 c31 c31ev ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep

dev = torch.device('cuda', 0)
epoch = int(sys.argv[1]) if len(sys.argv) > 1 else 0
batches = [int(x) for x in sys.argv[2:] if x.isdigit()] or [4, 7, 49]
VARIANTS = [x for x in sys.argv[2:] if not x.isdigit()] or ['py', 'c63', 'c127']


class A:
    pass


def measure(B, variant, reads, steps):
    a = A()
    a.views, a.H, a.W, a.blocks, a.fpp, a.txt = B, 300, 400, 10, 10, 256
    model, inp = bench.build_workload(a, dev)
    model.set_cur_epoch(epoch)
    model.sync_free = True
    kw = {'py': dict(use_c_step=False), 'c0': dict(fuse=0), 'c15': dict(fuse=15), 'c31': dict(fuse=31), 'c31ev': dict(fuse=31), 'c63': dict(fuse=63), 'c127': dict(fuse=127)}[variant]
    if variant == 'c0':          # the operator-level kernels need the caller's draws
        model._noise_override = torch.randn(10, device=dev)
        model._overlap_u_override = torch.rand(10, 1000, 3, device=dev)
    step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=227391, **kw)
    if step.cstep is not None:
        step.cstep.read_losses = reads
        step.cstep.sync_events = variant.endswith('ev')       # HIP events instead of polled words between the streams

    def read(out):
        if not reads:
            return
        if hasattr(out, 'host'):
            out.host()
        else:
            _ = {k: float(v) for k, v in out.items()}
    for _ in range(10):
        read(step(inp))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        read(step(inp))
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del step, model, inp
    torch.cuda.empty_cache()
    return dt / steps * 1e3, t_host / steps * 1e3


for B in batches:
    steps = 200 if B <= 8 else 50
    for variant in VARIANTS:
        row = []
        for reads in (False, True):
            for rep in range(2):
                ms, host = measure(B, variant, reads, steps)
            row.append(f'{"reads" if reads else "no reads"}: {ms:.4f} ms/step (host enqueue {host:.4f})')
        print(f'epoch {epoch} B={B:2d} {variant:7s} ' + ' | '.join(row), flush=True)
