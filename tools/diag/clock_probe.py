"""GPU helper: the shader clock the GPU actually runs at (tools/ubench/clock_probe.hip) at points of the small-batch loop: right after a
synchronisation, after idle periods of different length, between the steps of a running loop, and behind the perceptual network.
usage: clock_probe.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
L = ctypes.CDLL(os.path.join(ROOT, 'tools', 'ubench', 'libclock_probe.so'))
L.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda', 0)
out = torch.zeros(64, 3, device=dev)
n = [0]
def probe(iters=2000):
    i = n[0]; n[0] += 1
    L.clock_probe(out[i].data_ptr(), iters, torch.cuda.current_stream(dev).cuda_stream)
    return i
def show(tag, idx):
    torch.cuda.synchronize()
    print(f'{tag:60s} ' + '  '.join(f'{out[i, 0].item():6.0f} MHz ({out[i, 1].item():5.1f} us)' for i in idx), flush=True)
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = 4, 300, 400, 10, 10, 256
model, inp = bench.build_workload(a, dev)
model.sync_free = True
step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=1)
step.cstep.read_losses = True
for _ in range(20):
    step(inp).host()
torch.cuda.synchronize()
show('three probes back to back after a busy loop', [probe(), probe(), probe()])
for ms in (0.1, 0.5, 2, 10, 50):
    torch.cuda.synchronize(); time.sleep(ms * 1e-3)
    show(f'after {ms} ms of idle: three probes back to back', [probe(), probe(), probe()])
idx = []
for _ in range(8):
    idx.append(probe(400)); step(inp).host()
show('in front of each step of the batch-4 loop (reads every step)', idx)
x = torch.randn(8192, 8192, device=dev)
torch.cuda.synchronize()
idx = []
for _ in range(4):
    for _ in range(6): y = x @ x
    idx.append(probe(400)); step(inp); idx.append(probe(400)); torch.cuda.synchronize()
show('behind ~12 ms of GEMMs: probe, step, probe (x4)', idx)
