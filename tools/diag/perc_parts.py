"""GPU helper: where the iteration with the perceptual term spends its GPU time, unprofiled: torch events on the caller's stream around
phase 1 of the step, the network's forward + backward, phase 2.  usage: perc_parts.py"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib
from dbw_amd.lpips_vgg import LPIPSVGG
from dbw_amd.parallel import ShardedTrainStep
dev = torch.device('cuda', 0)
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = 4, 300, 400, 10, 10, 256
model, inp = bench.build_workload(a, dev)
lw = {'rgb': model.loss_weights['rgb'], 'perceptual': 0.1}
lw.update({k: v for k, v in model.loss_weights.items() if k != 'rgb'})
model.loss_weights = lw
torch.manual_seed(5)
net = LPIPSVGG(allow_random_init=True).to(dev)
CHEAP = os.environ.get('DBW_CHEAP_TERM', '0')
if CHEAP == '1':
    model.set_perceptual(lambda a, b: ((a - b) ** 2).mean())          # a trivial term instead of the network
elif CHEAP == '2':
    conv = torch.nn.Conv2d(3, 64, 3, padding=1).to(dev)
    model.set_perceptual(lambda a, b: ((conv(a) - conv(b)) ** 2).mean())        # one MIOpen convolution
else:
    model.set_perceptual(net)
model.sync_free = True
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391)
step.cstep.read_losses = True
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
PROBE = os.environ.get('DBW_PROBE', '0') == '1'
if PROBE:
    PL = ctypes.CDLL(os.path.join(ROOT, 'tools', 'ubench', 'libclock_probe.so'))
    PL.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    pout = torch.zeros(4, 3, device=dev)
    pev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    pacc = [0.0, 0.0]
orig_call = _lib.call
marks = []
def call(name, *args):
    if name == 'dbw_train_step_run':
        i = len(marks)
        if PROBE and i == 0:          # a tiny ALU-only kernel in front of phase 1, between two events of its own
            pev[0].record(); PL.clock_probe(pout[0].data_ptr(), 400, torch.cuda.current_stream(dev).cuda_stream); pev[1].record()
        ev[2 * i].record()
        r = orig_call(name, *args)
        ev[2 * i + 1].record()
        marks.append(i)
        if PROBE and i == 1:          # ... and one behind phase 2
            pev[2].record(); PL.clock_probe(pout[1].data_ptr(), 400, torch.cuda.current_stream(dev).cuda_stream); pev[3].record()
        return r
    return orig_call(name, *args)
import dbw_amd.c_step as CS
acc = [0.0, 0.0, 0.0, 0.0]
for it in range(25):
    marks.clear()
    CS._lib.call = call
    t0 = time.perf_counter()
    step(inp).host()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    CS._lib.call = orig_call
    if it >= 5:
        if PROBE:
            pacc[0] += pev[0].elapsed_time(pev[1]); pacc[1] += pev[2].elapsed_time(pev[3])
        acc[0] += ev[0].elapsed_time(ev[1]); acc[1] += ev[1].elapsed_time(ev[2]); acc[2] += ev[2].elapsed_time(ev[3]); acc[3] += wall * 1e3
n = 20
print(f'phase 1 (head + fg pass): {acc[0] / n:.3f} ms   network forward + backward between the phases: {acc[1] / n:.3f} ms   phase 2 (fg pass again, backward, tail, Adam): {acc[2] / n:.3f} ms   wall {acc[3] / n:.3f} ms')
if PROBE:
    print(f'probe in front of phase 1: {pacc[0] / n:.3f} ms between its events; probe behind phase 2: {pacc[1] / n:.3f} ms')
rec = torch.rand_like(inp['imgs']).requires_grad_(True)
for mode in ('back to back', 'synchronised every iteration'):
    for _ in range(3):
        torch.autograd.grad(net(inp['imgs'], rec), rec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(10):
        e0.record()
        torch.autograd.grad(net(inp['imgs'], rec), rec)
        e1.record()
        if mode != 'back to back':
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
    torch.cuda.synchronize()
    if mode == 'back to back':
        tot = e0.elapsed_time(e1) * 10
    print(f'network alone, {mode}: {tot / 10:.3f} ms (events around one forward + backward)')
