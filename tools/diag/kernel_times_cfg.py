"""GPU helper: HIP-event times of the render kernels for an arbitrary configuration and debug flags.
usage: kernel_times_cfg.py views H W blocks fpp txt epoch [flags ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib
class A: pass
args = A()
args.views, args.H, args.W, args.blocks, args.fpp, args.txt = [int(x) for x in sys.argv[1:7]]
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(int(sys.argv[7]))
model(inp, None)
lib = _lib.load()
for flags in [int(x, 0) for x in sys.argv[8:]] or [0]:
    lib.dbw_debug_set_flags(flags)
    kb = bench.kernel_breakdown(model, inp, reps=3)
    print(hex(flags), {k: round(v[0], 3) for k, v in kb.items()})
lib.dbw_debug_set_flags(0)
