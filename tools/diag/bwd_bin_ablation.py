"""GPU helper: ablation of the binned uv backward at epoch 800 (record stores / cursor atomics off via debug flags 1<<17, 1<<18).
With ONE cursor per bin the atomics were 0.43 ms of the 0.86 ms kernel (returning atomics on a hot address serialise); with the 16
cursors per bin of the product build they are no longer visible."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(int(sys.argv[1]) if len(sys.argv) > 1 else 800)
model(inp, None)
lib = _lib.load()
for flags in (0, 1 << 17, 1 << 18, (1 << 17) | (1 << 18)):
    lib.dbw_debug_set_flags(flags)
    kb = bench.kernel_breakdown(model, inp, reps=3)
    print(hex(flags), {k: round(v[0], 3) for k, v in kb.items() if 'bwd' in k or 'texbin' in k})
lib.dbw_debug_set_flags(0)
