"""GPU helper: record capacity of the texture bins at one GPU's share of config 5, full-resolution phase: kernel times and overflowed
sub-ranges with the capacity scaled by the factors given (default 1 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import ops
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 25, 1080, 1920, 50, 16, 512
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(800)
model(inp, None)
orig = ops.texbin_capacity
for scale in [float(x) for x in sys.argv[1:]] or [1.0, 4.0]:
    def cap(B, H, W, K, nbins, _s=scale):
        sub = ops.bin_subcursors()
        c = int(orig(B, H, W, K, nbins) * _s)
        return (c + sub - 1) // sub * sub
    ops.texbin_capacity = cap
    kb = bench.kernel_breakdown(model, inp, reps=1)
    print('capacity x%g:' % scale, {k.replace('render_', '').replace('_fused', ''): round(v[0], 3) for k, v in kb.items() if 'fg' in k}, bench.BIN_STATS.get('fg'))
    torch.cuda.empty_cache()
ops.texbin_capacity = orig
