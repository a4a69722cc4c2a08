"""GPU tuning helper: env-pass forward (K = 1) under tile-shape variants (dbw_debug_set_render_variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
model, inp = bench.build_workload(args, torch.device('cuda', 0))
model(inp, None)
lib = _lib.load()
for v, name in ((0, '16x16'), (1, '8x8'), (2, '16x8')):
    lib.dbw_debug_set_render_variant(v)
    kb = bench.kernel_breakdown(model, inp, reps=10)
    print(name, {k: round(x[0], 4) for k, x in kb.items() if 'env' in k})
lib.dbw_debug_set_render_variant(0)
