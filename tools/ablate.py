"""GPU profiling helper (not product code): times the kernels of the fg pass on the bench workload under ablation flags."""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch
import bench
from dbw_amd import _lib

class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
import sys
model.set_cur_epoch(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
model(inp, None)   # sets cameras
lib = _lib.load()
for flags in ([int(x) for x in sys.argv[2:]] or [0, 16]):   # 8 = force the non-LDS backward; 16 = rasteriser without its stores
    lib.dbw_debug_set_flags(flags)
    kb = bench.kernel_breakdown(model, inp, reps=5)
    print(flags, {k: round(v[0], 3) for k, v in kb.items()})
lib.dbw_debug_set_flags(0)
