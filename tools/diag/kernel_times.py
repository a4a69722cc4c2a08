"""GPU helper: HIP-event times of the four render kernels of the bench configuration (bench.kernel_breakdown) for the library in
DBW_HIP_LIB; usage: kernel_times.py [epoch] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
from dbw_amd import _lib
lib = _lib.load()
if os.environ.get('DBW_DEBUG_FLAGS'):            # e.g. 4096: no cell lists (include/dbw_hip.h: dbw_debug_set_flags)
    lib.dbw_debug_set_flags(int(os.environ['DBW_DEBUG_FLAGS']))
if os.environ.get('DBW_RENDER_VARIANT'):         # tile shape of the hard K = 1 pass (render_fused.hip: launch<1>)
    lib.dbw_debug_set_render_variant(int(os.environ['DBW_RENDER_VARIANT']))
model.set_cur_epoch(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
model(inp, None)
kb = bench.kernel_breakdown(model, inp, reps=int(sys.argv[2]) if len(sys.argv) > 2 else 5)
print(os.environ.get('DBW_HIP_LIB', 'product'), 'flags', os.environ.get('DBW_DEBUG_FLAGS'), 'variant', os.environ.get('DBW_RENDER_VARIANT'), {k: round(v[0], 4) for k, v in kb.items()})
