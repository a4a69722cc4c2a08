"""GPU helper: cycle accounting of the fused forward (fg pass of the bench config) from a library built with -DDBW_PROFILE_FWD
(tools/variants.sh fprof "-DDBW_PROFILE_FWD"; run with DBW_HIP_LIB=tools/variants/fprof.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib

class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
model(inp, None); torch.cuda.synchronize()
lib.dbw_debug_read_fwd_profile(buf, 1)
kb = bench.kernel_breakdown(model, inp, reps=1)
torch.cuda.synchronize()
lib.dbw_debug_read_fwd_profile(buf, 1)
v = list(buf)
launches = 2                    # t() runs the forward once to warm up and once timed; only soft (K > 1) passes are counted
tot = v[3]
print({'binning (list walk + LDS fill)': f'{100 * v[0] / tot:.1f}%', 'per-pixel evaluation of staged faces': f'{100 * v[1] / tot:.1f}%',
       'shading + stores': f'{100 * v[2] / tot:.1f}%'})
print('per launch (fg pass): staged (tile, face) pairs %.2f M, box-passing (pixel, face) evaluations %.1f M, inserts %.1f M' %
      (v[4] / launches / 1e6, v[5] / launches / 1e6, v[6] / launches / 1e6))
