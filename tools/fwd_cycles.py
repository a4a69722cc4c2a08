"""GPU helper: cycle / event accounting of the fused forward (fg pass of the bench config) from a library built with -DDBW_PROFILE_FWD
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
model.set_cur_epoch(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
model(inp, None); torch.cuda.synchronize()
lib.dbw_debug_read_fwd_profile(buf, 1)
kb = bench.kernel_breakdown(model, inp, reps=1)
torch.cuda.synchronize()
lib.dbw_debug_read_fwd_profile(buf, 1)
v = list(buf)
L = float(os.environ.get('DBW_FPROF_LAUNCHES', 3.0))                         # kernel_breakdown runs the soft forward three times (first call, warm-up, one timed repetition)
tot = v[3]
print({k: round(x[0], 3) for k, x in kb.items()})
print({'prologue': f'{100 * v[12] / tot:.1f}%', 'binning (list walk + staging)': f'{100 * v[0] / tot:.1f}%',
       'staged-face loop (evaluate + insert)': f'{100 * v[1] / tot:.1f}%', 'shading + stores': f'{100 * v[2] / tot:.1f}%'})
print('per launch (%s pass; counted per wave):' % os.environ.get('DBW_FPROF_PASS', 'fg') + ' tiles %.1f k (%.1f k with staged faces), staged (tile, face) pairs %.3f M (culled by the tile-vs-edge test: %.3f M), '
      'with a pixel in the box %.3f M, (pixel, face) evaluations %.2f M, kept %.2f M, wave re-evaluations with IEEE divisions %.4f M' %
      (v[9] / L / 1e3, v[10] / L / 1e3, v[4] / L / 1e6, v[11] / L / 1e6, v[5] / L / 1e6, v[6] / L / 1e6, v[7] / L / 1e6, v[8] / L / 1e6))
