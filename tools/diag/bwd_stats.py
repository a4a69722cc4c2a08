"""GPU helper: table traffic of the uv backward (fg pass of the bench config) from a library built with -DDBW_STATS_BWD
(tools/variants.sh stats "-DDBW_STATS_BWD"; run with DBW_HIP_LIB=tools/variants/stats.so).  Per wave and layer: lanes that update
the texel table (all taps), distinct texels of tap 0, taps with any lane, lanes that update the face table, distinct faces, and the
LDS-atomic replays (sum over 16-lane groups of the most contended address) of one face-table / texel-table value."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib

class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
for ep in [int(x) for x in sys.argv[1:]] or [0]:
    model.set_cur_epoch(ep); model(inp, None)
    torch.cuda.synchronize()
    lib.dbw_debug_read_profile(buf, 1)
    bench.kernel_breakdown(model, inp, reps=1)
    torch.cuda.synchronize()
    lib.dbw_debug_read_profile(buf, 1)
    wl = max(buf[0], 1)
    print('epoch %d: %d wave-layers (all launches of the breakdown); per wave-layer: texel-table lanes %.1f, distinct texels of tap 0 %.1f, '
          'taps %.2f, texel replays %.1f; face-table lanes %.1f, distinct faces %.2f, face replays %.1f'
          % (ep, wl, buf[1] / wl, buf[2] / wl, buf[3] / wl, buf[7] / wl, buf[4] / wl, buf[5] / wl, buf[6] / wl))
