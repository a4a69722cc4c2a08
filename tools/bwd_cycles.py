"""GPU helper: cycle accounting of the fused backward (fg pass of the bench config) from a library built with -DDBW_PROFILE_BWD
(tools/variants.sh prof "-DDBW_PROFILE_BWD"; run with DBW_HIP_LIB=tools/variants/prof.so).  Phases are per-wave s_memtime
deltas summed over all waves: 0 pass 0 (occupied layers), 1 pass 1 (alpha, transmittance), 2 pass 2: fragment load + footprint +
texel fetch + blend recurrences, 3 opacity table, 4 texel table / bins, 5 uv->barycentric + rasteriser backward, 6 face table,
7 whole kernel body."""
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
names = ['pass0 (layer bound)', 'pass1 (uv kernel: flushes)', 'p2 load+blend', 'p2 opacity table (generic kernel) / final sync + flushes (hard kernel) / wait for the slowest wave of the workgroup (uv kernel)', 'p2 footprint + texel table/bins', 'p2 distance / raster bwd math', 'p2 face (+ opacity) table', 'total']
for ep in [int(x) for x in sys.argv[1:]] or [0]:
    model.set_cur_epoch(ep); model(inp, None)
    torch.cuda.synchronize()
    lib.dbw_debug_read_profile(buf, 1)
    kb = bench.kernel_breakdown(model, inp, reps=1)          # 2 launches of each backward (fg + env share the counters)
    torch.cuda.synchronize()
    lib.dbw_debug_read_profile(buf, 1)
    tot = max(buf[7], 1)
    print('epoch', ep, {n: f'{100.0 * buf[i] / tot:.1f}%' for i, n in enumerate(names[:-1])}, 'wave-cycles', tot)
