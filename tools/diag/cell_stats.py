"""GPU helper: per-tile face-list statistics of the fg pass of a configuration (cell table of the pass's workspace) and the kernel times
with / without tile lists.  usage: cell_stats.py views H W blocks fpp txt"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib, ops
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = [int(x) for x in sys.argv[1:7]]
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(0)
model(inp, None)
lib = _lib.load()
B, H, W = args.views, args.H, args.W
with torch.no_grad():
    scene = model.build_blocks_scene(filter_transparent=False)
    alpha = model._alpha.detach().repeat_interleave(model.BNF).contiguous()
r = model.renderer
cfg = r._cfg(scene.faces.shape[0], lds_aggregate=True)
Kmat = r.cameras.K[0].contiguous()
cl = ops.project_clip(scene.verts.detach(), scene.faces, inp['R'], inp['T'], Kmat, cfg.eps, cfg.z_clip, cfg.persp)
fvc = cl['face_verts'].view(-1, 3, 3)
state = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), alpha, r._bg, 2, stage=1)
ws = state[0]
off = (ctypes.c_ulonglong * 6)()
lib.dbw_debug_cell_layout(ctypes.c_int64(fvc.shape[0]), B, H, W, off)
tiles = ((H + 7) // 8) * ((W + 7) // 8)
total = B * tiles
wsb = ws.view(torch.uint8)
torch.cuda.synchronize()
cell = wsb[off[1]:off[1] + total * 8].view(torch.int32).view(total, 2).cpu().numpy()
hdr = wsb[off[0]:off[0] + 4].view(torch.int32).cpu().numpy()
cnt = cell[:, 1].astype(np.int64)
print('tiles %d: walking the coarse bin %d (%.1f %%), empty %d, occupied %d; faces per occupied tile mean %.1f, p50 %d, p90 %d, p99 %d, max %d; '
      'pool used %d of %d entries' % (total, (cnt < 0).sum(), 100.0 * (cnt < 0).mean(), (cnt == 0).sum(), (cnt > 0).sum(), cnt[cnt > 0].mean(),
                                      np.percentile(cnt[cnt > 0], 50), np.percentile(cnt[cnt > 0], 90), np.percentile(cnt[cnt > 0], 99), cnt.max(), hdr[0], off[5]))
for flags in (0, 4096, 0, 4096):
    lib.dbw_debug_set_flags(flags)
    kb = bench.kernel_breakdown(model, inp, reps=3)
    print('flags %5d:' % flags, {k.replace('render_', '').replace('_fused', ''): round(v[0], 4) for k, v in kb.items()})
lib.dbw_debug_set_flags(0)
