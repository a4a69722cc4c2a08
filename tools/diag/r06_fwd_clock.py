"""Time line of ONE launch of the fg forward (operator level, initial scene) from a -DDBW_TILE_CLOCK -DDBW_DIAG build: workgroups in flight on
one XCD over time, run times and start times by list length.  usage: DBW_HIP_LIB=tools/variants/tclk.so r06_fwd_clock.py views H W blocks fpp txt
(at most 2^18 tiles are stamped)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib, ops
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = [int(x) for x in sys.argv[1:7]]
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(a, dev)
model.set_cur_epoch(0)
model(inp, None)
lib = _lib.load()
B, H, W = a.views, a.H, a.W
tx_, ty_ = (W + 7) // 8, (H + 7) // 8
tiles = tx_ * ty_
with torch.no_grad():
    scene = model.build_blocks_scene(filter_transparent=False)
    alpha = model._alpha.detach().repeat_interleave(model.BNF).contiguous()
r = model.renderer
cfg = r._cfg(scene.faces.shape[0], lds_aggregate=True)
Kmat = r.cameras.K[0].contiguous()
cl = ops.project_clip(scene.verts.detach(), scene.faces, inp['R'], inp['T'], Kmat, cfg.eps, cfg.z_clip, cfg.persp)
fvc = cl['face_verts'].view(-1, 3, 3)
for _ in range(2):
    ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), alpha, r._bg, 2)
torch.cuda.synchronize()
NB = min(1 << 18, 8 * ((B * tiles + 7) // 8))
buf = (ctypes.c_uint * (NB * 4))()
lib.dbw_debug_read_tile_clock(buf, NB)
t = np.frombuffer(buf, dtype=np.uint32).reshape(NB, 4)
ok = t[:, 2] != 0
ws = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), alpha, r._bg, 2, stage=1)[0]
off = (ctypes.c_ulonglong * 6)()
lib.dbw_debug_cell_layout(ctypes.c_int64(fvc.shape[0]), B, H, W, off)
torch.cuda.synchronize()
cell = ws.view(torch.uint8)[off[1]:off[1] + B * tiles * 8].view(torch.int32).view(B * tiles, 2).cpu().numpy()
t0 = int(t[ok, 3].astype(np.int64).min())
st, en = (t[:, 3].astype(np.int64) - t0) / 100.0, (t[:, 2].astype(np.int64) - t0) / 100.0
tid = (t[:, 0].astype(np.int64) * ty_ + (t[:, 1] >> 16)) * tx_ + (t[:, 1] & 0xffff)
cnt = cell[np.clip(tid, 0, B * tiles - 1), 1]
m = ok & (np.arange(NB) % 8 == 0)
step = max(10, int(en[m].max() / 30) // 10 * 10)
print('%d tiles, %d stamped; XCD 0: last end %.0f us; workgroups in flight every %d us: %s' % (B * tiles, int(ok.sum()), en[m].max(), step,
      ' '.join(str(int(((st[m] <= a0) & (en[m] > a0)).sum())) for a0 in range(0, int(en[m].max()) + step, step))))
for lo, hi in ((0, 0), (1, 3), (4, 7), (8, 15), (16, 31), (32, 63), (64, 127), (128, 9999)):
    mm = m & (cnt >= lo) & (cnt <= hi)
    if mm.any(): print('   lists of %d-%d faces: %d tiles, run time p50 %.1f p90 %.1f us, started p50 %.1f p90 %.1f us' %
                       (lo, hi, int(mm.sum()), *(np.percentile((en - st)[mm], q) for q in (50, 90)), *(np.percentile(st[mm], q) for q in (50, 90))))
