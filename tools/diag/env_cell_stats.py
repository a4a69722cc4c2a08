"""GPU helper: per-tile face-list statistics of the ENV scene (sky dome + ground) on 8x8 tiles, and how many of its clipped faces carry the
REC_CULL flag (the conservative tile-vs-edge test of the binning only applies to those).  usage: env_cell_stats.py [views]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib, ops
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = int(sys.argv[1]) if len(sys.argv) > 1 else 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(0)
model._ensure_cameras(inp)
lib = _lib.load()
B, H, W = args.views, args.H, args.W
with torch.no_grad():
    scene = model.build_env_scene()
r = model.renderer_env
cfg = r._cfg(scene.faces.shape[0], lds_aggregate=True, const_faces=getattr(scene, 'const_faces', 0))
Kmat = r.cameras.K[0].contiguous()
cl = ops.project_clip(scene.verts.detach(), scene.faces, inp['R'], inp['T'], Kmat, cfg.eps, cfg.z_clip, cfg.persp)
fvc = cl['face_verts'].view(-1, 3, 3)
lib.dbw_debug_set_render_variant(1)          # 8x8 tiles for the hard pass: per-tile lists are built
mode = ops.hard_layout(cfg, None, scene.map_desc)
state = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), None, r._bg, mode, stage=1)
lib.dbw_debug_set_render_variant(0)
ws = state[0]
off = (ctypes.c_ulonglong * 6)()
lib.dbw_debug_cell_layout(ctypes.c_int64(fvc.shape[0]), B, H, W, off)
tiles = ((H + 7) // 8) * ((W + 7) // 8)
total = B * tiles
wsb = ws.view(torch.uint8)
torch.cuda.synchronize()
cell = wsb[off[1]:off[1] + total * 8].view(torch.int32).view(total, 2).cpu().numpy()
cnt = cell[:, 1].astype(np.int64)
F = fvc.shape[0]
rec0 = (F * 16 + 255) // 256 * 256
recs = wsb[rec0:rec0 + F * 128].view(torch.int32).view(F, 32).cpu().numpy()
nf = cl['num_faces'].cpu().numpy()
first = cl['first_idx'].cpu().numpy() if 'first_idx' in cl else np.arange(B) * (F // B)
live = np.zeros(F, bool)
for b in range(B):
    live[first[b]:first[b] + nf[b]] = True
flags = recs[:, 28]
alive = live & (recs.view(np.float32)[:, 23] <= recs.view(np.float32)[:, 24])      # xlo <= xhi: not a dead face
print(f'env scene: {B} views, {int(nf.sum())} clipped faces, {int(alive.sum())} of them alive; REC_CULL on {int(((flags & 32) != 0)[alive].sum())} '
      f'({100.0 * ((flags & 32) != 0)[alive].mean():.1f} %), REC_FAST on {100.0 * ((flags & 1) != 0)[alive].mean():.1f} %')
hist = np.bincount(np.clip(cnt[cnt >= 0], 0, 12))
print('faces per tile list: ' + ', '.join(f'{i}{"+" if i == 12 else ""}: {100.0 * h / total:.1f} %' for i, h in enumerate(hist)) +
      f'; walking the coarse bin: {100.0 * (cnt < 0).mean():.2f} %; mean {cnt[cnt >= 0].mean():.2f}')
