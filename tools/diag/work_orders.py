"""GPU helper: the fused forward of the fg pass (stage 2 only, HIP events) under launch orders written by this script into the pass's
workspace (dbw_debug_cell_layout): what the order of the tiles is worth.  usage: work_orders.py [epoch]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib, ops
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
model(inp, None)
lib = _lib.load()
B, H, W = args.views, args.H, args.W
with torch.no_grad():
    scene = model.build_blocks_scene(filter_transparent=False)
    alpha = model._alpha.detach().repeat_interleave(model.BNF).contiguous()
r = model.renderer
cfg = r._cfg(scene.faces.shape[0], lds_aggregate=True)
Kmat = r.cameras.K[0].contiguous()
verts, maps = scene.verts.detach(), scene.maps.detach()
cl = ops.project_clip(verts, scene.faces, inp['R'], inp['T'], Kmat, cfg.eps, cfg.z_clip, cfg.persp)
fvc = cl['face_verts'].view(-1, 3, 3)
state = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, maps, alpha, r._bg, 2, stage=1)
ws = state[0]
fwd = lambda: ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, maps, alpha, r._bg, 2, stage=2, state=state)
off = (ctypes.c_ulonglong * 6)()
lib.dbw_debug_cell_layout(ctypes.c_int64(fvc.shape[0]), B, H, W, off)
tiles = ((H + 7) // 8) * ((W + 7) // 8)
total = B * tiles
wsb = ws.view(torch.uint8)
cell = wsb[off[1]:off[1] + total * 8].view(torch.int32).view(total, 2)
work = wsb[off[2]:off[2] + total * 4].view(torch.int32)
torch.cuda.synchronize()
cnt = cell[:, 1].cpu().numpy().astype(np.int64)
product = work.cpu().numpy().copy()
per = (total + 7) // 8
print('tiles %d, occupied %d, faces per occupied tile %.1f, max %d' % (total, (cnt > 0).sum(), cnt[cnt > 0].mean(), cnt.max()))

def t(reps=10):
    fwd(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fwd()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def spread(occ, emp, n):
    """occupied tiles (in the given order) take len(occ) slots spread evenly over n positions, the empty ones the rest"""
    out = np.empty(n, dtype=np.int64)
    O = len(occ)
    slot = np.zeros(n, dtype=bool)
    if O:
        p = ((np.arange(O) + 1) * n + O - 1) // O - 1
        slot[p] = True
        out[p] = occ
    out[~slot] = emp
    return out

def build(kind, rng):
    out = np.empty(total, dtype=np.int64)
    for x in range(8):
        L = np.arange(x * per, min((x + 1) * per, total))
        c = cnt[L]
        occ, emp = L[c > 0], L[c == 0]
        if kind == 'raster':
            o = L
        elif kind == 'occupied first, raster':
            o = np.concatenate([occ, emp])
        elif kind == 'raster, empties spread':
            o = spread(occ, emp, len(L))
        elif kind == 'heavy first (exact sort), empties spread':
            o = spread(occ[np.argsort(-cnt[occ], kind='stable')], emp, len(L))
        elif kind == 'heavy first (exact sort), empties last':
            o = np.concatenate([occ[np.argsort(-cnt[occ], kind='stable')], emp])
        elif kind == 'classes, raster inside, empties spread':
            cls = np.digitize(cnt[occ], [4, 8, 12, 16, 24, 32, 48, 64])
            o = spread(occ[np.argsort(-cls, kind='stable')], emp, len(L))
        elif kind == 'classes, random inside, empties spread':
            cls = np.digitize(cnt[occ], [4, 8, 12, 16, 24, 32, 48, 64]).astype(np.float64) + rng.random(len(occ)) * 0.5
            o = spread(occ[np.argsort(-cls, kind='stable')], emp, len(L))
        elif kind == 'two classes (>= 16 first), raster inside, empties spread':
            cls = (cnt[occ] >= 16).astype(np.int64)
            o = spread(occ[np.argsort(-cls, kind='stable')], emp, len(L))
        elif kind == 'random':
            o = rng.permutation(L)
        elif kind == 'light first (worst case)':
            o = np.concatenate([emp, occ[np.argsort(cnt[occ], kind='stable')]])
        out[x * per:x * per + len(L)] = o
    assert np.array_equal(np.sort(out), np.arange(total))
    return out

rng = np.random.default_rng(0)
print('%-62s %.4f ms' % ('product (work_scatter_kernel)', t()))
kinds = ['raster', 'occupied first, raster', 'raster, empties spread', 'heavy first (exact sort), empties spread', 'heavy first (exact sort), empties last',
         'classes, raster inside, empties spread', 'classes, random inside, empties spread', 'two classes (>= 16 first), raster inside, empties spread', 'random',
         'light first (worst case)']
for rep in range(2):
    for kind in kinds:
        work.copy_(torch.from_numpy(build(kind, rng).astype(np.int32)).to(dev))
        print('%-62s %.4f ms' % (kind, t()))
work.copy_(torch.from_numpy(product).to(dev))
print('%-62s %.4f ms' % ('product again', t()))
