"""The fg forward of every view on its own in a slow scene state, and the slowest view again with single faces taken out of the scene.
usage: r06_spike5.py slow_step"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import ops
from dbw_amd.parallel import ShardedTrainStep
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(a, dev)
model.set_cur_epoch(0); model.sync_free = True
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391)
slow_i = int(sys.argv[1])
for i in range(slow_i + 1):
    if i == slow_i: snap = step.params.flat.clone()
    step(inp)
torch.cuda.synchronize()
step.params.flat.copy_(snap)
with torch.no_grad():
    scene = model.build_blocks_scene(filter_transparent=False)
    alpha = model._alpha.detach().repeat_interleave(model.BNF).contiguous()
r = model.renderer
cfg = r._cfg(scene.faces.shape[0], lds_aggregate=True)
Kmat = r.cameras.K[0].contiguous()
def fwd_ms(verts, R, T, reps=5):
    cl = ops.project_clip(verts, scene.faces, R, T, Kmat, cfg.eps, cfg.z_clip, cfg.persp)
    fvc = cl['face_verts'].view(-1, 3, 3)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops._render_fwd_fused(fvc, cl, R.shape[0], cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), alpha, r._bg, 2)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, fvc
verts = scene.verts.detach()
print('all views: %.3f ms' % fwd_ms(verts, inp['R'], inp['T'])[0])
per = [fwd_ms(verts, inp['R'][v:v + 1].contiguous(), inp['T'][v:v + 1].contiguous())[0] for v in range(a.views)]
print('view by view (ms):', ' '.join('%.3f' % t for t in per))
w = max(range(a.views), key=lambda v: per[v])
R1, T1 = inp['R'][w:w + 1].contiguous(), inp['T'][w:w + 1].contiguous()
t_w, fvc = fwd_ms(verts, R1, T1)
v = fvc[:scene.faces.shape[0]].float()
x, y = v[:, :, 0], v[:, :, 1]
area = ((x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0]) - (y[:, 2] - y[:, 0]) * (x[:, 1] - x[:, 0])).abs()
small = torch.argsort(area)[:6]
print('slowest view %d: %.3f ms; its smallest faces (index: |area|):' % (w, t_w), ' '.join('%d:%.3g' % (int(f), float(area[f])) for f in small))
for f in small:
    # the face degenerated on purpose (its third vertex moved onto its first: zero area, dead in every view) -- through a copy of the vertex
    # array in which that face gets vertices of its own
    vv = torch.cat([verts, verts[scene.faces[f].long()]], 0)
    faces2 = scene.faces.clone()
    nv = verts.shape[0]
    faces2[f] = torch.tensor([nv, nv + 1, nv], device=dev, dtype=faces2.dtype)
    cl = ops.project_clip(vv, faces2, R1, T1, Kmat, cfg.eps, cfg.z_clip, cfg.persp)
    f2 = cl['face_verts'].view(-1, 3, 3)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops._render_fwd_fused(f2, cl, 1, cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), alpha, r._bg, 2)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    print('   without face %d: %.3f ms' % (int(f), best))
