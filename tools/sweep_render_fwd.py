"""GPU tuning helper: fused forward kernel (fg pass of the bench config) across tile shapes / bbox-load batching.
Needs a library built with -DDBW_TUNE_VARIANTS (python differentiable-blocksworld_amd/build.py --force -DDBW_TUNE_VARIANTS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib, ops
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.sync_free = True
model(inp, None)
lib = _lib.load()
with torch.no_grad():
    scene = model.build_blocks_scene()
r = model.renderer
cfg = r._cfg(scene.faces.shape[0])
cl = ops.project_clip(scene.verts.detach(), scene.faces, inp['R'], inp['T'], r.cameras.K[0].contiguous(), cfg.eps, cfg.z_clip, cfg.persp)
fvc = cl['face_verts'].view(-1, 3, 3)
alpha = model._alpha.detach().repeat_interleave(model.BNF).contiguous()
run = lambda: ops._render_fwd_fused(fvc, cl, 49, cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), alpha, r._bg)
names = {0: '8x8 G4 (default)', 1: '8x8 G1', 2: '8x8 G2', 3: '16x16 G1', 4: '16x16 G4', 5: '16x8 G2', 6: '8x16 G2'}
ref = None
for v in range(7):
    lib.dbw_debug_set_render_variant(v)
    out = run(); torch.cuda.synchronize()
    if ref is None: ref = out
    assert torch.equal(out[0], ref[0]) and torch.equal(out[3], ref[3])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print('%-18s %.3f ms' % (names[v], e0.elapsed_time(e1) / 10))
