"""GPU profiling helper: rasteriser forward with and without its global stores (compute floor)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch
import bench
from dbw_amd import _lib, ops

class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model(inp, None)
lib = _lib.load()
with torch.no_grad():
    scene = model.build_blocks_scene()
r = model.renderer
cfg = r._cfg(scene.faces.shape[0])
cl = ops.project_clip(scene.verts.detach(), scene.faces, inp['R'], inp['T'], r.cameras.K[0].contiguous(), cfg.eps, cfg.z_clip, cfg.persp)
fvc, nb = cl['face_verts'].view(-1, 3, 3), cl['neighbor'].view(-1)
def run():
    return ops._raster_fwd(fvc, cl['first_idx'], cl['num_faces'], nb, 49, 300, 400, 10, cfg.blur, True, True, False, need_zbuf=False)
p2f = run()[0]
valid = (p2f >= 0)
print('valid fragment slots: %.3f of P*K; pixels with >=1: %.3f; mean valid per covered pixel: %.2f' % (
    valid.float().mean().item(), valid[..., 0].float().mean().item(), valid.sum().item() / max(valid[..., 0].sum().item(), 1)))
for flags in (0, 16, 32, 48, 64, 80, 96, 112):   # bits 5-6: tile shape 16x16 / 8x8 / 16x8 / 8x16; 16 = no stores
    lib.dbw_debug_set_flags(flags)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print('flags', flags, 'raster_fwd ms', e0.elapsed_time(e1) / 10)
lib.dbw_debug_set_flags(0)
