"""GPU tuning helper: fused backward kernel (fg pass of the bench config) under fragment layouts / pixel mappings / aggregation."""
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
cfg = r._cfg(scene.faces.shape[0], lds_aggregate=True)
cl = ops.project_clip(scene.verts.detach(), scene.faces, inp['R'], inp['T'], r.cameras.K[0].contiguous(), cfg.eps, cfg.z_clip, cfg.persp)
fvc = cl['face_verts'].view(-1, 3, 3)
alpha = model._alpha.detach().repeat_interleave(model.BNF).contiguous()
maps = scene.maps.detach()
B, H, W, K = 49, 300, 400, 10
for tiled in (1, 0):
    p2f, bary, dists, img = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, maps, alpha, r._bg, bool(tiled))
    g_img = torch.rand(img.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    for flags, name in ((0, 'quadrant map'), (32, 'strip map'), (1, 'no texel agg'), (2, 'no alpha agg'), (3, 'no tex/alpha')):
        lib.dbw_debug_set_flags(flags)
        g_maps, g_fvc, g_alpha = torch.zeros_like(maps), torch.zeros_like(fvc), torch.zeros_like(alpha)
        def bwd():
            _lib.call('dbw_render_bwd_fused', *ops._shade_args(p2f, bary, dists, cl, scene.face_uvs, scene.face_map, scene.map_desc, maps, alpha, cfg.F,
                      cfg.sigma, r._bg, (B, H, W, K)), g_img.data_ptr(), fvc.data_ptr(), 1, 1, g_maps.data_ptr(), g_alpha.data_ptr(), g_fvc.data_ptr(), 1, tiled, 0, 0, 0, 0, 0, 0, 0, 0, ops._stream(fvc))        # (no texture bins, no layout, const faces 0, no grad scale, image_layout 0)
        bwd(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): bwd()
        e1.record(); torch.cuda.synchronize()
        print('tiled=%d %-14s %.3f ms' % (tiled, name, e0.elapsed_time(e1) / 10))
lib.dbw_debug_set_flags(0)
