"""GPU helper: does the bin reduction hide behind the binned uv backward when the two run side by side?  (fg pass of the bench
configuration at epoch 800; the reduction reads the records of an earlier launch while the backward writes a second set)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib, ops

class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.set_cur_epoch(800)
model(inp, None)
B, H, W = 49, 300, 400
r = model.renderer
with torch.no_grad():
    scene = model.build_blocks_scene(filter_transparent=False)
    alpha = model._alpha.detach().contiguous()
    cfg = r._cfg(scene.faces.shape[0], lds_aggregate=False, const_faces=0)
    K = cfg.K
    Kmat = r.cameras.K[0].contiguous()
    verts, maps = scene.verts.detach(), scene.maps.detach()
    cl = ops.project_clip(verts, scene.faces, inp['R'], inp['T'], Kmat, cfg.eps, cfg.z_clip, cfg.persp)
    fvc = cl['face_verts'].view(-1, 3, 3)
    p2f, bary, dists, img = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, maps, alpha, r._bg, 2)
g_img = torch.rand_like(img)
g_maps, g_fvc = torch.zeros_like(maps), torch.zeros_like(fvc)
g_alpha = torch.zeros(alpha.numel() * ops.ALPHA_SPREAD, device=dev)
bins = scene.texbins; nbins = bins[2]
cap = ops.texbin_capacity(B, H, W, K, nbins)
sets = [(torch.zeros(nbins * ops.BIN_SUBCURSORS, dtype=torch.int32, device=dev), torch.empty(nbins * cap * 8, dtype=torch.int32, device=dev)) for _ in range(2)]
LAY = ops.uniform_bin_layout(nbins, cap, dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def bwd(i, stream):
    cur, rec = sets[i]
    with torch.cuda.stream(stream):
        cur.zero_()
        _lib.call('dbw_render_bwd_fused', *ops._shade_args(p2f, bary, dists, cl, scene.face_uvs, scene.face_map, scene.map_desc, maps, alpha, cfg.F,
                  cfg.sigma, r._bg, (B, H, W, K)), g_img.data_ptr(), fvc.data_ptr(), int(cfg.persp), int(cfg.detach_bary), g_maps.data_ptr(),
                  g_alpha.data_ptr(), g_fvc.data_ptr(), 0, 2, bins[0].data_ptr(), cur.data_ptr(), rec.data_ptr(), cap, LAY.data_ptr(), 0, 0, 0, stream.cuda_stream)

def reduce(i, stream):
    cur, rec = sets[i]
    with torch.cuda.stream(stream):
        _lib.call('dbw_texbin_reduce', bins[1].data_ptr(), cur.data_ptr(), rec.data_ptr(), cap, LAY.data_ptr(), nbins, g_maps.data_ptr(), stream.cuda_stream)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
        s1.synchronize(); s2.synchronize()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

bwd(0, s1); bwd(1, s1); torch.cuda.synchronize()
def both():
    reduce(0, s2); bwd(1, s1)
for rep in range(3):            # (the first round also ramps the clocks up)
    t_b = timed(lambda: bwd(1, s1))
    t_r = timed(lambda: reduce(0, s2))
    t_both = timed(both)
    print('backward alone %.3f ms, reduction alone %.3f ms, side by side %.3f ms (sum %.3f)' % (t_b, t_r, t_both, t_b + t_r))
