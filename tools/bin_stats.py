"""GPU profiling helper (not product code): same-texel multiplicity of the texture-bin records of the fg pass."""
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
model.set_cur_epoch(800)
model(inp, None)
B, H, W = 49, 300, 400
r = model.renderer
scene = model.build_blocks_scene()
alpha = model._alpha.detach().repeat_interleave(model.BNF).contiguous()
cfg = r._cfg(scene.faces.shape[0], lds_aggregate=False)
K = cfg.K
Kmat = r.cameras.K[0].contiguous()
verts, maps = scene.verts.detach(), scene.maps.detach()
cl = ops.project_clip(verts, scene.faces, inp['R'], inp['T'], Kmat, cfg.eps, cfg.z_clip, cfg.persp)
fvc = cl['face_verts'].view(-1, 3, 3)
p2f, bary, dists, img = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, maps, alpha, r._bg, 2)
g_img = torch.rand_like(img); g_maps, g_fvc, g_alpha = torch.zeros_like(maps), torch.zeros_like(fvc), torch.zeros_like(alpha)
bins = scene.texbins; nbins = bins[2]
cap = ops.texbin_capacity(B, H, W, K, nbins)
cursor = torch.zeros(nbins * ops.BIN_SUBCURSORS, dtype=torch.int32, device=dev); records = torch.zeros(nbins * cap * 8, dtype=torch.int32, device=dev)
_lib.call('dbw_render_bwd_fused', *ops._shade_args(p2f, bary, dists, cl, scene.face_uvs, scene.face_map, scene.map_desc, maps, alpha, cfg.F,
          cfg.sigma, r._bg, (B, H, W, K)), g_img.data_ptr(), fvc.data_ptr(), int(cfg.persp), int(cfg.detach_bary), g_maps.data_ptr(),
          g_alpha.data_ptr(), g_fvc.data_ptr(), 0, 2, bins[0].data_ptr(), cursor.data_ptr(), records.data_ptr(), cap, ops.uniform_bin_layout(nbins, cap, dev).data_ptr(), 0, 0, 0, ops._stream(fvc))
torch.cuda.synchronize()
c = cursor.view(nbins, -1).clamp(max=cap // ops.BIN_SUBCURSORS).sum(1).cpu()        # (records sit in BIN_SUBCURSORS sub-ranges of each bin)
print('records', int(c.sum()), 'max', int(c.max()), 'nonempty bins', int((c > 0).sum()), 'top10', sorted(c.tolist())[-10:])
rec = records.view(nbins, cap, 8)
tot = mult = 0
for b in torch.argsort(c, descending=True)[:20].tolist():
    n = int(c[b]) // 64 * 64
    key = (rec[b, :n, 0] & 1023).view(-1, 64)
    srt = key.sort(dim=1).values
    uniq = 1 + (srt[:, 1:] != srt[:, :-1]).sum(1)
    # serialisation of a 64-lane LDS atomic = max multiplicity of one address
    mx = torch.stack([(key == key[:, i:i + 1]).sum(1) for i in range(64)], 1).max(1).values
    print('bin', b, 'n', n, 'unique texels / 64 records: mean', float(uniq.float().mean()), 'max multiplicity mean', float(mx.float().mean()))
