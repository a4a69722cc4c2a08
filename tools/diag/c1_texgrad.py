"""diagnostic: c1 epoch-0 texture-gradient error vs oracle under debug flags"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'differentiable-blocksworld_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch, oracle as O, dbw_amd
from dbw_amd import _lib
from test_gpu_configs import _cfg, rel_err
DEV = 'cuda:0'
H, W, nb, ts, fpp, V = 75, 100, 4, 256, 4, 4
for flags in (0,):
    _lib.load().dbw_debug_set_flags(flags)
    torch.manual_seed(227391)
    model = dbw_amd.create_model(_cfg(nb, ts, fpp), (H, W))
    orc = O.OracleDBW((H, W), n_blocks=nb, txt_size=ts, faces_per_pixel=fpp, seed=227391)
    with torch.no_grad():
        g = torch.Generator().manual_seed(11)
        for name, scale in (('sq_eps', 1.0), ('alpha_logit', 1.0), ('R_6d_ground', 0.03)):
            d = torch.randn(orc.p[name].shape, generator=g) * scale
            orc.p[name].add_(d); getattr(model, name).add_(d)
        orc.p['T'].mul_(0.6); model.T.mul_(0.6)
    model = model.to(DEV).train(); model.set_cur_epoch(0)
    R, T, Km = O.synthetic_cameras(V, R_world=orc.R_world[0])
    imgs = torch.rand(V, 3, H, W, generator=torch.Generator().manual_seed(2))
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3))
    u = torch.rand(nb, 1000, 3, generator=torch.Generator().manual_seed(4))
    inp = dict(imgs=imgs, R=R, T=T, K=Km)
    ref = orc.forward(inp, training=True, coarse=True, decimate=True, opacity_noise=noise, overlap_points=u, n_threads=16)
    ref['total'].backward()
    model._noise_override, model._overlap_u_override = noise.to(DEV), u.to(DEV)
    out = model({k: v.to(DEV) for k, v in inp.items()}, None)
    out['total'].backward()
    gh, gr = model.textures.grad.cpu(), orc.p['textures'].grad
    d = (gh - gr).abs()
    print('flags', flags, {k: rel_err(getattr(model, k).grad, v.grad) for k, v in orc.p.items() if v.grad is not None})
    i = d.flatten().argmax()
    idx = torch.unravel_index(i, d.shape)
    print('  max diff at', [int(x) for x in idx], float(gh.flatten()[i]), float(gr.flatten()[i]), 'grad absmax', float(gr.abs().max()),
          'n cells differing >1e-6*max:', int((d > 1e-6 * gr.abs().max()).sum()))
# where do the differences sit?
gh, gr = model.textures.grad.cpu(), orc.p['textures'].grad
d = (gh - gr).abs().amax(-1)            # (nb, 256, 256)
cells = d.view(nb, 32, 8, 32, 8).amax(dim=(2, 4))
thr = 1e-4 * gr.abs().max()
bad = (cells > thr).nonzero()
print('cells above 1e-4*max:', bad.shape[0], 'of', cells.numel())
import collections
print('by cell column:', sorted(collections.Counter(bad[:, 2].tolist()).items()))
print('by cell row:', sorted(collections.Counter(bad[:, 1].tolist()).items()))
print('by block:', sorted(collections.Counter(bad[:, 0].tolist()).items()))
# within-cell uniformity of the difference (decimated gradients are uniform per cell up to sigma')
b0 = bad[0].tolist() if bad.shape[0] else None
if b0:
    blk, cr, cc = b0
    print('example cell', b0, 'ours', gh[blk, cr*8:cr*8+2, cc*8:cc*8+2, 0], 'oracle', gr[blk, cr*8:cr*8+2, cc*8:cc*8+2, 0])
