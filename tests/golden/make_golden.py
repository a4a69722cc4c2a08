"""
Generates tests/golden/*.npz by importing the REAL reference functions from /root/reference/src
(read-only; nothing is written there, bytecode writing is disabled) with sys.modules stubs for the
third-party packages this image lacks (pytorch3d, lpips, torchvision, toolz, cv2, ...).

Run here (the container that has /root/reference):   python tests/golden/make_golden.py
The fixtures are data only (inputs + expected outputs); no reference source travels.

Reference functions exercised (the only ones on the hot path that are the reference's OWN torch code,
SURVEY.md 8c):
  model/renderer.py:241-273   layered_rgb_blend
  utils/superquadric.py:10-14 parametric_sq          utils/superquadric.py:17-38 implicit_sq(as_sdf=2)
  utils/pytorch.py:31-36      signed_pow / safe_pow
  model/loss.py:43-47         tv_norm_funcs['l2sq'] (implicit_misc.npz); all three norms + the criteria of loss.py:11-24 (criteria.npz)
  model/tools.py:173-207      elev/azim/roll_to_rotation_matrix (R_world, dbw.py:59)
  model/loss.py:28-29,124-156 mse2psnr, SSIMLoss (evaluation metrics)
  utils/mesh.py:78-89,127-169 point_to_uv_sphericalmap, get_icosphere_uvs post-processing
                              (fed an icosphere restated per SURVEY A.9 -- PyTorch3D's ico_sphere is absent)
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

REF = '/root/reference/src'
HERE = os.path.dirname(os.path.abspath(__file__))
STUB_ROOTS = ['pytorch3d', 'lpips', 'torchvision', 'toolz', 'cv2', 'trimesh', 'open3d', 'seaborn', 'imageio',
              'iopath', 'fvcore', 'visdom', 'nerfstudio', 'matplotlib', 'sklearn', 'pandas', 'scipy', 'tqdm', 'PIL']


class _AnyAttr(type):
    def __getattr__(cls, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return None


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        cls = _AnyAttr(name, (), {'__init__': lambda self, *a, **k: None, '__call__': lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


def import_reference():
    for k in list(sys.modules):
        if k.split('.')[0] in STUB_ROOTS:
            del sys.modules[k]
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF)
    mods = {}
    for name in ['utils.pytorch', 'utils.superquadric', 'utils.mesh', 'model.renderer', 'model.loss', 'model.tools', 'scheduler', 'optimizer']:
        mods[name] = importlib.import_module(name)
    return mods


def _np(t):
    return t.detach().cpu().numpy()


def main():
    sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
    import oracle as O     # only for the restated icosphere fed to the reference's uv post-processing
    m = import_reference()
    ren, sq, pt, mesh, loss, tools = (m['model.renderer'], m['utils.superquadric'], m['utils.pytorch'],
                                      m['utils.mesh'], m['model.loss'], m['model.tools'])
    g = torch.Generator().manual_seed(227391)

    # ---- 1. layered_rgb_blend: sigma in {1e-4, 5e-6, 0}, faces_alpha on/off, with grads ------------------
    class Frag:
        pass

    class BP:
        pass
    N, H, W, K, Ff = 2, 6, 7, 5, 12
    out = {}
    for tag, sigma, use_alpha in [('s1e-4_a', 1e-4, True), ('s1e-4', 1e-4, False), ('s5e-6_a', 5e-6, True),
                                  ('s0', 0.0, False), ('s0_a', 0.0, True)]:
        p2f = torch.randint(-1, N * Ff, (N, H, W, K), generator=g)
        p2f[torch.rand(N, H, W, K, generator=g) < 0.3] = -1
        scale = max(sigma, 1e-5) * 3
        dists = (torch.randn(N, H, W, K, generator=g) * scale).requires_grad_(True)
        colors = torch.rand(N, H, W, K, 3, generator=g).requires_grad_(True)
        fa = torch.rand(N * Ff, generator=g).requires_grad_(True) if use_alpha else None
        fr = Frag()
        fr.pix_to_face, fr.dists = p2f, dists
        bp = BP()
        bp.sigma, bp.background_color = sigma, (0.2, 0.3, 0.4) if 'a' in tag else (0, 0, 0)
        res = ren.layered_rgb_blend(colors, fr, bp, clip_inside=True, faces_alpha=fa)
        w = torch.rand(res.shape, generator=g)
        (res * w).sum().backward()
        out.update({f'{tag}/p2f': _np(p2f), f'{tag}/dists': _np(dists), f'{tag}/colors': _np(colors),
                    f'{tag}/sigma': np.float64(sigma), f'{tag}/bg': np.array(bp.background_color, dtype=np.float32),
                    f'{tag}/out': _np(res), f'{tag}/w': _np(w), f'{tag}/g_colors': _np(colors.grad)})
        if sigma > 0:
            out[f'{tag}/g_dists'] = _np(dists.grad)
        if use_alpha:
            out[f'{tag}/faces_alpha'] = _np(fa)
            out[f'{tag}/g_faces_alpha'] = _np(fa.grad)
    np.savez_compressed(os.path.join(HERE, 'blend.npz'), **out)

    # ---- 1b. the same with clip_inside=False: sigmoid(-d / sigma) instead of exp(-max(d, 0) / sigma) (renderer.py:257-258; no shipped config
    # sets it, the Renderer's constructor takes it) -- a file of its own, drawn from a generator of its own, so that blend.npz stays as it is
    g2 = torch.Generator().manual_seed(20240905)
    out = {}
    for tag, sigma, use_alpha in [('sig1e-4_a', 1e-4, True), ('sig1e-4', 1e-4, False), ('sig5e-6_a', 5e-6, True)]:
        p2f = torch.randint(-1, N * Ff, (N, H, W, K), generator=g2)
        p2f[torch.rand(N, H, W, K, generator=g2) < 0.3] = -1
        dists = (torch.randn(N, H, W, K, generator=g2) * sigma * 3).requires_grad_(True)
        colors = torch.rand(N, H, W, K, 3, generator=g2).requires_grad_(True)
        fa = torch.rand(N * Ff, generator=g2).requires_grad_(True) if use_alpha else None
        fr = Frag()
        fr.pix_to_face, fr.dists = p2f, dists
        bp = BP()
        bp.sigma, bp.background_color = sigma, (0.2, 0.3, 0.4) if use_alpha else (0, 0, 0)
        res = ren.layered_rgb_blend(colors, fr, bp, clip_inside=False, faces_alpha=fa)
        w = torch.rand(res.shape, generator=g2)
        (res * w).sum().backward()
        out.update({f'{tag}/p2f': _np(p2f), f'{tag}/dists': _np(dists), f'{tag}/colors': _np(colors),
                    f'{tag}/sigma': np.float64(sigma), f'{tag}/bg': np.array(bp.background_color, dtype=np.float32),
                    f'{tag}/out': _np(res), f'{tag}/w': _np(w), f'{tag}/g_colors': _np(colors.grad), f'{tag}/g_dists': _np(dists.grad)})
        if use_alpha:
            out[f'{tag}/faces_alpha'] = _np(fa)
            out[f'{tag}/g_faces_alpha'] = _np(fa.grad)
    np.savez_compressed(os.path.join(HERE, 'blend_sigmoid.npz'), **out)

    # ---- 2. parametric_sq on icosphere-1 angles, eps grid incl. grads ----------------------------------
    v1, _ = O.get_icosphere(1)
    eta, omega = torch.asin(v1[:, 1]), torch.atan2(v1[:, 0], v1[:, 2])
    out = {'eta': _np(eta), 'omega': _np(omega)}
    for i, e1 in enumerate([0.1, 0.5, 1.0, 1.9]):
        for j, e2 in enumerate([0.1, 0.5, 1.0, 1.9]):
            eps1 = torch.tensor([[e1]], requires_grad=True)
            eps2 = torch.tensor([[e2]], requires_grad=True)
            pts = sq.parametric_sq(eta[None], omega[None], eps1, eps2)
            w = torch.rand(pts.shape, generator=g)
            (pts * w).sum().backward()
            out.update({f'{i}{j}/eps': np.array([e1, e2], dtype=np.float32), f'{i}{j}/pts': _np(pts), f'{i}{j}/w': _np(w),
                        f'{i}{j}/g_eps1': _np(eps1.grad), f'{i}{j}/g_eps2': _np(eps2.grad)})
    np.savez_compressed(os.path.join(HERE, 'parametric_sq.npz'), **out)

    # ---- 3. implicit_sq(as_sdf=2), safe_pow, signed_pow, tv l2sq -----------------------------------------
    Nb, Np = 3, 64
    pts = (torch.randn(Nb, Np, 3, generator=g) * 2)
    pts[0, :4] = torch.tensor([[6., -7., 0.], [0., 0., 0.], [1e-4, 0., 5.], [-5., 5., -5.]])
    pts.requires_grad_(True)
    eps1 = torch.tensor([[0.1], [1.0], [1.9]], requires_grad=True)
    eps2 = torch.tensor([[1.9], [0.7], [0.1]], requires_grad=True)
    sdf = sq.implicit_sq(pts, eps1, eps2, as_sdf=2)
    w = torch.rand(sdf.shape, generator=g)
    (sdf * w).sum().backward()
    a = torch.rand(16, generator=g)
    a[:3] = torch.tensor([0., 1e-7, 1.])
    a.requires_grad_(True)
    sp = pt.safe_pow(a, 0.5)
    sp.sum().backward()
    t = torch.randn(32, generator=g)
    maps = torch.rand(2, 8, 9, 3, generator=g).requires_grad_(True)
    dx = loss.tv_norm_funcs['l2sq'](torch.diff(maps, dim=2, append=maps[:, :, 0:1]))
    dy = loss.tv_norm_funcs['l2sq'](torch.diff(maps, dim=1))
    tv = dx.sum(0).mean() + dy.sum(0).mean()
    tv.backward()
    np.savez_compressed(os.path.join(HERE, 'implicit_misc.npz'), pts=_np(pts), eps1=_np(eps1), eps2=_np(eps2),
                        sdf=_np(sdf), w=_np(w), g_pts=_np(pts.grad), g_eps1=_np(eps1.grad), g_eps2=_np(eps2.grad),
                        sp_in=_np(a), sp_out=_np(sp), sp_grad=_np(a.grad),
                        spow_in=_np(t), spow_out=_np(pt.signed_pow(t, torch.tensor(0.37))),
                        tv_maps=_np(maps), tv=_np(tv), tv_grad=_np(maps.grad))

    # ---- 4. Euler -> R_world (dbw.py:59) for DTU [115,0,0] and BMVS [130,50,0] ----------------------------
    out = {}
    for tag, (e, a_, r) in {'dtu': (115, 0, 0), 'bmvs': (130, 50, 0), 'mix': (20, -35, 70)}.items():
        Rw = tools.elev_to_rotation_matrix(e) @ tools.azim_to_rotation_matrix(a_) @ tools.roll_to_rotation_matrix(r)
        out[tag] = _np(Rw)
        out[tag + '_angles'] = np.array([e, a_, r], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, 'world_rotation.npz'), **out)

    # ---- 5. get_icosphere_uvs post-processing (mesh.py:127-169) on the restated icosphere ---------------
    out = {}
    for level in [1, 2]:
        mesh.get_icosphere = lambda level, **k: types.SimpleNamespace(
            get_mesh_verts_faces=lambda i, _l=level: O.get_icosphere(_l))
        f_uv, v_uv = mesh.get_icosphere_uvs(level=level, fix_continuity=True, fix_poles=True)
        f_raw, v_raw = mesh.get_icosphere_uvs(level=level)
        out.update({f'l{level}/faces_uvs': _np(f_uv), f'l{level}/verts_uvs': _np(v_uv), f'l{level}/verts_uvs_raw': _np(v_raw)})
    np.savez_compressed(os.path.join(HERE, 'icosphere_uvs.npz'), **out)
    # ---- 6. MultiStepLR (scheduler.py:26-69) stepped per epoch as trainer.py:127,163-169 does, Adam groups of optimizer.py:6-18
    sch, opt = m['scheduler'], m['optimizer']

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.S = torch.nn.Parameter(torch.zeros(2))
            self.textures = torch.nn.Parameter(torch.zeros(3))
            self.texture_bkg = torch.nn.Parameter(torch.zeros(1))
    out = {}
    for tag, kwargs, n_ep in [('dtu', dict(gamma=[0.1, 0.1], milestones=[1700]), 1800),
                              ('warm', dict(gamma=[0.5, 0.1], milestones=[5, 9, 9], warmup=3), 14),
                              ('scalar_gamma', dict(gamma=0.3, milestones=[2, 4]), 7)]:
        model = Tiny()
        cfg = {'training': {'optimizer': {'name': 'adam', 'lr': 5.0e-3, 'texture': {'lr': 5.0e-2}}}}
        optimizer = opt.create_optimizer(cfg, model)
        assert [len(g['params']) for g in optimizer.param_groups] == [1, 2]
        scheduler = sch.MultiStepLR(optimizer, **kwargs)
        lrs = [[g['lr'] for g in optimizer.param_groups]]
        for _ in range(n_ep):
            optimizer.step()
            scheduler.step()
            lrs.append([g['lr'] for g in optimizer.param_groups])
        out[tag] = np.array(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'lr_schedule.npz'), **out)

    # ---- 7. evaluation metrics of quantitative_eval (dbw.py:464-493): SSIMLoss (loss.py:124-156) with and without padding,
    #         mse2psnr (loss.py:28-29)
    loss = m['model.loss']
    g = torch.Generator().manual_seed(7)
    a = torch.rand(3, 3, 40, 52, generator=g)
    b = (a + 0.1 * torch.randn(3, 3, 40, 52, generator=g)).clamp(0, 1)
    out = {'img1': _np(a), 'img2': _np(b)}
    for pad in (False, True):
        f = loss.SSIMLoss(padding=pad)
        out[f'one_minus_ssim_pad{int(pad)}'] = _np(f(a, b))          # per image: mean(1 - ssim_map)
        out[f'ssim_map_pad{int(pad)}'] = _np(1 - f.ssim(a, b))
    mse = torch.tensor([1e-4, 3.7e-3, 0.25])
    out['mse'] = _np(mse)
    out['psnr'] = _np(loss.mse2psnr(mse))
    np.savez_compressed(os.path.join(HERE, 'ssim.npz'), **out)

    # ---- 8. the non-default keys of the loss registry the model's config accepts (loss.py:11-24: l1, huber next to mse / l2; loss.py:43-47:
    #         tv_norm_funcs l1 / l2 next to l2sq): the criteria on an image pair as dbw.py:367 calls them, the TV term as dbw.py:378-387
    #         assembles it from the norms (sky map, wrapped block maps, ground map x factor), values and gradients.  Its own generator: the
    #         sections above keep their draws
    g = torch.Generator().manual_seed(8)
    out = {}
    imgs = torch.rand(2, 3, 9, 11, generator=g)
    rec0 = (imgs + 0.3 * torch.randn(2, 3, 9, 11, generator=g)).clamp(0, 1)
    rec0[0, :, :2] += 1.7                           # (differences above 1: the quadratic / linear switch of the Huber criterion)
    out['crit_imgs'], out['crit_rec'] = _np(imgs), _np(rec0)
    for name in ('mse', 'l2', 'l1', 'huber'):
        rec = rec0.clone().requires_grad_(True)
        val = loss.get_loss(name)()(imgs, rec)
        val.backward()
        out[f'crit_{name}'], out[f'crit_{name}_grad'] = _np(val), _np(rec.grad)
    bkg, blocks, ground = (torch.rand(1, 6, 7, 3, generator=g), torch.rand(3, 5, 8, 3, generator=g), torch.rand(1, 4, 6, 3, generator=g))
    blocks[0, 2, 3] = blocks[0, 2, 4]               # a zero difference: where safe_pow's clamp decides the l2 norm's gradient
    out['tv_bkg'], out['tv_blocks'], out['tv_ground'] = _np(bkg), _np(blocks), _np(ground)
    for name in ('l1', 'l2', 'l2sq'):
        norm = loss.tv_norm_funcs[name]
        b_, m_, g_ = (t.clone().requires_grad_(True) for t in (bkg, blocks, ground))
        tv = sum([norm(torch.diff(b_, dim=k)).mean() for k in [1, 2]])
        dx = norm(torch.diff(m_, dim=2, append=m_[:, :, 0:1]))
        dy = norm(torch.diff(m_, dim=1))
        tv = tv + (dx.sum(0).mean() + dy.sum(0).mean())
        tv = tv + sum([norm(torch.diff(g_, dim=k)).mean() for k in [1, 2]]) * 0.1
        tv.backward()
        out[f'tv_{name}'] = _np(tv)
        out[f'tv_{name}_g_bkg'], out[f'tv_{name}_g_blocks'], out[f'tv_{name}_g_ground'] = _np(b_.grad), _np(m_.grad), _np(g_.grad)
    np.savez_compressed(os.path.join(HERE, 'criteria.npz'), **out)
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
