"""GPU parity of the model-level path and of the small kernels around the renderer (param -> mesh, texture prep,
regularisers, composite+MSE, Adam) against the CPU oracle / torch fp32 references.  `-m gpu`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import oracle as O                                              # noqa: E402  (checker only)
import dbw_amd                                                  # noqa: E402
from dbw_amd import ops                                         # noqa: E402
from trajectory import assert_same_trajectory                   # noqa: E402

DEV = 'cuda:0'
REL = 1e-4


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def test_texture_prep_and_decimation():
    torch.manual_seed(0)
    tex = torch.randn(3, 32, 48, 3)
    for d in (1, 8):
        t_ref = tex.clone().requires_grad_(True)
        sig = torch.sigmoid(t_ref)
        maps = sig if d == 1 else F.interpolate(F.avg_pool2d(sig.permute(0, 3, 1, 2), d, stride=d), scale_factor=d).permute(0, 2, 3, 1)
        w1, w2 = torch.rand_like(tex), torch.rand_like(tex)
        ((maps * w1).sum() + (sig * w2).sum()).backward()
        t = tex.to(DEV).requires_grad_(True)
        m_h, s_h = ops.texture_prep(t, d)          # decimated maps come back at cell resolution (nearest upsample = sampler shift)
        assert m_h.shape == (3, 32 // d, 48 // d, 3)
        m_up = m_h if d == 1 else m_h.repeat_interleave(d, 1).repeat_interleave(d, 2)
        ((m_up * w1.to(DEV)).sum() + (s_h * w2.to(DEV)).sum()).backward()
        assert rel_err(m_up, maps) < 1e-6 and rel_err(s_h, sig) < 1e-6
        assert rel_err(t.grad, t_ref.grad) < REL


def test_tv_l2sq_matches_reference_golden(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, 'implicit_misc.npz'))
    m = torch.from_numpy(g['tv_maps']).to(DEV).requires_grad_(True)
    tv = ops.tv_l2sq(m, wrap_x=True)
    tv.backward()
    assert abs(tv.item() - float(g['tv'])) < 1e-5 * abs(float(g['tv']))
    assert rel_err(m.grad, torch.from_numpy(g['tv_grad'])) < REL
    # non-wrapping variant (bkg / ground, dbw.py:380,386)
    mm = torch.rand(1, 16, 20, 3)
    ref = mm.clone().requires_grad_(True)
    tvr = sum(O.tv_l2sq(torch.diff(ref, dim=k)).mean() for k in [1, 2])
    tvr.backward()
    h = mm.to(DEV).requires_grad_(True)
    tvh = ops.tv_l2sq(h, wrap_x=False)
    tvh.backward()
    assert abs(tvh.item() - tvr.item()) < 1e-5 * tvr.item() and rel_err(h.grad, ref.grad) < REL


def test_sq_blocks_and_ground_match_oracle():
    m = O.OracleDBW((8, 8), n_blocks=5, txt_size=8, seed=2)
    with torch.no_grad():
        m.p['sq_eps'].copy_(torch.tensor([[-3., 3.], [0., 0.], [2., -1.], [4., 4.], [-4., -4.]]))   # eps from ~0.1 to ~1.9
    verts = m._world((m.get_blocks_verts() * (m.p['S'].exp() + m.scale_min)[:, None]) @ O.rotation_6d_to_matrix(m.p['R_6d']) + m.p['T'][:, None])
    w = torch.rand(verts.shape, generator=torch.Generator().manual_seed(0))
    (verts * w).sum().backward()
    hp = {k: v.detach().to(DEV).requires_grad_(True) for k, v in m.p.items() if k in ('sq_eps', 'S', 'R_6d', 'T', 'R_6d_ground', 'T_ground')}
    trig = torch.stack([torch.cos(m.sq_eta), torch.sin(m.sq_eta), torch.cos(m.sq_omega), torch.sin(m.sq_omega)], 0).to(DEV)
    Rw, Tw = m.R_world[0].to(DEV).contiguous(), m.T_world[0].to(DEV).contiguous()
    v_h = ops.sq_blocks(hp['sq_eps'], hp['S'], hp['R_6d'], hp['T'], trig, None, 5, m.ratio, m.scale_min, m.S_world, Rw, Tw)
    (v_h * w.to(DEV)).sum().backward()
    assert rel_err(v_h, verts) < 1e-5
    for k in ('sq_eps', 'S', 'R_6d', 'T'):
        assert rel_err(hp[k].grad, m.p[k].grad) < REL, k
    # kept subset is written densely
    keep = torch.tensor([1, 0, 1, 1, 0], dtype=torch.int32, device=DEV)
    v_k = ops.sq_blocks(hp['sq_eps'], hp['S'], hp['R_6d'], hp['T'], trig, keep, 3, m.ratio, m.scale_min, m.S_world, Rw, Tw)
    assert torch.equal(v_k, v_h[keep.bool()])
    # ground plane
    for v in m.p.values():
        v.grad = None
    gv = m._world(m.ground_verts[None] @ O.rotation_6d_to_matrix(m.p['R_6d_ground'] + torch.tensor([[0.1, -0.2, 0.05, 0.3, 0.0, 0.1]])) + m.p['T_ground'][:, None])[0]
    # (perturbed 6D vector so that the Gram-Schmidt backward is exercised off the identity)
    r6 = (m.p['R_6d_ground'].detach() + torch.tensor([[0.1, -0.2, 0.05, 0.3, 0.0, 0.1]])).requires_grad_(True)
    gv = m._world(m.ground_verts[None] @ O.rotation_6d_to_matrix(r6) + m.p['T_ground'][:, None])[0]
    wg = torch.rand(gv.shape, generator=torch.Generator().manual_seed(1))
    (gv * wg).sum().backward()
    r6h = r6.detach().to(DEV).requires_grad_(True)
    g_h = ops.posed_mesh(r6h, hp['T_ground'], m.ground_verts.to(DEV).contiguous(), m.S_world, Rw, Tw)
    (g_h * wg.to(DEV)).sum().backward()
    assert rel_err(g_h, gv) < 1e-5
    assert rel_err(r6h.grad, r6.grad) < REL and rel_err(hp['T_ground'].grad, m.p['T_ground'].grad) < REL


def test_overlap_loss_matches_oracle():
    m = O.OracleDBW((8, 8), n_blocks=4, txt_size=8, seed=4)
    with torch.no_grad():
        m.p['T'].mul_(0.2)                                  # pull the blocks together so that they do overlap
        m.p['sq_eps'].copy_(torch.tensor([[-1., 1.], [0., 0.], [2., -2.], [0.5, 0.3]]))
    m.build_blocks(True, True, False, None, kill_blocks=False)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(0))
    m.loss_weights = dict(overlap=1.0)
    m._bkg_maps = m._ground_maps = None
    ov = m.compute_losses(torch.zeros(1), torch.zeros(1), True, overlap_points=u)['overlap']
    assert ov.item() > 1e-4
    ov.backward()
    hp = {k: v.detach().to(DEV).requires_grad_(True) for k, v in m.p.items() if k in ('sq_eps', 'S', 'R_6d', 'T', 'alpha_logit')}
    alpha = torch.sigmoid(hp['alpha_logit'])
    ov_h = ops.overlap_loss(hp['sq_eps'], hp['S'], hp['R_6d'], hp['T'], alpha, u.to(DEV), m.ratio, m.scale_min)
    ov_h.backward()
    assert abs(ov_h.item() - ov.item()) < REL * ov.item(), (ov_h.item(), ov.item())
    for k in ('sq_eps', 'S', 'R_6d', 'T', 'alpha_logit'):
        assert rel_err(hp[k].grad, m.p[k].grad) < REL, (k, rel_err(hp[k].grad, m.p[k].grad))


def test_composite_mse_and_composite():
    torch.manual_seed(0)
    fg, env, img = torch.rand(2, 4, 9, 11), torch.rand(2, 4, 9, 11), torch.rand(2, 3, 9, 11)
    a, b = fg.clone().requires_grad_(True), env.clone().requires_grad_(True)
    rec = a[:, :3] * a[:, 3:4] + (1 - a[:, 3:4]) * b[:, :3]
    loss = F.mse_loss(img, rec)
    (loss * 3).backward()
    ah, bh = fg.to(DEV).requires_grad_(True), env.to(DEV).requires_grad_(True)
    lh = ops.composite_mse(ah, bh, img.to(DEV))
    (lh * 3).backward()
    assert abs(lh.item() - loss.item()) < 1e-6
    assert rel_err(ah.grad, a.grad) < REL and rel_err(bh.grad[:, :3], b.grad[:, :3]) < REL and torch.all(bh.grad[:, 3] == 0)
    ah.grad = bh.grad = None
    rh = ops.composite(ah, bh)
    assert rel_err(rh, rec) < 1e-6
    (rh * img.to(DEV)).sum().backward()
    a.grad = b.grad = None
    rec = a[:, :3] * a[:, 3:4] + (1 - a[:, 3:4]) * b[:, :3]
    (rec * img).sum().backward()
    assert rel_err(ah.grad, a.grad) < REL


def test_block_alpha_matches_torch():
    torch.manual_seed(0)
    logit, noise = torch.randn(12), torch.randn(12)
    for thresh in (-1.0, 0.5, 0.01):
        l_ref = logit.clone().requires_grad_(True)
        a = torch.sigmoid(l_ref + 0.3 * noise)
        mask = (torch.sigmoid(l_ref) > thresh) if thresh >= 0 else torch.ones(12, dtype=torch.bool)
        af = a * mask
        w1, w2 = torch.rand(12), torch.rand(12)
        ((a * w1).sum() + (af * w2).sum()).backward()
        l_h = logit.to(DEV).requires_grad_(True)
        ah, afh, keep = ops.block_alpha(l_h, noise.to(DEV), 0.3, thresh)
        ((ah * w1.to(DEV)).sum() + (afh * w2.to(DEV)).sum()).backward()
        assert torch.equal(keep.cpu().bool(), mask)
        assert rel_err(ah, a) < 1e-6 and rel_err(afh, af) < 1e-6 and rel_err(l_h.grad, l_ref.grad) < REL


def test_fused_losses_equal_the_operator_level_losses():
    """ops.fused_losses (one autograd node, weights folded into the kernels) against the separately tested operator-level
    losses combined with torch arithmetic, values and every gradient, with a non-trivial upstream gradient per term."""
    torch.manual_seed(1)
    Kb = 4
    mk = lambda *s: torch.rand(*s, device=DEV)
    base = dict(fg=mk(2, 4, 9, 11), env=mk(2, 4, 9, 11), alpha=mk(Kb) * 0.9 + 0.05, bkg=mk(1, 16, 16, 3), blk=mk(Kb, 8, 12, 3),
                gnd=mk(1, 16, 16, 3), sq_eps=mk(Kb, 2) * 0.5, S=mk(Kb, 3) * 0.2, R6=torch.randn(Kb, 6, device=DEV), T=mk(Kb, 3) * 0.4 - 0.2)
    imgs, u = mk(2, 3, 9, 11), mk(Kb, 200, 3)
    wts = dict(rgb=1.0, parsimony=0.01, tv=0.1, overlap=1.0)
    consts = (0.5, 0.2, 0.005, 1.95)
    up = torch.tensor([0.7, 1.3, 2.0, 0.5], device=DEV)

    def leaves():
        return {k: v.clone().requires_grad_(True) for k, v in base.items()}

    a = leaves()
    cfg = {'rgb': wts['rgb'], 'count': float(imgs.numel()), 'parsimony': wts['parsimony'], 'tv': wts['tv'] * 0.1, 'tv_ground_factor': 0.1,
           'overlap': wts['overlap'], 'overlap_consts': consts}
    vals = ops.fused_losses(a['fg'], a['env'], imgs, a['alpha'], a['bkg'], a['blk'], a['gnd'], a['sq_eps'], a['S'], a['R6'], a['T'], u, cfg)
    (vals * up).sum().backward()
    b = leaves()
    ref = torch.stack([
        wts['rgb'] * ops.composite_mse(b['fg'], b['env'], imgs),
        wts['parsimony'] * b['alpha'].clamp(1e-6).pow(0.5).mean(),
        wts['tv'] * 0.1 * (ops.tv_l2sq(b['bkg']) + ops.tv_l2sq(b['blk'], wrap_x=True) + ops.tv_l2sq(b['gnd']) * 0.1),
        wts['overlap'] * ops.overlap_loss(b['sq_eps'], b['S'], b['R6'], b['T'], b['alpha'], u, *consts)])
    (ref * up).sum().backward()
    assert rel_err(vals, ref) < 1e-5
    for k in base:
        assert rel_err(a[k].grad, b[k].grad) < REL, k
    # disabled terms: zero value, no gradient
    c = leaves()
    cfg2 = dict(cfg, parsimony=None, overlap=None)
    v2 = ops.fused_losses(c['fg'], c['env'], imgs, c['alpha'], c['bkg'], c['blk'], c['gnd'], c['sq_eps'], c['S'], c['R6'], c['T'], None, cfg2)
    v2.sum().backward()
    assert v2[1].item() == 0 and v2[3].item() == 0 and c['alpha'].grad is None and c['S'].grad is None
    assert rel_err(v2[[0, 2]], ref[[0, 2]]) < 1e-5


def test_fused_adam_matches_torch():
    torch.manual_seed(0)
    p0 = torch.randn(1000)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=5e-3)
    ph, m, v = p0.to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step in range(1, 6):
        g = torch.randn(1000)
        p.grad = g.clone()
        opt.step()
        ops.adam_step_(ph, g.to(DEV), m, v, 5e-3, step)
    assert rel_err(ph, p) < 1e-5


def test_texture_sets_and_grouped_adam_equal_the_single_launches():
    """The multi-set launches of the native step (one launch for the blocks', sky and ground maps; one Adam launch for both
    learning-rate groups) are the single-tensor kernels run per set: bit-identical results."""
    from dbw_amd import _lib
    torch.manual_seed(3)
    shapes, decims, wraps, scales = [(1, 64, 64, 3), (3, 32, 32, 3), (1, 64, 64, 3)], [8, 8, 1], [0, 1, 0], [0.1, 0.1, 0.01]
    texs = [torch.randn(s, device=DEV) for s in shapes]
    gmaps = [torch.randn(s[0], s[1] // d, s[2] // d, 3, device=DEV) for s, d in zip(shapes, decims)]
    st = ops._stream(texs[0])
    single, sets, keep = [], [], []
    loss1, loss2 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    for t, d, wr, sc, gmp in zip(texs, decims, wraps, scales, gmaps):
        n, h, w, _ = t.shape
        maps, sig, gsig, gtex = torch.empty_like(gmp), torch.empty_like(t), torch.empty_like(t), torch.empty_like(t)
        _lib.call('dbw_texture_prep_fwd', t.data_ptr(), n, h, w, d, maps.data_ptr(), sig.data_ptr(), st)
        _lib.call('dbw_tv_l2sq', sig.data_ptr(), n, h, w, wr, sc, loss1.data_ptr(), gsig.data_ptr(), st)
        _lib.call('dbw_texture_prep_bwd', t.data_ptr(), n, h, w, d, gmp.data_ptr(), gsig.data_ptr(), gtex.data_ptr(), st)
        single.append((maps, sig, gsig, gtex))
        m2, s2, gs2, gt2 = torch.empty_like(gmp), torch.empty_like(t), torch.empty_like(t), torch.empty_like(t)
        keep.append((m2, s2, gs2, gt2))
        sets.append(dict(texture=t.data_ptr(), n=n, h=h, w=w, decim=d, maps=m2.data_ptr(), sig=s2.data_ptr(), wrap_x=wr, tv_scale=sc,
                         grad_sig_out=gs2.data_ptr(), grad_maps=gmp.data_ptr(), grad_sig=gs2.data_ptr(), grad_texture=gt2.data_ptr()))
    arr, k = _lib.texture_sets(sets)
    _lib.call('dbw_texture_prep_fwd_sets', arr, k, st)
    _lib.call('dbw_tv_l2sq_sets', arr, k, loss2.data_ptr(), st)
    _lib.call('dbw_texture_prep_bwd_sets', arr, k, st)
    for a, b in zip(single, keep):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert abs(loss1.item() - loss2.item()) < 1e-6 * abs(loss1.item())          # (atomic order differs)
    # Adam: two groups, two learning rates
    n0, n1 = 700, 300
    p0, g = torch.randn(n0 + n1, device=DEV), torch.randn(n0 + n1, device=DEV)
    pa, ma, va = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pb, mb, vb = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in (1, 2, 3):
        ops.adam_step_(pa[:n0], g[:n0], ma[:n0], va[:n0], 5e-3, step)
        ops.adam_step_(pa[n0:], g[n0:], ma[n0:], va[n0:], 5e-2, step)
        scratch = torch.full((4099,), 7, dtype=torch.uint8, device=DEV)[:4096]          # the launch also clears the next step's zero arena
        ops.adam_step_groups_(pb, g, mb, vb, [n0, n0 + n1], [5e-3, 5e-2], step, zero=scratch)
        assert int(scratch.sum()) == 0
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)


def test_camera_ingest_feeds_the_hip_render_path_on_the_pixels_of_the_projection_matrix():
    """N3 end to end on the GPU: an OpenCV-style projection matrix P goes through dbw_amd.cameras.pytorch3d_KRT_from_proj
    (dtu.py:75-115) into the HIP projection + rasteriser, and a small facet placed at a world point X is drawn on the pixel that
    P @ X names -- for every facet, with every covered pixel within its few-pixel extent."""
    from dbw_amd.cameras import pytorch3d_KRT_from_proj
    rng = np.random.RandomState(5)
    H, W = 300, 400
    for _ in range(3):
        Q, _r = np.linalg.qr(rng.randn(3, 3))
        R_w2c = Q * np.sign(np.linalg.det(Q))
        C = rng.randn(3) * 0.5 + np.array([0, 0, -3.0])
        Kcv = np.array([[720 + 50 * rng.rand(), 0.0, W / 2 + 10 * rng.randn()], [0, 715.0, H / 2 + 10 * rng.randn()], [0, 0, 1]])
        P = Kcv @ np.concatenate([R_w2c, (-R_w2c @ C)[:, None]], 1) * 2.3
        # target pixels on a grid, random depths; X = C + R^T (depth * Kcv^-1 [u, v, 1])
        uu, vv = np.meshgrid(np.linspace(30, W - 30, 6), np.linspace(30, H - 30, 5))
        uv = np.stack([uu.ravel() + rng.rand(30), vv.ravel() + rng.rand(30)], 1)
        depth = 2.0 + 2.0 * rng.rand(30)
        Kinv = np.linalg.inv(Kcv)
        def unproject(px, d):
            return C + R_w2c.T @ (d * (Kinv @ np.array([px[0], px[1], 1.0])))
        verts, faces = [], []
        for i in range(30):                                      # a facet of +-5 px around the target, parallel to the image plane
            for du, dv in ((-5.0, -4.0), (6.0, -3.0), (-1.0, 7.0)):
                verts.append(unproject(uv[i] + [du, dv], depth[i]))
            faces.append([3 * i, 3 * i + 1, 3 * i + 2])
        verts = torch.tensor(np.array(verts), dtype=torch.float32, device=DEV)
        faces_f = torch.tensor(faces, dtype=torch.int32, device=DEV)
        faces_b = faces_f[:, [0, 2, 1]].contiguous()             # both windings: the rasteriser does not cull
        Kp, Rp, Tp = pytorch3d_KRT_from_proj(P, (H, W))
        for fc in (faces_f, faces_b):
            cfg = ops.RenderCfg(H, W, 1, 0.0, 0.001, True, False, fc.shape[0], 1e-8)
            cl, p2f, _, _, _ = ops.render_fragments(verts, fc, Rp[None].to(DEV).contiguous(), Tp[None].to(DEV).contiguous(), Kp.to(DEV), cfg)
            first = p2f[0, :, :, 0].long()
            orig = torch.where(first >= 0, cl['c2o'].view(-1).long()[first.clamp(min=0)], torch.full_like(first, -1)).cpu()
            for i in range(30):
                u, v = uv[i]
                assert int(orig[int(np.floor(v)), int(np.floor(u))]) == i, (i, u, v)      # pixel (x, y) covers [x, x+1) x [y, y+1)
                ys, xs = torch.nonzero(orig == i, as_tuple=True)
                assert 20 <= len(ys) <= 90                                                 # area of the facet: 60 px^2 ... at its depth
                assert float((xs.float() + 0.5 - u).abs().max()) <= 7.0 and float((ys.float() + 0.5 - v).abs().max()) <= 8.0
            assert int((orig >= 0).sum()) == sum(int((orig == i).sum()) for i in range(30))


def _dtu_like_cfg(n_blocks=4, ts=32, fpp=6):
    return {'model': {'name': 'dbw',
                      'mesh': {'n_blocks': n_blocks, 'S_world': 0.5, 'R_world': [115, 0, 0], 'txt_size': ts},
                      'renderer': {'faces_per_pixel': fpp, 'cameras': {'name': 'perspective'}, 'detach_bary': True, 'z_clip': 0.001},
                      'rend_optim': {'coarse_learning': 1500, 'decimate_txt': 750, 'decimate_factor': 8, 'kill_blocks': True,
                                     'decouple_rendering': True, 'opacity_noise': True},
                      'loss': {'rgb_weight': 1, 'perceptual_weight': 0, 'parsimony_weight': 0.01, 'tv_weight': 0.1, 'overlap_weight': 1}}}


@pytest.mark.parametrize('epoch,decimate', [(0, True), (800, False), (1600, False)])
def test_model_losses_and_param_grads_match_oracle(epoch, decimate):
    """One full training iteration (dbw.py:198-200 -> trainer.py:141-142): all losses and the gradient of every one of
    the 10 parameter tensors, coarse+decimated (epoch 0), coarse (epoch 800) and fine (epoch 1600) phases."""
    H, W, nb, ts, fpp = 48, 64, 4, 32, 6
    torch.manual_seed(227391)
    model = dbw_amd.create_model(_dtu_like_cfg(nb, ts, fpp), (H, W))
    orc = O.OracleDBW((H, W), n_blocks=nb, txt_size=ts, faces_per_pixel=fpp, seed=227391)
    for k, v in orc.p.items():                                       # same seed, same draw order -> identical init
        assert torch.equal(v.detach(), getattr(model, k).detach()), k
    with torch.no_grad():                                            # move off the symmetric init: varied shapes/opacities
        g = torch.Generator().manual_seed(1)
        for name, scale in (('sq_eps', 1.5), ('alpha_logit', 1.0), ('R_6d_ground', 0.05), ('T', 0.0)):
            d = torch.randn(orc.p[name].shape, generator=g) * scale
            orc.p[name].add_(d)
            getattr(model, name).add_(d)
        orc.p['T'].mul_(0.5)
        model.T.mul_(0.5)
        if epoch >= 1500:
            orc.p['alpha_logit'][0] = -6.0                           # one block below the 0.5 filter, one killed
            model.alpha_logit[0] = -6.0
    model = model.to(DEV)
    model.train()
    model.set_cur_epoch(epoch)
    coarse = epoch < 1500
    R, T, Km = O.synthetic_cameras(3, R_world=orc.R_world[0])
    imgs = torch.rand(3, 3, H, W, generator=torch.Generator().manual_seed(2))
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3))
    u = torch.rand(nb, 1000, 3, generator=torch.Generator().manual_seed(4))
    inp = dict(imgs=imgs, R=R, T=T, K=Km)
    ref = orc.forward(inp, training=True, coarse=coarse, decimate=decimate, opacity_noise=noise, overlap_points=u, n_threads=8)
    ref['total'].backward()
    model._noise_override, model._overlap_u_override = noise.to(DEV), u.to(DEV)
    out = model({k: v.to(DEV) for k, v in inp.items()}, None)
    out['total'].backward()
    assert set(out) == set(ref)
    for k in ref:
        assert abs(out[k].item() - ref[k].item()) <= REL * max(abs(ref[k].item()), 1e-3), (k, out[k].item(), ref[k].item())
    for k, v in orc.p.items():
        gh = getattr(model, k).grad
        if gh is None or v.grad is None:      # no path in this phase (e.g. opacities in the fine phase): both must agree
            assert (gh is None or gh.abs().max() == 0) and (v.grad is None or v.grad.abs().max() == 0), k
            continue
        assert rel_err(gh, v.grad) < REL, (k, rel_err(gh, v.grad))


def test_sync_free_block_culling_equals_host_packed_path():
    """kill_blocks with dead blocks collapsed on device (no .item()) == the reference's host-side packing."""
    H, W = 48, 64
    R, T, Km = O.synthetic_cameras(3, R_world=O.world_rotation(115, 0, 0))
    inp = {k: v.to(DEV) for k, v in dict(imgs=torch.rand(3, 3, H, W, generator=torch.Generator().manual_seed(2)), R=R, T=T, K=Km).items()}
    res = []
    for sync_free in (False, True, 'overlap'):
        torch.manual_seed(7)
        model = dbw_amd.create_model(_dtu_like_cfg(5, 32, 6), (H, W)).to(DEV).train()
        with torch.no_grad():
            model.alpha_logit[1] = -8.0          # sigmoid < 0.01 -> killed
            model.alpha_logit[3] = -7.0
        model.sync_free = bool(sync_free)
        model.overlap_passes = sync_free == 'overlap'      # env pass on a side stream: same numbers
        model._noise_override = torch.zeros(5, device=DEV)
        model._overlap_u_override = torch.rand(5, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
        out = model(inp, None)
        out['total'].backward()
        res.append((out, {k: v.grad.clone() for k, v in model.named_parameters()}))
    for other in (1, 2):
        for k in res[0][0]:
            assert abs(res[0][0][k].item() - res[other][0][k].item()) <= 1e-6 * max(1.0, abs(res[0][0][k].item())), k
        for k in res[0][1]:
            assert rel_err(res[other][1][k], res[0][1][k]) < 1e-5, k
    assert res[1][1]['S'][1].abs().max() == 0 and res[1][1]['R_6d'][3].abs().max() == 0     # dead blocks get no pose gradient


def test_a_ragged_batch_reuses_the_plan_of_the_full_batch_and_equals_the_launch_by_launch_step():
    """One plan (one workspace) per phase whatever the batch size: the ragged last mini-batch of an epoch runs on the plan of the full
    batch (B < max_views), its random numbers keyed on the optimisation-step count -- and the sequence full, ragged, full, ragged equals the
    launch-by-launch native step on the same draws."""
    from dbw_amd.parallel import ShardedTrainStep
    H, W = 48, 64
    R, T, Km = O.synthetic_cameras(3, R_world=O.world_rotation(115, 0, 0))
    full = {k: v.to(DEV) for k, v in dict(imgs=torch.rand(3, 3, H, W, generator=torch.Generator().manual_seed(2)), R=R, T=T, K=Km).items()}
    ragged = {k: v[:1].contiguous() for k, v in full.items()}
    finals = []
    for c_step in (False, True):
        torch.manual_seed(7)
        model = dbw_amd.create_model(_dtu_like_cfg(5, 32, 6), (H, W)).to(DEV).train()
        model.sync_free = True
        model._noise_override = torch.zeros(5, device=DEV)
        model._overlap_u_override = torch.rand(5, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
        step = ShardedTrainStep(model, use_c_step=c_step)
        seen = []
        for it in range(6):
            losses = step(full if it % 2 == 0 else ragged)
            seen.append(float(losses['total']))
        if c_step:
            assert step.cstep is not None and len(step.cstep._plans) == 1 and step.cstep._cur[3] == 3
        torch.cuda.synchronize()
        finals.append((seen, step.params.flat.clone()))
    for a, b in zip(*[f[0] for f in finals]):
        assert abs(a - b) < 1e-4 * abs(a), (finals[0][0], finals[1][0])
    assert rel_err(finals[1][1], finals[0][1]) < 5e-3


def test_predict_returns_reference_shaped_image_and_state_dict_roundtrip():
    model = dbw_amd.create_model(_dtu_like_cfg(), (48, 64)).to(DEV)
    model.eval()
    R, T, Km = O.synthetic_cameras(2, R_world=model.R_world[0].cpu())
    inp = dict(imgs=torch.rand(2, 3, 48, 64), R=R, T=T, K=Km)
    with torch.no_grad():
        rec = model.predict({k: v.to(DEV) for k, v in inp.items()}, None)
    assert rec.shape == (2, 3, 48, 64) and torch.isfinite(rec).all() and rec.min() >= 0 and rec.max() <= 1 + 1e-5
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    other = dbw_amd.create_model(_dtu_like_cfg(), (48, 64))
    other.load_state_dict({'module.' + k.replace('sq_', 'spq_'): v for k, v in sd.items()})
    for k, v in other.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_trainer_driver_optimises_schedules_and_checkpoints():
    """N1: the loop of src/trainer.py:109-147,163-169 on the HIP path: mini-batches over the views, fused Adam with the two
    lr groups, per-epoch MultiStepLR + model.step(), checkpoint round trip (trainer.py:201-209)."""
    from dbw_amd.trainer import Trainer
    H, W, V = 48, 64, 8
    cfg = _dtu_like_cfg(4, 32, 6)
    cfg['model']['rend_optim'].update(coarse_learning=True, decimate_txt=False, opacity_noise=False)
    cfg['training'] = {'batch_size': 4, 'n_epoches': 6, 'seed': 123,
                       'optimizer': {'name': 'adam', 'lr': 5.0e-3, 'texture': {'lr': 5.0e-2}},
                       'scheduler': {'name': 'multi_step', 'gamma': [0.1, 0.1], 'milestones': [4]}}
    torch.manual_seed(5)
    target = dbw_amd.create_model(cfg, (H, W)).to(DEV).eval()
    R, T, Km = O.synthetic_cameras(V, R_world=target.R_world[0].cpu())
    views = {k: v.to(DEV) for k, v in dict(imgs=torch.zeros(V, 3, H, W), R=R, T=T, K=Km).items()}
    with torch.no_grad():
        target.textures.add_(torch.randn_like(target.textures))
        target.alpha_logit.add_(3.0)
        views['imgs'] = target.predict(views, None).clamp(0, 1).contiguous()
    torch.manual_seed(6)
    model = dbw_amd.create_model(cfg, (H, W)).to(DEV)
    tr = Trainer(cfg, model, views)
    assert tr.step_fn.lrs == (5.0e-3, 5.0e-2)
    first = tr.run_epoch()['rgb'].item()
    for _ in range(3):
        last = tr.run_epoch()
    assert tr.n_iters == 8 and tr.epoch == 5 and model.cur_epoch == 4
    assert tr.step_fn.lrs == pytest.approx((5.0e-4, 5.0e-3))              # milestone 4 reached: both groups x 0.1
    assert last['rgb'].item() < 0.9 * first, (first, last['rgb'].item())
    assert tr.time_per_img > 0
    ckpt = tr.state_dict()
    assert set(ckpt) == {'epoch', 'batch', 'model_name', 'model_kwargs', 'model_state', 'optimizer_state', 'scheduler_state'}
    ref_next = tr.run_epoch(shuffle=False)
    torch.manual_seed(7)
    model2 = dbw_amd.create_model(cfg, (H, W)).to(DEV)
    tr2 = Trainer(cfg, model2, views)
    tr2.load_state_dict(ckpt)
    assert tr2.epoch == 5 and model2.cur_epoch == 4 and tr2.step_fn.lrs == pytest.approx(tr.step_fn.lrs)
    got_next = tr2.run_epoch(shuffle=False)
    assert abs(got_next['total'].item() - ref_next['total'].item()) < 1e-4 * abs(ref_next['total'].item())


def test_quantitative_eval_hard_render_of_joined_scene_matches_oracle():
    """quantitative_eval (dbw.py:464-493): the hard, 4x supersampled render of the JOINED scene (background + ground + opaque
    blocks, one face per pixel) against the oracle rendering the same joined scene; PSNR / SSIM of that image against the
    metrics module applied to the oracle's image."""
    from dbw_amd import metrics
    H, W, nb, ts = 40, 56, 4, 32
    torch.manual_seed(227391)
    model = dbw_amd.create_model(_dtu_like_cfg(nb, ts, 6), (H, W))
    orc = O.OracleDBW((H, W), n_blocks=nb, txt_size=ts, faces_per_pixel=6, seed=227391)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        for name, scale in (('sq_eps', 1.0), ('alpha_logit', 2.0), ('textures', 1.0), ('texture_bkg', 1.0), ('texture_ground', 1.0)):
            d = torch.randn(orc.p[name].shape, generator=g) * scale
            orc.p[name].add_(d)
            getattr(model, name).add_(d)
        orc.p['alpha_logit'][1] = -3.0                                # transparent: filtered out of the evaluation scene
        model.alpha_logit[1] = -3.0
        orc.p['alpha_logit'][2] = 3.0
        model.alpha_logit[2] = 3.0
    model = model.to(DEV)
    R, T, Km = O.synthetic_cameras(3, R_world=orc.R_world[0])
    with torch.no_grad():
        env, blk = orc.build_env(False, False), orc.build_blocks(False, False, False, None, filter_transparent=True)
        nv, nm = env['verts'].shape[0], len(env['maps'])
        joined = dict(verts=torch.cat([env['verts'], blk['verts']]), faces=torch.cat([env['faces'], blk['faces'] + nv]),
                      face_uvs=torch.cat([env['face_uvs'], blk['face_uvs']]), face_map=torch.cat([env['face_map'], blk['face_map'] + nm]),
                      maps=env['maps'] + blk['maps'])
        ref = torch.nn.functional.avg_pool2d(O.render(joined, R, T, Km[0], (4 * H, 4 * W), 0.0, 1, False, None, 0.001, n_threads=8), 4, 4)[:, :3]
    imgs = (ref + 0.05 * torch.randn(ref.shape, generator=torch.Generator().manual_seed(9))).clamp(0, 1)
    loader = [(dict(imgs=imgs[:2], R=R[:2], T=T[:2], K=Km[:2]), None), (dict(imgs=imgs[2:], R=R[2:], T=T[2:], K=Km[2:]), None)]
    res = model.quantitative_eval(loader, DEV, hard_inference=True)
    assert list(res)[:6] == ['n_blocks', 'L_tot', 'L_rec', 'PSNR', 'SSIM', 'LPIPS'] and f'alpha{nb - 1}' in res
    assert res['n_blocks'] == int((torch.sigmoid(orc.p['alpha_logit']) > 0.5).sum())
    psnr = [metrics.mse2psnr(F.mse_loss(imgs[s], ref[s])).item() for s in (slice(0, 2), slice(2, 3))]
    ssim = [metrics.ssim(imgs[s], ref[s]).mean().item() for s in (slice(0, 2), slice(2, 3))]
    assert abs(res['PSNR'] - (2 * psnr[0] + psnr[1]) / 3) < 2e-3 and abs(res['SSIM'] - (2 * ssim[0] + ssim[1]) / 3) < 1e-4
    assert res['LPIPS'] != res['LPIPS'] and model.training              # no perceptual network: NaN; training mode restored
    # and the image itself
    model.eval()
    with torch.no_grad():
        scene = model.build_scene(filter_transparent=True)
        img = model.renderer.render_packed(scene, R.to(DEV), T.to(DEV), viz_purpose=True)[:, :3]
    assert rel_err(img, ref) < REL


def test_edge_overlays_and_log_tick_members():
    """N4: Renderer.render_edges / draw_edges (renderer.py:134-175) on the HIP rasteriser against the oracle's
    fragments, and the model members src/trainer.py's log tick calls (trainer.py:181-198): predict(w_edges=True),
    predict(filter_transparent=True), predict_synthetic, get_arranged_block_txt."""
    import torch.nn.functional as Fn
    from dbw_amd.structures import PackedScene
    H, W, nb, ts = 40, 56, 5, 16
    torch.manual_seed(227391)
    model = dbw_amd.create_model(_dtu_like_cfg(nb, ts, 6), (H, W))
    orc = O.OracleDBW((H, W), n_blocks=nb, txt_size=ts, faces_per_pixel=6, seed=227391)
    with torch.no_grad():
        orc.p['alpha_logit'][1] = -3.0
        model.alpha_logit[1] = -3.0
    model = model.to(DEV).eval()
    R, T, Km = O.synthetic_cameras(2, R_world=orc.R_world[0])
    inp = {k: v.to(DEV) for k, v in dict(imgs=torch.rand(2, 3, H, W), R=R, T=T, K=Km).items()}
    model._ensure_cameras(inp)
    # ---- render_edges of the env scene at 4x: mask and nearest-face ids against the oracle's clip + rasterise ----
    with torch.no_grad():
        env_o = orc.build_env(False, False)
        scene = model.build_env_scene()
        mask, p2f = model.renderer.render_edges(scene, inp['R'], inp['T'], image_size=(4 * H, 4 * W), linewidth=4, return_pix2face=True)
        Fs = env_o['faces'].shape[0]
        ndc = O.transform_to_ndc(env_o['verts'], R, T, Km[0], eps=1e-8)
        fv = ndc[:, env_o['faces']].reshape(2 * Fs, 3, 3)
        cl = O.clip_faces(fv, torch.arange(2) * Fs, torch.full((2,), Fs), 0.001, True)
        p2f_c, _, bary_c, dists = O.rasterize(cl['face_verts'], cl['first_idx'], cl['num_faces'], cl['neighbor'], (4 * H, 4 * W), 0.0, 1, True, True, 8)
        p2f_o, _ = O.convert_clipped_to_original(p2f_c, bary_c, cl)
    ref_mask = (-dists < (4 * 2 / min(4 * H, 4 * W)) ** 2).float()[:, None].max(-1)[0]
    # vertices come from two libms (device / host): allow the handful of pixels an ulp moves across a threshold
    assert (mask.cpu() != ref_mask).float().mean() < 1e-4 and (p2f.cpu() != p2f_o[..., 0]).float().mean() < 1e-4
    assert 0.01 < mask.mean() < 0.6
    # ---- draw_edges = img * (1 - mask) + mask * colour, with the 4x mask average-pooled ----
    img = torch.rand(2, 3, H, W, device=DEV)
    colors = torch.rand(2 * scene.faces.shape[0], 3, device=DEV)
    out = model.renderer.draw_edges(img, scene, inp['R'], inp['T'], colors=colors)
    m4, c4 = Fn.avg_pool2d(mask, 4, 4), Fn.avg_pool2d(colors[p2f].permute(0, 3, 1, 2), 4, 4)
    assert torch.allclose(out, img * (1 - m4) + m4 * c4, atol=1e-6)
    red = model.renderer.draw_edges(img, scene, inp['R'], inp['T'], antialias=False)
    m1 = model.renderer.render_edges(scene, inp['R'], inp['T'])
    assert torch.allclose(red, img * (1 - m1) + m1 * torch.tensor([1., 0., 0.], device=DEV).view(1, 3, 1, 1))
    # ---- the log tick ----
    rec = model.predict(inp, None)
    rec_e = model.predict(inp, None, w_edges=True)
    assert rec_e.shape == rec.shape == (2, 3, H, W) and torch.isfinite(rec_e).all() and (rec_e != rec).any()
    assert rec_e.min() >= 0 and rec_e.max() <= 1 + 1e-5
    hard = model.predict(inp, None, filter_transparent=True)
    assert hard.shape == (2, 3, H, W)
    syn = model.predict_synthetic(inp, None)
    assert syn.shape == (2, 3, H, W) and syn.min() >= 0 and syn.max() <= 1 + 1e-5 and (syn == 1).float().mean() > 0.2
    txt = model.get_arranged_block_txt()
    assert txt.shape == (1, 3, ts * (nb // 5), ts * 5)
    cols = model.get_scene_face_colors(filter_transparent=True)
    assert cols.shape == (model.env_n_faces + (nb - 1) * model.BNF, 3)


def test_log_tick_members_with_sync_free_and_a_killed_block_equal_the_host_packed_results():
    """The trainer and the bench run with model.sync_free = True, where culled blocks stay in the scene (collapsed to a point) and face
    ids run over all n_blocks blocks; the visualisation members build colour tables for the KEPT blocks only, so they pack the scene
    on the host themselves -- same images as with sync_free = False, with a block in the middle killed (kill_blocks threshold) in
    training mode and another one filtered (opacity < 0.5) by the hard renders."""
    H, W, nb, ts = 40, 56, 5, 16
    torch.manual_seed(227391)
    model = dbw_amd.create_model(_dtu_like_cfg(nb, ts, 6), (H, W))
    with torch.no_grad():
        model.alpha_logit[1] = -8.0          # sigmoid < 0.01: killed
        model.alpha_logit[3] = -1.0          # kept by training renders, filtered by the hard ones
    model = model.to(DEV)
    R, T, Km = O.synthetic_cameras(2, R_world=O.world_rotation(115, 0, 0))
    inp = {k: v.to(DEV) for k, v in dict(imgs=torch.rand(2, 3, H, W), R=R, T=T, K=Km).items()}
    outs = {}
    for sf in (False, True):
        model.sync_free = sf
        model.eval()
        outs[sf] = (model.predict(inp, None, w_edges=True), model.predict(inp, None, w_edges=True, filter_transparent=True),
                    model.predict_synthetic(inp, None))
        model.train()
        model._noise_override = torch.zeros(nb, device=DEV)
        outs[sf] += (model.predict(inp, None, w_edges=True),)
        assert model.sync_free == sf
    for a, b in zip(outs[False], outs[True]):
        assert torch.isfinite(b).all() and float((a - b).abs().max()) < 1e-5
    assert float((outs[True][0] - outs[True][1]).abs().max()) > 1e-3       # the filtered block does change the picture


@pytest.mark.parametrize('epoch', [0, 1600])
def test_non_decoupled_rendering_matches_the_oracle_render_of_the_joined_scene(epoch):
    """dbw.py:225-232 (`decouple_rendering: False`, no shipped config): sky dome + ground + blocks as ONE scene in ONE soft pass, env faces
    with opacity 1 next to the blocks' learned opacities (coarse phase) / no opacities (fine phase).  Image and the gradient of an
    MSE against noise w.r.t. the scene's vertices, maps and opacities against the oracle's render of the same joined scene; then the
    model's forward / backward end to end (all loss terms present, every parameter receives a finite gradient)."""
    H, W, nb, ts, fpp = 48, 64, 4, 32, 6
    cfg = _dtu_like_cfg(nb, ts, fpp)
    cfg['model']['rend_optim']['decouple_rendering'] = False
    torch.manual_seed(227391)
    model = dbw_amd.create_model(cfg, (H, W)).to(DEV).train()
    model.set_cur_epoch(epoch)
    model._noise_override = torch.zeros(nb, device=DEV)
    R, T, Km = O.synthetic_cameras(2, R_world=O.world_rotation(115, 0, 0))
    imgs = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(2))
    inp = {k: v.to(DEV) for k, v in dict(imgs=imgs, R=R, T=T, K=Km).items()}
    coarse = epoch < 1500
    out4 = model.render_joined(inp)
    assert out4.shape == (2, 4, H, W)
    scene = model.build_scene(filter_transparent=not coarse)
    n_env = model.env_n_faces
    alpha = None
    if coarse:
        alpha = torch.cat([torch.ones(n_env, device=DEV), model._alpha.detach().repeat_interleave(model.BNF)]).cpu()
    # the oracle on the very same scene tensors (vertices, faces, uvs, maps): verts / maps / opacities as leaves
    desc = scene.map_desc.cpu()
    flat = scene.maps.detach().cpu()
    leaves, maps = [], []                  # leaves: the unpadded maps as the product stores them; the oracle samples the circularly padded ones
    for off, h, w, pl, pr, sh, _, _ in desc.tolist():
        hs, ws_ = h >> sh, w >> sh                 # decimated maps are stored at cell resolution (nearest upsampling = the sampler's shift)
        leaf = flat[off:off + hs * ws_ * 3].view(hs, ws_, 3).clone().requires_grad_(True)
        leaves.append(leaf)
        full = leaf.repeat_interleave(1 << sh, 0).repeat_interleave(1 << sh, 1) if sh else leaf
        maps.append(torch.nn.functional.pad(full.permute(2, 0, 1)[None], (pl, pr, 0, 0), mode='circular')[0].permute(1, 2, 0) if (pl or pr) else full)
    verts_o = scene.verts.detach().cpu().requires_grad_(True)
    alpha_o = None if alpha is None else alpha.clone().requires_grad_(True)
    sc = dict(verts=verts_o, faces=scene.faces.cpu().long(), face_uvs=scene.face_uvs.cpu(), face_map=scene.face_map.cpu().long(), maps=maps)
    sigma = 1e-4 if coarse else 5e-6
    ref = O.render(sc, R, T, Km[0], (H, W), sigma, fpp, True, None if alpha_o is None else alpha_o.repeat(2), 0.001, n_threads=8)
    assert rel_err(out4, ref) < REL
    # gradients of sum((rgb - noise)^2) through the joined render: product op vs oracle autograd
    from dbw_amd.structures import PackedScene
    v_h = scene.verts.detach().clone().requires_grad_(True)
    m_h = scene.maps.detach().clone().requires_grad_(True)
    a_h = None if alpha is None else alpha.to(DEV).requires_grad_(True)
    sc_h = PackedScene(v_h, scene.faces, scene.face_uvs, scene.face_map, scene.map_desc, m_h)
    renderer = model.renderer if coarse else model.renderer_fine
    img_h = renderer.render_packed(sc_h, inp['R'], inp['T'], faces_alpha=a_h)
    ((img_h[:, :3] - inp['imgs']) ** 2).sum().backward()
    ((ref[:, :3] - imgs) ** 2).sum().backward()
    assert rel_err(v_h.grad, verts_o.grad) < REL
    if a_h is not None:
        assert rel_err(a_h.grad[n_env:], alpha_o.grad[n_env:]) < REL
    assert rel_err(m_h.grad, torch.cat([leaf.grad.reshape(-1) for leaf in leaves])) < REL
    # end to end through the model
    losses = model(inp, None)
    assert set(losses) == {'rgb', 'parsimony', 'tv', 'overlap', 'total'}
    assert abs(float(losses['rgb']) - float(((out4[:, :3] - inp['imgs']) ** 2).mean())) < 1e-5 * float(losses['rgb'])
    losses['total'].backward()
    for n in ('textures', 'texture_bkg', 'texture_ground', 'S', 'T', 'R_6d', 'sq_eps', 'R_6d_ground', 'T_ground') + (('alpha_logit',) if coarse else ()):
        g = model.get_parameter(n).grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0, n
    # predict = the joined render's colour channels
    model.eval()
    assert model.predict(inp, None).shape == (2, 3, H, W)


@pytest.mark.parametrize('epoch', [0, 800, 1600])
def test_loss_epilogue_path_equals_layered_path(epoch):
    """The training forward with composite + MSE as the epilogue of the fg pass (ops.render_decoupled_mse: no fg image, no composite
    kernel, upstream gradient applied inside the backward kernels) against the layered path (two render nodes + dbw_composite_mse):
    same losses, same gradients for all 10 parameter tensors, also under a non-unit upstream gradient."""
    H, W, nb, ts, fpp = 48, 64, 4, 32, 6
    R, T, Km = O.synthetic_cameras(3, R_world=O.world_rotation(115, 0, 0))
    imgs = torch.rand(3, 3, H, W, generator=torch.Generator().manual_seed(2))
    inp = {k: v.to(DEV) for k, v in dict(imgs=imgs, R=R, T=T, K=Km).items()}
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3)).to(DEV)
    u = torch.rand(nb, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    res = []
    for fused in (True, False):
        torch.manual_seed(227391)
        model = dbw_amd.create_model(_dtu_like_cfg(nb, ts, fpp), (H, W)).to(DEV).train()
        with torch.no_grad():
            model.T.mul_(0.5)
            model.alpha_logit.add_(torch.tensor([1.0, -0.5, 0.3, 2.0], device=DEV))
        model.set_cur_epoch(epoch)
        model.fused_loss_epilogue = fused
        model._noise_override, model._overlap_u_override = noise, u
        out = model(inp, None)
        (out['total'] * 1.7).backward()
        res.append(({k: v.item() for k, v in out.items()}, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert set(res[0][0]) == set(res[1][0]) and set(res[0][1]) == set(res[1][1])
    for k in res[0][0]:
        assert abs(res[0][0][k] - res[1][0][k]) <= 2e-6 * max(abs(res[1][0][k]), 1e-3), (k, res[0][0][k], res[1][0][k])
    for n in res[0][1]:
        assert rel_err(res[0][1][n], res[1][1][n]) < 1e-5, (n, rel_err(res[0][1][n], res[1][1][n]))


@pytest.mark.parametrize('epoch', [0, 800, 1600])
def test_native_step_equals_autograd_step(epoch):
    """native_step.NativeStep (forward + backward kernels in a fixed order, gradients accumulated straight into the flat buffer, one
    opacity per block instead of one per face) against the autograd iteration `model(inp); total.backward()`: the same flat gradient
    after one step and the same parameters after three, in each training phase, opacity noise and overlap samples on."""
    from dbw_amd.parallel import ShardedTrainStep
    H, W, nb, ts, fpp = 48, 64, 4, 32, 6
    R, T, Km = O.synthetic_cameras(3, R_world=O.world_rotation(115, 0, 0))
    imgs = torch.rand(3, 3, H, W, generator=torch.Generator().manual_seed(2))
    inp = {k: v.to(DEV) for k, v in dict(imgs=imgs, R=R, T=T, K=Km).items()}
    res = []
    for native in (True, False):
        torch.manual_seed(227391)
        model = dbw_amd.create_model(_dtu_like_cfg(nb, ts, fpp), (H, W)).to(DEV).train()
        with torch.no_grad():
            model.T.mul_(0.5)
            model.alpha_logit.add_(torch.tensor([1.0, -6.0, 0.3, 2.0], device=DEV))       # block 1 is killed / filtered
        model.set_cur_epoch(epoch)
        model.sync_free = True
        # (the launch-by-launch form of the native step: use_c_step=False; the C step is held to it in tests/test_gpu_c_step.py)
        step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99, use_native=native, use_c_step=False)
        assert (step.native is not None) == native and (not native or step.native.supported()) and step.cstep is None
        out = step(inp)
        grad1 = step.params.grad.clone()
        vals = {k: float(v) for k, v in out.items()}
        for _ in range(2):
            step(inp)
        res.append((vals, grad1, step.params.flat.clone(), step.params.names))
    assert set(res[0][0]) == set(res[1][0])
    for k in res[0][0]:
        assert abs(res[0][0][k] - res[1][0][k]) <= 2e-6 * max(abs(res[1][0][k]), 1e-3), (k, res[0][0][k], res[1][0][k])
    for n, off, k in res[0][3]:
        a, b = res[0][1][off:off + k], res[1][1][off:off + k]
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, (n, float((a - b).abs().max()), float(b.abs().max()))
    # parameters after three Adam steps: the same for all but a handful of elements.  Not for ALL of them, for two reasons that have
    # nothing to do with the step being native or not: Adam divides by sqrt(v), so an element whose gradient is at the rounding noise
    # of the float atomics moves by +-lr whatever the noise was; and from the second step on the two runs' parameters differ in the
    # last bits, which can move a borderline fragment in or out of a pixel's list and shift the gradient of the texels / vertices
    # under that pixel by one pixel's worth (tests/test_gpu_configs.py::_fragment_flips)
    assert_same_trajectory(res[0][2], res[1][2])          # (the bounds and the measurements behind them: tests/trajectory.py)


def test_perceptual_term_with_the_lpips_vgg_module_runs_through_the_model():
    """N4: `perceptual_weight > 0` + `model.set_perceptual(LPIPSVGG(...))` (random weights here: none exist offline): the term appears in the
    losses with its phase factor (dbw.py:370), the HIP render path stays differentiable through it (MIOpen convolutions behind the
    composite), and without a network the model refuses instead of silently dropping the term."""
    from dbw_amd.lpips_vgg import LPIPSVGG
    H, W = 48, 64
    cfg = _dtu_like_cfg(4, 32, 6)
    cfg['model']['loss']['perceptual_weight'] = 0.1
    R, T, Km = O.synthetic_cameras(2, R_world=O.world_rotation(115, 0, 0))
    inp = {k: v.to(DEV) for k, v in dict(imgs=torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(2)), R=R, T=T, K=Km).items()}
    torch.manual_seed(227391)
    model = dbw_amd.create_model(cfg, (H, W)).to(DEV).train()
    with pytest.raises(RuntimeError, match='set_perceptual'):
        model(inp, None)
    net = LPIPSVGG(allow_random_init=True).to(DEV)
    model.set_perceptual(net)
    out = model(inp, None)
    assert set(out) == {'rgb', 'perceptual', 'parsimony', 'tv', 'overlap', 'total'}
    rec = model.predict(inp, None).detach()
    pv = float(out['perceptual'].detach())
    assert abs(pv / 0.1 - float(net(inp['imgs'], rec))) < 0.2 * pv / 0.1 + 1e-3   # (opacity noise differs between the two renders)
    out['total'].backward()
    for n in ('textures', 'S', 'T', 'alpha_logit', 'texture_ground'):
        g = model.get_parameter(n).grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0, n
