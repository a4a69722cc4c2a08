"""CPU: the oracle's restatements of the reference's OWN torch code against golden vectors produced by the real
reference functions (tests/golden/make_golden.py), plus self-consistency of the restated PyTorch3D pieces
(flagged 'parity unpinned' -- SURVEY.md 8c)."""
import os

import numpy as np
import pytest
import torch

import oracle as O


def _load(golden_dir, name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, name)).items()}


@pytest.mark.parametrize('tag', ['s1e-4_a', 's1e-4', 's5e-6_a', 's0', 's0_a', 'sig1e-4_a', 'sig1e-4', 'sig5e-6_a'])
def test_layered_rgb_blend_matches_reference(golden_dir, tag):
    clip_inside = not tag.startswith('sig')          # sig*: the real function with clip_inside=False (sigmoid opacity), blend_sigmoid.npz
    g = _load(golden_dir, 'blend.npz' if clip_inside else 'blend_sigmoid.npz')
    colors = g[f'{tag}/colors'].clone().requires_grad_(True)
    dists = g[f'{tag}/dists'].clone().requires_grad_(True)
    fa = g[f'{tag}/faces_alpha'].clone().requires_grad_(True) if f'{tag}/faces_alpha' in g else None
    sigma = float(g[f'{tag}/sigma'])
    out = O.layered_rgb_blend(colors, g[f'{tag}/p2f'], dists, sigma, tuple(g[f'{tag}/bg'].tolist()), fa, clip_inside)
    assert torch.equal(out, g[f'{tag}/out'])                       # same torch ops -> bit-identical
    (out * g[f'{tag}/w']).sum().backward()
    torch.testing.assert_close(colors.grad, g[f'{tag}/g_colors'], rtol=1e-6, atol=1e-7)
    if sigma > 0:
        torch.testing.assert_close(dists.grad, g[f'{tag}/g_dists'], rtol=1e-5, atol=1e-4)
    if fa is not None:
        torch.testing.assert_close(fa.grad, g[f'{tag}/g_faces_alpha'], rtol=1e-5, atol=1e-6)


def test_parametric_sq_matches_reference(golden_dir):
    g = _load(golden_dir, 'parametric_sq.npz')
    for i in range(4):
        for j in range(4):
            e = g[f'{i}{j}/eps']
            e1 = e[0].reshape(1, 1).clone().requires_grad_(True)
            e2 = e[1].reshape(1, 1).clone().requires_grad_(True)
            pts = O.parametric_sq(g['eta'][None], g['omega'][None], e1, e2)
            assert torch.equal(pts, g[f'{i}{j}/pts'])
            (pts * g[f'{i}{j}/w']).sum().backward()
            torch.testing.assert_close(e1.grad, g[f'{i}{j}/g_eps1'], equal_nan=True)
            torch.testing.assert_close(e2.grad, g[f'{i}{j}/g_eps2'], equal_nan=True)


def test_implicit_sq_safe_pow_tv_match_reference(golden_dir):
    g = _load(golden_dir, 'implicit_misc.npz')
    pts = g['pts'].clone().requires_grad_(True)
    e1 = g['eps1'].clone().requires_grad_(True)
    e2 = g['eps2'].clone().requires_grad_(True)
    sdf = O.implicit_sq_sdf2(pts, e1, e2)
    assert torch.equal(sdf, g['sdf'])
    (sdf * g['w']).sum().backward()
    torch.testing.assert_close(pts.grad, g['g_pts'])
    torch.testing.assert_close(e1.grad, g['g_eps1'])
    torch.testing.assert_close(e2.grad, g['g_eps2'])
    a = g['sp_in'].clone().requires_grad_(True)
    sp = O.safe_pow(a, 0.5)
    assert torch.equal(sp, g['sp_out'])
    sp.sum().backward()
    torch.testing.assert_close(a.grad, g['sp_grad'])
    assert torch.equal(O.signed_pow(g['spow_in'], torch.tensor(0.37)), g['spow_out'])
    m = g['tv_maps'].clone().requires_grad_(True)
    tv = O.tv_l2sq(torch.diff(m, dim=2, append=m[:, :, 0:1])).sum(0).mean() + O.tv_l2sq(torch.diff(m, dim=1)).sum(0).mean()
    torch.testing.assert_close(tv, g['tv'])
    tv.backward()
    torch.testing.assert_close(m.grad, g['tv_grad'])


@pytest.mark.parametrize('side', ['oracle', 'product'])
def test_registry_criteria_and_tv_norms_match_reference(golden_dir, side):
    """The non-default keys of the loss registry the config schema accepts (loss.py:11-24: l1, huber next to mse / l2; loss.py:43-47:
    tv_norm_funcs l1 / l2 next to l2sq), as dbw.py:367 and dbw.py:378-387 use them: the oracle's restatement and the product's own host-side
    tables (dbw_amd/dbw.py runs these keys through torch on the rendered image / the prepared maps) against vectors of the REAL functions
    (tests/golden/criteria.npz): values and gradients, incl. Huber's quadratic / linear switch and the l2 norm's clamp at a zero difference."""
    g = _load(golden_dir, 'criteria.npz')
    if side == 'oracle':
        crit, norms = O.CRITERIA, O.TV_NORMS
    else:
        from dbw_amd import dbw as D
        crit, norms = D._CRITERIA, D._TV_NORMS
    for name in ('mse', 'l2', 'l1', 'huber'):
        rec = g['crit_rec'].clone().requires_grad_(True)
        val = crit[name](g['crit_imgs'], rec)
        val.backward()
        torch.testing.assert_close(val.detach(), g[f'crit_{name}'], rtol=1e-6, atol=0)
        torch.testing.assert_close(rec.grad, g[f'crit_{name}_grad'], rtol=1e-6, atol=1e-9)
    for name in ('l1', 'l2', 'l2sq'):
        norm = norms[name]
        b_, m_, g_ = (g[k].clone().requires_grad_(True) for k in ('tv_bkg', 'tv_blocks', 'tv_ground'))
        tv = sum(norm(torch.diff(b_, dim=k)).mean() for k in (1, 2))
        tv = tv + norm(torch.diff(m_, dim=2, append=m_[:, :, 0:1])).sum(0).mean() + norm(torch.diff(m_, dim=1)).sum(0).mean()
        tv = tv + sum(norm(torch.diff(g_, dim=k)).mean() for k in (1, 2)) * 0.1
        tv.backward()
        torch.testing.assert_close(tv.detach(), g[f'tv_{name}'], rtol=1e-6, atol=0)
        for t, k in ((b_, 'bkg'), (m_, 'blocks'), (g_, 'ground')):
            torch.testing.assert_close(t.grad, g[f'tv_{name}_g_{k}'], rtol=1e-5, atol=1e-8)


def test_world_rotation_matches_reference(golden_dir):
    g = _load(golden_dir, 'world_rotation.npz')
    for tag in ['dtu', 'bmvs', 'mix']:
        torch.testing.assert_close(O.world_rotation(*g[tag + '_angles'].tolist()), g[tag], rtol=0, atol=1e-7)


def test_icosphere_uvs_match_reference(golden_dir):
    g = _load(golden_dir, 'icosphere_uvs.npz')
    for level in [1, 2]:
        f, v = O.get_icosphere_uvs(level, True, True)
        assert torch.equal(f, g[f'l{level}/faces_uvs'])
        torch.testing.assert_close(v, g[f'l{level}/verts_uvs'], rtol=0, atol=1e-7)
        vr, _ = O.get_icosphere(level)
        torch.testing.assert_close(O.point_to_uv_sphericalmap(vr), g[f'l{level}/verts_uvs_raw'], rtol=0, atol=0)


def test_topology_known_answers():
    """SURVEY A.10 [PROBED] values."""
    v, f = O.get_icosphere(1)
    assert v.shape == (42, 3) and f.shape == (80, 3)
    assert f[:3].tolist() == [[0, 16, 13], [0, 13, 12], [0, 12, 14]] and f[-1].tolist() == [19, 20, 40]
    v2, f2 = O.get_icosphere(2)
    assert v2.shape == (162, 3) and f2.shape == (320, 3)
    for ts, pr, wp in [(128, 12, 140), (256, 23, 279), (512, 46, 558)]:
        m = O.OracleDBW((8, 8), n_blocks=1, txt_size=ts, seed=0)
        assert m.txt_padding == (0, pr) and ts + pr == wp
        assert m.ground_verts.shape == (81, 3) and m.ground_faces.shape == (128, 3) and m.BNF == 80
        assert m.block_verts_uvs.shape == (63, 2)


# ---------------------------------------------------------------------------------------------------------------------
# restated PyTorch3D pieces: self-consistency only (parity unpinned)
# ---------------------------------------------------------------------------------------------------------------------
def _tri_scene():
    # vertex coordinates deliberately off the pixel-centre lattice (a centre ON a vertex/edge is a kink of the
    # clipped barycentrics where finite differences are meaningless)
    fv = torch.tensor([[[-0.53, -0.41, 2.0], [0.63, -0.33, 3.0], [0.13, 0.71, 2.5]],
                       [[-0.23, -0.61, 1.0], [0.73, 0.11, 1.5], [-0.61, 0.53, 4.0]]])
    return fv, torch.tensor([0]), torch.tensor([2])


def test_raster_single_triangle_properties():
    fv, first, num = _tri_scene()
    H, W, K = 24, 32, 3
    p2f, zbuf, bary, dists = O.rasterize_fwd_raw(fv, first, num, None, (H, W), 1e-3, K)
    valid = p2f >= 0
    assert valid.any() and (~valid).any()
    assert torch.all(zbuf[~valid] == -1) and torch.all(dists[~valid] == -1) and torch.all(bary[~valid] == -1)
    b = bary[valid]
    torch.testing.assert_close(b.sum(-1), torch.ones(b.shape[0]), rtol=0, atol=1e-5)
    assert torch.all(b >= 0)
    # z-sorted front to back, ties impossible here
    z = torch.where(valid, zbuf, torch.full_like(zbuf, float('inf')))
    assert torch.all(z[..., 1:] >= z[..., :-1])
    # zbuf is the bary-interpolated depth
    zf = fv[p2f.clamp(0)][..., 2]
    torch.testing.assert_close((bary * zf).sum(-1)[valid], zbuf[valid], rtol=1e-5, atol=1e-6)
    # inside pixels have negative distance; centre pixel of face 0 is covered
    assert (dists[valid] < 0).any() and (dists[valid] > 0).any()
    assert torch.all(dists[valid] < 1e-3)


def test_raster_threads_and_double_agree():
    fv, first, num = _tri_scene()
    a = O.rasterize_fwd_raw(fv, first, num, None, (20, 28), 1e-3, 3, n_threads=1)
    b = O.rasterize_fwd_raw(fv, first, num, None, (20, 28), 1e-3, 3, n_threads=4)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    d = O.rasterize_fwd_raw(fv.double(), first, num, None, (20, 28), 1e-3, 3)
    assert (d[0] != a[0]).float().mean() < 0.01
    m = (d[0] == a[0]) & (a[0] >= 0)
    torch.testing.assert_close(d[3][m].float(), a[3][m], rtol=1e-3, atol=1e-6)


def test_raster_backward_finite_difference_double():
    """Hand-restated backward vs central differences in fp64 with the face selection frozen."""
    torch.manual_seed(0)
    fv, first, num = _tri_scene()
    fv = fv.double()
    H, W, K = 10, 12, 2
    p2f, zbuf, bary, dists = O.rasterize_fwd_raw(fv, first, num, None, (H, W), 5e-2, K)
    gz, gb, gd = torch.randn_like(zbuf), torch.randn_like(bary), torch.randn_like(dists)
    g = O.rasterize_bwd_raw(fv, p2f, gz, gb, gd)
    lib = O.lib()
    import ctypes

    def loss(fvx):
        tot = 0.0
        out5 = torch.zeros(5, dtype=torch.float64)
        for y in range(H):
            yf = lib.dbw_ref_pix_to_ndc_f64(H - 1 - y, H, W)
            for x in range(W):
                xf = lib.dbw_ref_pix_to_ndc_f64(W - 1 - x, W, H)
                for k in range(K):
                    f = int(p2f[0, y, x, k])
                    if f < 0:
                        continue
                    one = fvx[f].contiguous()
                    lib.dbw_ref_eval_pair_f64(ctypes.c_void_p(one.data_ptr()), ctypes.c_double(xf), ctypes.c_double(yf),
                                              1, 1, ctypes.c_void_p(out5.data_ptr()))
                    tot += float(out5[0] * gz[0, y, x, k] + (out5[1:4] * gb[0, y, x, k]).sum() + out5[4] * gd[0, y, x, k])
        return tot
    num_g = torch.zeros_like(fv)
    h = 1e-6
    for idx in range(fv.numel()):
        d = torch.zeros(fv.numel(), dtype=torch.float64)
        d[idx] = h
        num_g.view(-1)[idx] = (loss(fv + d.view_as(fv)) - loss(fv - d.view_as(fv))) / (2 * h)
    torch.testing.assert_close(g, num_g, rtol=1e-4, atol=1e-5)


def test_clip_faces_cases_and_bary_conversion():
    """A.4: straddling triangles (case 3 and case 4) keep covering the same pixels with consistent original-face
    barycentrics; neighbours are linked and de-duplicated in the top-K."""
    c = 0.5
    fv = torch.tensor([[[-0.8, -0.8, 2.0], [0.8, -0.8, 2.0], [0.0, 0.8, 0.2]],     # 1 behind -> case 4
                       [[-0.8, 0.6, 0.2], [0.8, 0.6, 0.3], [0.0, -0.9, 3.0]],     # 2 behind -> case 3
                       [[-0.3, -0.3, 0.1], [0.3, -0.3, 0.1], [0.0, 0.3, 0.2]],     # 3 behind -> dropped
                       [[-0.9, -0.9, 5.0], [0.9, -0.9, 5.0], [0.0, 0.9, 5.0]]])    # untouched
    cl = O.clip_faces(fv, torch.tensor([0]), torch.tensor([4]), c)
    assert cl['face_verts'].shape[0] == 4 and cl['num_faces'].tolist() == [4]
    assert cl['clipped_to_orig'].tolist() == [0, 0, 1, 3]
    assert cl['neighbor'].tolist() == [1, 0, -1, -1]
    assert cl['has_conv'].tolist() == [True, True, True, False]
    assert torch.all(cl['face_verts'][:, :, 2] >= c - 1e-6)
    p2f_c, zbuf, bary_c, dists = O.rasterize_fwd_raw(cl['face_verts'], cl['first_idx'], cl['num_faces'], cl['neighbor'],
                                                     (32, 32), 0.0, 4)
    p2f, bary = O.convert_clipped_to_original(p2f_c, bary_c, cl)
    valid = p2f >= 0
    # each pixel sees an original face at most once (t1/t2 de-duplicated)
    srt = torch.where(valid, p2f, torch.arange(-4, 0).expand_as(p2f)).sort(-1)[0]
    assert torch.all(srt[..., 1:] != srt[..., :-1])
    assert set(p2f[valid].unique().tolist()) == {0, 1, 3}
    # converted barycentrics still sum to one and reproduce the clipped-triangle depth through the ORIGINAL verts'
    # reciprocal-depth interpolation
    torch.testing.assert_close(bary[valid].sum(-1), torch.ones(int(valid.sum())), rtol=0, atol=1e-5)


def test_render_and_model_gradients_flow():
    torch.manual_seed(0)
    m = O.OracleDBW((24, 32), n_blocks=3, txt_size=16, faces_per_pixel=4, seed=3)
    R, T, Km = O.synthetic_cameras(2, R_world=m.R_world[0] * 1.0)
    inp = dict(imgs=torch.rand(2, 3, 24, 32), R=R, T=T, K=Km)
    losses = m.forward(inp, opacity_noise=torch.randn(3) * 0.1, overlap_points=torch.rand(3, 1000, 3), n_threads=4)
    assert set(losses) == {'rgb', 'parsimony', 'tv', 'overlap', 'total'}
    losses['total'].backward()
    for k, v in m.p.items():
        assert v.grad is not None and torch.isfinite(v.grad).all(), k
        assert v.grad.abs().sum() > 0, k


def _tiny_scene(g, name):
    fv = g[f'{name}/face_verts']
    first, num, nbr = torch.tensor([0]), torch.tensor([fv.shape[0]]), None
    if float(g[f'{name}/zclip']) >= 0:
        cl = O.clip_faces(fv, first, num, float(g[f'{name}/zclip']), True)
        assert torch.equal(cl['face_verts'], g[f'{name}/clipped']) and torch.equal(cl['neighbor'], g[f'{name}/neighbor'])
        assert torch.equal(cl['clipped_to_orig'], g[f'{name}/clipped_to_orig'])
        fv, first, num, nbr = cl['face_verts'], cl['first_idx'], cl['num_faces'], cl['neighbor']
    size = tuple(int(x) for x in g[f'{name}/size'])
    return fv, first, num, nbr, size, float(g[f'{name}/blur']), int(g[f'{name}/K'])


@pytest.mark.parametrize('name', ['front', 'straddle', 'ties'])
def test_raster_tiny_scenes_are_frozen(golden_dir, name):
    """SURVEY.md 8(c) golden vector (7): the oracle's rasteriser + clipping on three tiny scenes (K > faces, a triangle straddling the
    near plane both ways, depth ties) against the committed fixture -- SELF-CONSISTENCY (parity unpinned vs PyTorch3D): it freezes the
    restatement bit for bit; the host build of the product's arithmetic and the HIP kernels are held to the same file."""
    g = _load(golden_dir, 'raster_tiny.npz')
    fv, first, num, nbr, size, blur, K = _tiny_scene(g, name)
    p2f, zbuf, bary, dists = O.rasterize_fwd_raw(fv, first, num, nbr, size, blur, K)
    for got, key in ((p2f, 'p2f'), (zbuf, 'zbuf'), (bary, 'bary'), (dists, 'dists')):
        assert torch.equal(got, g[f'{name}/{key}']), key
    gfv = O.rasterize_bwd_raw(fv, p2f, g[f'{name}/g_zbuf'], g[f'{name}/g_bary'], g[f'{name}/g_dists'])
    assert torch.equal(gfv, g[f'{name}/g_face_verts'])
    if name == 'ties':                                          # equal depth everywhere: the smaller face id comes first
        v = g['ties/p2f'][..., 0] >= 0
        assert bool((g['ties/p2f'][..., 0][v] < g['ties/p2f'][..., 1][v]).all() or (g['ties/p2f'][..., 1][v] < 0).any())


def test_keps_double_vs_float_reading_is_quantified_on_a_config2_view_and_the_tiny_scenes(golden_dir):
    """PyTorch3D's kEpsilon is a double; the one place it enters arithmetic is the barycentric denominator area + kEpsilon (header of
    oracle/raster_ref.c).  Rounds 1-2 added 1e-8f in float.  This test counts what the two readings differ by, so that the residual
    risk of the (unpinned) restatement is a number: nothing on the tiny scenes, and on a full config-2 view of the blocks (300x400,
    K = 10, 10 superquadric blocks) no face index and at most a few last-bit barycentric / depth values per million slots."""
    import math
    g = _load(golden_dir, 'raster_tiny.npz')
    for name in ('front', 'straddle', 'ties'):
        fv, first, num, nbr, size, blur, K = _tiny_scene(g, name)
        a = O.rasterize_fwd_raw(fv, first, num, nbr, size, blur, K)
        b = O.rasterize_fwd_raw(fv, first, num, nbr, size, blur, K, keps_float=True)
        for x, y in zip(a, b):
            assert torch.equal(x, y), name
    m = O.OracleDBW((300, 400), n_blocks=10, txt_size=16, faces_per_pixel=10, seed=227391)
    R, T, Km = O.synthetic_cameras(49, R_world=m.R_world[0])
    with torch.no_grad():
        sc = m.build_blocks(training=True, coarse=True, decimate=False, opacity_noise=None)
    verts, faces = sc['verts'], sc['faces']
    moved = {'p2f': 0, 'zbuf': 0, 'bary': 0, 'dists': 0}
    slots = 0
    for v in (0, 17, 33):
        ndc = O.transform_to_ndc(verts, R[v:v + 1], T[v:v + 1], Km[0])
        fv = ndc[:, faces].reshape(-1, 3, 3)
        first, num = torch.tensor([0]), torch.tensor([fv.shape[0]])
        cl = O.clip_faces(fv, first, num, 0.001, True)
        blur = math.log(1. / 1e-4 - 1.) * 1e-4
        a = O.rasterize_fwd_raw(cl['face_verts'], cl['first_idx'], cl['num_faces'], cl['neighbor'], (300, 400), blur, 10, n_threads=8)
        b = O.rasterize_fwd_raw(cl['face_verts'], cl['first_idx'], cl['num_faces'], cl['neighbor'], (300, 400), blur, 10, n_threads=8,
                                keps_float=True)
        slots += int((a[0] >= 0).sum())
        for key, x, y in zip(('p2f', 'zbuf', 'bary', 'dists'), a, b):
            moved[key] += int((x != y).sum())
        # where a value moves it moves by an ulp
        for x, y in ((a[1], b[1]), (a[2], b[2])):
            d = (x - y).abs()
            assert float(d.max()) <= 2.4e-7 * float(x.abs().max())
    assert slots > 200_000
    print('kEpsilon double vs float over %d occupied slots of three config-2 views:' % slots, moved)
    assert moved['p2f'] == 0 and moved['dists'] == 0            # no face index, no distance (the area is not part of the distance)
    assert moved['zbuf'] <= 50 and moved['bary'] <= 150         # at most a few last-bit values per million (measured: none)
    # ... and the count is not vacuous: on slivers whose area is within two decades of kEpsilon, where 1e-8 (double) and 1e-8f differ
    # by a sizeable fraction of an ulp of the sum, the two readings do part ways
    gen = torch.Generator().manual_seed(5)
    c = (torch.rand(4000, 1, 2, generator=gen) * 2 - 1) * 0.9
    fv = torch.cat([c + (torch.rand(4000, 3, 2, generator=gen) * 2 - 1) * 4e-4, torch.rand(4000, 3, 1, generator=gen) + 1.0], -1).contiguous()
    first, num = torch.tensor([0]), torch.tensor([4000])
    a = O.rasterize_fwd_raw(fv, first, num, None, (64, 64), 1e-3, 8, n_threads=8)
    b = O.rasterize_fwd_raw(fv, first, num, None, (64, 64), 1e-3, 8, n_threads=8, keps_float=True)
    diff = int((a[2] != b[2]).sum())
    print('slivers (area ~ 1e-7): %d of %d barycentric values differ between the readings' % (diff, int((a[0] >= 0).sum()) * 3))
    assert diff > 0
