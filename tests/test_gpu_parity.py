"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs.  Bars (BASELINE.json north_star): face indices bit-exact; rendered RGB and gradients
within 1e-4 relative fp32.  The rasteriser/clipper arithmetic is additionally required to be BIT-exact in its float
outputs (same op order, no FMA contraction)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O                                              # noqa: E402  (checker only)
from dbw_amd import ops                                         # noqa: E402
from dbw_amd.structures import PackedScene                      # noqa: E402

DEV = 'cuda:0'
REL = 1e-4


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def random_faces(n_faces, seed, zmin=0.5, zmax=5.0, spread=1.2, size=0.5):
    g = torch.Generator().manual_seed(seed)
    c = (torch.rand(n_faces, 1, 2, generator=g) * 2 - 1) * spread
    xy = c + (torch.rand(n_faces, 3, 2, generator=g) * 2 - 1) * size
    z = torch.rand(n_faces, 3, 1, generator=g) * (zmax - zmin) + zmin
    return torch.cat([xy, z], -1).contiguous()


@pytest.fixture
def raster_flags():
    """Sets the library's debug switches for one test (include/dbw_hip.h: dbw_debug_set_flags) and restores the product setting."""
    from dbw_amd import _lib
    lib = _lib.load()
    yield lib.dbw_debug_set_flags
    lib.dbw_debug_set_flags(0)


# ---------------------------------------------------------------------------------------------------------------------
# rasteriser: operator-level drop-in
# ---------------------------------------------------------------------------------------------------------------------
def test_shared_reciprocal_division_equals_ieee_division_on_this_gpu():
    """div_fast (raster_math.h) against `/` with the real v_rcp_f32: 64 M operand pairs spanning the guarded range (numerators 0 or
    2^-60..2^62 of either sign, denominators 2^-27..2^40 of either sign), plus quotients that sit next to a rounding boundary."""
    from dbw_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(1)
    n_ = 1 << 26
    def spread(lo, hi):
        e = torch.rand(n_, device=DEV, generator=g) * (hi - lo) + lo
        m = torch.rand(n_, device=DEV, generator=g) + 1.0
        sgn = (torch.rand(n_, device=DEV, generator=g) < 0.5).float() * 2 - 1
        return (sgn * m * torch.exp2(e)).float()
    num, den = spread(-60, 62), spread(-27, 40)
    num[:4096] = 0.0
    q = torch.rand(1 << 22, device=DEV, generator=g) * 1.5 + 0.5
    d2 = torch.rand(1 << 22, device=DEV, generator=g) * 7.5 + 0.5
    n2 = (d2.double() * q.double()).float()
    n2 = torch.nextafter(n2, torch.where(torch.rand(1 << 22, device=DEV, generator=g) < 0.5, n2 * 2, n2 * 0))
    num, den = torch.cat([num, n2]).contiguous(), torch.cat([den, d2]).contiguous()
    bad = torch.zeros(1, dtype=torch.int64, device=DEV)
    import device_checks
    device_checks.call('dbwt_divcheck', num.data_ptr(), den.data_ptr(), num.numel(), bad.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert int(bad.item()) == 0


@pytest.mark.parametrize('H,W,K,nf,blur,persp,clipb', [
    (33, 47, 4, 60, 1e-3, True, True),        # ragged image size (partial tiles)
    (64, 48, 10, 300, math.log(1e4 - 1) * 1e-4, True, True),   # the coarse renderer's setting, H > W
    (40, 40, 1, 100, 0.0, True, True),        # hard raster (env pass)
    (32, 64, 16, 20, 5e-3, False, False),     # K > faces hit, no perspective correction / clipping
    (48, 80, 25, 700, 2e-4, True, True),      # > LIST_CAP faces per tile flush path, max K
])
@pytest.mark.parametrize('flags', [0, 256, 512, 768, 4096])     # product / IEEE divisions / no tile culling / neither / no per-tile face lists (include/dbw_hip.h)
def test_rasterize_forward_bit_exact(H, W, K, nf, blur, persp, clipb, flags, raster_flags):
    raster_flags(flags)
    fv = random_faces(nf, seed=nf + K)
    # two meshes packed back to back, the second one a shifted copy
    fv = torch.cat([fv, fv * torch.tensor([0.9, -1.1, 1.0])], 0)
    first, num = torch.tensor([0, nf]), torch.tensor([nf, nf])
    ref = O.rasterize_fwd_raw(fv, first, num, None, (H, W), blur, K, persp, clipb, n_threads=8)
    out = ops.rasterize_meshes(fv.to(DEV), first.to(DEV), num.to(DEV), None, (H, W), blur, K, 0, 0, persp, clipb, False)
    assert out[0].dtype == torch.int64
    assert torch.equal(out[0].cpu(), ref[0]), f'pix_to_face mismatch on {(out[0].cpu() != ref[0]).sum().item()} slots'
    for name, a, b in zip(['zbuf', 'bary', 'dists'], out[1:], ref[1:]):
        assert torch.equal(a.cpu(), b), f'{name}: max abs diff {(a.cpu() - b).abs().max().item()}'


@pytest.mark.parametrize('H,W,K,nf,blur', [(64, 48, 10, 300, math.log(1e4 - 1) * 1e-4), (40, 40, 1, 100, 0.0)])
@pytest.mark.parametrize('flags', [0, 256, 4096])
def test_rasterize_forward_bit_exact_with_backface_culling(H, W, K, nf, blur, flags, raster_flags):
    """`cull_backfaces=True` of the `_C.rasterize_meshes` signature (no shipped config sets it; the boundary declares it): faces whose
    NDC winding is negative are dropped -- by the per-face record on the device (REC box emptied: never binned, never evaluated), per pixel
    in the oracle -- and what is left is bit-exact, soft and hard.  Random faces are about half back-facing, so the culled and the unculled
    lists differ substantially (asserted)."""
    raster_flags(flags)
    fv = random_faces(nf, seed=nf + K)
    fv = torch.cat([fv, fv * torch.tensor([0.9, -1.1, 1.0])], 0)         # (the mirrored copy flips every winding)
    first, num = torch.tensor([0, nf]), torch.tensor([nf, nf])
    ref = O.rasterize_fwd_raw(fv, first, num, None, (H, W), blur, K, True, True, cull_backfaces=True, n_threads=8)
    unculled = O.rasterize_fwd_raw(fv, first, num, None, (H, W), blur, K, True, True, cull_backfaces=False, n_threads=8)
    assert (ref[0] != unculled[0]).float().mean() > 0.05 and (ref[0] >= 0).sum() > 100
    out = ops.rasterize_meshes(fv.to(DEV), first.to(DEV), num.to(DEV), None, (H, W), blur, K, 0, 0, True, True, True)
    assert torch.equal(out[0].cpu(), ref[0]), f'pix_to_face mismatch on {(out[0].cpu() != ref[0]).sum().item()} slots'
    for name, a, b in zip(['zbuf', 'bary', 'dists'], out[1:], ref[1:]):
        assert torch.equal(a.cpu(), b), f'{name}: max abs diff {(a.cpu() - b).abs().max().item()}'


def test_two_level_binning_is_bit_identical_to_the_full_scan(monkeypatch):
    """Coarse 64x64-pixel bins (default) vs every tile scanning every face: same candidate order, so every output bit matches,
    on an image spanning several bins with ragged borders and uneven meshes (one empty)."""
    H, W, K = 150, 200, 10
    fv = torch.cat([random_faces(900, seed=1, size=0.15), random_faces(300, seed=2, size=0.6)], 0).to(DEV)
    first, num = torch.tensor([0, 900, 900]).to(DEV), torch.tensor([900, 0, 300]).to(DEV)
    blur = math.log(1e4 - 1) * 1e-4
    outs = []
    for on in (True, False):
        monkeypatch.setattr(ops, 'COARSE_BINS', on)
        outs.append(ops.rasterize_meshes(fv, first, num, None, (H, W), blur, K, 0, 0, True, True, False))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert (outs[0][0] >= 0).sum() > 1000


@pytest.mark.parametrize('H,W,nf,size,K', [
    (64, 64, 900, 3.0, 6),        # one bin: 900 faces that each cover most of the image -> ~30 000 list entries for a pool of 8 192
    (150, 200, 1500, 1.5, 10),    # twelve bins with ragged borders, all of them far over their share of the pool
    (130, 70, 3600, 0.8, 4),      # some bins fit, some do not (whichever reserves last): the result must not depend on it
])
def test_per_tile_face_lists_fall_back_to_the_coarse_bins_when_the_pool_is_full(H, W, nf, size, K, raster_flags):
    """cell_bin_kernel splits every 64x64-pixel bin into the face lists of its 8x8-pixel tiles; the lists live in a pool sized for
    DBW_CELL_POOL_PER_TILE = 128 faces per tile.  Scenes of large faces overflow it: the bins that do not fit mark their tiles and those
    tiles walk the coarse bin as before -- bit-exact against the oracle either way, and identical to the run without tile lists."""
    fv = random_faces(nf, seed=nf + H, size=size, spread=0.8)
    first, num = torch.tensor([0]), torch.tensor([nf])
    blur = 1e-3
    ref = O.rasterize_fwd_raw(fv, first, num, None, (H, W), blur, K, True, True, n_threads=8)
    outs = []
    for flags in (0, 4096):
        raster_flags(flags)
        out = ops.rasterize_meshes(fv.to(DEV), first.to(DEV), num.to(DEV), None, (H, W), blur, K, 0, 0, True, True, False)
        outs.append(out)
        assert torch.equal(out[0].cpu(), ref[0]), (flags, (out[0].cpu() != ref[0]).sum().item())
        for name, a, b in zip(['zbuf', 'bary', 'dists'], out[1:], ref[1:]):
            assert torch.equal(a.cpu(), b), (flags, name)
    # which bins overflow is a race between their workgroups, and so is every rank a tile takes in the launch order: the order has to be a
    # permutation of the tiles whatever the interleaving (a tile launched twice or never shows here, or as a memory fault)
    raster_flags(0)
    for rep in range(12):
        out = ops.rasterize_meshes(fv.to(DEV), first.to(DEV), num.to(DEV), None, (H, W), blur, K, 0, 0, True, True, False)
        assert torch.equal(out[0].cpu(), ref[0]), rep
    # the scene does overflow the pool: NDC boxes of the faces against the 8x8-pixel tiles, on the host
    s = max(H, W) / min(H, W)
    xr, yr = (s if W > H else 1.0), (s if H > W else 1.0)
    lo, hi = fv[..., :2].min(1).values, fv[..., :2].max(1).values
    tx = ((hi[:, 0].clamp(-xr, xr) - lo[:, 0].clamp(-xr, xr)) / (2 * xr) * W / 8).clamp(min=1)
    ty = ((hi[:, 1].clamp(-yr, yr) - lo[:, 1].clamp(-yr, yr)) / (2 * yr) * H / 8).clamp(min=1)
    tiles = ((H + 7) // 8) * ((W + 7) // 8)
    assert float((tx * ty).sum()) > 1.5 * tiles * 128           # DBW_CELL_POOL_PER_TILE (csrc/raster_common.h)


def test_per_tile_face_lists_on_sparse_multi_view_scenes_match_the_coarse_walk(raster_flags):
    """Several views with different face counts (one empty), small faces: most tiles are empty, the work list of every view puts
    the occupied tiles first -- outputs identical to the walk of the coarse bins, bit for bit, for the soft and the hard pass."""
    H, W = 150, 200
    fv = torch.cat([random_faces(400, seed=1, size=0.08), random_faces(30, seed=2, size=0.3), random_faces(200, seed=3, size=0.12)], 0).to(DEV)
    first, num = torch.tensor([0, 400, 400, 430]).to(DEV), torch.tensor([400, 0, 30, 200]).to(DEV)
    for K, blur in ((10, math.log(1e4 - 1) * 1e-4), (1, 0.0), (4, 1e-3)):
        outs = []
        for flags in (0, 4096):
            raster_flags(flags)
            outs.append(ops.rasterize_meshes(fv, first, num, None, (H, W), blur, K, 0, 0, True, True, False))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        assert (outs[0][0] >= 0).sum() > 500
        assert (outs[0][0][1] >= 0).sum() == 0


@pytest.mark.parametrize('name', ['front', 'straddle', 'ties'])
def test_rasterize_reproduces_the_frozen_tiny_scenes(golden_dir, name):
    """The HIP rasteriser (forward bit-exact, backward within 1e-5) on the committed tiny scenes of tests/golden/raster_tiny.npz
    (SURVEY.md 8c golden vector 7: K > faces, a triangle straddling the near plane both ways, depth ties)."""
    from test_oracle_golden import _load, _tiny_scene
    g = _load(golden_dir, 'raster_tiny.npz')
    fv, first, num, nbr, size, blur, K = _tiny_scene(g, name)
    fvd = fv.to(DEV).requires_grad_(True)
    nb = None if nbr is None else nbr.to(torch.int32).to(DEV)
    p2f, zbuf, bary, dists = ops.rasterize_meshes(fvd, first.to(torch.int32).to(DEV), num.to(torch.int32).to(DEV), nb, size, blur, K, 0, 0,
                                                  True, True, False)
    for got, key in ((p2f, 'p2f'), (zbuf, 'zbuf'), (bary, 'bary'), (dists, 'dists')):
        assert torch.equal(got.cpu(), g[f'{name}/{key}']), key
    ((zbuf * g[f'{name}/g_zbuf'].to(DEV)).sum() + (bary * g[f'{name}/g_bary'].to(DEV)).sum() + (dists * g[f'{name}/g_dists'].to(DEV)).sum()).backward()
    ref = g[f'{name}/g_face_verts']
    assert float((fvd.grad.cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_rasterize_ties_and_degenerates():
    """Coincident faces (identical z everywhere: tie broken by face id), zero-area faces, faces behind the camera."""
    base = random_faces(8, seed=5)
    fv = torch.cat([base, base, base[:2] * torch.tensor([1., 1., 0.]) + torch.tensor([0., 0., -1.])], 0)
    fv[3, 2] = fv[3, 1]                                          # zero-area face
    first, num = torch.tensor([0]), torch.tensor([fv.shape[0]])
    ref = O.rasterize_fwd_raw(fv, first, num, None, (37, 29), 1e-3, 6)
    out = ops.rasterize_meshes(fv.to(DEV), first.to(DEV), num.to(DEV), None, (37, 29), 1e-3, 6, 0, 0, True, True, False)
    assert torch.equal(out[0].cpu(), ref[0])
    assert torch.equal(out[3].cpu(), ref[3])
    p = ref[0]
    both = (p[..., 0] >= 0) & (p[..., 1] >= 0)
    assert both.any() and torch.all(p[..., 1][both] != p[..., 0][both])


def test_rasterize_empty_and_errors():
    fv = random_faces(4, seed=1).to(DEV)
    first, num = torch.tensor([0, 4], device=DEV), torch.tensor([4, 0], device=DEV)      # second mesh is empty
    p2f, zbuf, bary, dists = ops.rasterize_meshes(fv, first, num, None, (16, 16), 1e-3, 3, 0, 0, True, True)
    assert torch.all(p2f[1] == -1) and torch.all(zbuf[1] == -1) and torch.all(bary[1] == -1) and torch.all(dists[1] == -1)
    with pytest.raises(ValueError):
        ops.rasterize_meshes(fv, first, num, None, (16, 16), 1e-3, 26)
    with pytest.raises(RuntimeError):
        ops.rasterize_meshes(fv.cpu(), first, num, None, (16, 16), 1e-3, 3)       # no CPU fallback
    with pytest.raises(RuntimeError):
        ops.rasterize_meshes(fv, first, num, None, (16, 16), -1.0, 3)             # C ABI rejects blur < 0


def test_rasterize_operators_take_pytorch3d_positional_signatures():
    """`_C.rasterize_meshes(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, image_size,
    blur_radius, faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct, clip_barycentric_coords, cull_backfaces)` and
    `_C.rasterize_meshes_backward(face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists, perspective_correct,
    clip_barycentric_coords)` (PyTorch3D 0.7.1), called positionally in exactly that order, against the oracle;
    bin_size / max_faces_per_bin take PyTorch3D's values (None, 0, heuristics) without changing a bit."""
    H, W, K, nf = 40, 56, 5, 80
    fv = random_faces(nf, seed=11)
    first, num = torch.tensor([0]), torch.tensor([nf])
    ref = O.rasterize_fwd_raw(fv, first, num, None, (H, W), 2e-3, K)
    fvd, firstd, numd = fv.to(DEV), first.to(DEV), num.to(DEV)
    for bin_size, max_faces_per_bin in ((0, 0), (None, None), (16, 10000), (32, 18)):
        out = ops.rasterize_meshes(fvd, firstd, numd, None, (H, W), 2e-3, K, bin_size, max_faces_per_bin, True, True, False)
        assert len(out) == 4 and out[0].dtype == torch.int64
        for a, b in zip(out, ref):
            assert torch.equal(a.cpu(), b)
    g = torch.Generator().manual_seed(3)
    gz, gb, gd = torch.randn(ref[1].shape, generator=g), torch.randn(ref[2].shape, generator=g), torch.randn(ref[3].shape, generator=g)
    grad = ops.rasterize_meshes_backward(fvd, out[0], gz.to(DEV), gb.to(DEV), gd.to(DEV), True, True)
    assert grad.shape == fv.shape and rel_err(grad, O.rasterize_bwd_raw(fv, ref[0], gz, gb, gd)) < REL
    with pytest.raises(ValueError):
        ops.rasterize_meshes(fvd, firstd, numd, None, (H, W), 2e-3, K, -1, 0, True, True, False)
    with pytest.raises(ValueError):
        ops.rasterize_meshes_backward(fvd, out[0], gz.to(DEV), gd.to(DEV), gd.to(DEV), True, True)


def test_rasterize_backward_matches_oracle():
    H, W, K, nf = 40, 56, 5, 80
    fv = random_faces(nf, seed=11)
    first, num = torch.tensor([0]), torch.tensor([nf])
    g = torch.Generator().manual_seed(3)
    ref = O.rasterize_fwd_raw(fv, first, num, None, (H, W), 2e-3, K)
    gz, gb, gd = torch.randn(ref[1].shape, generator=g), torch.randn(ref[2].shape, generator=g), torch.randn(ref[3].shape, generator=g)
    g_ref = O.rasterize_bwd_raw(fv, ref[0], gz, gb, gd)
    fvd = fv.to(DEV).requires_grad_(True)
    out = ops.rasterize_meshes(fvd, first.to(DEV), num.to(DEV), None, (H, W), 2e-3, K, 0, 0, True, True, False)
    (out[1] * gz.to(DEV) + (out[2] * gb.to(DEV)).sum(-1) + out[3] * gd.to(DEV)).sum().backward()
    assert rel_err(fvd.grad, g_ref) < REL
    # each gradient stream alone (NULL pointers for the others)
    for sel in range(3):
        fvd.grad = None
        out = ops.rasterize_meshes(fvd, first.to(DEV), num.to(DEV), None, (H, W), 2e-3, K, 0, 0, True, True, False)
        z = torch.zeros
        parts = [gz, gb, gd]
        (out[1 + sel] * parts[sel].to(DEV)).sum().backward()
        zeros = [z(gz.shape), z(gb.shape), z(gd.shape)]
        zeros[sel] = parts[sel]
        assert rel_err(fvd.grad, O.rasterize_bwd_raw(fv, ref[0], *zeros)) < REL


# ---------------------------------------------------------------------------------------------------------------------
# camera transform + z clipping
# ---------------------------------------------------------------------------------------------------------------------
def _camera_inside_scene(seed, B=3):
    """A closed icosphere around the cameras (like the sky dome): many faces straddle z = z_clip (cases 3 and 4)."""
    torch.manual_seed(seed)
    verts, faces = O.get_icosphere(2, flip_faces=True)
    verts = verts * 3.0 + 0.05 * torch.randn_like(verts)
    C = torch.randn(B, 3) * 0.4
    R, T = O.look_at_cameras(C, at=(0.3, 0.2, 2.5))
    Kmat = torch.tensor([[2.1, 0, 0.05, 0], [0, 2.1, -0.03, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=torch.float32)
    return verts, faces, R, T, Kmat


@pytest.mark.parametrize('persp', [True, False])
def test_project_clip_bit_exact(persp):
    verts, faces, R, T, Kmat = _camera_inside_scene(0)
    B, Fs = R.shape[0], faces.shape[0]
    zc = 0.25
    ndc = O.transform_to_ndc(verts, R, T, Kmat, 1e-8)
    fv = ndc[:, faces].reshape(B * Fs, 3, 3)
    ref = O.clip_faces(fv, torch.arange(B) * Fs, torch.full((B,), Fs), zc, persp)
    cl = ops.project_clip(verts.to(DEV), faces.to(torch.int32).to(DEV), R.to(DEV), T.to(DEV), Kmat.to(DEV), 1e-8, zc, persp)
    num = cl['num_faces'].cpu().long()
    assert torch.equal(num, ref['num_faces'])
    assert (ref['neighbor'] >= 0).any() and ref['has_conv'].any(), 'test scene must exercise cases 3 and 4'
    for b in range(B):
        n, s = int(num[b]), int(ref['first_idx'][b])
        got = cl['face_verts'][b, :n].cpu()
        exp = ref['face_verts'][s:s + n]
        assert torch.equal(got, exp), f'view {b}: max diff {(got - exp).abs().max().item()}'
        assert torch.equal(cl['c2o'][b, :n].cpu().long(), ref['clipped_to_orig'][s:s + n] - b * Fs)
        nb_ref = ref['neighbor'][s:s + n]
        nb_ref = torch.where(nb_ref >= 0, nb_ref - s + b * 2 * Fs, nb_ref)
        assert torch.equal(cl['neighbor'][b, :n].cpu().long(), nb_ref)
        assert torch.equal(cl['clip_code'][b, :n].cpu() >= 0, ref['has_conv'][s:s + n])


def test_project_clip_disabled_is_plain_projection():
    verts, faces, R, T, Kmat = _camera_inside_scene(1)
    cl = ops.project_clip(verts.to(DEV), faces.to(torch.int32).to(DEV), R.to(DEV), T.to(DEV), Kmat.to(DEV), 1e-8, None, True)
    ndc = O.transform_to_ndc(verts, R, T, Kmat, 1e-8)
    Fs = faces.shape[0]
    assert torch.all(cl['num_faces'].cpu() == Fs)
    assert torch.equal(cl['face_verts'][:, :Fs].cpu(), ndc[:, faces])


def test_project_clip_backward_lds_table_equals_global_atomics_path():
    """dbw_project_clip_bwd accumulates in an LDS table when the mesh has at most 4096 vertices and straight into memory otherwise:
    the same mesh with 5000 unused vertices appended must give the same gradient (clipped views: slots shifted between views)."""
    verts, faces, R, T, Kmat = _camera_inside_scene(3, B=7)
    B, Fs = R.shape[0], faces.shape[0]
    fi = faces.to(torch.int32).to(DEV)
    args = (R.to(DEV), T.to(DEV), Kmat.to(DEV))
    g = torch.randn(B, 2 * Fs, 3, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    out = []
    for extra in (0, 5000):
        v = torch.cat([verts, torch.randn(extra, 3)]).to(DEV)
        cl = ops.project_clip(v, fi, *args, 1e-8, 0.25, True)
        assert int(cl['num_faces'].min()) != int(cl['num_faces'].max()), 'views must clip differently'
        out.append(ops.project_clip_bwd(v, fi, *args, cl, g, 1e-8, 0.25, True).clone())
    nv = verts.shape[0]
    assert float(out[1][nv:].abs().max()) == 0.0
    assert rel_err(out[0], out[1][:nv]) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# full render pass (project -> clip -> raster -> shade -> blend) forward + backward
# ---------------------------------------------------------------------------------------------------------------------
def _packed(scene, pads=None):
    maps = scene['maps']
    pads = pads or [(0, 0)] * len(maps)
    desc, _ = PackedScene.describe_maps([m.shape[:2] for m in maps], pads, DEV)
    flat = torch.cat([m.reshape(-1) for m in maps]).detach().to(DEV)
    return PackedScene(scene['verts'].detach().to(DEV), scene['faces'].to(torch.int32).to(DEV), scene['face_uvs'].float().to(DEV),
                       scene['face_map'].to(torch.int32).to(DEV), desc, flat)


def _render_both(scene, R, T, Kmat, H, W, sigma, K, detach_bary, faces_alpha, z_clip=0.001, bg=(0., 0., 0.), seed=0, lds=False, clip_inside=True):
    """-> dict of (hip, oracle) pairs: image, grad verts, grad maps, grad alpha."""
    verts_o = scene['verts'].detach().clone().requires_grad_(True)
    maps_o = [m.detach().clone().requires_grad_(True) for m in scene['maps']]
    fa_o = None if faces_alpha is None else faces_alpha.detach().clone().requires_grad_(True)
    sc = dict(scene, verts=verts_o, maps=maps_o)
    fa_rep = None if fa_o is None else fa_o.repeat(R.shape[0])          # the reference passes alpha.repeat(B) (dbw.py:219)
    img_o = O.render(sc, R, T, Kmat, (H, W), sigma, K, detach_bary, fa_rep, z_clip, bg, n_threads=8, clip_inside=clip_inside)
    w = torch.rand(img_o.shape, generator=torch.Generator().manual_seed(seed))
    (img_o * w).sum().backward()

    ps = _packed(scene)
    ps.verts.requires_grad_(True)
    ps.maps.requires_grad_(True)
    fa_h = None if faces_alpha is None else faces_alpha.detach().to(DEV).requires_grad_(True)
    cfg = ops.RenderCfg(H, W, K, sigma, z_clip, True, detach_bary, scene['faces'].shape[0], lds_aggregate=lds, clip_inside=clip_inside)
    img_h = ops.render_scene(ps.verts, ps.maps, fa_h, ps.faces, R.to(DEV), T.to(DEV), Kmat.to(DEV), ps.face_uvs, ps.face_map,
                             ps.map_desc, ops.make_bg(bg), cfg)
    (img_h * w.to(DEV)).sum().backward()
    res = {'image': (img_h, img_o), 'g_maps': (ps.maps.grad, torch.cat([(m.grad if m.grad is not None else torch.zeros_like(m)).reshape(-1) for m in maps_o]))}
    if verts_o.grad is not None:
        res['g_verts'] = (ps.verts.grad, verts_o.grad)
    if fa_o is not None:
        res['g_alpha'] = (fa_h.grad, fa_o.grad)
    return res


def _model(seed=3, n_blocks=4, ts=32, hw=(48, 64), fpp=6):
    m = O.OracleDBW(hw, n_blocks=n_blocks, txt_size=ts, faces_per_pixel=fpp, seed=seed)
    R, T, Km = O.synthetic_cameras(3, R_world=m.R_world[0], dist=2.8)
    return m, R, T, Km


@pytest.mark.parametrize('detach_bary,lds,fused', [(True, True, True), (True, False, True), (False, False, False)])
def test_render_with_the_sigmoid_opacity_of_clip_inside_false_matches_oracle(detach_bary, lds, fused, monkeypatch):
    """`clip_inside=False` of the reference's Renderer (renderer.py:41,257-258; no shipped config): the blend opacity is
    sigmoid(-d / sigma) -- a fragment keeps a gradient to its distance INSIDE its face too.  Image and every gradient (vertices through the
    distances, and through the barycentrics when they are attached; maps; opacities) against the oracle's render with the same switch;
    uv-fragments and plain fragments, LDS tables and atomics.
    Attached barycentrics only through the operator-level kernels, whose arithmetic is the oracle's operation by operation: a soft pass has
    fragments in the blur band OUTSIDE their face, and where such a pixel falls on the line on which the perspective-correct denominator
    vanishes (barycentrics of 1e10 before clipping) the gradient through the barycentrics is rounding noise amplified by 1e6 -- the
    fused backward (v_exp_f32 opacities, 1e-7 off) then differs from ANY other evaluation order on that one fragment.  No pass of the
    reference attaches the barycentrics of a soft pass (every shipped config sets detach_bary: True for the blocks; dbw.py:137 attaches them for
    the environment's hard K = 1 pass only, whose fragments all lie inside their face)."""
    monkeypatch.setattr(ops, 'FUSED_FORWARD', fused)
    monkeypatch.setattr(ops, 'FUSED_BACKWARD', fused)
    m, R, T, Km = _model(fpp=6)
    with torch.no_grad():
        scene = m.build_blocks(True, True, False, None, kill_blocks=False)
    fa = (torch.rand(scene['faces'].shape[0] // m.BNF, generator=torch.Generator().manual_seed(1)) * 0.8 + 0.1).repeat_interleave(m.BNF)
    res = _render_both(scene, R, T, Km[0], 48, 64, 1e-4, 6, detach_bary, fa, bg=(0.1, 0.2, 0.3), lds=lds, clip_inside=False)
    ref = _render_both(scene, R, T, Km[0], 48, 64, 1e-4, 6, detach_bary, fa, bg=(0.1, 0.2, 0.3), lds=lds)
    for k, (a, b) in res.items():
        assert rel_err(a, b) < REL, f'{k}: rel err {rel_err(a, b)}'
    assert res['g_verts'][1].abs().max() > 0
    assert rel_err(res['image'][1], ref['image'][1]) > 1e-2          # (the two opacities do differ on this scene)


@pytest.mark.parametrize('fpp', [6, 16, 25])
def test_render_blocks_soft_pass_matches_oracle(fpp):
    """fg pass: sigma=1e-4, learned opacities, detach_bary=True (geometry gradient through dists only); every compiled
    faces_per_pixel bucket of the fused kernels (<=10, <=16, <=25)."""
    m, R, T, Km = _model(fpp=fpp)
    with torch.no_grad():
        scene = m.build_blocks(True, True, False, None, kill_blocks=False)
    fa = (torch.rand(scene['faces'].shape[0] // m.BNF, generator=torch.Generator().manual_seed(1)) * 0.8 + 0.1).repeat_interleave(m.BNF)
    res = _render_both(scene, R, T, Km[0], 48, 64, 1e-4, fpp, True, fa, bg=(0., 0., 0.))
    for k, (a, b) in res.items():
        assert rel_err(a, b) < REL, f'{k}: rel err {rel_err(a, b)}'
    assert res['g_verts'][1].abs().max() > 0


def test_render_fine_pass_no_alpha_matches_oracle():
    m, R, T, Km = _model(seed=5)
    with torch.no_grad():
        scene = m.build_blocks(True, False, False, None, kill_blocks=False)
    res = _render_both(scene, R, T, Km[0], 48, 64, 5e-6, 6, True, None)
    for k, (a, b) in res.items():
        assert rel_err(a, b) < REL, f'{k}: rel err {rel_err(a, b)}'


@pytest.mark.parametrize('rig', [(30.0, 4.82), (5.0, 1.5)])          # cameras looking down at the ground / seeing half sky, half ground
@pytest.mark.parametrize('lds', [False, True])
def test_render_env_hard_pass_with_clipping_matches_oracle(lds, rig):
    """env pass: sigma=0, 1 face per pixel, detach_bary=False -> geometry gradient only through barycentrics -> uv;
    the camera sits inside the dome so z-clipping (cases 3/4 + barycentric back-conversion) is live.
    lds=True is the training path's form: hard uv-fragments (frag_layout 3) + the specialised backward kernel."""
    m, R, T, Km = _model(seed=7, ts=16)
    R, T, Km = O.synthetic_cameras(3, R_world=m.R_world[0], dist=2.8, elev_deg=rig[0], f_ndc=rig[1])
    with torch.no_grad():
        scene = m.build_env(True, False)
    res = _render_both(scene, R, T, Km[0], 48, 64, 0.0, 1, False, None, bg=(0.1, 0.2, 0.3), lds=lds)
    for k, (a, b) in res.items():
        assert rel_err(a, b) < REL, f'{k}: rel err {rel_err(a, b)}'
    assert res['g_verts'][1].abs().max() > 0


def test_constant_geometry_faces_get_no_vertex_gradient_and_the_others_the_same():
    """const_faces (the sky dome's vertices are a buffer): the vertices only those faces use get no gradient, the gradient through
    every other face and the texture gradient are unchanged -- both fused backward kernels of the hard pass."""
    m, R, T, Km = _model(seed=7, ts=16)
    R, T, Km = O.synthetic_cameras(3, R_world=m.R_world[0], dist=2.8, elev_deg=5.0, f_ndc=1.5)       # half sky, half ground
    with torch.no_grad():
        scene = m.build_env(True, False)
    nsky = int((scene['face_map'] == 0).sum())
    assert 0 < nsky < scene['faces'].shape[0] and bool((scene['face_map'][:nsky] == 0).all())
    sky_only = torch.ones(scene['verts'].shape[0], dtype=torch.bool)
    sky_only[scene['faces'][nsky:].reshape(-1).long()] = False
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    for lds in (False, True):
        out = []
        for cf in (0, nsky):
            ps = _packed(scene)
            ps.maps.requires_grad_(True)
            ps.verts.requires_grad_(True)
            cfg = ops.RenderCfg(48, 64, 1, 0.0, 0.001, True, False, scene['faces'].shape[0], lds_aggregate=lds, const_faces=cf)
            img = ops.render_scene(ps.verts, ps.maps, None, ps.faces, *args, ps.face_uvs, ps.face_map, ps.map_desc, None, cfg)
            (img * torch.rand(img.shape, generator=torch.Generator().manual_seed(5)).to(DEV)).sum().backward()
            out.append((ps.maps.grad.cpu(), ps.verts.grad.cpu()))
        assert rel_err(out[1][0], out[0][0]) < 1e-6
        assert float(out[0][1][sky_only].abs().max()) > 0 and float(out[1][1][sky_only].abs().max()) == 0.0
        # vertices shared by sky and ground faces do not exist (two meshes), so the ground's gradient is simply the same
        assert rel_err(out[1][1][~sky_only], out[0][1][~sky_only]) < 1e-5


def test_render_pixel_faces_bit_exact_through_clipping():
    """Face indices after mapping clipped -> original faces are identical to the oracle's converted pix_to_face."""
    m, R, T, Km = _model(seed=9, ts=16)
    with torch.no_grad():
        scene = m.build_env(False, False)
        _, frag = O.render(scene, R, T, Km[0], (48, 64), 0.0, 2, False, None, 0.001, n_threads=8, return_fragments=True)
    ps = _packed(scene)
    cfg = ops.RenderCfg(48, 64, 2, 0.0, 0.001, True, False, scene['faces'].shape[0])
    cl, p2f, zbuf, bary, dists = ops.render_fragments(ps.verts, ps.faces, R.to(DEV), T.to(DEV), Km[0].to(DEV), cfg)
    Fs = scene['faces'].shape[0]
    c2o = cl['c2o'].view(-1).long()
    view = torch.arange(R.shape[0], device=DEV).view(-1, 1, 1, 1)
    orig = torch.where(p2f >= 0, c2o[p2f.clamp(min=0).long()] + view * Fs, torch.full_like(p2f, -1).long())
    assert torch.equal(orig.cpu(), frag['pix_to_face'])
    assert torch.equal(dists.cpu(), frag['dists']) and torch.equal(zbuf.cpu(), frag['zbuf'])


def test_circular_padding_by_index_wrap_equals_padded_copy():
    """dbw.py:339-341: sampling the UNPADDED map with (pad_left, pad_right) wrap == sampling the circularly padded copy."""
    m, R, T, Km = _model(seed=11, ts=32)
    with torch.no_grad():
        scene = m.build_blocks(False, True, False, None, kill_blocks=False)      # maps are padded copies here
    pl, pr = m.txt_padding
    assert pr > 0
    ps_pad = _packed(scene)
    unpadded = [mp[:, pl:mp.shape[1] - pr] for mp in scene['maps']]
    ps_wrap = _packed(dict(scene, maps=unpadded), pads=[(pl, pr)] * len(unpadded))
    cfg = ops.RenderCfg(48, 64, 6, 1e-4, 0.001, True, True, scene['faces'].shape[0])
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    a = ops.render_scene(ps_pad.verts, ps_pad.maps, None, ps_pad.faces, *args, ps_pad.face_uvs, ps_pad.face_map, ps_pad.map_desc, None, cfg)
    b = ops.render_scene(ps_wrap.verts, ps_wrap.maps, None, ps_wrap.faces, *args, ps_wrap.face_uvs, ps_wrap.face_map, ps_wrap.map_desc, None, cfg)
    assert torch.equal(a, b)
    assert a[:, 3].max() > 0.5


@pytest.mark.parametrize('agg', [False, True])
def test_decimated_maps_at_cell_resolution_match_upsampled_copy(agg):
    """dbw.py:331-334: rendering from cell-resolution maps + descriptor shift == rendering from the nearest-upsampled copy,
    forward and gradients (gradient of a cell = sum over its texels), with and without LDS pre-aggregation."""
    m, R, T, Km = _model(seed=13, ts=32)
    with torch.no_grad():
        scene = m.build_blocks(False, True, False, None, kill_blocks=False)
    pl, pr = m.txt_padding
    d, sh = 8, 3
    g = torch.Generator().manual_seed(0)
    cells = [torch.rand(32 // d, 32 // d, 3, generator=g) for _ in scene['maps']]
    full = [c.repeat_interleave(d, 0).repeat_interleave(d, 1) for c in cells]
    ps_full = _packed(dict(scene, maps=full), pads=[(pl, pr)] * len(full))
    ps_cell = _packed(dict(scene, maps=cells), pads=[(pl, pr)] * len(cells))
    ps_cell.map_desc = PackedScene.describe_maps([(32, 32)] * len(cells), [(pl, pr)] * len(cells), DEV, shift=sh)[0]
    fa = torch.full((scene['faces'].shape[0],), 0.7, device=DEV)
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    outs = []
    for ps, lds in ((ps_full, False), (ps_cell, agg)):
        ps.maps.requires_grad_(True)
        fa_ = fa.clone().requires_grad_(True)
        cfg = ops.RenderCfg(48, 64, 6, 1e-4, 0.001, True, True, scene['faces'].shape[0], lds_aggregate=lds)
        img = ops.render_scene(ps.verts, ps.maps, fa_, ps.faces, *args, ps.face_uvs, ps.face_map, ps.map_desc, None, cfg)
        w = torch.rand(img.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        (img * w).sum().backward()
        outs.append((img.detach(), ps.maps.grad, fa_.grad))
    assert rel_err(outs[1][0], outs[0][0]) < 1e-6
    g_full = outs[0][1].view(len(full), 32 // d, d, 32 // d, d, 3).sum((2, 4)).reshape(-1)
    assert rel_err(outs[1][1], g_full) < REL
    assert rel_err(outs[1][2], outs[0][2]) < REL


@pytest.mark.parametrize('which', ['fg', 'env'])
def test_operator_level_kernels_equal_fused_path(which, monkeypatch):
    """The stand-alone rasterise / shade-blend / raster-backward kernels (operator-level ABI, (N,H,W,K) fragments) and the
    fused forward/backward kernels (8x8-tile planar fragments) produce the same image and gradients.  The hard pass is bit-equal;
    the soft uv-fragment pass keeps 12 B payloads (b2 = 1 - b0 - b1, raster_math.h: PAY3), i.e. agrees to a few fp32 ulp."""
    m, R, T, Km = _model(seed=21, ts=16)
    with torch.no_grad():
        scene = m.build_blocks(False, True, False, None, kill_blocks=False) if which == 'fg' else m.build_env(False, False)
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    fa = None if which == 'env' else torch.full((scene['faces'].shape[0],), 0.6, device=DEV)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, 'FUSED_FORWARD', fused)
        monkeypatch.setattr(ops, 'FUSED_BACKWARD', fused)
        ps = _packed(scene)
        ps.maps.requires_grad_(True)
        ps.verts.requires_grad_(True)
        fa_ = None if fa is None else fa.clone().requires_grad_(True)
        cfg = ops.RenderCfg(45, 61, 6 if which == 'fg' else 1, 1e-4 if which == 'fg' else 0.0, 0.001, True, which == 'fg', scene['faces'].shape[0])
        img = ops.render_scene(ps.verts, ps.maps, fa_, ps.faces, *args, ps.face_uvs, ps.face_map, ps.map_desc, None, cfg)
        (img * torch.rand(img.shape, generator=torch.Generator().manual_seed(5)).to(DEV)).sum().backward()
        outs.append((img.detach(), ps.maps.grad, ps.verts.grad, None if fa_ is None else fa_.grad))
    if which == 'env':
        assert torch.equal(outs[0][0], outs[1][0])
    assert rel_err(outs[0][0], outs[1][0]) < 1e-5
    for a, b in zip(outs[0][1:], outs[1][1:]):
        if a is not None:
            assert rel_err(a, b) < 1e-5


def test_renderer_class_with_meshes_api_and_viz_purpose():
    """The reference-shaped entry: Renderer(img_size, **cfg.model.renderer).forward(Meshes.extend(B), R, T, faces_alpha=...)
    (renderer.py:84-98) through Meshes / TexturesUV / join_meshes_as_scene, against the oracle; plus viz_purpose=True
    (4x supersampled hard render + avg_pool, renderer.py:56-60,178-183)."""
    from dbw_amd import Renderer, Meshes, TexturesUV, join_meshes_as_scene
    m, R, T, Km = _model(seed=23, ts=16, hw=(40, 56))
    with torch.no_grad():
        bkg_maps = torch.sigmoid(m.p['texture_bkg']).to(DEV)
        g_maps = torch.sigmoid(m.p['texture_ground']).to(DEV)
        env_o = m.build_env(False, False)
    nb = m.bkg_verts.shape[0]
    bkg = Meshes(env_o['verts'][:nb].to(DEV), m.bkg_faces.to(DEV), TexturesUV(bkg_maps, m.bkg_faces.to(DEV), m.bkg_verts_uvs.to(DEV)))
    ground = Meshes(env_o['verts'][nb:].to(DEV), m.ground_faces.to(DEV), TexturesUV(g_maps, m.ground_faces.to(DEV), m.ground_verts_uvs.to(DEV)))
    scene = join_meshes_as_scene([bkg, ground])
    r = Renderer((40, 56), faces_per_pixel=1, cameras={'name': 'perspective'}, sigma=0, z_clip=0.001, detach_bary=False)
    with pytest.raises(NotImplementedError):
        r(scene.extend(3), R.to(DEV), T.to(DEV))                   # K not set yet (dbw.py:204-208)
    r.update_cameras(device=DEV, K=Km[0:1].to(DEV))
    assert r.cameras.K.shape == (1, 4, 4) and r.img_size == (40, 56) and r.init_kwargs['faces_per_pixel'] == 1
    img = r(scene.extend(3), R.to(DEV), T.to(DEV))
    ref = O.render(env_o, R, T, Km[0], (40, 56), 0.0, 1, False, None, 0.001, n_threads=8)
    assert img.shape == (3, 4, 40, 56) and rel_err(img, ref) < REL
    viz = r(scene.extend(3), R.to(DEV), T.to(DEV), viz_purpose=True)
    ref4 = O.render(env_o, R, T, Km[0], (160, 224), 0.0, 1, False, None, 0.001, n_threads=8)
    assert rel_err(viz, torch.nn.functional.avg_pool2d(ref4, 4, 4)) < REL
    # blocks with the circular u padding passed symbolically (dbw.py:339-342) and per-face opacities packed per view
    with torch.no_grad():
        blk_o = m.build_blocks(False, True, False, None, kill_blocks=False)
        verts = blk_o['verts'].reshape(m.n_blocks, -1, 3).to(DEV)
        maps = torch.sigmoid(m.p['textures']).to(DEV)
    blocks = Meshes(verts, m.block_faces[None].expand(m.n_blocks, -1, -1).to(DEV),
                    TexturesUV(maps, m.block_faces_uvs.to(DEV), m.block_verts_uvs.to(DEV), circular_pad=m.txt_padding))
    fg_scene = join_meshes_as_scene(blocks)
    rf = Renderer((40, 56), faces_per_pixel=6, cameras={'name': 'perspective'}, z_clip=0.001, detach_bary=True)
    rf.update_cameras(device=DEV, K=Km[0:1].to(DEV))
    alpha = torch.rand(m.n_blocks, generator=torch.Generator().manual_seed(0)).repeat_interleave(m.BNF).repeat(3)
    out = rf(fg_scene.extend(3), R.to(DEV), T.to(DEV), faces_alpha=alpha.to(DEV))
    ref = O.render(blk_o, R, T, Km[0], (40, 56), 1e-4, 6, True, alpha, 0.001, n_threads=8)
    assert rel_err(out, ref) < REL


def test_texture_space_binning_equals_atomic_scatter(monkeypatch):
    """Full-resolution maps under minification: the binned (record append + per-bin LDS reduce) texel-gradient path equals the
    atomic scatter, including circularly wrapped footprints and a deliberately tiny bin capacity (overflow -> atomics)."""
    m, R, T, Km = _model(seed=29, ts=64, hw=(72, 96), fpp=8)
    with torch.no_grad():
        scene = m.build_blocks(False, True, False, None, kill_blocks=False)
    pl, pr = m.txt_padding
    unpadded = [mp[:, pl:mp.shape[1] - pr] for mp in scene['maps']]
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    w = torch.rand(3, 4, 72, 96, generator=torch.Generator().manual_seed(5)).to(DEV)
    grads = []
    for binned, counted in ((False, True), (True, True), (True, False)):
        monkeypatch.setattr(ops, 'TEXTURE_BINS', binned)
        ps = _packed(dict(scene, maps=unpadded), pads=[(pl, pr)] * len(unpadded))
        assert int(ps.map_desc[0, 6]) == len(unpadded)
        if not counted:                       # a descriptor table without its row count (include/dbw_hip.h: allowed): the kernels
            ps.map_desc = ps.map_desc.clone()  # read the descriptors from memory instead of their LDS copy
            ps.map_desc[0, 6] = 0
        ps.maps.requires_grad_(True)
        bins = PackedScene.describe_bins([(64, 64)] * len(unpadded), DEV)
        cfg = ops.RenderCfg(72, 96, 8, 1e-4, 0.001, True, True, scene['faces'].shape[0], lds_aggregate=False, texbins=bins)
        img = ops.render_scene(ps.verts, ps.maps, None, ps.faces, *args, ps.face_uvs, ps.face_map, ps.map_desc, None, cfg)
        (img * w).sum().backward()
        grads.append(ps.maps.grad)
    assert bins[2] == 4 * len(unpadded) and grads[0].abs().max() > 0
    assert rel_err(grads[1], grads[0]) < 1e-5
    assert rel_err(grads[2], grads[0]) < 1e-5


def test_bin_reduction_in_fixed_point_holds_long_record_runs_and_wide_dynamic_range(monkeypatch):
    """`texbin_reduce_kernel` accumulates a bin in int32 fixed point with a block exponent and an overflow BUDGET (shade_blend.hip): a
    workgroup may add 2047 records blindly, then has to look at its tile.  Here ONE 32x32 bin per map takes every record of a 300x400
    render -- tens of thousands per workgroup, dozens of budget checks -- and the image gradient spans nine decades (a few pixels weigh
    1e6, most 1e-3: batches of small gradients arrive before and after the large ones, so the tile's exponent is raised on the way and
    small addends meet a coarse grid): against the atomic scatter, which adds the same fp32 products in fp32, to 1e-5 of the largest
    entry, and texel by texel for the texels only small pixels touch (their sums must not drown in the large pixels' quantum)."""
    m, R, T, Km = _model(seed=31, ts=32, hw=(300, 400), fpp=8)
    with torch.no_grad():
        scene = m.build_blocks(False, True, False, None, kill_blocks=False)
    pl, pr = m.txt_padding
    unpadded = [mp[:, pl:mp.shape[1] - pr] for mp in scene['maps']]
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    g = torch.Generator().manual_seed(6)
    w = torch.rand(3, 4, 300, 400, generator=g) * 1e-3
    w[:, :, 140:150, 190:200] *= 1e9                       # a 10x10 patch of heavy pixels in the middle of the blocks
    w = w.to(DEV)
    grads = []
    for binned in (False, True):
        monkeypatch.setattr(ops, 'TEXTURE_BINS', binned)
        ps = _packed(dict(scene, maps=unpadded), pads=[(pl, pr)] * len(unpadded))
        ps.maps.requires_grad_(True)
        bins = PackedScene.describe_bins([(32, 32)] * len(unpadded), DEV)
        cfg = ops.RenderCfg(300, 400, 8, 1e-4, 0.001, True, True, scene['faces'].shape[0], lds_aggregate=False, texbins=bins)
        img = ops.render_scene(ps.verts, ps.maps, None, ps.faces, *args, ps.face_uvs, ps.face_map, ps.map_desc, None, cfg)
        (img * w).sum().backward()
        grads.append(ps.maps.grad.clone())
    assert bins[2] == len(unpadded)
    ref, got = grads
    assert float(ref.abs().max()) > 1e3 and rel_err(got, ref) < 1e-5
    # every bin holds heavy and light texels: a light texel's sum is held to the quantum of ITS bin's largest pixel gradient (2^-21 of it per
    # addend), i.e. here to an absolute error far below the heavy entries' 1e-5 -- and bins the heavy patch does not reach keep full precision
    per_map = ref.numel() // len(unpadded)
    for k in range(len(unpadded)):
        a, b = got[k * per_map:(k + 1) * per_map], ref[k * per_map:(k + 1) * per_map]
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12, k


def test_lds_aggregation_is_equivalent_on_magnified_env_pass():
    m, R, T, Km = _model(seed=17, ts=16)
    with torch.no_grad():
        scene = m.build_env(False, False)
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    grads = []
    for lds in (False, True):
        ps = _packed(scene)
        ps.maps.requires_grad_(True)
        ps.verts.requires_grad_(True)
        cfg = ops.RenderCfg(48, 64, 1, 0.0, 0.001, True, False, scene['faces'].shape[0], lds_aggregate=lds)
        img = ops.render_scene(ps.verts, ps.maps, None, ps.faces, *args, ps.face_uvs, ps.face_map, ps.map_desc, None, cfg)
        (img * torch.rand(img.shape, generator=torch.Generator().manual_seed(5)).to(DEV)).sum().backward()
        grads.append((ps.maps.grad, ps.verts.grad))
    assert rel_err(grads[1][0], grads[0][0]) < 1e-5 and rel_err(grads[1][1], grads[0][1]) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# size-independent properties at BASELINE config-2 resolution (300x400, K=10, 10 blocks)
# ---------------------------------------------------------------------------------------------------------------------
def test_full_size_properties():
    torch.manual_seed(0)
    m = O.OracleDBW((300, 400), n_blocks=10, txt_size=64, faces_per_pixel=10, seed=227391)
    R, T, Km = O.synthetic_cameras(6, R_world=m.R_world[0])
    with torch.no_grad():
        scene = m.build_blocks(False, True, False, None, kill_blocks=False)
    ps = _packed(scene)
    cfg = ops.RenderCfg(300, 400, 10, 1e-4, 0.001, True, True, scene['faces'].shape[0])
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    cl, p2f, zbuf, bary, dists = ops.render_fragments(ps.verts, ps.faces, *args, cfg)
    cl2, p2f2, zbuf2, _, _ = ops.render_fragments(ps.verts, ps.faces, *args, cfg)
    assert torch.equal(p2f, p2f2) and torch.equal(zbuf, zbuf2)                      # deterministic / idempotent
    valid = p2f >= 0
    assert 0.02 < valid[..., 0].float().mean() < 0.9                                # blocks cover part of the image
    z = torch.where(valid, zbuf, torch.full_like(zbuf, float('inf')))
    assert torch.all(z[..., 1:] >= z[..., :-1])                                     # front-to-back order
    assert torch.all(valid[..., 1:] <= valid[..., :-1])                             # no holes in the lists
    torch.testing.assert_close(bary[valid].sum(-1), torch.ones(int(valid.sum()), device=DEV), rtol=0, atol=2e-6)
    assert torch.all(dists[valid] < cfg.blur)
    srt = torch.where(valid, p2f, -torch.arange(1, 11, device=DEV, dtype=torch.int32).expand_as(p2f)).sort(-1)[0]
    assert torch.all(srt[..., 1:] != srt[..., :-1])                                 # a face appears once per pixel
    first = cl['first_idx'].long().view(-1, 1, 1, 1)
    assert torch.all((p2f[valid] >= 0)) and torch.all(((p2f - first) < cl['num_faces'].view(-1, 1, 1, 1))[valid])
    # image-level: alpha in [0,1]; zero opacity -> empty image; opacity scales linearly for single-layer pixels
    img = ops.render_scene(ps.verts, ps.maps, None, ps.faces, *args, ps.face_uvs, ps.face_map, ps.map_desc, None, cfg)
    assert img[:, 3].min() >= 0 and img[:, 3].max() <= 1 + 1e-6 and torch.isfinite(img).all()
    zero = torch.zeros(scene['faces'].shape[0], device=DEV)
    img0 = ops.render_scene(ps.verts, ps.maps, zero, ps.faces, *args, ps.face_uvs, ps.face_map, ps.map_desc, None, cfg)
    assert torch.all(img0 == 0)
    # a strided subset of the views against the oracle (one view, bit-exact indices at full resolution)
    ref = O.render(scene, R[:1], T[:1], Km[0], (300, 400), 1e-4, 10, True, None, 0.001, n_threads=8, return_fragments=True)[1]
    c2o = cl['c2o'].view(-1).long()
    orig = torch.where(p2f[:1] >= 0, c2o[p2f[:1].clamp(min=0).long()], torch.full_like(p2f[:1], -1).long())
    assert torch.equal(orig.cpu(), ref['pix_to_face'])


def test_full_size_gradient_paths_agree(monkeypatch):
    """BASELINE config-2 geometry at full size (300x400, K = 10, 256^2 textures, 6 views): the alternative implementations of one
    step must agree -- two-level binning vs full face scan (bit-identical image), texture-space bins vs atomic scatter of the
    texel gradients (same sums up to the order of the fp32/fp64 additions), and the texel-gradient total equals the total
    weight the blend assigns to the texture samples (bilinear weights sum to one: a conservation law of the scatter)."""
    m = O.OracleDBW((300, 400), n_blocks=10, txt_size=256, faces_per_pixel=10, seed=227391)
    R, T, Km = O.synthetic_cameras(6, R_world=m.R_world[0])
    with torch.no_grad():
        scene = m.build_blocks(False, True, False, None, kill_blocks=False)
    shapes = [tuple(t.shape[:2]) for t in scene['maps']]
    F_ = scene['faces'].shape[0]
    alpha = (torch.rand(10, generator=torch.Generator().manual_seed(3)) * 0.8 + 0.1).repeat_interleave(F_ // 10).to(DEV)
    g_img = torch.rand(6, 4, 300, 400, generator=torch.Generator().manual_seed(4)).to(DEV)
    g_img[:, 3] = 0                                    # colour gradients only: then sum(grad_maps[c]) = sum_pix g_c * sum_k T_k a_k
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    out = {}
    for tag, coarse, bins in (('default', True, True), ('full_scan', False, True), ('atomics', True, False)):
        monkeypatch.setattr(ops, 'COARSE_BINS', coarse)
        ps = _packed(scene)
        ps.maps.requires_grad_(True)
        ps.verts.requires_grad_(True)
        bb, bi, nb = PackedScene.describe_bins(shapes, DEV)
        cfg = ops.RenderCfg(300, 400, 10, 1e-4, 0.001, True, True, F_, lds_aggregate=False, texbins=(bb, bi, nb) if bins else None)
        img = ops.render_scene(ps.verts, ps.maps, alpha, ps.faces, *args, ps.face_uvs, ps.face_map, ps.map_desc, None, cfg)
        (img * g_img).sum().backward()
        out[tag] = (img.detach(), ps.maps.grad.clone(), ps.verts.grad.clone())
    assert torch.equal(out['default'][0], out['full_scan'][0])
    assert rel_err(out['default'][1], out['atomics'][1]) < 2e-5 and rel_err(out['default'][2], out['atomics'][2]) < 2e-5
    # conservation: the colour part of the image is sum_k T_k a_k c_k; with unit textures it equals sum_k T_k a_k = 1 - T_K = alpha
    # channel, so the texel gradients of channel c add up to sum_pix g_c * image_alpha
    expect = (g_img[:, :3] * out['default'][0][:, 3:4]).sum(dim=(0, 2, 3)).double()
    got = out['default'][1].view(-1, 3).double().sum(0)
    assert torch.allclose(got, expect, rtol=2e-4), (got, expect)


def test_rasterize_slivers_follow_the_double_reading_of_kepsilon():
    """area = EdgeFunction + kEpsilon with kEpsilon a DOUBLE (PyTorch3D; header of oracle/raster_ref.c): on slivers whose area is
    within two decades of 1e-8 the float and the double sum round differently; face_setup_kernel adds in double (DBW_AREA_EPS) and
    reproduces the oracle's canonical reading bit for bit -- and not the float reading of rounds 1-2."""
    g = torch.Generator().manual_seed(5)
    c = (torch.rand(4000, 1, 2, generator=g) * 2 - 1) * 0.9
    fv = torch.cat([c + (torch.rand(4000, 3, 2, generator=g) * 2 - 1) * 4e-4, torch.rand(4000, 3, 1, generator=g) + 1.0], -1).contiguous()
    first, num = torch.tensor([0]), torch.tensor([4000])
    ref = O.rasterize_fwd_raw(fv, first, num, None, (64, 64), 1e-3, 8, n_threads=8)
    old = O.rasterize_fwd_raw(fv, first, num, None, (64, 64), 1e-3, 8, n_threads=8, keps_float=True)
    assert (ref[2] != old[2]).sum() > 0
    out = ops.rasterize_meshes(fv.to(DEV), first.to(DEV), num.to(DEV), None, (64, 64), 1e-3, 8, 0, 0, True, True, False)
    assert torch.equal(out[0].cpu(), ref[0])
    for name, a, b in zip(['zbuf', 'bary', 'dists'], out[1:], ref[1:]):
        assert torch.equal(a.cpu(), b), name


@pytest.mark.parametrize('steps', [1, 2, 3, 4])
def test_lane_merge_preserves_the_per_key_sums(steps):
    """lane_merge (csrc/dbw_common.h): neighbouring lanes that update the same key hand their values over in registers before the LDS
    tables of the backward see them.  Whatever the key pattern, the sums per key over the ACTIVE lanes must not change (integer-valued
    floats: exact in any order), no lane may become active, inactive lanes must not contribute, and runs of equal keys must shrink:
    2^steps aligned neighbours of one key leave one active lane."""
    from dbw_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(steps)
    waves = 64
    n = waves * 64
    keys = torch.randint(0, 7, (n,), generator=g)
    run = 1 << steps
    keys[:16 * 64] = (torch.arange(16 * 64) // run) % 5                 # aligned runs of 2^steps equal keys
    keys[16 * 64:24 * 64] = 3                                           # one key for whole waves
    active = (torch.rand(n, generator=g) < 0.8).int()
    active[:24 * 64] = 1
    vals = torch.randint(-8, 9, (n, 3), generator=g).float()
    k_d, a_d, v_d = keys.int().to(DEV), active.to(DEV), vals.to(DEV)
    a_out, v_out = torch.zeros_like(a_d), torch.zeros_like(v_d)
    import device_checks
    device_checks.call('dbwt_lane_merge', k_d.data_ptr(), a_d.data_ptr(), v_d.data_ptr(), waves, steps, a_out.data_ptr(), v_out.data_ptr(), 0)
    torch.cuda.synchronize()
    a_out, v_out = a_out.cpu(), v_out.cpu()
    assert bool(((a_out == 1) <= (active == 1)).all())                  # nobody wakes up
    wave = torch.arange(n) // 64
    idx = wave * 8 + keys                                               # merging never crosses a wave
    before = torch.zeros(waves * 8, 3).index_add_(0, idx, vals * active[:, None].float())
    after = torch.zeros(waves * 8, 3).index_add_(0, idx, v_out * a_out[:, None].float())
    assert torch.equal(before, after)
    assert int(a_out[:16 * 64].sum()) == 16 * 64 // run                 # aligned runs: one lane left per run
    assert int(a_out[16 * 64:24 * 64].sum()) == 8 * 64 // min(run, 16)  # (a wave of one key: 2^steps lanes into one, 16 at most)
    assert int(a_out.sum()) < int(active.sum())


def test_bin_layout_by_demand_is_a_partition_and_gives_the_same_gradients(monkeypatch):
    """Record sub-ranges sized by the previous launch's demand (ops.BinDemand, dbw_bin_layout): the layout is a partition of the record
    array -- sub-ranges back to back, none empty, their sum within the total -- proportional to the demand, and a backward pass that
    uses it gives the gradients of the atomic scatter (a tiny total forces overflow into the exact fallback as well)."""
    from dbw_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    n = 16 * 37
    asked = torch.randint(0, 5000, (n,), generator=g).int()
    asked[::5] = 0
    total = 400000.0
    lay = torch.zeros(n, 2, dtype=torch.int32, device=DEV)
    assert lib.dbw_bin_layout(asked.to(DEV).data_ptr(), n, total, 64, lay.data_ptr(), 0) == 0, lib.dbw_last_error()
    first, caps = (lay.cpu().long() & 0xffffffff).unbind(1)
    assert int(first[0]) == 0 and bool((first[1:] == (first + caps)[:-1]).all()) and int(caps.min()) >= 1
    assert total - n <= float(first[-1] + caps[-1]) <= total
    want = asked.clamp(min=64).double() * 1.25      # one record each, the rest in proportion (so that the sum can never exceed the total)
    assert float(((caps.double() - 1 - want * ((total - n) / want.sum())).abs()).max()) <= 1.0
    # a demand so skewed that most shares round to nothing: still a partition inside the total (it used to overrun it)
    skew = torch.zeros(n, dtype=torch.int32)
    skew[0] = 2 ** 30
    assert lib.dbw_bin_layout(skew.to(DEV).data_ptr(), n, float(4 * n), 1, lay.data_ptr(), 0) == 0, lib.dbw_last_error()
    first, caps = (lay.cpu().long() & 0xffffffff).unbind(1)
    assert int(caps.min()) >= 1 and int(first[-1] + caps[-1]) <= 4 * n and bool((first[1:] == (first + caps)[:-1]).all())
    # the same gradients with and without it (two launches: the second one runs on the first one's demand)
    m, R, T, Km = _model(seed=29, ts=64, hw=(72, 96), fpp=8)
    with torch.no_grad():
        scene = m.build_blocks(False, True, False, None, kill_blocks=False)
    pl, pr = m.txt_padding
    unpadded = [mp[:, pl:mp.shape[1] - pr] for mp in scene['maps']]
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    ps = _packed(dict(scene, maps=unpadded), pads=[(pl, pr)] * len(unpadded))
    bins = PackedScene.describe_bins([(64, 64)] * len(unpadded), DEV)
    B, H, W, K = 3, 72, 96, 8
    cfg = ops.RenderCfg(H, W, K, 1e-4, 0.001, True, True, scene['faces'].shape[0], lds_aggregate=False, texbins=bins)
    cl = ops.project_clip(ps.verts, ps.faces, *args, cfg.eps, cfg.z_clip, cfg.persp)
    fvc = cl['face_verts'].view(-1, 3, 3)
    p2f, bary, dists, img = ops._render_fwd_fused(fvc, cl, B, cfg, ps.face_uvs, ps.face_map, ps.map_desc, ps.maps, None, None, 2)
    g_img = torch.rand(img.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    outs = []
    for capacity in (None, 4096, 32):
        if capacity is not None:
            monkeypatch.setattr(ops, 'texbin_capacity', lambda *a, _c=capacity: _c)
        for demand in ((None,) if capacity is None else (ops.BinDemand(),)):
            for launch in range(1 if demand is None else 3):
                gm, _, gv = ops._fused_bwd(p2f, bary, dists, cl, ps.face_uvs, ps.face_map, ps.map_desc, ps.maps, None, cfg, None, 2, g_img, B, None,
                                           bin_demand=demand)
                outs.append((gm.clone(), gv.clone()))
            if demand is not None:
                assert demand.ready and int(demand.cursors[1 - demand.turn].sum()) > 0
    for gm, gv in outs[1:]:
        assert rel_err(gm, outs[0][0]) < 1e-5 and rel_err(gv, outs[0][1]) < 1e-5
