"""CPU tests of the product's rasteriser arithmetic (csrc/raster_math.h, the header the HIP kernels compile) built for the host
with g++ -ffp-contract=off (tests/host_raster_math.cpp) against the oracle: per-face records, shared-reciprocal divisions with
their guards, conservative tile culling and the key/payload top-K list must reproduce oracle/raster_ref.c bit for bit.
The GPU parity tests (tests/test_gpu_parity.py) hold the kernels themselves to the same bar."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        out = os.path.join(HERE, '_build')
        os.makedirs(out, exist_ok=True)
        so = os.path.join(out, 'libhost_raster_math.so')
        srcs = [os.path.join(HERE, 'host_raster_math.cpp'),
                os.path.join(HERE, '..', 'differentiable-blocksworld_amd', 'csrc', 'raster_math.h')]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', srcs[0], '-o', so])
        _LIB = ctypes.CDLL(so)
        _LIB.host_divcheck.restype = ctypes.c_longlong
    return _LIB


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def host_rasterize(fv, first, num, nbr, size, blur, K, persp=True, clipb=True, cull=False, fastdiv=1, tile=8, perturb=0, exact_k=0, bounded=1):
    H, W = size
    N = first.numel()
    fv = fv.contiguous()
    first, num = first.to(torch.int64).contiguous(), num.to(torch.int64).contiguous()
    nbr = None if nbr is None else nbr.to(torch.int64).contiguous()
    p2f = torch.empty(N, H, W, K, dtype=torch.int64)
    zbuf, dists = torch.empty(N, H, W, K), torch.empty(N, H, W, K)
    bary = torch.empty(N, H, W, K, 3)
    stats = torch.zeros(4, dtype=torch.int64)
    rc = lib().host_rasterize(_p(fv), _p(first), _p(num), _p(nbr), N, H, W, K, ctypes.c_float(blur), int(persp), int(clipb), int(cull),
                              int(fastdiv), int(tile), int(perturb), int(exact_k), int(bounded), _p(p2f), _p(zbuf), _p(bary), _p(dists), _p(stats))
    assert rc == 0
    return (p2f, zbuf, bary, dists), dict(zip(('evals', 'unsafe', 'culled', 'staged'), stats.tolist()))


def random_faces(n_faces, seed, zmin=0.5, zmax=5.0, spread=1.2, size=0.5):
    g = torch.Generator().manual_seed(seed)
    c = (torch.rand(n_faces, 1, 2, generator=g) * 2 - 1) * spread
    xy = c + (torch.rand(n_faces, 3, 2, generator=g) * 2 - 1) * size
    z = torch.rand(n_faces, 3, 1, generator=g) * (zmax - zmin) + zmin
    return torch.cat([xy, z], -1).contiguous()


def assert_same(out, ref):
    assert torch.equal(out[0], ref[0]), f'pix_to_face differs on {(out[0] != ref[0]).sum().item()} slots'
    for name, a, b in zip(('zbuf', 'bary', 'dists'), out[1:], ref[1:]):
        assert torch.equal(a, b), f'{name}: max abs diff {(a - b).abs().max().item()}'


@pytest.mark.parametrize('H,W,K,nf,blur,persp,clipb,exact_k', [
    (33, 47, 4, 60, 1e-3, True, True, 1),
    (33, 47, 4, 60, 1e-3, True, True, 0),                       # K < KMAX: the generic list instantiation
    (40, 32, 10, 200, math.log(1e4 - 1) * 1e-4, True, True, 1),  # the coarse renderer's setting
    (24, 24, 1, 100, 0.0, True, True, 1),                       # hard pass: sign pre-reject
    (20, 36, 16, 20, 5e-3, False, False, 0),                    # no perspective correction / clipping: negative barycentrics
    (24, 40, 25, 300, 2e-4, True, True, 0),
])
@pytest.mark.parametrize('fastdiv,tile,perturb', [(0, 0, 0), (1, 0, 0), (1, 8, 0), (1, 16, 1), (1, 8, -1)])
def test_host_build_of_the_kernel_arithmetic_is_bit_exact_to_the_oracle(H, W, K, nf, blur, persp, clipb, exact_k, fastdiv, tile, perturb):
    fv = random_faces(nf, seed=nf + K)
    fv = torch.cat([fv, fv * torch.tensor([0.9, -1.1, 1.0])], 0)
    first, num = torch.tensor([0, nf]), torch.tensor([nf, nf])
    ref = O.rasterize_fwd_raw(fv, first, num, None, (H, W), blur, K, persp, clipb, n_threads=4)
    out, st = host_rasterize(fv, first, num, None, (H, W), blur, K, persp, clipb, fastdiv=fastdiv, tile=tile, perturb=perturb, exact_k=exact_k)
    assert_same(out, ref)
    assert st['evals'] > 0
    if fastdiv:
        assert st['unsafe'] < 0.02 * st['evals']        # the guarded fast path is the common case, not the exception
    if tile and nf >= 60:
        assert st['culled'] > 0                         # the tile test does cull -- without changing a bit


def test_sibling_rule_ties_degenerates_and_tiny_faces():
    """Split-quad siblings (neighbour links), coincident faces (ties broken by face id), zero-area and behind-camera faces, faces
    with a degenerate edge, faces far outside the REC_FAST coordinate range (slow path) -- all against the oracle."""
    base = random_faces(12, seed=5)
    fv = torch.cat([base, base, base[:2] * torch.tensor([1., 1., 0.]) + torch.tensor([0., 0., -1.])], 0)
    fv[3, 2] = fv[3, 1]                                            # zero-area face
    fv[5, 1, :2] = fv[5, 0, :2] + 1e-5                             # degenerate edge (l2 <= eps), non-zero area? (thin sliver)
    fv[7] = fv[7] * torch.tensor([3000., 3000., 1.])               # coordinates beyond 1024: IEEE path
    nf = fv.shape[0]
    nbr = torch.full((nf,), -1, dtype=torch.int64)
    for a, b in ((0, 12), (1, 13), (4, 6), (8, 9)):                # overlapping pairs declared siblings
        nbr[a], nbr[b] = b, a
    first, num = torch.tensor([0]), torch.tensor([nf])
    for K in (2, 6, 10):
        for blur in (1e-3, 2e-2):
            ref = O.rasterize_fwd_raw(fv, first, num, nbr, (37, 29), blur, K)
            for exact_k in (0, 1):
                for bounded in (0, 1):
                    out, st = host_rasterize(fv, first, num, nbr, (37, 29), blur, K, exact_k=exact_k, bounded=bounded)
                    assert_same(out, ref)
    p = ref[0]
    both = (p[..., 0] >= 0) & (p[..., 1] >= 0)
    assert both.any()


def test_ordered_insert_on_meshes_with_shared_vertices_equal_depths_and_overfull_lists():
    """TopK::insert_ordered (32-bit depth compare + median shift, candidates in ascending face id) against the oracle and against the
    64-bit rank-and-shift insert where it matters: closed meshes whose fans clamp to a shared vertex (several faces with EXACTLY the
    same depth at a pixel, ordered by face id), duplicated meshes (every depth twice), and far more candidates than K."""
    from dbw_amd import mesh as M
    verts, faces = M.ico_sphere(1)
    g = torch.Generator().manual_seed(11)
    fvs = []
    for i in range(6):
        c = torch.tensor([(i % 3 - 1) * 0.45, (i // 3 - 0.5) * 0.5, 2.5 + 0.3 * i])
        v = verts * (0.35 + 0.05 * i) + c
        ndc = torch.stack([v[:, 0] / v[:, 2] * 2.0, v[:, 1] / v[:, 2] * 2.0, v[:, 2]], -1)
        fvs.append(ndc[faces])
    fv = torch.cat(fvs + [fvs[0], fvs[3]], 0).contiguous()            # two meshes twice: coincident faces, ties on every slot
    nf = fv.shape[0]
    first, num = torch.tensor([0]), torch.tensor([nf])
    blur = math.log(1e4 - 1) * 1e-4
    for K, exact_k in ((10, 1), (4, 1), (6, 0), (25, 0)):
        ref = O.rasterize_fwd_raw(fv, first, num, None, (48, 64), blur, K, n_threads=4)
        zb = ref[1]
        ties = ((zb[..., 1:] == zb[..., :-1]) & (ref[0][..., 1:] >= 0)).sum().item()
        assert ties > 50, ties                                          # the scene does produce equal depths in one list
        for bounded in (0, 1):
            out, _ = host_rasterize(fv, first, num, None, (48, 64), blur, K, exact_k=exact_k, bounded=bounded)
            assert_same(out, ref)


def test_slivers_whose_area_is_close_to_kepsilon_follow_the_double_reading():
    """area = EdgeFunction + kEpsilon with kEpsilon a double (PyTorch3D; oracle/raster_ref.c header): on faces whose area is within two
    decades of 1e-8 the float and the double sum round differently -- the product's record (make_face_rec: DBW_AREA_EPS) must follow the
    oracle's canonical (double) reading bit for bit, and must NOT equal the float reading everywhere."""
    g = torch.Generator().manual_seed(5)
    c = (torch.rand(4000, 1, 2, generator=g) * 2 - 1) * 0.9
    fv = torch.cat([c + (torch.rand(4000, 3, 2, generator=g) * 2 - 1) * 4e-4, torch.rand(4000, 3, 1, generator=g) + 1.0], -1).contiguous()
    first, num = torch.tensor([0]), torch.tensor([4000])
    ref = O.rasterize_fwd_raw(fv, first, num, None, (64, 64), 1e-3, 8, n_threads=4)
    old = O.rasterize_fwd_raw(fv, first, num, None, (64, 64), 1e-3, 8, n_threads=4, keps_float=True)
    assert (ref[2] != old[2]).sum() > 0
    for fastdiv in (0, 1):
        out, _ = host_rasterize(fv, first, num, None, (64, 64), 1e-3, 8, fastdiv=fastdiv, tile=8)
        assert_same(out, ref)


def test_large_faces_with_near_plane_coordinates():
    """Env-pass-like geometry: few huge faces, some with NDC coordinates in the thousands (clipped at z = 1e-3)."""
    g = torch.Generator().manual_seed(3)
    fv = random_faces(40, seed=9, zmin=1e-3, zmax=30.0, spread=3.0, size=6.0)
    fv[:8, :, :2] *= 400.0
    first, num = torch.tensor([0]), torch.tensor([40])
    for blur, K in ((0.0, 1), (1e-4, 3)):
        ref = O.rasterize_fwd_raw(fv, first, num, None, (30, 40), blur, K)
        out, st = host_rasterize(fv, first, num, None, (30, 40), blur, K, tile=16, exact_k=1)
        assert_same(out, ref)


def test_pix_to_ndc_with_shared_reciprocal_is_bit_exact():
    lib().host_ndccheck.restype = ctypes.c_longlong
    for S1, S2 in ((300, 400), (400, 300), (75, 100), (100, 75), (576, 768), (768, 576), (1080, 1920), (1920, 1080), (33, 47), (1, 7),
                   (4096, 4096), (8191, 17)):
        for perturb in (0, 1, -1):
            assert lib().host_ndccheck(S1, S2, perturb) == 0, (S1, S2, perturb)


def test_div_fast_equals_ieee_division_inside_the_guarded_range():
    """div_fast == `/` bit for bit for operands inside the guards (|n| in {0} U [2^-60, 2^69], d in [2^-27, 2^40]) whatever the last
    bit of the reciprocal seed; the device-side twin of this test runs the real v_rcp_f32 (tests/test_gpu_parity.py)."""
    rng = np.random.default_rng(0)
    n_ = 400_000
    sign = rng.choice([-1.0, 1.0], n_)
    num = (sign * np.exp2(rng.uniform(-60, 69, n_)) * rng.uniform(1, 2, n_)).astype(np.float32)
    num[:1000] = 0.0
    num[1000:2000] = np.float32(2.0 ** -60)
    den = (rng.choice([-1.0, 1.0], n_) * np.exp2(rng.uniform(-27, 40, n_)) * rng.uniform(1, 2, n_)).astype(np.float32)
    # hard cases: quotients next to a rounding boundary (n = d * q with q a float, +- 1 ulp of n)
    q = rng.uniform(0.5, 2.0, 100_000).astype(np.float32)
    d2 = rng.uniform(0.5, 8.0, 100_000).astype(np.float32)
    n2 = (d2.astype(np.float64) * q.astype(np.float64)).astype(np.float32)
    n2 = np.nextafter(n2, np.float32(np.inf) * rng.choice([-1, 1], 100_000).astype(np.float32))
    num, den = np.concatenate([num, n2]), np.concatenate([den, d2])
    tn, td = torch.from_numpy(num), torch.from_numpy(den)
    for perturb in (0, 1, -1):
        assert lib().host_divcheck(_p(tn), _p(td), ctypes.c_longlong(tn.numel()), perturb) == 0


@pytest.mark.parametrize('name', ['front', 'straddle', 'ties'])
def test_host_build_reproduces_the_frozen_tiny_scenes(golden_dir, name):
    """The product's rasteriser arithmetic (host build of csrc/raster_math.h) on the committed tiny scenes of tests/golden/raster_tiny.npz."""
    from test_oracle_golden import _load, _tiny_scene
    g = _load(golden_dir, 'raster_tiny.npz')
    fv, first, num, nbr, size, blur, K = _tiny_scene(g, name)
    for fastdiv in (0, 1):
        (p2f, zbuf, bary, dists), _ = host_rasterize(fv, first, num, nbr, size, blur, K, fastdiv=fastdiv)
        for got, key in ((p2f, 'p2f'), (zbuf, 'zbuf'), (bary, 'bary'), (dists, 'dists')):
            assert torch.equal(got, g[f'{name}/{key}']), (key, fastdiv)


def _work_positions(length, occupied):
    pos = np.empty(length, dtype=np.int32)
    lib().host_work_positions(int(length), int(occupied), ctypes.c_void_p(pos.ctypes.data))
    return pos


def test_launch_order_is_a_permutation_of_every_segment():
    """work_position (raster_math.h, what work_scatter_kernel evaluates per tile): the occupied and the empty tiles of a segment take every
    position exactly once -- whole windows of 64, a partial last window, segments shorter than a window, all or none occupied."""
    rng = np.random.default_rng(3)
    cases = [(1, 0), (1, 1), (5, 2), (63, 20), (64, 16), (65, 1), (200, 199), (11638, 2909), (11638, 2902), (11634, 2693), (101250, 30000)]
    cases += [(int(n), int(rng.integers(0, n + 1))) for n in rng.integers(1, 5000, size=40)]
    for length, occ in cases:
        pos = _work_positions(length, occ)
        assert np.array_equal(np.sort(pos), np.arange(length)), (length, occ)


def test_launch_order_keeps_heavy_tiles_first_and_has_no_comb():
    """Heaviest first at the scale of the scramble's windows; and the reason for the scramble: with one tile in 2, 3, 4, 8 or 16 occupied --
    exactly, or to within one tile -- the occupied tiles' positions fall evenly into the residues modulo 2 ... 128 (an even comb put all
    of them into ONE residue modulo 4 at a time: every occupied tile on the same shader engine of its XCD, profiles/r06_experiments.md)."""
    for length in (11638, 11634, 32400, 4096):
        for ratio in (2, 3, 4, 8, 16):
            for occ in (length // ratio - 1, length // ratio, length // ratio + 1):
                pos = _work_positions(length, occ)[:occ]
                assert np.all(np.diff(pos // 64) >= 0), 'the windows follow the ranks'
                for m in (2, 4, 8, 16, 32, 128):
                    share = np.bincount(pos % m, minlength=m) / occ
                    slack = max(0.5 / m, 4.0 * math.sqrt(1.0 / (m * occ)))       # (half the even share, or four sigma of a random draw)
                    assert np.abs(share - 1.0 / m).max() <= slack, (length, occ, m, share)
