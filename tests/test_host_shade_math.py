"""CPU tests of the product's shading arithmetic (csrc/shade_math.h, the header the HIP kernels compile) built for the host with g++
(tests/host_shade_math.cpp):
  * the bilinear texture footprint (v flip, circular u padding by index wrap, decimated maps kept at cell resolution, border clamping)
    and its two backward branches against torch's grid_sample on the texture the reference would have materialised
    (dbw.py:276-278,331-341; TexturesUV.sample_textures);
  * the layered blend, forward and the division-free backward recurrences, against golden vectors of the REAL reference function
    `layered_rgb_blend` (tests/golden/blend.npz, renderer.py:241-273);
  * the barycentric back-conversion of clipped faces against the oracle's 3x3 conversion matrices.
The GPU tests hold the kernels that inline these functions to the same references."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        out = os.path.join(HERE, '_build')
        os.makedirs(out, exist_ok=True)
        so = os.path.join(out, 'libhost_shade_math.so')
        csrc = os.path.join(HERE, '..', 'differentiable-blocksworld_amd', 'csrc')
        srcs = [os.path.join(HERE, 'host_shade_math.cpp'), os.path.join(csrc, 'shade_math.h'), os.path.join(csrc, 'raster_math.h')]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', srcs[0], '-o', so])
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def host_sample(stored, h, w, pl, pr, sh, uv, gc):
    desc = torch.tensor([0, h, w, pl, pr, sh, 0, 0], dtype=torch.int32)
    maps = stored.contiguous().reshape(-1)
    n = uv.shape[0]
    rgb, guv, gmaps = torch.empty(n, 3), torch.empty(n, 2), torch.zeros_like(maps)
    assert lib().host_sample(_p(maps), _p(desc), n, _p(uv.contiguous()), _p(gc.contiguous()), _p(rgb), _p(guv), _p(gmaps)) == 0
    return rgb, guv, gmaps.view(stored.shape)


@pytest.mark.parametrize('h,w,pl,pr,sh', [(16, 16, 0, 0, 0), (16, 24, 3, 5, 0), (32, 32, 0, 0, 2), (32, 64, 6, 12, 3), (8, 8, 8, 8, 0)])
def test_footprint_matches_grid_sample_on_the_materialised_texture(h, w, pl, pr, sh):
    g = torch.Generator().manual_seed(h * 131 + w * 7 + pl + sh)
    stored = torch.rand(h >> sh, w >> sh, 3, generator=g).requires_grad_(True)
    n = 4000
    uv = torch.rand(n, 2, generator=g) * 1.3 - 0.15                       # beyond [0, 1]: border clamping
    uv[:50] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [0.5, 0.0], [0.0, 1.0], [1.0, 0.25]]).repeat(10, 1)   # exactly on the border
    uv_ref = uv.clone().requires_grad_(True)
    gc = torch.randn(n, 3, generator=g)
    # what the reference materialises: nearest upsampling of the decimated map, circular padding along u, v flipped
    full = stored.repeat_interleave(1 << sh, 0).repeat_interleave(1 << sh, 1)
    if pl or pr:
        full = F.pad(full.permute(2, 0, 1)[None], (pl, pr, 0, 0), mode='circular')[0].permute(1, 2, 0)
    t = full.permute(2, 0, 1)[None].flip(2)
    ref = F.grid_sample(t, (uv_ref * 2 - 1)[None, :, None, :], mode='bilinear', align_corners=True, padding_mode='border')[0, :, :, 0].t()
    (ref * gc).sum().backward()
    rgb, guv, gmaps = host_sample(stored.detach(), h, w, pl, pr, sh, uv, gc)
    torch.testing.assert_close(rgb, ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gmaps, stored.grad, rtol=1e-4, atol=1e-5)
    # the gradient to the coordinates: identical where the bilinear cell is unambiguous (away from texel boundaries, where the one-sided
    # derivatives of the two implementations may pick different cells)
    wp = w + pl + pr
    fx, fy = uv[:, 0] * (wp - 1), uv[:, 1] * (h - 1)
    inner = ((fx - fx.round()).abs() > 1e-3) & ((fy - fy.round()).abs() > 1e-3)
    torch.testing.assert_close(guv[inner], uv_ref.grad[inner], rtol=1e-3, atol=1e-4)
    assert int(inner.sum()) > n // 2


@pytest.mark.parametrize('tag', ['s1e-4_a', 's1e-4', 's5e-6_a', 's0', 's0_a', 'sig1e-4_a', 'sig1e-4', 'sig5e-6_a'])
def test_blend_recurrences_match_the_reference_golden(golden_dir, tag):
    sigmoid = tag.startswith('sig')                  # the real layered_rgb_blend with clip_inside=False: the library's sigma < 0
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, 'blend_sigmoid.npz' if sigmoid else 'blend.npz')).items()}
    p2f, dists, colors = g[f'{tag}/p2f'].contiguous(), g[f'{tag}/dists'].contiguous(), g[f'{tag}/colors'].contiguous()
    fa = g[f'{tag}/faces_alpha'].contiguous() if f'{tag}/faces_alpha' in g else None
    sigma, bg, w = float(g[f'{tag}/sigma']), g[f'{tag}/bg'].contiguous(), g[f'{tag}/w'].contiguous()
    N, H, W, K = p2f.shape
    out, g_colors, g_dists = torch.empty(N, 4, H, W), torch.empty_like(colors), torch.empty_like(dists)
    g_fa = None if fa is None else torch.zeros_like(fa)
    assert lib().host_blend(_p(p2f), _p(dists), _p(colors), _p(fa), N, H, W, K, ctypes.c_float(-sigma if sigmoid else sigma), _p(bg), _p(w), _p(out), _p(g_colors),
                            _p(g_dists), _p(g_fa)) == 0
    torch.testing.assert_close(out, g[f'{tag}/out'], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(g_colors, g[f'{tag}/g_colors'], rtol=1e-5, atol=1e-6)
    if sigma > 0:
        # d/d dist = -a / sigma * d/da: 1e4 .. 2e5 times the other gradients, compare relative to its scale
        ref = g[f'{tag}/g_dists']
        assert float((g_dists - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    if fa is not None:
        torch.testing.assert_close(g_fa, g[f'{tag}/g_faces_alpha'], rtol=1e-4, atol=1e-5)


def test_barycentric_back_conversion_of_clipped_faces():
    """convert_bary = the 3x3 barycentric_conversion of SURVEY.md A.4 (columns = original-triangle barycentrics of the clipped
    triangle's vertices) for the three clipped-triangle kinds and every rotation of the vertex roles; its backward = the transpose."""
    import oracle as O
    rng = np.random.RandomState(0)
    for kind in range(3):
        for i1 in range(3):
            w2, w3 = float(rng.rand()), float(rng.rand())
            cd = i1 | (kind << 2)
            M = np.zeros((3, 3), dtype=np.float64)            # bo = M @ b
            for col in range(3):
                b = torch.zeros(3); b[col] = 1.0
                bo, gb = torch.empty(3), torch.empty(3)
                lib().host_convert_bary(cd, ctypes.c_float(w2), ctypes.c_float(w3), _p(b), _p(bo), _p(torch.zeros(3)), _p(gb))
                M[:, col] = bo.numpy()
            i2, i3 = (i1 + 1) % 3, (i1 + 2) % 3
            e = np.eye(3)
            p4 = (1 - w2) * e[i1] + w2 * e[i2]
            p5 = (1 - w3) * e[i1] + w3 * e[i3]
            cols = {0: (p4, p5, e[i1]), 1: (p4, e[i2], p5), 2: (p5, e[i2], e[i3])}[kind]
            np.testing.assert_allclose(M, np.stack(cols, 1), atol=1e-6)
            go = torch.from_numpy(rng.randn(3).astype(np.float32))
            bo, gb = torch.empty(3), torch.empty(3)
            lib().host_convert_bary(cd, ctypes.c_float(w2), ctypes.c_float(w3), _p(torch.ones(3) / 3), _p(bo), _p(go), _p(gb))
            np.testing.assert_allclose(gb.numpy(), M.T @ go.numpy().astype(np.float64), rtol=1e-5, atol=1e-6)
