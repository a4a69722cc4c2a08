"""CPU tests of the product's loss / optimiser arithmetic (csrc/loss_math.h, the header texture.hip compiles) built for the host with g++
(tests/host_loss_math.cpp): the TV term against the golden vectors of the REAL reference (tests/golden/implicit_misc.npz: tv_maps / tv /
tv_grad from dbw.py:378-387 + loss.py:46, with the u-wrap), the decoupled composite + MSE against autograd of dbw.py:223,366-367, and
the Adam update against torch.optim.Adam (optimizer.py:6-18)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        out = os.path.join(HERE, '_build')
        os.makedirs(out, exist_ok=True)
        so = os.path.join(out, 'libhost_loss_math.so')
        csrc = os.path.join(HERE, '..', 'differentiable-blocksworld_amd', 'csrc')
        srcs = [os.path.join(HERE, 'host_loss_math.cpp'), os.path.join(csrc, 'loss_math.h'), os.path.join(csrc, 'raster_math.h')]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', srcs[0], '-o', so])
        _LIB = ctypes.CDLL(so)
        _LIB.host_tv.restype = ctypes.c_double
        _LIB.host_composite_mse.restype = ctypes.c_double
    return _LIB


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def host_tv(maps, wrap):
    n, h, w, _ = maps.shape
    grad = torch.empty_like(maps)
    return lib().host_tv(_p(maps.contiguous()), n, h, w, int(wrap), _p(grad)), grad


def test_tv_matches_the_reference_golden_and_the_unwrapped_form(golden_dir):
    g = np.load(os.path.join(golden_dir, 'implicit_misc.npz'))
    maps = torch.from_numpy(g['tv_maps'])
    tv, grad = host_tv(maps, True)                                   # blocks' maps: the u direction wraps (dbw.py:383)
    assert abs(tv - float(g['tv'])) < 1e-5 * abs(float(g['tv']))
    torch.testing.assert_close(grad, torch.from_numpy(g['tv_grad']), rtol=1e-4, atol=1e-6)
    mm = torch.rand(1, 16, 20, 3, generator=torch.Generator().manual_seed(0))       # sky / ground: no wrap (dbw.py:380,386)
    ref = mm.clone().requires_grad_(True)
    tvr = sum(O.tv_l2sq(torch.diff(ref, dim=k)).mean() for k in [1, 2])
    tvr.backward()
    tv, grad = host_tv(mm, False)
    assert abs(tv - tvr.item()) < 1e-5 * tvr.item()
    torch.testing.assert_close(grad, ref.grad, rtol=1e-4, atol=1e-6)


def test_decoupled_composite_and_mse_match_autograd():
    gen = torch.Generator().manual_seed(5)
    P = 500
    fg, env, tgt = torch.rand(4, P, generator=gen), torch.rand(4, P, generator=gen), torch.rand(3, P, generator=gen)
    scale = 1.0 / (3 * P)
    rec, g_fg, g_env = torch.empty(3, P), torch.empty(4, P), torch.empty(3, P)
    loss = lib().host_composite_mse(_p(fg), _p(env), _p(tgt), P, ctypes.c_float(scale), _p(rec), _p(g_fg), _p(g_env))
    f, e = fg.clone().requires_grad_(True), env.clone().requires_grad_(True)
    r = f[:3] * f[3:] + (1 - f[3:]) * e[:3]                       # dbw.py:223 (premultiplied colour times the mask again)
    ref = ((r - tgt) ** 2).mean()                                  # nn.MSELoss, dbw.py:366-367
    ref.backward()
    torch.testing.assert_close(rec, r.detach(), rtol=1e-6, atol=1e-7)
    assert abs(loss - ref.item()) < 1e-5 * ref.item()
    torch.testing.assert_close(g_fg, f.grad, rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(g_env, e.grad[:3], rtol=1e-4, atol=1e-8)


def test_adam_update_matches_torch_adam():
    torch.manual_seed(0)
    p0 = torch.randn(1000)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=5e-3)
    ph, m, v = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    for step in range(1, 6):
        g = torch.randn(1000)
        p.grad = g.clone()
        opt.step()
        lib().host_adam(_p(ph), _p(g), _p(m), _p(v), 1000, ctypes.c_float(5e-3), ctypes.c_float(0.9), ctypes.c_float(0.999), ctypes.c_float(1e-8), step)
    torch.testing.assert_close(ph, p.detach(), rtol=1e-5, atol=1e-6)
