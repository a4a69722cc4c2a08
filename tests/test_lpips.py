"""The perceptual criterion (src/model/loss.py:32-40 -> lpips==0.1.4, absent here, weights absent too).
oracle/lpips_ref.py restates the published forward independently; tests/golden/lpips_random.npz freezes it on seeded random weights
(float64).  The product module a user loads real weights into (dbw_amd/lpips_vgg.py: LPIPSVGG) must reproduce the fixture with the same
weights -- forward and gradient -- on the CPU here and on the GPU (MIOpen convolutions) under -m gpu, and the model's perceptual term
must be weight x phase factor x batch share x that value.  REAL-WEIGHT PARITY STAYS UNPINNED: what is pinned is the architecture."""
import os

import numpy as np
import pytest
import torch

import lpips_ref as L                                   # oracle/ (checker only)


def _fixture(golden_dir):
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, 'lpips_random.npz')).items()}
    vgg, lin = L.random_weights(int(g['seed']))
    return g, vgg, lin


def test_restated_lpips_forward_is_frozen_by_the_fixture(golden_dir):
    g, vgg, lin = _fixture(golden_dir)
    vgg64, lin64 = L.random_weights(int(g['seed']), torch.float64)
    r = g['rec'].double().requires_grad_(True)
    per = L.lpips_vgg(g['imgs'].double(), r, vgg64, lin64)
    per.mean().backward()
    assert torch.allclose(per.view(-1), g['per_sample'], rtol=1e-12, atol=0) and abs(float(per.mean()) - float(g['loss'])) < 1e-15
    assert torch.allclose(r.grad, g['g_rec'], rtol=1e-9, atol=1e-16)
    assert float(g['per_sample'][2]) == 0.0 and float(g['per_sample'][:2].min()) > 1e-3           # identical pair / different pairs
    # the same function in float32 (what runs in practice) stays within 1e-5 of it
    per32 = L.lpips_vgg(g['imgs'], g['rec'], vgg, lin).view(-1).double()
    assert float((per32 - g['per_sample']).abs().max()) < 1e-5 * float(g['per_sample'].max())


def _check_module(net, g, dev, rel):
    imgs, rec = g['imgs'].to(dev), g['rec'].to(dev).requires_grad_(True)
    per = torch.stack([net(imgs[i:i + 1], rec[i:i + 1]) for i in range(imgs.shape[0])])
    ref = g['per_sample'].to(dev)
    assert float((per.double() - ref).abs().max()) < rel * float(ref.max()), (per.tolist(), ref.tolist())
    loss = net(imgs, rec)                                                  # the reference's .mean() over the batch
    assert abs(float(loss) - float(g['loss'])) < rel * float(g['loss'])
    loss.backward()
    gr, gref = rec.grad.double().cpu(), g['g_rec']
    assert float((gr - gref).abs().max()) < rel * float(gref.abs().max())
    assert float(gr[2].abs().max()) <= 1e-3 * float(gref.abs().max())     # (the identical pair: zero gradient up to 0/0-guard noise)


def test_product_lpips_module_reproduces_the_fixture_with_the_same_weights(golden_dir):
    from dbw_amd.lpips_vgg import LPIPSVGG
    g, vgg, lin = _fixture(golden_dir)
    net = LPIPSVGG().load_weights(vgg, lin)
    _check_module(net, g, 'cpu', 1e-4)


@pytest.mark.gpu
def test_product_lpips_module_on_the_gpu_reproduces_the_fixture(golden_dir):
    from dbw_amd.lpips_vgg import LPIPSVGG
    g, vgg, lin = _fixture(golden_dir)
    net = LPIPSVGG().load_weights(vgg, lin).to('cuda')
    _check_module(net, g, 'cuda', 1e-4)


@pytest.mark.gpu
def test_model_perceptual_term_is_weight_times_phase_factor_times_batch_share_times_lpips(golden_dir):
    """dbw.py:370: perceptual_weight x (1 coarse | 0.1 fine) x LPIPS(imgs, rec) with rec the composite the HIP path renders; under
    view-sharded data parallelism a rank's term carries its share of the global batch (the gradients of the ranks are summed).  The
    value is checked against the ORACLE's restatement evaluated on the very images the model handed to the criterion."""
    import dbw_amd
    import oracle as O
    from dbw_amd.lpips_vgg import LPIPSVGG
    from test_gpu_model import _dtu_like_cfg
    DEV = 'cuda'
    g, vgg, lin = _fixture(golden_dir)
    H, W = 48, 64
    cfg = _dtu_like_cfg(4, 32, 6)
    cfg['model']['loss']['perceptual_weight'] = 0.1
    R, T, Km = O.synthetic_cameras(2, R_world=O.world_rotation(115, 0, 0))
    inp = {k: v.to(DEV) for k, v in dict(imgs=torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(2)), R=R, T=T, K=Km).items()}
    torch.manual_seed(227391)
    model = dbw_amd.create_model(cfg, (H, W)).to(DEV).train()
    net = LPIPSVGG().load_weights(vgg, lin).to(DEV)
    seen = {}

    def criterion(imgs, rec):
        seen['imgs'], seen['rec'] = imgs.detach().cpu(), rec.detach().cpu()
        return net(imgs, rec)
    model.set_perceptual(criterion)
    for epoch, factor in ((0, 1.0), (1600, 0.1)):
        model.set_cur_epoch(epoch)
        for world, count, share in ((1, None, 1.0), (4, 4 * inp['imgs'].numel(), 0.25)):
            model.world_size, model._global_count = world, count
            out = model(inp, None)
            ref = float(L.lpips_loss(seen['imgs'], seen['rec'], vgg, lin))
            got = float(out['perceptual'])
            assert abs(got - 0.1 * factor * share * ref) <= 1e-4 * 0.1 * factor * share * ref, (epoch, world, got, ref)
    model.world_size, model._global_count = 1, None


def test_cached_targets_do_not_change_value_or_gradient(golden_dir):
    """LPIPSVGG.cache_targets: the normalised features of the (fixed) training images, computed once, gathered by view id -- the value and
    the gradient to `rec` are those of the plain call, whatever the order and repetition of the ids; target_bytes is what the cache holds."""
    from dbw_amd.lpips_vgg import LPIPSVGG
    g, vgg, lin = _fixture(golden_dir)
    net = LPIPSVGG().load_weights(vgg, lin)
    imgs_all = g['imgs']
    V, _, H, W = imgs_all.shape
    ids = torch.tensor([2, 0, 0, 1][:max(V, 1) + 1]) % V
    rec = g['rec'][ids].clone().requires_grad_(True)
    ref = net(imgs_all[ids], rec)
    g_ref, = torch.autograd.grad(ref, rec)
    net.cache_targets(imgs_all, chunk=2)
    assert sum(c.numel() * 4 for c in net.target_cache) == net.target_bytes(H, W, V)
    got = net(torch.full_like(imgs_all[ids], float('nan')), rec, view_ids=ids)          # (the images are not read any more)
    g_got, = torch.autograd.grad(got, rec)
    assert abs(float(got) - float(ref)) <= 1e-6 * float(ref) and float((g_got - g_ref).abs().max()) <= 1e-6 * float(g_ref.abs().max())
    with pytest.raises(ValueError):
        net(imgs_all[ids], rec, view_ids=ids[:1])
    net.cache_targets(None)
    assert net.target_cache is None


@pytest.mark.gpu
def test_trainer_with_cached_perceptual_targets_takes_the_same_steps(golden_dir):
    """Trainer(cache_perceptual_targets=True), the default: the criterion's target features are computed once for the rank's views and every
    batch carries its view ids through the C step's two phases (c_step.py) -- losses and parameters after two epochs equal those of a
    trainer that recomputes the targets' features every step, as the reference does."""
    import dbw_amd
    import oracle as O
    from dbw_amd.lpips_vgg import LPIPSVGG
    from dbw_amd.trainer import Trainer
    from test_gpu_model import _dtu_like_cfg
    DEV = 'cuda'
    g, vgg, lin = _fixture(golden_dir)
    H, W, V = 48, 64, 6
    cfg = _dtu_like_cfg(4, 32, 6)
    cfg['model']['loss']['perceptual_weight'] = 0.1
    cfg['training'] = {'batch_size': 4, 'n_epoches': 2, 'seed': 11, 'optimizer': {'name': 'adam', 'lr': 5.0e-3, 'texture': {'lr': 5.0e-2}},
                       'scheduler': {'name': 'multi_step', 'gamma': [0.1], 'milestones': [100]}}
    R, T, Km = O.synthetic_cameras(V, R_world=O.world_rotation(115, 0, 0))
    views = {k: v.to(DEV) for k, v in dict(imgs=torch.rand(V, 3, H, W, generator=torch.Generator().manual_seed(2)), R=R, T=T, K=Km).items()}
    results = []
    for cached in (False, True):
        torch.manual_seed(227391)
        model = dbw_amd.create_model(cfg, (H, W)).to(DEV)
        net = LPIPSVGG().load_weights(vgg, lin).to(DEV)
        calls = []
        orig = net.forward
        net.forward = lambda imgs, rec, view_ids=None, _o=orig: (calls.append(view_ids is not None), _o(imgs, rec, view_ids=view_ids))[1]
        model.set_perceptual(net)
        tr = Trainer(cfg, model, views, cache_perceptual_targets=cached)
        assert tr.view_ids == cached and (net.target_cache is not None) == cached
        assert tr.step_fn.cstep is not None
        vals = []
        for _ in range(2):
            last = tr.run_epoch()
            vals.append({k: float(v) for k, v in last.items()})
        assert calls and all(c == cached for c in calls)          # (the ragged second batch of an epoch too)
        results.append((vals, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (va, pa), (vb, pb) = results
    for k in va[0]:          # (the first epoch's last step is the second Adam step: compared tightly; the later ones through the parameters)
        assert abs(va[0][k] - vb[0][k]) <= 1e-4 * max(abs(va[0][k]), 1e-6), (k, va[0][k], vb[0][k])
    from trajectory import assert_same_trajectory          # (four Adam steps apart: two samples of a slightly chaotic system, tests/trajectory.py)
    flat = [torch.cat([p[k].reshape(-1).float() for k in sorted(p) if p[k].is_floating_point()]) for p in (pa, pb)]
    assert_same_trajectory(flat[0], flat[1])


@pytest.mark.gpu
@pytest.mark.parametrize('cached', [False, True])
def test_fused_head_on_the_device_equals_the_torch_formulation(golden_dir, cached):
    """csrc/lpips_head.hip (dbw_lpips_head_fwd / _bwd through the C-ABI) against the same module with the head written in torch ops: value and
    gradient to `rec` at 1e-5, with the targets' features computed in the call and gathered from the cache (ids out of order and repeated),
    at an image size whose taps are no multiple of the workgroup; a view id outside the cache poisons the value instead of reading out of bounds."""
    from dbw_amd.lpips_vgg import LPIPSVGG
    g, vgg, lin = _fixture(golden_dir)
    net = LPIPSVGG().load_weights(vgg, lin).to('cuda')
    gen = torch.Generator().manual_seed(3)
    imgs_all = torch.rand(3, 3, 52, 76, generator=gen).cuda()
    ids = torch.tensor([2, 0, 2, 1], device='cuda')
    rec = (imgs_all[ids] * 0.7 + 0.3 * torch.rand(4, 3, 52, 76, generator=gen).cuda()).requires_grad_(True)
    if cached:
        net.cache_targets(imgs_all)
    kw = dict(view_ids=ids) if cached else {}
    out = {}
    for fused in (False, True):
        net.fused_head = fused
        v = net(imgs_all[ids], rec, **kw)
        gr, = torch.autograd.grad(v * 3.0, rec)
        out[fused] = (float(v), gr)
    (v0, g0), (v1, g1) = out[False], out[True]
    assert v0 > 1e-3 and abs(v1 - v0) <= 1e-5 * v0, (v0, v1)
    assert float((g1 - g0).abs().max()) <= 1e-5 * float(g0.abs().max())
    if cached:
        bad = net(imgs_all[ids], rec, view_ids=torch.tensor([0, 1, 3, 1], device='cuda'))
        assert torch.isnan(bad)
        host_ids = net(imgs_all[ids], rec, view_ids=ids.cpu().int())          # ids on the host, 32-bit: moved and widened by the module
        assert abs(float(host_ids) - v1) <= 1e-6 * v1


@pytest.mark.gpu
def test_bias_relu_and_maxpool_kernels_equal_torch_bit_for_bit():
    """dbw_bias_relu / dbw_maxpool2_fwd / dbw_maxpool2_bwd (the layers between the frozen network's convolutions) against torch's own ops on
    the device: same bits forward and backward, at odd sizes (floor mode drops the last row / column), with windows that are ties -- zeros
    behind a ReLU: the gradient goes to the FIRST maximum of the window, as torch's forward picks it."""
    import torch.nn.functional as F
    from dbw_amd.lpips_vgg import _BiasReLU, _MaxPool2
    g = torch.Generator().manual_seed(7)
    for shape in ((2, 5, 13, 19), (3, 4, 8, 12), (1, 3, 37, 50)):
        x = torch.randn(*shape, generator=g).cuda()
        b = torch.randn(shape[1], generator=g).cuda()
        x0 = x.clone().requires_grad_(True)
        ref = F.relu(x0 + b.view(1, -1, 1, 1))
        x1 = x.clone().requires_grad_(True)
        got = _BiasReLU.apply(x1 * 1.0, b)                       # (in place on the product, a non-leaf like a convolution's output)
        w = torch.randn(*shape, generator=g).cuda()
        (ref * w).sum().backward()
        (got * w).sum().backward()
        assert torch.equal(ref, got) and torch.equal(x0.grad, x1.grad)
        y0 = ref.detach().clone().requires_grad_(True)             # behind a ReLU: many all-zero windows
        y1 = ref.detach().clone().requires_grad_(True)
        p0, p1 = F.max_pool2d(y0, 2, 2), _MaxPool2.apply(y1)
        wp = torch.randn(*p0.shape, generator=g).cuda()
        (p0 * wp).sum().backward()
        (p1 * wp).sum().backward()
        assert torch.equal(p0, p1) and torch.equal(y0.grad, y1.grad)
        assert (y0.grad != 0).sum() > 0 and float((ref == 0).float().mean()) > 0.2
