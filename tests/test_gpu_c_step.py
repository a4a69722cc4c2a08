"""GPU: the optimisation iteration behind one C-ABI call (dbw_train_step_*, csrc/train_step.hip) against the launch-by-launch native step
(dbw_amd/native_step.py, itself held to the autograd iteration and through it to the oracle): same loss values, same flat gradient, same
parameters after Adam -- for the operator-level kernels enqueued from C (fuse = 0), for each fused kernel alone (1, 2, 4, 1 + 8) and for all
of them (15), and with the env layer folded into the fg pass (31), in the three training phases; the step's own random numbers against the host build of the same generator; the loss values
the step copies to host memory; fresh mini-batches; the reference's own operating point (4 views).  `-m gpu`."""
import ctypes
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O                                              # noqa: E402  (checker only)
import dbw_amd                                                  # noqa: E402
from dbw_amd import _lib, ops                                   # noqa: E402
from dbw_amd.parallel import ShardedTrainStep                   # noqa: E402
from trajectory import assert_same_trajectory                   # noqa: E402

DEV = 'cuda:0'
HERE = os.path.dirname(os.path.abspath(__file__))


def _cfg(n_blocks=4, ts=32, fpp=6):
    return {'model': {'name': 'dbw', 'mesh': {'n_blocks': n_blocks, 'S_world': 0.5, 'R_world': [115, 0, 0], 'txt_size': ts},
                      'renderer': {'faces_per_pixel': fpp, 'cameras': {'name': 'perspective'}, 'detach_bary': True, 'z_clip': 0.001},
                      'rend_optim': {'coarse_learning': 1500, 'decimate_txt': 750, 'decimate_factor': 8, 'kill_blocks': True,
                                     'decouple_rendering': True, 'opacity_noise': True},
                      'loss': {'rgb_weight': 1, 'perceptual_weight': 0, 'parsimony_weight': 0.01, 'tv_weight': 0.1, 'overlap_weight': 1}}}


def _inputs(n_views, H, W, seed=2):
    R, T, Km = O.synthetic_cameras(n_views, R_world=O.world_rotation(115, 0, 0))
    imgs = torch.rand(n_views, 3, H, W, generator=torch.Generator().manual_seed(seed))
    return {k: v.to(DEV) for k, v in dict(imgs=imgs, R=R, T=T, K=Km).items()}


def _model(epoch, nb=4, ts=32, fpp=6, H=48, W=64, kill=True):
    torch.manual_seed(227391)
    model = dbw_amd.create_model(_cfg(nb, ts, fpp), (H, W)).to(DEV).train()
    with torch.no_grad():
        model.T.mul_(0.5)
        if kill:
            model.alpha_logit.add_(torch.tensor(([1.0, -6.0, 0.3, 2.0] * nb)[:nb], device=DEV))       # block 1 is killed / filtered
    model.set_cur_epoch(epoch)
    model.sync_free = True
    return model


def _run(model, inp, steps, noise, u, **kw):
    model._noise_override, model._overlap_u_override = noise, u
    step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99, **kw)
    out = step(inp)
    torch.cuda.synchronize()
    vals = {k: float(v) for k, v in out.items()}
    grad1 = step.params.grad.clone()
    for _ in range(steps - 1):
        step(inp)
    torch.cuda.synchronize()
    return step, vals, grad1, step.params.flat.clone()


def _compare(a, b, names):
    (_, va, ga, pa), (_, vb, gb, pb) = a, b
    assert set(va) == set(vb)
    for k in va:
        assert abs(va[k] - vb[k]) <= 2e-6 * max(abs(vb[k]), 1e-3), (k, va[k], vb[k])
    for n, off, k in names:
        x, y = ga[off:off + k], gb[off:off + k]
        assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max()) + 1e-12, (n, float((x - y).abs().max()), float(y.abs().max()))
    assert_same_trajectory(pa, pb)          # (later steps: tests/trajectory.py)


@pytest.mark.parametrize('epoch', [0, 800, 1600])
@pytest.mark.parametrize('fuse', [0, 1, 2, 4, 9, 15, 31, 63, 127])
def test_c_step_equals_native_step(epoch, fuse):
    inp = _inputs(3, 48, 64)
    noise = torch.randn(4, generator=torch.Generator().manual_seed(3)).to(DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    ref = _run(_model(epoch), inp, 3, noise, u, use_c_step=False)
    assert ref[0].cstep is None and ref[0].native is not None
    got = _run(_model(epoch), inp, 3, noise, u, use_c_step=True, fuse=fuse)
    assert got[0].cstep is not None and got[0].cstep.supported() and got[0].cstep._cur is not None, 'the C step did not run'
    _compare(got, ref, ref[0].params.names)


@pytest.mark.parametrize('epoch', [0, 800])
def test_c_step_at_the_benchmark_geometry_fused_equals_operator_level_kernels(epoch):
    """Config-2 geometry (300x400, K = 10, 10 blocks of 80 faces: four chunks of faces per view in the set-up kernel, per-tile lists of real
    density, the ground plane crossing the near plane in every view) on 5 views: every fused kernel against the kernels it replaces, and
    against the launch-by-launch native step."""
    inp = _inputs(5, 300, 400)
    nb = 10
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3)).to(DEV)
    u = torch.rand(nb, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    mk = lambda: _model(epoch, nb=nb, ts=256, fpp=10, H=300, W=400, kill=False)
    ref = _run(mk(), inp, 2, noise, u, use_c_step=False)
    for fuse in (0, 15, 31, 127):
        got = _run(mk(), inp, 2, noise, u, use_c_step=True, fuse=fuse)
        _compare(got, ref, ref[0].params.names)


def test_c_step_draws_its_noise_and_samples_from_the_counter_based_generator():
    """Without the caller's draws the step makes its own (rng_math.h): the opacities of step t are sigmoid(logit + std * normal(seed, t, k))
    of the HOST build of the same generator, identical on a second replica (what data-parallel ranks need), different from step to
    step; and the overlap term over the step's own samples is the overlap term over torch's samples up to Monte-Carlo noise."""
    import test_host_model_math as HM
    L = HM.lib()
    L.host_step_noise.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
    inp = _inputs(2, 48, 64)
    alphas = []
    for rep in range(2):
        model = _model(0, kill=False)
        step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=1234)
        logit = model.alpha_logit.detach().cpu().clone()
        got = []
        for t in range(3):
            out = step(inp)
            torch.cuda.synchronize()
            got.append(step.cstep.view('alpha', numel=4).cpu().clone())
            z = torch.empty(4)
            L.host_step_noise(1234, t, 4, ctypes.c_void_p(z.data_ptr()))
            want = torch.sigmoid(logit + float(model.opacity_noise) * z)
            assert float((got[-1] - want).abs().max()) < 2e-6, (t, got[-1], want)
        assert not torch.equal(got[0], got[1])
        ov = float(out['overlap'])
        # the counter is the optimisation step, not a per-plan count: a ragged batch and the next phase's plan (epoch 800) go on with t = 3, 4
        for t, (e, nv) in ((3, (0, 1)), (4, (800, 2))):
            model.set_cur_epoch(e)
            step({k: v[:nv].contiguous() for k, v in inp.items()})
            torch.cuda.synchronize()
            got.append(step.cstep.view('alpha', numel=4).cpu().clone())
            z = torch.empty(4)
            L.host_step_noise(1234, t, 4, ctypes.c_void_p(z.data_ptr()))
            assert float((got[-1] - torch.sigmoid(logit + float(model.opacity_noise) * z)).abs().max()) < 2e-6, t
        alphas.append(torch.stack(got))
    assert torch.equal(alphas[0], alphas[1])
    # overlap with torch's samples (the launch-by-launch step): same scene, frozen parameters
    model = _model(0, kill=False)
    model._noise_override = torch.zeros(4, device=DEV)
    ref = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=1234, use_c_step=False)
    ovs = [float(ref(inp)['overlap']) for _ in range(8)]
    mean, spread = sum(ovs) / len(ovs), max(ovs) - min(ovs)
    assert abs(ov - mean) <= max(3 * spread, 0.1 * abs(mean)) + 1e-6, (ov, ovs)


def test_c_step_copies_the_loss_values_to_host_memory_itself():
    """read_losses: the step leaves rgb / parsimony / tv / overlap / total in pinned host memory with ONE copy (src/trainer.py:143 reads six
    scalars one by one); they are the values on the device, and total is their sum."""
    inp = _inputs(3, 48, 64)
    model = _model(0)
    step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=7)
    step.cstep.read_losses = True
    for _ in range(3):
        out = step(inp)
        host = out.host()
        dev = {k: float(v) for k, v in out.items()}
        assert set(host) == set(dev) == {'rgb', 'parsimony', 'tv', 'overlap', 'total'}
        for k in host:
            assert host[k] == dev[k], (k, host[k], dev[k])
        assert abs(host['total'] - (host['rgb'] + host['parsimony'] + host['tv'] + host['overlap'])) < 1e-6 * abs(host['total'])
        assert host['rgb'] > 0 and host['tv'] > 0


def test_c_step_on_fresh_minibatches_of_the_reference_batch_size():
    """configs/dtu/default.yml:28: batch_size 4, a different set of views every iteration (src/trainer.py:137-147).  The step tiles a fresh
    mini-batch's targets itself and sizes nothing by the first batch it saw: a plan made for 4 views runs 4, then a ragged last batch of
    2, then 4 again -- each equal to the launch-by-launch step on the same views."""
    H, W = 48, 64
    allv = _inputs(10, H, W, seed=5)
    nz = torch.zeros(4, device=DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    res = []
    for c in (True, False):
        model = _model(0)
        model._noise_override, model._overlap_u_override = nz, u
        step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=7, use_c_step=c)
        grads = []
        for a, b in ((0, 4), (4, 8), (8, 10), (2, 6)):
            inp = {k: (v[a:b].clone() if k != 'K' else v[a:b]) for k, v in allv.items()}
            out = step(inp)
            torch.cuda.synchronize()
            grads.append((step.params.grad.clone(), float(out['total'])))
        res.append(grads)
    for (ga, la), (gb, lb) in zip(*res):
        assert abs(la - lb) <= 2e-6 * abs(lb), (la, lb)
        assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()), float((ga - gb).abs().max())


@pytest.mark.parametrize('epoch', [0, 800])
def test_env_layer_folded_into_the_fg_pass_leaves_the_fragments_of_the_env_pass(epoch):
    """fuse bit 4: every 8x8 tile of the fg pass rasterises and shades its pixel of the env scene itself (per-tile lists of the env faces,
    the env pass's own per-pair arithmetic) instead of reading the image of a separate env pass.  What the env backward consumes -- the
    hard uv-fragments: clipped face id, u, v, face | map per pixel -- must be what the env pass writes, bit for bit; config-2 geometry, the
    ground plane crossing the near plane (split quads: the sibling rule)."""
    inp = _inputs(5, 300, 400)
    nb = 10
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3)).to(DEV)
    u = torch.rand(nb, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    frags = []
    for fuse in (15, 31):
        model = _model(epoch, nb=nb, ts=256, fpp=10, H=300, W=400, kill=False)
        model._noise_override, model._overlap_u_override = noise, u
        step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=99, fuse=fuse)
        out = step(inp)
        torch.cuda.synchronize()
        tiles = 5 * ((300 + 7) // 8) * ((400 + 7) // 8)
        p2f = step.cstep.view('p2f_env', torch.int32, tiles * 64).clone()
        uvj = step.cstep.view('uvj_env', torch.int32, tiles * 192).view(tiles, 3, 64).clone()
        frags.append((p2f, uvj, {k: float(v) for k, v in out.items()}, step.params.grad.clone()))
    (pa, ua, la, ga), (pb, ub, lb, gb) = frags
    # rows / columns of the last tiles that lie beyond the image are never written: compare the pixels of the image
    ty, tx = (300 + 7) // 8, (400 + 7) // 8
    yy = (torch.arange(ty, device=DEV)[:, None, None] * 8 + torch.arange(64, device=DEV)[None, None, :] // 8).expand(ty, tx, 64)
    xx = (torch.arange(tx, device=DEV)[None, :, None] * 8 + torch.arange(64, device=DEV)[None, None, :] % 8).expand(ty, tx, 64)
    inside = ((yy < 300) & (xx < 400)).reshape(1, ty * tx, 64).expand(5, -1, -1).reshape(-1)
    assert torch.equal(pa[inside], pb[inside]) and int((pa[inside] >= 0).sum()) > 0.9 * int(inside.sum())
    valid = (inside & (pa >= 0)).view(-1, 1, 64).expand(-1, 3, -1)
    assert torch.equal(ua[valid], ub[valid])
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-6 * max(abs(la[k]), 1e-3), (k, la[k], lb[k])
    assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max())


@pytest.mark.parametrize('epoch', [0, 1600])
def test_c_step_with_the_perceptual_term_equals_the_autograd_iteration(epoch):
    """configs/dtu/default.yml:23 trains with perceptual_weight 0.1 (src/model/dbw.py:369-371, loss.py:32-40): LPIPS on the composite.  The
    network stays outside the library; the step runs in two phases around it -- composite out, d term / d rec in -- and must give the
    losses, the flat gradient and the parameters of the autograd iteration `model(inp); total.backward(); Adam` with the same network
    (seeded LPIPS-VGG16 weights: none exist offline; the architecture is what tests/test_lpips.py pins)."""
    from dbw_amd.lpips_vgg import LPIPSVGG
    H, W = 48, 64
    inp = _inputs(3, H, W)
    noise = torch.randn(4, generator=torch.Generator().manual_seed(3)).to(DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    torch.manual_seed(5)
    net = LPIPSVGG(allow_random_init=True).to(DEV)
    res = []
    for c in (True, False):
        torch.manual_seed(227391)
        cfg = _cfg(4, 32, 6)
        cfg['model']['loss']['perceptual_weight'] = 0.1
        model = dbw_amd.create_model(cfg, (H, W)).to(DEV).train()
        with torch.no_grad():
            model.T.mul_(0.5)
            model.alpha_logit.add_(torch.tensor([1.0, -6.0, 0.3, 2.0], device=DEV))
        model.set_cur_epoch(epoch)
        model.sync_free = True
        model.set_perceptual(net)
        model._noise_override, model._overlap_u_override = noise, u
        step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99, use_c_step=c, use_native=c)
        assert (step.cstep is not None and step.cstep.supported()) == c
        out = step(inp)
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in out.items()}
        assert set(vals) == {'rgb', 'perceptual', 'parsimony', 'tv', 'overlap', 'total'} and vals['perceptual'] > 0
        if c:
            assert step.cstep._cur is not None and out.host() == pytest.approx(vals, rel=1e-6)
        grad1 = step.params.grad.clone()
        step(inp)
        torch.cuda.synchronize()
        res.append((step, vals, grad1, step.params.flat.clone()))
    _compare(res[0], res[1], res[0][0].params.names)


def test_c_step_at_a_config_4_like_geometry_equals_the_native_step():
    """BASELINE config 4's shape at a small size: 20 blocks, faces_per_pixel 16 (the K = 16 instantiation of the fused forward with the env
    layer inside it, 1600 block faces: seven chunks of faces per view in the set-up kernel), non-square image whose sides are not
    multiples of 16, full-resolution textures (texture bins, two consecutive steps: the second one's record sub-ranges follow the
    first one's demand)."""
    H, W, nb = 72, 100, 20
    inp = _inputs(3, H, W)
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3)).to(DEV)
    u = torch.rand(nb, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    for epoch in (0, 800):
        mk = lambda: _model(epoch, nb=nb, ts=64, fpp=16, H=H, W=W, kill=False)
        ref = _run(mk(), inp, 3, noise, u, use_c_step=False)
        got = _run(mk(), inp, 3, noise, u, use_c_step=True)
        assert got[0].cstep is not None and got[0].cstep.fuse == 127
        _compare(got, ref, ref[0].params.names)


@pytest.mark.parametrize('epoch', [0, 800])
def test_c_step_at_config_5_geometry_equals_the_native_step(epoch):
    """BASELINE config 5 at FULL size -- 1080 x 1920, 50 blocks (4 000 faces: sixteen chunks of faces per view in the set-up kernel), faces_per_pixel
    16, 512^2 textures (texture bins at epoch 800; int32 texel offsets of 52 maps), 2 of the 25 views a GPU holds -- through `dbw_train_step_run`,
    the entry point bench.py's `configs.c5` leg measures, INCLUDING the plan's first run (the one that starts behind the driver's clearing of
    the freshly allocated workspace and goes through events; it met a real voided step in round 5): loss values, the whole flat gradient and
    the parameters after two Adam steps against the launch-by-launch native step.  (The oracle at this scene is held by
    test_gpu_configs.py: one full view of indices / depths / distances bit-exact, and -- at a size the CPU finishes -- the C step at 50
    blocks, 25 faces per pixel against OracleDBW.)"""
    H, W, nb = 1080, 1920, 50
    inp = _inputs(2, H, W)
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3)).to(DEV)
    u = torch.rand(nb, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    mk = lambda: _model(epoch, nb=nb, ts=512, fpp=16, H=H, W=W, kill=False)
    got = _run(mk(), inp, 2, noise, u, use_c_step=True)          # (first: its plan's first run is then the first thing to touch new memory)
    assert got[0].cstep is not None and got[0].cstep.fuse == 127 and got[0].cstep.sync_timeouts() == 0 and got[0].cstep.voided_runs() == 0
    torch.cuda.empty_cache()
    ref = _run(mk(), inp, 2, noise, u, use_c_step=False)
    assert ref[0].cstep is None and ref[0].native is not None
    _compare(got, ref, ref[0].params.names)
    del got, ref
    torch.cuda.empty_cache()


def test_sharded_step_with_the_sigmoid_opacity_takes_the_autograd_path():
    """`clip_inside=False` (renderer.py:41,257-258) is implemented by the generic kernels of the autograd path only: a ShardedTrainStep
    on such a model must say so (neither the C step nor the launch-by-launch native step claims it) and run the iteration through
    autograd -- not hand a negative sigma to dbw_train_step_create and raise."""
    inp = _inputs(2, 48, 64)
    noise = torch.zeros(4, device=DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)

    def mk():
        torch.manual_seed(227391)
        cfg = _cfg()
        cfg['model']['renderer']['clip_inside'] = False
        model = dbw_amd.create_model(cfg, (48, 64)).to(DEV).train()
        model.set_cur_epoch(0)
        model.sync_free = True
        model._noise_override, model._overlap_u_override = noise, u
        return model
    model = mk()
    assert not model.renderer.clip_inside
    step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=99)
    assert step.cstep is None or not step.cstep.supported()
    assert step.native is None or not step.native.supported()
    out = step(inp)
    torch.cuda.synchronize()
    ref_model = mk()
    ref = ref_model(inp, None)
    ref['total'].backward()
    assert abs(float(out['total']) - float(ref['total'].detach())) <= 1e-6 * abs(float(ref['total'].detach()))
    for n, off, k in step.params.names:
        g = getattr(ref_model, n).grad.reshape(-1)
        assert float((step.params.grad[off:off + k] - g).abs().max()) <= 1e-5 * float(g.abs().max()) + 1e-12, n


def test_c_step_one_stream_equals_side_streams_and_every_launch_is_there():
    """dbw_step_inputs.single_stream: everything in order on the caller's stream (what a per-kernel profile wants) gives the same step as
    the three-stream schedule."""
    inp = _inputs(3, 48, 64)
    noise = torch.zeros(4, device=DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    res = []
    for side in (True, False):
        model = _model(0)
        model._noise_override, model._overlap_u_override = noise, u
        step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99)
        step.cstep.use_side_stream = side
        out = step(inp)
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in out.items()}
        grad1 = step.params.grad.clone()
        step(inp)
        torch.cuda.synchronize()
        res.append((step, vals, grad1, step.params.flat.clone()))
    _compare(res[0], res[1], res[0][0].params.names)


@pytest.mark.parametrize('epoch', [0, 800])
def test_c_step_streams_wait_through_memory_words_as_through_events(epoch):
    """dbw_step_desc.sync_events: the plan's streams wait for each other through polled words in device memory (default: a one-thread store
    kernel behind the producer, a one-thread poll kernel in front of the consumer) or through HIP events -- the same step either way, over
    several iterations (a wait that does not hold shows up as a race), and no poll ever gives up."""
    inp = _inputs(3, 48, 64)
    noise = torch.zeros(4, device=DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    res = []
    for events in (False, True):
        model = _model(epoch)
        model._noise_override, model._overlap_u_override = noise, u
        step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99)
        step.cstep.sync_events = events
        out = step(inp)
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in out.items()}
        grad1 = step.params.grad.clone()
        for _ in range(6):
            step(inp)
        torch.cuda.synchronize()
        assert step.cstep.sync_timeouts() == 0
        res.append((step, vals, grad1, step.params.flat.clone()))
    _compare(res[0], res[1], res[0][0].params.names)


@pytest.mark.parametrize('epoch', [0, 800, 1600])
def test_c_step_with_the_texture_tail_deferred_equals_the_step_in_one_call(epoch):
    """dbw_step_inputs.defer_textures + dbw_train_step_finish: what a data-parallel rank runs around its all-reduce of the prepared maps'
    gradient (here: no other rank, nothing to add) is the step in one call -- loss values, gradient buffer, parameters after Adam; the
    range it would reduce holds the maps' gradient and nothing else of the arena."""
    inp = _inputs(3, 48, 64)
    noise = torch.zeros(4, device=DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    res = []
    for defer in (False, True):
        model = _model(epoch)
        model._noise_override, model._overlap_u_override = noise, u
        step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99)
        cs = step.cstep
        grads = None
        for it in range(3):
            adam = (it + 1, step.lrs, step.betas, step.eps)
            if defer:
                out = cs(inp, None, adam=None, defer_textures=True)
                if it == 0:
                    mg = cs.map_grads()
                    decim = 8 if epoch < 750 else 1
                    assert mg.numel() >= (4 * 32 * 32 * 3 + 2 * 32 * 32 * 3) // decim ** 2 and float(mg.abs().max()) > 0
                cs.finish(adam=adam)
            else:
                out = cs(inp, None, adam=adam)
            torch.cuda.synchronize()
            if it == 0:
                vals, grads = {k: float(v) for k, v in out.items()}, step.params.grad.clone()
        res.append((step, vals, grads, step.params.flat.clone()))
    _compare(res[0], res[1], res[0][0].params.names)


def test_c_step_through_the_three_training_phases_and_back_no_wait_ever_gives_up():
    """A few hundred iterations across the phase changes of the schedule (a plan per phase, the library's side streams shared by all of
    them, the host running ahead of the GPU and reading the loss values now and then): finite, decreasing, and none of the polls through
    which the streams wait for each other gave up."""
    inp = _inputs(4, 48, 64)
    model = _model(0, kill=False)
    step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=5)
    step.cstep.read_losses = True
    first = {}
    for epoch in (0, 800, 1600, 0, 1600):
        model.set_cur_epoch(epoch)
        for it in range(60):
            out = step(inp)
            if it % 7 == 0:
                vals = out.host()
                assert all(v == v and abs(v) < 1e3 for v in vals.values()), (epoch, it, vals)
                first.setdefault(epoch, vals['rgb'])
                last = vals['rgb']
        assert last < first[epoch] * 1.05, (epoch, first[epoch], last)
        torch.cuda.synchronize()
        assert step.cstep.sync_timeouts() == 0
    assert len(step.cstep._plans) == 3


def test_c_step_at_the_full_benchmark_batch_equals_the_native_step():
    """BASELINE config 2 at full size -- 49 views of 300x400, 10 blocks, faces_per_pixel 10, 256^2 textures, the first training phase: the
    one-call step with every fusion on against the launch-by-launch native step (itself held to the autograd iteration and the oracle at
    sizes the oracle finishes)."""
    inp = _inputs(49, 300, 400)
    nb = 10
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3)).to(DEV)
    u = torch.rand(nb, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    mk = lambda: _model(0, nb=nb, ts=256, fpp=10, H=300, W=400, kill=False)
    ref = _run(mk(), inp, 2, noise, u, use_c_step=False)
    got = _run(mk(), inp, 2, noise, u, use_c_step=True)
    assert got[0].cstep is not None and got[0].cstep._cur is not None and got[0].cstep.sync_timeouts() == 0
    _compare(got, ref, ref[0].params.names)


def test_c_step_voids_a_step_whose_wait_gave_up_and_goes_on_through_events():
    """A poll between the step's streams that gives up must not let the step update anything: forced here for real (the join of the run
    polls for a value that never comes, 0.05 s) -- the step's Adam launch sees the void flag and moves no parameter and no moment, the
    next run notices the word in mapped host memory, switches the plan to HIP events and goes on; from then on the plan equals a plan
    created on events, run for run."""
    import warnings
    inp = _inputs(2, 48, 64)
    noise = torch.zeros(4, device=DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)

    def mk(events):
        model = _model(0)
        model._noise_override, model._overlap_u_override = noise, u
        step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99)
        step.cstep.sync_events = events
        return step
    ref = mk(True)
    ref(inp)
    ref(inp)                               # reference: two applied steps on events
    step = mk(False)
    step(inp)
    torch.cuda.synchronize()
    assert step.cstep.sync_timeouts() == 0 and step.cstep.voided_runs() == 0
    # (a plan's FIRST run goes through events whatever sync_events says -- it is the run that waits for the driver to clear freshly allocated
    # memory, a second at BASELINE config 5 -- so no counter has been asked for yet; the polls start with the second run)
    seen, asked = (ctypes.c_uint * 12)(), (ctypes.c_uint * 12)()
    _lib.call('dbw_debug_train_step_counters', step.cstep._cur[0], seen, asked)
    assert not any(asked) and step.cstep.last_timeout() is None
    before = (step.params.flat.clone(), step.exp_avg.clone(), step.exp_avg_sq.clone())
    _lib.call('dbw_debug_train_step_force_timeout', step.cstep._cur[0])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        step(inp)                          # voided: its join gives up
        torch.cuda.synchronize()
        assert step.cstep.sync_timeouts() == 1 and step.cstep.voided_runs() == 1
        which = step.cstep.last_timeout()
        assert which[0] == 'env chain' and which[3]['prologue'] == (1, 1)          # the join's poll, and what the counters stood at then
        for a, b in zip(before, (step.params.flat, step.exp_avg, step.exp_avg_sq)):
            assert torch.equal(a, b)       # nothing moved
        step.n_steps -= 1                  # (test only: line the Adam step count up with the reference's two applied steps)
        out = step(inp)                    # goes on: through events from here
        torch.cuda.synchronize()
    assert any('gave up' in str(x.message) for x in w)
    assert step.cstep.sync_timeouts() == 1 and step.cstep.voided_runs() == 1
    assert all(torch.isfinite(v).all() for v in out.values())
    assert_same_trajectory(step.params.flat, ref.params.flat)


def test_c_step_side_polls_that_give_up_in_front_of_a_stalled_main_stream_void_the_step():
    """The descheduling case the void flag exists for: the caller's stream is stalled (here: 0.25 s of spinning enqueued in front of the
    step) while the plan's side streams -- which wait for the step's prologue through polled words -- give up (limit forced down to 0.02 s,
    their real value) and run ahead on stale inputs.  When the main stream finally runs, nothing it executes may wipe the flag those polls
    raised (a clear at the head of the run, ordered on the main stream, did exactly that): the step's Adam launch moves no parameter and
    no moment, the next run notices, goes on through events, and equals the reference from there."""
    import warnings
    inp = _inputs(2, 48, 64)
    noise = torch.zeros(4, device=DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)

    def mk(events):
        model = _model(0)
        model._noise_override, model._overlap_u_override = noise, u
        step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99)
        step.cstep.sync_events = events
        return step
    ref = mk(True)
    ref(inp)
    ref(inp)                               # reference: two applied steps on events
    step = mk(False)
    step(inp)                              # (a plan's first run goes through events; the polls start with the second)
    torch.cuda.synchronize()
    assert step.cstep.sync_timeouts() == 0 and step.cstep.voided_runs() == 0
    # how many spin cycles are 0.25 s on this GPU
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
    cycles = int(20_000_000 * 250.0 / max(e0.elapsed_time(e1), 1e-3))
    before = (step.params.flat.clone(), step.exp_avg.clone(), step.exp_avg_sq.clone())
    _lib.call('dbw_debug_train_step_hasty_prologue_wait', step.cstep._cur[0])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        torch.cuda._sleep(cycles)          # the main stream is busy for 0.25 s; the side streams are not
        step(inp)                          # voided: the side streams' polls for the prologue give up after 0.02 s
        torch.cuda.synchronize()
        assert step.cstep.sync_timeouts() >= 1 and step.cstep.voided_runs() == 1
        assert step.cstep.last_timeout()[0] == 'prologue'
        for a, b in zip(before, (step.params.flat, step.exp_avg, step.exp_avg_sq)):
            assert torch.equal(a, b)       # nothing moved, although the main stream ran its whole chain AFTER the polls had given up
        step.n_steps -= 1                  # (test only: line the Adam step count up with the reference's two applied steps)
        out = step(inp)                    # goes on: through events from here
        torch.cuda.synchronize()
    assert any('gave up' in str(x.message) for x in w)
    assert step.cstep.voided_runs() == 1
    assert all(torch.isfinite(v).all() for v in out.values())
    assert_same_trajectory(step.params.flat, ref.params.flat)


def test_arena_cleaned_by_the_caller_counts_for_the_plan_it_cleaned_only():
    """A caller that runs Adam itself (data parallel, whole-buffer flow) clears the CURRENT plan's zero arena; a run of ANOTHER plan --
    the next phase -- must still open with its own fill: the mark is per plan."""
    inp = _inputs(2, 48, 64)
    noise = torch.zeros(4, device=DEV)
    u = torch.rand(4, 1000, 3, generator=torch.Generator().manual_seed(4)).to(DEV)
    model = _model(0)
    model._noise_override, model._overlap_u_override = noise, u
    step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=99)
    cs = step.cstep
    want0 = {k: float(v) for k, v in step(inp).items()}
    g0 = step.params.grad.clone()
    model.set_cur_epoch(800)
    want1 = {k: float(v) for k, v in step(inp).items()}
    g1 = step.params.grad.clone()
    cs(inp, adam=None)                                        # the epoch-800 plan is left dirty (no Adam, nobody cleared its arena)
    model.set_cur_epoch(0)
    cs(inp, adam=None)                                        # ... and so is the epoch-0 plan; the caller now cleans THIS one:
    cs.arena().zero_()
    cs.arena_cleaned_by_caller()
    model.set_cur_epoch(800)
    got1 = {k: float(v) for k, v in cs(inp, adam=None).items()}      # must not inherit the mark
    torch.cuda.synchronize()
    for k in want1:
        assert abs(got1[k] - want1[k]) <= 2e-6 * max(abs(want1[k]), 1e-3), (k, got1[k], want1[k])
    assert float((step.params.grad - g1).abs().max()) <= 1e-5 * float(g1.abs().max())
    model.set_cur_epoch(0)
    got0 = {k: float(v) for k, v in cs(inp, adam=None).items()}      # the cleaned plan runs without a fill, on a clean arena
    torch.cuda.synchronize()
    for k in want0:
        assert abs(got0[k] - want0[k]) <= 2e-6 * max(abs(want0[k]), 1e-3), (k, got0[k], want0[k])
    assert float((step.params.grad - g0).abs().max()) <= 1e-5 * float(g0.abs().max())
