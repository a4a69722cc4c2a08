"""GPU parity at the shapes of BASELINE.json's configs (run with `-m gpu`):
  c1  DTU scan24, 4 views, 100x75 render, 4 blocks, faces_per_pixel 4: one full training iteration (all losses, all 10 parameter
      gradients) against the oracle, at the exact shape, in the three training phases;
  c4  BlendedMVS-like, 768x576, 20 blocks, faces_per_pixel 16 (the KMAX = 16 instantiations, 1600 faces): size-independent
      properties, one full view of face indices / distances bit-exact against the oracle, fused vs operator-level gradients;
  c5  Nerfstudio-like, 1920x1080, 50 blocks, 512^2 textures, faces_per_pixel 16 (4000 faces, 20-bit face | 11-bit map packing of
      the uv-fragments, int32 texel offsets of 52 x 512^2 x 3 floats, texture-space bins): properties, binned vs atomic texel
      gradients, the conservation law of the scatter, and one view of oracle indices.
c2 / c3 (300x400, 10 blocks, faces_per_pixel 10) are covered by tests/test_gpu_parity.py::test_full_size_* and bench.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O                                              # noqa: E402  (checker only)
import dbw_amd                                                  # noqa: E402
from dbw_amd import ops                                         # noqa: E402
from dbw_amd.structures import PackedScene                      # noqa: E402

DEV = 'cuda:0'
REL = 1e-4


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def off_entries(a, b, rel=REL, floor=1e-2):
    """Element-wise form of the bar: entries with |a - b| > rel * |b| + rel * floor * max|b| (the max-norm check above leaves small
    entries unconstrained; this one holds every entry to 1e-4 of ITSELF, with a floor of 1e-6 of the largest entry so that values four
    orders of magnitude down are not held to their own last bits).  -> (number of entries off, worst excess as a multiple of the bar)"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    tol = rel * b.abs() + rel * floor * b.abs().max()
    ex = (a - b).abs() / tol.clamp(min=1e-300)
    return int((ex > 1).sum()), float(ex.max())


def _packed(scene, pads=None):
    maps = scene['maps']
    pads = pads or [(0, 0)] * len(maps)
    desc, _ = PackedScene.describe_maps([m.shape[:2] for m in maps], pads, DEV)
    flat = torch.cat([m.reshape(-1) for m in maps]).detach().to(DEV)
    return PackedScene(scene['verts'].detach().to(DEV), scene['faces'].to(torch.int32).to(DEV), scene['face_uvs'].float().to(DEV),
                       scene['face_map'].to(torch.int32).to(DEV), desc, flat)


def _cfg(nb, ts, fpp, bkg_upscale=1, criterion='mse', tv_type='l2sq'):
    return {'model': {'name': 'dbw', 'mesh': {'n_blocks': nb, 'S_world': 0.5, 'R_world': [115, 0, 0], 'txt_size': ts, 'txt_bkg_upscale': bkg_upscale},
                      'renderer': {'faces_per_pixel': fpp, 'cameras': {'name': 'perspective'}, 'detach_bary': True, 'z_clip': 0.001},
                      'rend_optim': {'coarse_learning': 1500, 'decimate_txt': 750, 'decimate_factor': 8, 'kill_blocks': True,
                                     'decouple_rendering': True, 'opacity_noise': True},
                      'loss': {'name': criterion, 'tv_type': tv_type, 'rgb_weight': 1, 'perceptual_weight': 0, 'parsimony_weight': 0.01, 'tv_weight': 0.1,
                               'overlap_weight': 1}}}


# ---------------------------------------------------------------------------------------------------------------------
# config 1: exact shape, whole iteration
# ---------------------------------------------------------------------------------------------------------------------
def _fragment_flips(model, orc, inp, coarse, decimate, noise):
    """Number of fragment slots whose face differs when the SAME HIP rasteriser is fed the oracle's vertices instead of the model's.
    Parameters -> vertices goes through pow / sin / cos / exp, which differ by an ulp between the host's and the device's libm (each
    side is held to the oracle at 1e-5 elsewhere); the face lists are a discontinuous function of the vertices, so an ulp can move
    one borderline (pixel, face) pair in or out of a top-K list -- the one legitimate source of a > 1e-4 difference downstream."""
    flips = 0
    with torch.no_grad():
        fine = not coarse
        o_fg = orc.build_blocks(True, coarse, decimate, noise, filter_transparent=fine)
        o_env = orc.build_env(True, decimate)
        m_fg = model.build_blocks_scene(filter_transparent=fine)
        m_env = model.build_env_scene()
        R, T, Km = inp['R'].to(DEV), inp['T'].to(DEV), inp['K'][0].to(DEV)
        for mine, theirs, r in ((m_fg, o_fg, model.renderer_fine if fine else model.renderer), (m_env, o_env, model.renderer_env)):
            if mine is None:
                continue
            cfg = r._cfg(mine.faces.shape[0])
            a = ops.render_fragments(mine.verts, mine.faces, R, T, Km, cfg)[1]
            b = ops.render_fragments(theirs['verts'].to(DEV), mine.faces, R, T, Km, cfg)[1]
            flips += int((a != b).sum())
    return flips


def _oracle_in_double(orc):
    """A float64 twin of an OracleDBW (every float32 tensor it holds, parameters included): the 'true' value both fp32
    implementations approximate.  Used to judge gradients that are ill-conditioned sums (see the test below)."""
    import copy
    d = copy.copy(orc)
    for k, v in list(vars(orc).items()):
        if torch.is_tensor(v) and v.dtype == torch.float32:
            setattr(d, k, v.detach().double())
    d.p = {k: v.detach().double().requires_grad_(True) for k, v in orc.p.items()}
    return d


def _iteration(shape, seed, epoch, decimate, c_step=False, bkg_upscale=1, device_vertices=False, criterion='mse', tv_type='l2sq'):
    H, W, nb, ts, fpp, V = shape
    torch.manual_seed(227391)
    model = dbw_amd.create_model(_cfg(nb, ts, fpp, bkg_upscale, criterion, tv_type), (H, W))
    orc = O.OracleDBW((H, W), n_blocks=nb, txt_size=ts, txt_bkg_upscale=bkg_upscale, faces_per_pixel=fpp, seed=227391, criterion=criterion,
                      tv_type=tv_type)
    for k, v in orc.p.items():                                       # same seed, same draw order -> identical init
        assert torch.equal(v.detach(), getattr(model, k).detach()), k
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed)
        for name, scale in (('sq_eps', 1.0), ('alpha_logit', 1.0), ('R_6d_ground', 0.03)):
            d = torch.randn(orc.p[name].shape, generator=g) * scale
            orc.p[name].add_(d)
            getattr(model, name).add_(d)
        orc.p['T'].mul_(0.6)
        model.T.mul_(0.6)
    model = model.to(DEV).train()
    model.set_cur_epoch(epoch)
    coarse = epoch < 1500
    R, T, Km = O.synthetic_cameras(V, R_world=orc.R_world[0])
    imgs = torch.rand(V, 3, H, W, generator=torch.Generator().manual_seed(2))
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3))
    u = torch.rand(nb, 1000, 3, generator=torch.Generator().manual_seed(4))
    inp = dict(imgs=imgs, R=R, T=T, K=Km)
    model._noise_override, model._overlap_u_override = noise.to(DEV), u.to(DEV)
    verts_at = None
    if device_vertices:
        # large scenes: the oracle is evaluated AT the device's fp32 vertices (OracleDBW._verts_through: same values bit for bit, the
        # oracle's own graph behind them) -- at 4 000 faces x 25 layers an ulp of libm / summation-order difference in a vertex always moves
        # some borderline fragment, and then nothing downstream can be held to 1e-4; with identical vertices the rasteriser is bit-exact
        with torch.no_grad():
            m_fg, m_env = model.build_blocks_scene(filter_transparent=not coarse), model.build_env_scene()
            verts_at = {'env': m_env.verts.cpu(), 'blocks': None if m_fg is None else m_fg.verts.cpu()}
    ref = orc.forward(inp, training=True, coarse=coarse, decimate=decimate, opacity_noise=noise, overlap_points=u, n_threads=16, verts_at=verts_at)
    ref['total'].backward()
    flips_c = None
    if c_step:
        # (the fragment-flip count is taken from the model as the autograd test sees it -- host-packed blocks, its own vertex count --
        # BEFORE the parameters are re-homed into the flat buffers and the model is made sync-free for the step)
        flips_c = 0 if device_vertices else _fragment_flips(model, orc, inp, coarse, decimate, noise if coarse else None)
        # the benchmarked entry point: ONE call into the library per iteration (dbw_train_step_run, every fusion on: env layer folded into
        # the fg pass, fused set-up / tails); learning rates 0, so Adam runs and nothing moves -- gradients are read from the flat buffer
        from dbw_amd.parallel import ShardedTrainStep
        from dbw_amd.c_step import FUSE_ALL
        model.sync_free = True
        step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=227391)
        assert step.cstep is not None and step.cstep.supported() and step.cstep.fuse == FUSE_ALL
        out = step({k: v.to(DEV) for k, v in inp.items()})
        torch.cuda.synchronize()
        assert step.cstep._cur is not None and step.cstep.sync_timeouts() == 0
        out = {k: v.clone() for k, v in out.items()}
    else:
        out = model({k: v.to(DEV) for k, v in inp.items()}, None)
        out['total'].backward()
    assert set(out) == set(ref)
    errs = {}
    for k in ref:
        errs['loss ' + k] = abs(out[k].item() - ref[k].item()) / max(abs(ref[k].item()), 1e-3)
    for k, v in orc.p.items():
        gh = getattr(model, k).grad
        if gh is None or v.grad is None:
            assert (gh is None or gh.abs().max() == 0) and (v.grad is None or v.grad.abs().max() == 0), k
            continue
        errs['grad ' + k] = rel_err(gh, v.grad)
        if v.grad.numel() <= 64:          # the small pose / shape / opacity gradients: every entry, not just the largest (floor: 1e-5
            n_off, worst = off_entries(gh, v.grad, floor=0.1)          # of the largest entry -- each entry is a sum of ~10^4 terms of
            if n_off:                                                   # either sign, accumulated in fp32 on both sides)
                errs['grad ' + k] = max(errs['grad ' + k], REL * worst)
    if device_vertices:
        flips = 0
    elif flips_c is not None:
        flips = flips_c if max(errs.values()) >= REL else 0
    else:
        flips = _fragment_flips(model, orc, inp, coarse, decimate, noise if coarse else None) if max(errs.values()) >= REL else 0
    if max(errs.values()) >= REL:
        print(f'seed {seed}, epoch {epoch}: {flips} fragment flips; off: ' + ', '.join(f'{k} {v:.2e}' for k, v in errs.items() if v >= REL))
    if flips == 0 and max(errs.values()) >= REL:
        # No fragment differs, yet a gradient is off by more than 1e-4 of its largest entry: then it must be an ill-conditioned sum
        # (the ground pose only receives gradient through barycentrics -> uv -> texel differences of a noise texture: tens of
        # thousands of terms of either sign that cancel to a small total).  Judge both fp32 results against the oracle run in
        # float64: the HIP gradient has to be within 1e-4 of THAT (measured: 2e-5, while the fp32 CPU oracle, which
        # accumulates in fp32, sits 1e-3 away from its own float64 twin).
        o64 = _oracle_in_double(orc)
        inp64 = {k: v.double() for k, v in inp.items()}
        r64 = o64.forward(inp64, training=True, coarse=coarse, decimate=decimate, opacity_noise=noise.double(), overlap_points=u.double(),
                          n_threads=16, verts_at=verts_at)
        r64['total'].backward()
        for k, v in orc.p.items():
            key = 'grad ' + k
            if errs.get(key, 0.0) >= REL:
                truth = o64.p[k].grad.float()
                e_hip, e_orc = rel_err(getattr(model, k).grad, truth), rel_err(v.grad, truth)
                cond = float(v.grad.abs().max() / truth.abs().max().clamp(min=1e-30))
                print(f'{k}: |hip - fp64| = {e_hip:.2e}, |fp32 oracle - fp64| = {e_orc:.2e} (relative to max |fp64|), seed {seed}')
                # within 1e-4 of the float64 value -- or, where fp32 accumulation itself cannot get that close (the fp32 oracle is
                # further away than that), at least as close to it as the fp32 oracle is
                assert e_hip <= max(REL, 1.2 * e_orc), (k, e_hip, e_orc, cond)
                errs[key] = 0.0
    return errs, flips


C1 = (75, 100, 4, 256, 4, 4)          # BASELINE configs[0]: H, W, blocks, texture size, faces_per_pixel, views


def _report_draws(record_property, what, tried):
    """How many parameter draws had to be replaced because of a verified borderline fragment flip is part of the result: a kernel that
    moves more borderline fragments than an ulp of libm explains shows up here long before all five draws are used up.  The count goes
    into the junit properties AND, when it is not zero, into pytest's warnings summary -- which `pytest -q` prints, so that it is on
    record in the driver's GPU test log; more than two replaced draws fail."""
    import warnings
    record_property('draws_replaced', len(tried))
    if tried:
        warnings.warn(f'{what}: draws_replaced = {len(tried)} of 5 (borderline fragment flips, verified: {[(t[0], t[3]) for t in tried]})')
    assert len(tried) <= 2, tried


@pytest.mark.parametrize('epoch,decimate', [(0, True), (800, False), (1600, False)])
def test_config1_training_iteration_matches_oracle(epoch, decimate, record_property):
    """BASELINE configs[0] exactly: 4 views, 100x75 (W x H), 4 superquadric blocks (the model has no cube primitive),
    faces_per_pixel 4, configs/dtu/default.yml otherwise (256^2 textures, opacity noise, kill_blocks, decimation until 750), in each
    of the three training phases: every loss term and the gradient of every parameter tensor within 1e-4.  A draw of the perturbed
    parameters for which an ulp of libm difference flips a borderline fragment (see _fragment_flips: verified, not assumed) is
    replaced by the next draw -- at this size most draws have none (at 300x400 with 10 layers nearly every draw has a few among its
    2.4 M slots, which is why the larger configs are compared from identical vertices down, see the next tests)."""
    shape = C1
    tried = []
    for seed in (12, 13, 14, 15, 11):       # (seed 11 has a verified borderline flip in every phase: kept, as the last resort)
        errs, flips = _iteration(shape, seed, epoch, decimate)
        worst = max(errs, key=errs.get)
        if errs[worst] < REL:
            # how many draws had to be replaced is part of the result: a regression in the flip frequency (a kernel that moves more
            # borderline fragments than an ulp of libm explains) shows up here long before all five draws are used up
            _report_draws(record_property, f'config 1, epoch {epoch}', tried)
            return
        assert flips > 0, f'seed {seed}: {worst} off by {errs[worst]:.2e} although the fragment lists are identical'
        tried.append((seed, worst, errs[worst], flips))
    pytest.fail(f'no parameter draw without a borderline fragment flip: {tried}')


@pytest.mark.parametrize('epoch,decimate', [(0, True), (800, False), (1600, False)])
def test_config1_c_step_matches_oracle(epoch, decimate, record_property):
    """The same bar for the entry point the bench measures: BASELINE configs[0] through `ShardedTrainStep` -> `dbw_train_step_run` (one
    C-ABI call, fuse 127: the env layer inside the fg pass, the fused prologue / set-up / bins / tails, the loss values reduced by the
    step) against `OracleDBW` DIRECTLY, in the three training phases: every loss term and the gradient of every parameter tensor within
    1e-4 -- no chain of pairwise comparisons in between."""
    tried = []
    for seed in (12, 13, 14, 15, 11):       # (seed 11 has a verified borderline flip in every phase: kept, as the last resort)
        errs, flips = _iteration(C1, seed, epoch, decimate, c_step=True)
        worst = max(errs, key=errs.get)
        if errs[worst] < REL:
            _report_draws(record_property, f'config 1 through the C step, epoch {epoch}', tried)
            return
        assert flips > 0, f'seed {seed}: {worst} off by {errs[worst]:.2e} although the fragment lists are identical'
        tried.append((seed, worst, errs[worst], flips))
    pytest.fail(f'no parameter draw without a borderline fragment flip: {tried}')


@pytest.mark.parametrize('criterion,tv_type,epoch,decimate', [('l1', 'l2', 0, True), ('huber', 'l1', 800, False)])
def test_config1_with_the_registrys_other_criteria_matches_oracle(criterion, tv_type, epoch, decimate):
    """The config keys loss.name / loss.tv_type beyond the defaults (loss.py:11-24: l1, huber; loss.py:43-47: l1, l2 -- no shipped config sets
    them): the rendered image comes from the HIP kernels, the criterion and the TV norm run in torch on it and on the prepared maps, autograd
    hands their gradients to the HIP backward.  BASELINE configs[0] at the oracle's vertices (the l1 criterion's gradient is a sign: a
    pixel an ulp on the other side of its target flips it, so nothing upstream may differ): every loss term and every gradient at 1e-4."""
    errs, _ = _iteration(C1, 12, epoch, decimate, device_vertices=True, criterion=criterion, tv_type=tv_type)
    worst = max(errs, key=errs.get)
    assert errs[worst] < REL, (worst, errs[worst], {k: v for k, v in errs.items() if v >= REL})


# ---------------------------------------------------------------------------------------------------------------------
# configs/bmvs/gundam_50.yml: the one shipped configuration outside K <= 16 / background maps x 1
# ---------------------------------------------------------------------------------------------------------------------
GUNDAM = (288, 384, 50, 128, 25, 2)   # configs/bmvs/gundam_50.yml:8-14 (n_blocks 50, txt_size 128, txt_bkg_upscale 2, faces_per_pixel 25), 2 views


@pytest.mark.parametrize('epoch,decimate', [(0, True), (800, False)])
def test_gundam_50_c_step_matches_oracle(epoch, decimate, record_property):
    """The shipped configuration with 50 blocks, 25 faces per pixel (the KMAX = 25 training instantiation of the fused forward), 128^2
    block textures and sky / ground maps at twice that resolution (txt_bkg_upscale 2: dbw.py:117-119), through the model and the one-call
    C step against `OracleDBW`: every loss term and the gradient of every parameter tensor within 1e-4.  The oracle is evaluated at the
    device's vertices (device_vertices: same fp32 values, its own graph behind them), so the rasteriser is bit-exact and nothing may
    differ beyond rounding -- no draw is replaced here, the first one has to hold."""
    errs, _ = _iteration(GUNDAM, 11, epoch, decimate, c_step=True, bkg_upscale=2, device_vertices=True)
    worst = max(errs, key=errs.get)
    assert errs[worst] < REL, (worst, errs[worst], {k: v for k, v in errs.items() if v >= REL})


def test_gundam_50_autograd_path_matches_oracle():
    errs, _ = _iteration(GUNDAM, 12, 0, True, c_step=False, bkg_upscale=2, device_vertices=True)
    worst = max(errs, key=errs.get)
    assert errs[worst] < REL, (worst, errs[worst], {k: v for k, v in errs.items() if v >= REL})


# ---------------------------------------------------------------------------------------------------------------------
# config 2 (= the per-rank workload of config 3): both render passes at full size from IDENTICAL vertices / maps / opacities
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('phase', ['coarse+decimated', 'coarse', 'fine'])
def test_config2_render_passes_match_oracle_at_full_size(phase):
    """400x300, 10 blocks, faces_per_pixel 10, 256^2 textures, 2 of the 49 views (what the CPU oracle finishes in seconds): the fg
    pass of each training phase and the env pass, image and every gradient (vertices, texture maps, opacities) within 1e-4 of the
    oracle when both start from the same fp32 scene -- then the rasteriser is bit-exact and no fragment can differ."""
    from test_gpu_parity import _render_both
    H, W, nb, ts, fpp, V = 300, 400, 10, 256, 10, 2
    coarse, decimate = phase != 'fine', phase == 'coarse+decimated'
    m = O.OracleDBW((H, W), n_blocks=nb, txt_size=ts, faces_per_pixel=fpp, seed=227391)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        m.p['sq_eps'].add_(torch.randn(nb, 2, generator=g))
        m.p['alpha_logit'].add_(torch.randn(nb, generator=g) + 1.0)
        m.p['R_6d_ground'].add_(torch.randn(1, 6, generator=g) * 0.03)
        m.p['T'].mul_(0.7)
    R, T, Km = O.synthetic_cameras(V, R_world=m.R_world[0])
    noise = torch.randn(nb, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        blocks = m.build_blocks(True, coarse, decimate, noise, filter_transparent=not coarse)
        env = m.build_env(True, decimate)
        alpha = None if not coarse else m._alpha.detach().repeat_interleave(m.BNF)
    res = _render_both(blocks, R, T, Km[0], H, W, 1e-4 if coarse else 5e-6, fpp, True, alpha)
    assert set(res) == ({'image', 'g_maps', 'g_verts', 'g_alpha'} if coarse else {'image', 'g_maps', 'g_verts'})
    for k, (a, b) in res.items():
        assert rel_err(a, b) < REL, f'fg {k}: rel err {rel_err(a, b)}'
        if k in ('image', 'g_alpha'):           # element-wise too: every pixel of the image, every opacity gradient
            n_off, worst = off_entries(a, b)
            assert n_off == 0, f'fg {k}: {n_off} entries off, worst {worst:.2f} x the element-wise bar'
    if phase != 'coarse':                       # the env pass does not depend on coarse / fine, only on the decimation
        res = _render_both(env, R, T, Km[0], H, W, 0.0, 1, False, None)
        for k, (a, b) in res.items():
            assert rel_err(a, b) < REL, f'env {k}: rel err {rel_err(a, b)}'
            if k == 'image':
                n_off, worst = off_entries(a, b)
                assert n_off == 0, f'env {k}: {n_off} entries off, worst {worst:.2f} x the element-wise bar'


# ---------------------------------------------------------------------------------------------------------------------
# configs 4 and 5: properties + oracle indices + gradient-path agreement at full size
# ---------------------------------------------------------------------------------------------------------------------
def _check_fragment_properties(cl, p2f, zbuf, bary, dists, cfg):
    K = p2f.shape[-1]
    valid = p2f >= 0
    assert 0.02 < valid[..., 0].float().mean() < 0.95                                # blocks cover part of the image
    z = torch.where(valid, zbuf, torch.full_like(zbuf, float('inf')))
    assert torch.all(z[..., 1:] >= z[..., :-1])                                     # front-to-back order
    assert torch.all(valid[..., 1:] <= valid[..., :-1])                             # no holes in the lists
    torch.testing.assert_close(bary[valid].sum(-1), torch.ones(int(valid.sum()), device=DEV), rtol=0, atol=2e-6)
    assert torch.all(dists[valid] < cfg.blur)
    srt = torch.where(valid, p2f, -torch.arange(1, K + 1, device=DEV, dtype=torch.int32).expand_as(p2f)).sort(-1)[0]
    assert torch.all(srt[..., 1:] != srt[..., :-1])                                 # a face appears once per pixel
    first = cl['first_idx'].long().view(-1, 1, 1, 1)
    assert torch.all(((p2f - first) < cl['num_faces'].view(-1, 1, 1, 1))[valid]) and torch.all((p2f - first)[valid] >= 0)
    return valid


@pytest.mark.parametrize('name,H,W,nb,ts,fpp,V', [('c4', 576, 768, 20, 256, 16, 3), ('c5', 1080, 1920, 50, 512, 16, 2)])
def test_large_configs_properties_oracle_indices_and_gradient_paths(name, H, W, nb, ts, fpp, V, monkeypatch):
    m = O.OracleDBW((H, W), n_blocks=nb, txt_size=ts, faces_per_pixel=fpp, seed=227391)
    with torch.no_grad():
        m.p['T'].mul_(0.8)
    R, T, Km = O.synthetic_cameras(V, R_world=m.R_world[0])
    with torch.no_grad():
        scene = m.build_blocks(False, True, False, None, kill_blocks=False)
    F_ = scene['faces'].shape[0]
    assert F_ == nb * 80
    args = (R.to(DEV), T.to(DEV), Km[0].to(DEV))
    ps = _packed(scene)
    cfg = ops.RenderCfg(H, W, fpp, 1e-4, 0.001, True, True, F_)
    # ---- fragments: determinism + structural properties + one full view against the oracle (indices, depths, distances) ----
    cl, p2f, zbuf, bary, dists = ops.render_fragments(ps.verts, ps.faces, *args, cfg)
    cl2, p2f2, zbuf2, _, dists2 = ops.render_fragments(ps.verts, ps.faces, *args, cfg)
    assert torch.equal(p2f, p2f2) and torch.equal(zbuf, zbuf2) and torch.equal(dists, dists2)
    valid = _check_fragment_properties(cl, p2f, zbuf, bary, dists, cfg)
    assert valid[..., fpp - 1].any()                                               # some lists are full: truncation at K is exercised
    with torch.no_grad():         # the oracle's camera transform + clipping + rasteriser for view 0 (no texture sampling: indices only)
        ndc = O.transform_to_ndc(scene['verts'], R[:1], T[:1], Km[0], eps=1e-8)
        fv = ndc[:, scene['faces']].reshape(F_, 3, 3)
        cl_o = O.clip_faces(fv, torch.zeros(1, dtype=torch.int64), torch.full((1,), F_, dtype=torch.int64), 0.001, True)
        p2f_c, zbuf_o, bary_c, dists_o = O.rasterize(cl_o['face_verts'], cl_o['first_idx'], cl_o['num_faces'], cl_o['neighbor'], (H, W),
                                                     cfg.blur, fpp, True, True, 64)
        p2f_o, _ = O.convert_clipped_to_original(p2f_c, bary_c, cl_o)
    c2o = cl['c2o'].view(-1).long()
    orig = torch.where(p2f[:1] >= 0, c2o[p2f[:1].clamp(min=0).long()], torch.full_like(p2f[:1], -1).long())
    assert torch.equal(orig.cpu(), p2f_o)
    assert torch.equal(dists[:1].cpu(), dists_o) and torch.equal(zbuf[:1].cpu(), zbuf_o)
    del cl2, p2f2, zbuf2, dists2, p2f_c, zbuf_o, bary_c, dists_o, p2f_o
    # ---- images: the fused training path (uv-fragments) against the operator-level kernels, value and gradients ----
    shapes = [tuple(t.shape[:2]) for t in scene['maps']]
    alpha = (torch.rand(nb, generator=torch.Generator().manual_seed(3)) * 0.8 + 0.1).repeat_interleave(F_ // nb).to(DEV)
    g_img = torch.rand(V, 4, H, W, generator=torch.Generator().manual_seed(4)).to(DEV)
    g_img[:, 3] = 0                                    # colour gradients only: then sum(grad_maps[c]) = sum_pix g_c * sum_k T_k a_k
    out = {}
    for tag, fused, bins in (('fused+bins', True, True), ('fused+atomics', True, False), ('operators', False, False)):
        monkeypatch.setattr(ops, 'FUSED_FORWARD', fused)
        monkeypatch.setattr(ops, 'FUSED_BACKWARD', fused)
        q = _packed(scene)
        q.maps.requires_grad_(True)
        q.verts.requires_grad_(True)
        fa = alpha.clone().requires_grad_(True)
        bb, bi, nbins = PackedScene.describe_bins(shapes, DEV)
        c = ops.RenderCfg(H, W, fpp, 1e-4, 0.001, True, True, F_, lds_aggregate=False, texbins=(bb, bi, nbins) if bins else None)
        img = ops.render_scene(q.verts, q.maps, fa, q.faces, *args, q.face_uvs, q.face_map, q.map_desc, None, c)
        (img * g_img).sum().backward()
        out[tag] = (img.detach(), q.maps.grad.clone(), q.verts.grad.clone(), fa.grad.clone())
        del img, q
        torch.cuda.empty_cache()
    img = out['fused+bins'][0]
    assert torch.isfinite(img).all() and img[:, 3].min() >= 0 and img[:, 3].max() <= 1 + 1e-6
    assert torch.equal(img, out['fused+atomics'][0])
    assert rel_err(img, out['operators'][0]) < 1e-5          # 12 B payloads: b2 = 1 - b0 - b1 (raster_math.h: PAY3)
    for i, what in ((1, 'maps'), (2, 'verts'), (3, 'alpha')):
        assert rel_err(out['fused+bins'][i], out['fused+atomics'][i]) < 2e-5, (what, rel_err(out['fused+bins'][i], out['fused+atomics'][i]))
        assert rel_err(out['fused+bins'][i], out['operators'][i]) < REL, (what, rel_err(out['fused+bins'][i], out['operators'][i]))
    # conservation law of the texel scatter: bilinear weights sum to one, so the texel gradients of channel c add up to
    # sum_pix g_c * (1 - T_K) = sum_pix g_c * image alpha
    expect = (g_img[:, :3] * img[:, 3:4]).sum(dim=(0, 2, 3)).double()
    got = out['fused+bins'][1].view(-1, 3).double().sum(0)
    assert torch.allclose(got, expect, rtol=2e-4), (got, expect)
    # zero opacity -> empty image
    q = _packed(scene)
    img0 = ops.render_scene(q.verts, q.maps, torch.zeros(F_, device=DEV), q.faces, *args, q.face_uvs, q.face_map, q.map_desc, None, cfg)
    assert torch.all(img0 == 0)
