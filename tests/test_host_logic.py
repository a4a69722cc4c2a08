"""CPU: host-side logic of the product package (topology, containers, model construction, config handling) -- everything
that does not launch a kernel.  The product's topology generators are checked against the oracle's independent ones."""
import numpy as np
import pytest
import torch
import yaml

import oracle as O
import dbw_amd
from dbw_amd import mesh as M
from dbw_amd.structures import Meshes, PackedScene, TexturesUV, join_meshes_as_scene


def test_topology_generators_agree_with_oracle():
    for level in (0, 1, 2):
        a, b = O.get_icosphere(level)
        c, d = M.get_icosphere(level)
        assert torch.equal(a, c) and torch.equal(b, d)
    assert torch.equal(O.get_icosphere(2, True)[1], M.get_icosphere(2, True)[1])
    for level in (1, 2):
        fa, ua = O.get_icosphere_uvs(level)
        fb, ub = M.get_icosphere_uvs(level, True, True)
        assert torch.equal(fa, fb) and torch.equal(ua, ub)
    v, f = O.get_plane()
    v2, f2 = M.get_plane()
    for _ in range(3):
        v, f = O.subdivide(v, f)
        v2, f2 = M.subdivide_mesh(v2, f2)
    assert torch.equal(v, v2) and torch.equal(f, f2)
    assert torch.equal(O.world_rotation(115, 0, 0), M.world_rotation(115, 0, 0))
    assert torch.equal(O.world_rotation(130, 50, 0), M.world_rotation(130, 50, 0))
    torch.manual_seed(1)
    r1 = O.random_rotations(4)
    torch.manual_seed(1)
    r2 = M.random_rotations(4)
    assert torch.equal(r1, r2)
    d6 = torch.randn(5, 6)
    assert torch.equal(O.rotation_6d_to_matrix(d6), M.rotation_6d_to_matrix(d6))
    C = torch.randn(3, 3) + 2
    for a, b in zip(O.look_at_cameras(C), M.look_at_view_transform(C)):
        assert torch.equal(a, b)


def _cfg():
    cfg = yaml.safe_load('''
model:
  name: dbw
  mesh: {n_blocks: 10, S_world: 0.5, R_world: [115, 0, 0], txt_size: 256}
  renderer: {faces_per_pixel: 10, cameras: {name: perspective}, detach_bary: True, z_clip: 0.001}
  rend_optim: {coarse_learning: 1500, decimate_txt: 750, decimate_factor: 8, kill_blocks: True, decouple_rendering: True, opacity_noise: True}
  loss: {rgb_weight: 1, perceptual_weight: 0, parsimony_weight: 0.01, tv_weight: 0.1, overlap_weight: 1}
''')
    return cfg


def test_model_matches_reference_state_layout_and_oracle_init():
    """Parameter names/shapes of dbw.py:84,99-119 + buffers and same-seed initialisation (draw order)."""
    torch.manual_seed(227391)
    m = dbw_amd.create_model(_cfg(), (300, 400))
    shapes = {k: tuple(v.shape) for k, v in m.named_parameters()}
    assert shapes == {'sq_eps': (10, 2), 'R_6d_ground': (1, 6), 'T_ground': (1, 3), 'S': (10, 3), 'R_6d': (10, 6), 'T': (10, 3),
                      'alpha_logit': (10,), 'texture_bkg': (1, 256, 256, 3), 'texture_ground': (1, 256, 256, 3),
                      'textures': (10, 256, 256, 3)}
    assert set(m.state_dict()) == set(shapes) | {'R_world', 'T_world', 'bkg_verts_uvs', 'ground_verts_uvs', 'sq_eta', 'sq_omega',
                                                   'block_faces_uvs', 'block_verts_uvs'}
    assert m.loss_names == ['loss_rgb', 'loss_parsimony', 'loss_tv', 'loss_overlap', 'loss_total']
    assert m.BNF == 80 and m.txt_padding == (0, 23) and m.env_n_faces == 448 and m.blocks_n_faces == 800
    o = O.OracleDBW((300, 400), seed=227391)
    for k, v in o.p.items():
        assert torch.equal(v.detach(), getattr(m, k).detach()), k
    assert torch.equal(m.block_verts_uvs, o.block_verts_uvs) and torch.equal(m.R_world, o.R_world)
    # texture params go to their own lr group by name (optimizer.py:11-13)
    assert [n for n, _ in m.named_parameters() if n.startswith('texture')] == ['texture_bkg', 'texture_ground', 'textures']
    # milestones (dbw.py:457-462)
    assert m.is_live('coarse_learning') and m.is_live('decimate_txt')
    m.set_cur_epoch(750)
    assert m.is_live('coarse_learning') and not m.is_live('decimate_txt')
    m.set_cur_epoch(1499)
    m.step()
    assert not m.is_live('coarse_learning')


def test_config_key_guards_like_the_reference():
    cfg = _cfg()
    cfg['model']['mesh']['unknown_key'] = 1
    with pytest.raises(AssertionError):                      # dbw.py:71 `assert len(kwargs) == 0`
        dbw_amd.create_model(cfg, (8, 8))
    cfg = _cfg()
    cfg['model']['renderer']['shading_type'] = 'phong'
    with pytest.raises(NotImplementedError):
        dbw_amd.create_model(cfg, (8, 8))
    with pytest.raises(KeyError):
        dbw_amd.create_model({'model': {'name': 'nope'}}, (8, 8))


def test_render_path_refuses_to_run_without_gpu():
    """There is no CPU fallback: CPU tensors are rejected loudly."""
    m = dbw_amd.create_model(_cfg(), (16, 16))
    R, T, K = O.synthetic_cameras(1)
    with pytest.raises(RuntimeError, match='GPU'):
        m(dict(imgs=torch.rand(1, 3, 16, 16), R=R, T=T, K=K), None)


def test_meshes_join_and_packed_scene_layout():
    """join order = list order, faces offset, packed face id of copy b = b*F + j."""
    v1, f1 = M.get_icosphere(0)
    v2, f2 = M.get_plane()
    t1 = TexturesUV(torch.rand(1, 4, 6, 3), f1, torch.rand(12, 2))
    t2 = TexturesUV(torch.rand(1, 8, 8, 3), f2, torch.rand(4, 2), circular_pad=(1, 2))
    scene = join_meshes_as_scene([Meshes(v1, f1, t1), Meshes(v2, f2, t2)])
    assert len(scene) == 1 and len(scene.extend(5)) == 5
    verts, faces = scene.get_mesh_verts_faces(0)
    assert verts.shape == (16, 3) and faces.shape == (22, 3) and faces[20:].min() >= 12
    ps = PackedScene.from_meshes(scene.extend(3))
    assert ps.faces.dtype == torch.int32 and ps.face_uvs.shape == (22, 3, 2)
    assert ps.face_map.tolist() == [0] * 20 + [1] * 2
    assert ps.map_desc.tolist() == [[0, 4, 6, 0, 0, 0, 2, 0], [72, 8, 8, 1, 2, 0, 0, 0]]       # (row 0, 7th int: the row count)
    assert PackedScene.describe_maps([(16, 16)] * 2, [(0, 3)] * 2, 'cpu', shift=3)[0].tolist() == [[0, 16, 16, 0, 3, 3, 2, 0], [12, 16, 16, 0, 3, 3, 0, 0]]
    assert ps.maps.numel() == 72 + 192
    batch = Meshes(torch.rand(3, 12, 3), f1[None].expand(3, -1, -1), TexturesUV(torch.rand(3, 4, 4, 3), f1, torch.rand(12, 2)))
    js = join_meshes_as_scene(batch)
    assert js.get_mesh_verts_faces(0)[1].max() == 35 and len(js.textures.maps) == 3


def test_multistep_lr_matches_reference_scheduler(golden_dir):
    """dbw_amd.trainer.MultiStepLR against LR traces of the REAL src/scheduler.py MultiStepLR + src/optimizer.py grouping
    (tests/golden/lr_schedule.npz), including the warm-up quirk and list/scalar gamma."""
    import os
    import numpy as np
    from dbw_amd.trainer import MultiStepLR
    g = np.load(os.path.join(golden_dir, 'lr_schedule.npz'))
    for tag, kwargs in [('dtu', dict(gamma=[0.1, 0.1], milestones=[1700])),
                        ('warm', dict(gamma=[0.5, 0.1], milestones=[5, 9, 9], warmup=3)),
                        ('scalar_gamma', dict(gamma=0.3, milestones=[2, 4]))]:
        sch = MultiStepLR([5.0e-3, 5.0e-2], **kwargs)
        trace = [sch.get_last_lr()]
        for _ in range(g[tag].shape[0] - 1):
            trace.append(list(sch.step()))
        np.testing.assert_allclose(np.array(trace), g[tag], rtol=1e-12, atol=0)


def test_multistep_lr_resumes_from_a_torch_scheduler_state_past_the_milestone(golden_dir):
    """A checkpoint of the REFERENCE's trainer carries a torch scheduler state ('last_epoch', '_last_lr', '_step_count', ...), not this
    trainer's {'last_epoch', 'lrs'}: loading it past the milestone must continue at the DECAYED rates (from '_last_lr', or replayed
    from the milestones when that key is missing too) -- not at the base values, i.e. ten times too high."""
    import os
    import numpy as np
    from dbw_amd.trainer import MultiStepLR
    g = np.load(os.path.join(golden_dir, 'lr_schedule.npz'))
    for tag, kwargs in [('dtu', dict(gamma=[0.1, 0.1], milestones=[1700])), ('warm', dict(gamma=[0.5, 0.1], milestones=[5, 9, 9], warmup=3))]:
        trace = g[tag]
        for epoch in (0, 2, 4, 6, 10, trace.shape[0] - 2):
            if epoch >= trace.shape[0] - 1:
                continue
            for state in ({'last_epoch': epoch, '_last_lr': list(trace[epoch]), '_step_count': epoch + 1}, {'last_epoch': epoch}):
                sch = MultiStepLR([5.0e-3, 5.0e-2], **kwargs)
                sch.load_state_dict(state)
                np.testing.assert_allclose(sch.get_last_lr(), trace[epoch], rtol=1e-12)
                np.testing.assert_allclose(sch.step(), trace[epoch + 1], rtol=1e-12)
        own = MultiStepLR([5.0e-3, 5.0e-2], **kwargs)
        for _ in range(7):
            own.step()
        twin = MultiStepLR([5.0e-3, 5.0e-2], **kwargs)
        twin.load_state_dict(own.state_dict())
        assert twin.get_last_lr() == own.get_last_lr() and twin.step() == own.step()


def test_lazy_losses_behave_like_the_dict_they_stand_for():
    """native_step.LazyLosses reduces the loss values on first access: every way of reading a dict has to trigger that, not only
    __getitem__ (get / copy / repr / pickling used to see the empty underlying dict)."""
    import copy
    import pickle
    from dbw_amd.native_step import LazyLosses
    def make():
        vals = torch.zeros(8)
        vals[1], vals[2], vals[3] = 0.25, 0.5, 0.125
        return LazyLosses(vals, torch.tensor([1.0, 2.0, 3.0]), 0.5, ['rgb', 'parsimony', 'tv', 'overlap'])
    want = {'rgb': 3.0, 'parsimony': 0.25, 'tv': 0.5, 'overlap': 0.125, 'total': 3.875}
    assert float(make().get('rgb')) == 3.0 and make().get('nope', 7) == 7
    assert {k: float(v) for k, v in make().copy().items()} == want
    assert 'rgb' in repr(make()) and len(make()) == 5 and 'total' in make()
    assert {k: float(v) for k, v in pickle.loads(pickle.dumps(make())).items()} == want
    assert {k: float(v) for k, v in copy.deepcopy(make()).items()} == want


def test_camera_ingest_round_trip_and_pixel_consistency():
    """N3: P = K_cv [R|t] -> (K_ndc, R, T) -> the render path's NDC projection lands on the same pixels as P."""
    import numpy as np
    from dbw_amd.cameras import opencv_KRT_from_proj, pytorch3d_KRT_from_proj
    rng = np.random.RandomState(0)
    H, W = 300, 400
    for _ in range(5):
        A = rng.randn(3, 3)
        Q, _ = np.linalg.qr(A)
        R_w2c = Q * np.sign(np.linalg.det(Q))
        C = rng.randn(3) * 0.5 + np.array([0, 0, -3.0])
        Kcv = np.array([[720 + 50 * rng.rand(), 0.0, W / 2 + 10 * rng.randn()], [0, 715.0, H / 2 + 10 * rng.randn()], [0, 0, 1]])
        P = Kcv @ np.concatenate([R_w2c, (-R_w2c @ C)[:, None]], 1) * 3.7          # arbitrary projective scale
        K4, R_c2w, Cc = opencv_KRT_from_proj(P)
        np.testing.assert_allclose(K4[:3, :3], Kcv / Kcv[2, 2], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(R_c2w, R_w2c.T, atol=1e-5)
        np.testing.assert_allclose(Cc, C, atol=1e-4)
        Kp, Rp, Tp = pytorch3d_KRT_from_proj(P, (H, W))
        X = torch.from_numpy(R_w2c.T @ (rng.rand(3, 20) * [[1.0], [0.8], [2.0]] + [[-0.5], [-0.4], [1.5]]) + C[:, None]).float().T
        ndc = O.transform_to_ndc(X, Rp[None], Tp[None], Kp)[0]
        uvw = (torch.from_numpy(P).float() @ torch.cat([X, torch.ones(20, 1)], 1).T).T
        u, v = uvw[:, 0] / uvw[:, 2], uvw[:, 1] / uvw[:, 2]
        s = min(H, W) / 2
        torch.testing.assert_close(W / 2 - ndc[:, 0] * s, u, rtol=1e-4, atol=5e-2)        # NDC +X points left, +Y up
        torch.testing.assert_close(H / 2 - ndc[:, 1] * s, v, rtol=1e-4, atol=5e-2)
        assert torch.all(ndc[:, 2] > 0)


def test_idr_cameras_npz_layout_round_trip(tmp_path):
    """N3: the IDR cameras.npz layout of the DTU / BlendedMVS scenes (dtu.py:42-44): world_mat_i @ scale_mat_i -> (K, R, T); a point
    of the normalised frame lands on the pixel its projection matrix names."""
    import numpy as np
    from dbw_amd.cameras import load_idr_cameras
    rng = np.random.RandomState(3)
    H, W, N = 1200, 1600, 4
    arrays, Ps = {}, []
    S = np.diag([350.0, 350.0, 350.0, 1.0]); S[:3, 3] = [10.0, -25.0, 600.0]                  # scale_mat: normalised -> world (mm)
    for i in range(N):
        Q, _ = np.linalg.qr(rng.randn(3, 3))
        R_w2c = Q * np.sign(np.linalg.det(Q))
        C = np.array([10.0, -25.0, 600.0]) + R_w2c.T @ np.array([0.0, 0.0, -900.0]) + rng.randn(3) * 30
        Kcv = np.array([[2892.3, 0.0, 823.2], [0.0, 2883.2, 619.1], [0.0, 0.0, 1.0]])
        Wm = np.eye(4); Wm[:3] = Kcv @ np.concatenate([R_w2c, (-R_w2c @ C)[:, None]], 1)
        arrays[f'world_mat_{i}'], arrays[f'scale_mat_{i}'] = Wm, S
        arrays[f'world_mat_inv_{i}'] = np.linalg.inv(Wm)                                      # present in the real files: must not be counted
        Ps.append((Wm @ S)[:3, :4])
    path = tmp_path / 'cameras.npz'
    np.savez(path, **arrays)
    cams = load_idr_cameras(str(path), (H, W))
    assert cams['K'].shape == (N, 4, 4) and cams['R'].shape == (N, 3, 3) and cams['T'].shape == (N, 3)
    np.testing.assert_allclose(cams['scale_mat'].numpy(), S, rtol=1e-6)
    X = torch.from_numpy(rng.rand(50, 3) * 1.2 - 0.6).float()                                 # points of the unit normalised scene
    for i in range(N):
        ndc = O.transform_to_ndc(X, cams['R'][i][None], cams['T'][i][None], cams['K'][i])[0]
        uvw = (torch.from_numpy(Ps[i]).float() @ torch.cat([X, torch.ones(50, 1)], 1).T).T
        u, v = uvw[:, 0] / uvw[:, 2], uvw[:, 1] / uvw[:, 2]
        s = min(H, W) / 2
        torch.testing.assert_close(W / 2 - ndc[:, 0] * s, u, rtol=1e-4, atol=0.2)
        torch.testing.assert_close(H / 2 - ndc[:, 1] * s, v, rtol=1e-4, atol=0.2)
        assert torch.all(ndc[:, 2] > 0)


def test_eval_metrics_match_reference_ssim_and_psnr(golden_dir):
    """metrics.ssim (separable 1-D Gaussian filtering) and mse2psnr against the reference's own SSIMLoss / mse2psnr outputs
    (tests/golden/ssim.npz, generated by make_golden.py from src/model/loss.py)."""
    import os
    from dbw_amd import metrics
    g = np.load(os.path.join(golden_dir, 'ssim.npz'))
    a, b = torch.from_numpy(g['img1']), torch.from_numpy(g['img2'])
    for pad in (0, 1):
        m = metrics.ssim_map(a, b, padding=bool(pad))
        assert m.shape == g[f'ssim_map_pad{pad}'].shape
        assert np.abs(m.numpy() - g[f'ssim_map_pad{pad}']).max() < 2e-5
        assert np.abs((1 - metrics.ssim(a, b, padding=bool(pad))).numpy() - g[f'one_minus_ssim_pad{pad}']).max() < 1e-6
    assert np.allclose(metrics.mse2psnr(torch.from_numpy(g['mse'])).numpy(), g['psnr'], rtol=1e-6)
    assert abs(metrics.ssim(a, a).mean().item() - 1.0) < 1e-6
    meter = metrics.AverageMeter()
    meter.update(torch.tensor(2.0), N=3); meter.update(4.0, N=1)
    assert abs(meter.avg - 2.5) < 1e-12


def test_packed_scene_join_rebases_faces_and_maps():
    from dbw_amd.structures import PackedScene
    d1, n1 = PackedScene.describe_maps([(4, 4)], [(0, 0)], 'cpu')
    d2, n2 = PackedScene.describe_maps([(2, 2), (2, 4)], [(1, 1), (0, 0)], 'cpu')
    s1 = PackedScene(torch.rand(5, 3), torch.tensor([[0, 1, 2], [2, 3, 4]], dtype=torch.int32), torch.rand(2, 3, 2),
                     torch.zeros(2, dtype=torch.int32), d1, torch.rand(n1))
    s2 = PackedScene(torch.rand(4, 3), torch.tensor([[0, 1, 3]], dtype=torch.int32), torch.rand(1, 3, 2), torch.ones(1, dtype=torch.int32),
                     d2, torch.rand(n2))
    j = PackedScene.join([s1, s2])
    assert j.verts.shape == (9, 3) and j.faces.tolist() == [[0, 1, 2], [2, 3, 4], [5, 6, 8]] and j.face_map.tolist() == [0, 0, 2]
    assert j.map_desc[:, 0].tolist() == [0, n1, n1 + 12] and j.map_desc[1, 3:5].tolist() == [1, 1]
    assert torch.equal(j.maps, torch.cat([s1.maps, s2.maps])) and j.faces.dtype == torch.int32
    assert j.map_desc[:, 6].tolist() == [3, 0, 0]                 # the row count of the JOINED table, in its row 0 only


def test_fancy_cmap_structure():
    """utils/plot.py:77-87 restated without seaborn / matplotlib: gold at 0, a 256-entry table, hues in palette order."""
    import numpy as np
    from dbw_amd import mesh as M
    cmap = M.get_fancy_cmap()
    c = cmap(np.array([0.0, 1.0, 0.5, 1.0 / 3]))
    assert c.shape == (4, 3) and np.allclose(c[0], (1.0, 0.8431372549, 0.0)) and (c >= 0).all() and (c <= 1).all()
    fine = cmap(np.linspace(0, 1, 1000))
    assert len(np.unique(fine.round(6), axis=0)) == 256                      # matplotlib's default quantisation
    assert np.abs(np.diff(fine, axis=0)).max() < 0.12                        # piecewise-linear, no jumps
    import torch
    assert np.allclose(cmap(torch.tensor([0.25, 0.75])), cmap(np.array([0.25, 0.75])))


def test_lpips_vgg_architecture_and_weight_layout():
    """N4: the LPIPS-VGG criterion as a plain-torch module (parity unpinned: no weights offline).  Structure checks: the state layout a user
    has to bring (torchvision vgg16.features + lpips linear heads), refusal to run without weights, zero distance of an image to
    itself, non-negativity with non-negative heads, gradient to the reconstruction only."""
    from dbw_amd.lpips_vgg import LPIPSVGG, _VGG16_CONVS, _CHANNELS
    net = LPIPSVGG()
    a, b = torch.rand(2, 3, 32, 48), torch.rand(2, 3, 32, 48)
    with pytest.raises(RuntimeError, match='no weights'):
        net(a, b)
    gen = torch.Generator().manual_seed(0)
    vgg = {}
    for i, cin, cout in _VGG16_CONVS:
        vgg[f'{i}.weight'] = torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
        vgg[f'{i}.bias'] = torch.zeros(cout)
    lin = {f'lin{k}.model.1.weight': torch.rand(1, c, 1, 1, generator=gen) for k, c in enumerate(_CHANNELS)}
    net.load_weights(vgg, lin)
    assert [t.shape[1] for t in net.features(a)] == _CHANNELS and [t.shape[2] for t in net.features(a)] == [32, 16, 8, 4, 2]
    assert float(net(a, a)) == 0.0
    rec = b.clone().requires_grad_(True)
    d = net(a, rec)
    assert float(d) > 0
    d.backward()
    assert rec.grad is not None and float(rec.grad.abs().sum()) > 0 and all(not p.requires_grad for p in net.parameters())
    # plugs into the model as the perceptual callable (src/model/loss.py:39-40: mean over the batch)
    assert d.dim() == 0
