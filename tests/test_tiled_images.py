"""The 8x8-tile planar image layout (include/dbw_hip.h: image_layout 1) the training step keeps its intermediate images in."""
import pytest
import torch

from dbw_amd import ops


def test_tile_untile_round_trip_and_kernel_addressing():
    for H, W in ((300, 400), (75, 100), (48, 64), (37, 29)):
        x = torch.rand(3, 4, H, W)
        t = ops.tile_image(x)
        ty, tx = (H + 7) // 8, (W + 7) // 8
        assert t.shape == (3, ty, tx, 4, 64) and torch.equal(ops.untile_image(t, H, W), x)
        for n, c, yi, xi in ((2, 1, H - 1, W - 3), (0, 3, 0, 0), (1, 0, 9, 17)):          # csrc/shade_common.h: img_addr
            flat = ((n * ty + yi // 8) * tx + xi // 8) * 4 * 64 + c * 64 + (((yi & 7) << 3) | (xi & 7))
            assert t.view(-1)[flat] == x[n, c, yi, xi]


@pytest.mark.gpu
def test_fused_passes_with_tiled_images_equal_the_plane_layout():
    """dbw_render_fwd_fused / dbw_render_fwd_fused_mse / dbw_render_bwd_fused with image_layout 1 against image_layout 0 on a ragged
    image size (tiles that stick out of the image on both axes): env-like hard pass image, loss partials, both gradient images and
    every gradient of the two backward passes are the same numbers in the other layout."""
    import dbw_amd
    import oracle as O
    from test_gpu_model import _dtu_like_cfg
    DEV = 'cuda'
    H, W, nb = 45, 70, 4
    torch.manual_seed(227391)
    model = dbw_amd.create_model(_dtu_like_cfg(nb, 32, 6), (H, W)).to(DEV).train()
    R, T, Km = O.synthetic_cameras(3, R_world=O.world_rotation(115, 0, 0))
    imgs = torch.rand(3, 3, H, W, generator=torch.Generator().manual_seed(2)).to(DEV)
    inp = dict(imgs=imgs, R=R.to(DEV), T=T.to(DEV), K=Km.to(DEV))
    model._noise_override = torch.zeros(nb, device=DEV)
    model(inp, None)                                               # cameras, scene caches
    B = 3
    with torch.no_grad():
        env, blocks = model.build_env_scene(), model.build_blocks_scene(filter_transparent=False)
    Kmat = model.renderer.cameras.K[0].contiguous()
    res = {}
    for tiled in (False, True):
        cfg_e = model.renderer_env._cfg(env.faces.shape[0], lds_aggregate=True, const_faces=getattr(env, 'const_faces', 0))
        cl_e = ops.project_clip(env.verts.detach(), env.faces, inp['R'], inp['T'], Kmat, cfg_e.eps, cfg_e.z_clip, cfg_e.persp)
        lay = ops.hard_layout(cfg_e, None, env.map_desc)
        p2f_e, bary_e, dists_e, img_e = ops._render_fwd_fused(cl_e['face_verts'].view(-1, 3, 3), cl_e, B, cfg_e, env.face_uvs, env.face_map, env.map_desc,
                                                              env.maps.detach(), None, model.renderer_env._bg, lay, img_tiled=tiled)
        cfg_f = model.renderer._cfg(blocks.faces.shape[0], lds_aggregate=True)
        cl_f = ops.project_clip(blocks.verts.detach(), blocks.faces, inp['R'], inp['T'], Kmat, cfg_f.eps, cfg_f.z_clip, cfg_f.persp)
        fa = model._alpha.detach().contiguous()
        target = ops.tile_image(imgs) if tiled else imgs
        p2f, bary, dists, part, g_fg, g_env = ops.render_fwd_fused_mse(cl_f, B, cfg_f, blocks.face_uvs, blocks.face_map, blocks.map_desc, blocks.maps.detach(),
                                                                      fa, model.renderer._bg, img_e, target, 1.0 / imgs.numel(), img_tiled=tiled)
        gm_f, ga_f, gv_f = ops._fused_bwd(p2f, bary, dists, cl_f, blocks.face_uvs, blocks.face_map, blocks.map_desc, blocks.maps.detach(), fa, cfg_f,
                                          model.renderer._bg, 2, g_fg, B, None, img_tiled=tiled)
        gm_e, _, gv_e = ops._fused_bwd(p2f_e, bary_e, dists_e, cl_e, env.face_uvs, env.face_map, env.map_desc, env.maps.detach(), None, cfg_e,
                                       model.renderer_env._bg, lay, g_env, B, None, img_tiled=tiled)
        un = (lambda t: ops.untile_image(t, H, W)) if tiled else (lambda t: t)
        res[tiled] = dict(img_e=un(img_e), part=part.clone(), g_fg=un(g_fg), g_env=un(g_env), gm_f=gm_f.clone(), ga_f=ga_f.clone(), gv_f=gv_f.clone(),
                          gm_e=gm_e.clone(), gv_e=gv_e.clone())
    for k in res[False]:
        a, b = res[False][k], res[True][k]
        assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-12, k
    assert float(res[True]['part'].sum()) > 0 and float(res[True]['g_env'].abs().sum()) > 0
