"""DifferentiableBlocksWorld -- host-side mirror of the reference model (src/model/dbw.py:38-462): same constructor
keywords (cfg.model.{mesh,renderer,rend_optim,loss}), same 10 parameters and buffer names/shapes (checkpoints
interchange), same `forward(inp, labels) -> {'rgb','parsimony','tv','overlap','total'}` contract that
src/trainer.py:137-147 drives, with every per-iteration computation executed by libdbw_hip.so:

  build_blocks / build_ground / build_bkg  -> dbw_sq_blocks_*, dbw_posed_mesh_*, dbw_texture_prep_*
  Renderer (x3: coarse, fine, env)         -> dbw_project_clip_*, dbw_rasterize_*, dbw_shade_blend_*
  decoupled composite + MSE                -> dbw_composite_mse
  TV / overlap regularisers                -> dbw_tv_l2sq, dbw_overlap_loss
Only O(K)-scalar glue (opacity sigmoid/noise, parsimony over K numbers, loss weighting) stays in torch.

The perceptual term needs the third-party `lpips` package + VGG weights (absent here): it is a pluggable callable
(`perceptual_fn`), excluded from the HIP path as SURVEY.md 8(a) A10 prescribes."""
import warnings
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import mesh as M
from . import ops
from .renderer import Renderer
from .structures import PackedScene

DECIMATE_FACTOR = 8            # dbw.py:32
OVERLAP_N_POINTS = 1000        # dbw.py:33
OVERLAP_N_BLOCKS = 1.95        # dbw.py:34
OVERLAP_TEMPERATURE = 0.005    # dbw.py:35


_FUSED_SLOT = {'rgb': 0, 'parsimony': 1, 'tv': 2, 'overlap': 3}      # ops.fused_losses output layout


def safe_pow(t, exponent, eps=1e-6):      # utils/pytorch.py:35-36
    return t.clamp(eps).pow(exponent)


# loss.py:11-24: the registry's criteria on an image pair, called as criterion(imgs, rec) with their default (mean) reduction
_CRITERIA = {'mse': F.mse_loss, 'l2': F.mse_loss, 'l1': F.l1_loss, 'huber': F.smooth_l1_loss}
# loss.py:43-47 (tv_norm_funcs): norm of a difference tensor over its channel axis
_TV_NORMS = {'l1': lambda t: t.abs().sum(-1), 'l2': lambda t: safe_pow(t.pow(2).sum(-1), 0.5), 'l2sq': lambda t: t.pow(2).sum(-1)}


class DifferentiableBlocksWorld(nn.Module):
    name = 'dbw'

    def __init__(self, img_size, **kwargs):
        super().__init__()
        self._init_kwargs = deepcopy(kwargs)
        self._init_kwargs['img_size'] = img_size
        self.img_size = tuple(img_size) if not isinstance(img_size, int) else (img_size, img_size)
        self._init_blocks(**kwargs.get('mesh', {}))
        self._init_renderer(self.img_size, **kwargs.get('renderer', {}))
        self._init_rend_optim(**kwargs.get('rend_optim', {}))
        self._init_loss(**kwargs.get('loss', {}))
        self.cur_epoch = 0
        self.perceptual_fn = None
        self.world_size, self.rank = 1, 0     # view-sharded data parallel (parallel.py)
        self._noise_override = None
        self._overlap_u_override = None
        # sync_free=True: dead blocks (kill_blocks / filter_transparent) keep their slot and are collapsed to a point by the
        # kernel instead of being packed away on the host -> no device->host sync per iteration (dbw.py:322 `.item()`), same
        # images/losses/gradients, and the whole iteration becomes hipGraph-capturable.
        self.sync_free = False
        self.overlap_passes = False   # render the env pass on a side stream, concurrently with the fg pass (layered path only)
        self.fused_loss_epilogue = True   # training forward: composite + MSE as the epilogue of the fg pass (ops.render_decoupled_mse)

    @property
    def init_kwargs(self):
        return deepcopy(self._init_kwargs)

    # ------------------------------------------------------------------------------------------------ init (dbw.py:55-119)
    def _init_blocks(self, **kwargs):
        self.n_blocks = kwargs.pop('n_blocks', 1)
        self.S_world = kwargs.pop('S_world', 1)
        elev, azim, roll = kwargs.pop('R_world', [0, 0, 0])
        self.register_buffer('R_world', M.world_rotation(elev, azim, roll)[None])
        self.register_buffer('T_world', torch.Tensor(kwargs.pop('T_world', [0., 0., 0.]))[None])
        self.z_far = kwargs.pop('z_far', 10)
        self.ratio_block_scene = kwargs.pop('ratio_block_scene', 1 / 4)
        self.txt_size = kwargs.pop('txt_size', 256)
        self.txt_bkg_upscale = kwargs.pop('txt_bkg_upscale', 1)
        self.scale_min = kwargs.pop('scale_min', 0.2)
        opacity_init = kwargs.pop('opacity_init', 0.5)
        T_range = kwargs.pop('T_range', [1, 1, 1])
        T_init_mode = kwargs.pop('T_init_mode', 'gauss')
        assert len(kwargs) == 0, kwargs
        N, TS = self.n_blocks, self.txt_size

        # sky dome + ground plane (dbw.py:74-79)
        bkg_v, bkg_f = M.get_icosphere(level=2, flip_faces=True)
        bkg_v = bkg_v * self.z_far
        self.register_buffer('bkg_verts_uvs', M.point_to_uv_sphericalmap(bkg_v))
        g_v, g_f = M.get_plane()
        g_v = g_v * torch.Tensor([self.z_far, 1, self.z_far])[None]
        for _ in range(3):
            g_v, g_f = M.subdivide_mesh(g_v, g_f)
        self.register_buffer('ground_verts_uvs', (g_v[:, [0, 2]] / self.z_far + 1) / 2)

        # block primitive = icosphere-1 superquadric (dbw.py:82-96)
        b_v, b_f = M.get_icosphere(level=1)
        self.sq_eps = nn.Parameter(torch.zeros(N, 2))
        self.register_buffer('sq_eta', torch.asin(b_v[:, 1])[None].repeat(N, 1))
        self.register_buffer('sq_omega', torch.atan2(b_v[:, 0], b_v[:, 2])[None].repeat(N, 1))
        faces_uvs, verts_uvs = M.get_icosphere_uvs(level=1, fix_continuity=True, fix_poles=True)
        p_left = abs(int(np.floor(verts_uvs.min(0)[0][0].item() * TS)))
        p_right = int(np.ceil((verts_uvs.max(0)[0][0].item() - 1) * TS))
        verts_u = (verts_uvs[..., 0] * TS + p_left) / (TS + p_left + p_right)
        verts_uvs = torch.stack([verts_u, verts_uvs[..., 1]], dim=-1)
        self.txt_padding = p_left, p_right
        self.BNF = len(faces_uvs)
        self.register_buffer('block_faces_uvs', faces_uvs)
        self.register_buffer('block_verts_uvs', verts_uvs)

        # learnable pose parameters; RNG draw order = dbw.py:99-119 (same seed -> same init as the reference)
        self.R_6d_ground = nn.Parameter(torch.Tensor([[1., 0., 0., 0., 1., 0.]]))
        self.T_ground = nn.Parameter(torch.Tensor([[0., -0.9 * T_range[1], 0.]]))
        S_init = (torch.rand(N, 3) + 0.5 - self.scale_min).log()
        R_6d_init = M.matrix_to_rotation_6d(M.random_rotations(N))
        if T_init_mode == 'gauss':
            T_init = torch.randn(N, 3) / 2 * torch.Tensor(T_range)
        elif T_init_mode == 'uni':
            T_init = (2 * torch.rand(N, 3) - 1) * torch.Tensor(T_range)
        else:
            raise NotImplementedError
        self.S = nn.Parameter(S_init.clone())
        self.R_6d = nn.Parameter(R_6d_init.clone())
        self.T = nn.Parameter(T_init.clone())
        self.alpha_logit = nn.Parameter(torch.logit(torch.ones(N) * opacity_init) + 1e-3)
        u = self.txt_bkg_upscale
        self.texture_bkg = nn.Parameter(torch.randn(1, TS * u, TS * u, 3) / 10)
        self.texture_ground = nn.Parameter(torch.randn(1, TS * u, TS * u, 3) / 10)
        self.textures = nn.Parameter(torch.randn(N, TS, TS, 3) / 10)

        # ---- constant device tables for the kernels (non-persistent: not part of checkpoints) ----
        nvb = bkg_v.shape[0]
        reg = lambda n, t: self.register_buffer(n, t.contiguous(), persistent=False)
        reg('_bkg_verts', bkg_v)
        reg('_ground_base', g_v)
        reg('_env_faces', torch.cat([bkg_f, g_f + nvb], 0).to(torch.int32))
        reg('_env_face_uvs', torch.cat([self.bkg_verts_uvs[bkg_f], self.ground_verts_uvs[g_f]], 0).float())
        reg('_env_face_map', torch.cat([torch.zeros(len(bkg_f)), torch.ones(len(g_f))]).to(torch.int32))
        reg('_env_map_desc', PackedScene.describe_maps([(TS * u, TS * u)] * 2, [(0, 0)] * 2, 'cpu')[0])
        self._n_bkg_faces, self._n_ground_faces = len(bkg_f), len(g_f)
        # cos/sin tables of the constant angle buffers, evaluated once on the host (include/dbw_hip.h: `trig`)
        reg('_trig', torch.stack([torch.cos(self.sq_eta), torch.sin(self.sq_eta), torch.cos(self.sq_omega), torch.sin(self.sq_omega)], 0))
        nv = b_v.shape[0]
        self._block_nv = nv
        reg('_block_faces_all', torch.cat([b_f + k * nv for k in range(N)], 0).to(torch.int32))
        reg('_block_face_uvs_all', verts_uvs[faces_uvs].repeat(N, 1, 1).float())
        reg('_block_face_map_all', torch.arange(N).repeat_interleave(self.BNF).to(torch.int32))
        reg('_block_map_desc_all', PackedScene.describe_maps([(TS, TS)] * N, [self.txt_padding] * N, 'cpu')[0])
        bb, bi, self._bins_per_block = PackedScene.describe_bins([(TS, TS)], 'cpu')
        bb, bi, _ = PackedScene.describe_bins([(TS, TS)] * N, 'cpu')
        reg('_block_bin_base', bb)
        reg('_block_bin_info', bi)
        reg('_block_faces_one', b_f)

    def _init_rend_optim(self, **kwargs):          # dbw.py:121-129
        self.opacity_noise = kwargs.pop('opacity_noise', False)
        self.decouple_rendering = kwargs.pop('decouple_rendering', False)
        self.coarse_learning = kwargs.pop('coarse_learning', True)
        self.decimate_txt = kwargs.pop('decimate_txt', False)
        self.decim_factor = kwargs.pop('decimate_factor', DECIMATE_FACTOR)
        self.kill_blocks = kwargs.pop('kill_blocks', False)
        assert len(kwargs) == 0, kwargs
        d = int(self.decim_factor)
        if d < 1 or (d & (d - 1)) or self.txt_size % d:
            raise NotImplementedError(f'decimate_factor={d}: the sampler keeps decimated maps at cell resolution and needs a '
                                      'power of two that divides txt_size')
        # descriptors of the decimated maps: same (h, w), stored at (h >> shift, w >> shift)
        shift, TS, u, N = d.bit_length() - 1, self.txt_size, self.txt_bkg_upscale, self.n_blocks
        dev = self._env_map_desc.device
        self.register_buffer('_env_map_desc_dec', PackedScene.describe_maps([(TS * u, TS * u)] * 2, [(0, 0)] * 2, dev, shift)[0], persistent=False)
        self.register_buffer('_block_map_desc_dec', PackedScene.describe_maps([(TS, TS)] * N, [self.txt_padding] * N, dev, shift)[0],
                             persistent=False)

    def _init_renderer(self, img_size, **kwargs):  # dbw.py:131-143 (renderer_light is visualisation-only: not built)
        kwargs = deepcopy(kwargs)
        self.renderer = Renderer(img_size, **kwargs)
        kwargs['sigma'] = 5e-6
        self.renderer_fine = Renderer(img_size, **kwargs)
        kwargs['faces_per_pixel'] = 1
        kwargs['sigma'] = 0
        kwargs['detach_bary'] = False
        self.renderer_env = Renderer(img_size, **kwargs)

    def _init_loss(self, **kwargs):                # dbw.py:145-163
        weights = {'rgb': kwargs.pop('rgb_weight', 1.0), 'perceptual': kwargs.pop('perceptual_weight', 0),
                   'parsimony': kwargs.pop('parsimony_weight', 0), 'scale': kwargs.pop('scale_weight', 0),
                   'tv': kwargs.pop('tv_weight', 0), 'overlap': kwargs.pop('overlap_weight', 0)}
        name = kwargs.pop('name', 'mse')
        kwargs.pop('perceptual_name', 'lpips')
        tv_type = kwargs.pop('tv_type', 'l2sq')
        assert len(kwargs) == 0, kwargs
        # loss.py:11-24,43-47: the image criteria of the registry that are criteria on an RGB image pair (mse / l2, l1, huber = SmoothL1)
        # and the three total-variation norms.  The defaults of every shipped config (mse, l2sq) run inside the HIP kernels (loss epilogue of
        # the fg pass, tv_l2sq_sets); the others through torch on the rendered image / the prepared maps -- autograd then reaches the HIP
        # backward the same way (general path of compute_losses; the one-call C step and the fused loss epilogue stand aside)
        if name not in _CRITERIA:
            raise NotImplementedError(f"loss '{name}': an image criterion out of {sorted(_CRITERIA)} (loss.py:11-24; bce / cosine / ssim / "
                                      'chamfer / tv / lpips are not reconstruction criteria on an RGB image pair)')
        if tv_type not in _TV_NORMS:
            raise NotImplementedError(f"tv_type '{tv_type}': one of {sorted(_TV_NORMS)} (loss.py:43-47)")
        self.criterion_name, self.tv_type = name, tv_type
        self.default_criteria = name in ('mse', 'l2') and tv_type == 'l2sq'
        self.loss_weights = {k: v for k, v in weights.items() if v > 0}
        self.loss_names = [f'loss_{n}' for n in list(self.loss_weights.keys()) + ['total']]

    def set_perceptual(self, fn):
        """fn(imgs, rec) -> scalar, e.g. lpips.LPIPS(net='vgg') with normalize=True (loss.py:32-40)."""
        self.perceptual_fn = fn

    # ------------------------------------------------------------------------------------------------ bookkeeping
    def set_cur_epoch(self, epoch):
        self.cur_epoch = epoch

    def step(self):
        self.cur_epoch += 1

    def to(self, device):
        super().to(device)
        for r in (self.renderer, self.renderer_fine, self.renderer_env):
            r.to(device)
        return self

    def release_graph(self):
        """Drop the tensors cached by the last forward (they keep its autograd graph -- and the parameters' AccumulateGrad
        nodes, which are bound to the stream they were created on -- alive)."""
        for n in ('_alpha', '_alpha_full', '_blocks_maps', '_bkg_maps', '_ground_maps', '_keep_mask'):
            if hasattr(self, n):
                setattr(self, n, None)

    def is_live(self, name):                        # dbw.py:457-462
        milestone = getattr(self, name)
        if isinstance(milestone, bool):
            return milestone
        return True if self.cur_epoch < milestone else False

    @property
    def bkg_n_faces(self):
        return self._n_bkg_faces

    @property
    def ground_n_faces(self):
        return self._n_ground_faces

    @property
    def env_n_faces(self):
        return self._n_bkg_faces + self._n_ground_faces

    @property
    def blocks_n_faces(self):
        return self.n_blocks * self.BNF

    def get_opacities(self):                        # dbw.py:410-414
        alpha = torch.sigmoid(self.alpha_logit)
        if self.kill_blocks:
            alpha = alpha * (alpha > 0.01)
        return alpha

    @torch.no_grad()
    def get_nb_opaque_blocks(self):
        return (self.get_opacities() > 0.5).sum().item()

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict=False):   # tolerant by-name copy, spq_ -> sq_ (dbw.py:440-455)
        state = self.state_dict()
        missing = []
        for name, param in state_dict.items():
            name = name.replace('module.', '').replace('spq_', 'sq_')
            if name in state:
                state[name].copy_(param.data if isinstance(param, nn.Parameter) else param)
            else:
                missing.append(name)
        if missing:
            warnings.warn(f'load_state_dict: {missing} not found')

    # ------------------------------------------------------------------------------------------------ scene building
    def _world_consts(self):
        return float(self.S_world), self.R_world[0].contiguous(), self.T_world[0].contiguous()

    def build_env_scene(self):
        """join(build_bkg(world_coord=True), build_ground(world_coord=True))  (dbw.py:214,267-295) as a PackedScene."""
        S_w, R_w, T_w = self._world_consts()
        # constant geometry (no parameter involved): cached until the world transform changes (load_state_dict copies into the
        # R_world / T_world buffers in place -> their version counters move; S_world is a plain attribute) or the model moves
        key = (float(S_w), self.R_world._version, self.T_world._version, self.R_world.data_ptr(), self._bkg_verts.device)
        if getattr(self, '_bkg_world_key', None) != key:
            self._bkg_world = ((self._bkg_verts * S_w) @ R_w + T_w).detach()
            self._bkg_world_key = key
        bkg_v = self._bkg_world
        ground_v = ops.posed_mesh(self.R_6d_ground, self.T_ground, self._ground_base, S_w, R_w, T_w)
        decim = self.decim_factor if (self.training and self.is_live('decimate_txt')) else 1
        bkg_maps, self._bkg_maps = ops.texture_prep(self.texture_bkg, decim)
        g_maps, self._ground_maps = ops.texture_prep(self.texture_ground, decim)
        verts = torch.cat([bkg_v, ground_v], 0)
        maps = torch.cat([bkg_maps.reshape(-1), g_maps.reshape(-1)])
        desc = self._env_map_desc if decim == 1 else self._env_map_desc_dec
        scene = PackedScene(verts, self._env_faces, self._env_face_uvs, self._env_face_map, desc, maps)
        scene.const_faces = self._n_bkg_faces       # the sky dome is a buffer (dbw.py:74-76): no gradient flows to its vertices
        return scene

    def get_blocks_verts(self):
        """Block-frame superquadric vertices * ratio (dbw.py:348-352), for callers that want them unposed."""
        dev = self.sq_eps.device
        ident6 = torch.tensor([[1., 0., 0., 0., 1., 0.]], device=dev).repeat(self.n_blocks, 1)
        zeros = torch.zeros(self.n_blocks, 3, device=dev)
        logS = torch.log(torch.full((self.n_blocks, 3), 1.0 - self.scale_min, device=dev))
        eye = torch.eye(3, device=dev)
        return ops.sq_blocks(self.sq_eps, logS, ident6, zeros, self._trig, None, self.n_blocks, self.ratio_block_scene,
                             self.scale_min, 1.0, eye, None)

    def build_blocks_scene(self, filter_transparent=False):
        """build_blocks(filter_transparent, as_scene=True) (dbw.py:297-346) as a PackedScene, or None if no block is left.
        Sets self._alpha (live blocks), self._alpha_full, self._blocks_maps like the reference."""
        coarse = self.training and self.is_live('coarse_learning')
        noise, noise_scale = None, 0.0
        if self.opacity_noise and coarse:
            noise = self._noise_override if self._noise_override is not None else self._shared_randn_like(self.alpha_logit)
            noise_scale = float(self.opacity_noise)
        masked = filter_transparent or self.kill_blocks
        thresh = (0.5 if filter_transparent else 0.01) if masked else -1.0
        # sigmoid(alpha_logit + noise), the transparency mask on the noise-free opacity and alpha * mask: one launch
        self._alpha, self._alpha_full, mask_i32 = ops.block_alpha(self.alpha_logit, noise, noise_scale, thresh)
        keep, nb = None, self.n_blocks
        if masked:
            if self.sync_free:
                keep = mask_i32
            else:
                nb = int(mask_i32.sum().item())                      # same host sync as dbw.py:322
                if nb < self.n_blocks:
                    keep = mask_i32
                    self._alpha = self._alpha[mask_i32.bool()]
        decim = self.decim_factor if (coarse and self.is_live('decimate_txt')) else 1
        maps_all, self._blocks_maps = ops.texture_prep(self.textures, decim)
        self._keep_mask = keep
        if nb == 0:
            return None
        S_w, R_w, T_w = self._world_consts()
        verts = ops.sq_blocks(self.sq_eps, self.S, self.R_6d, self.T, self._trig, keep, nb, self.ratio_block_scene,
                              self.scale_min, S_w, R_w, T_w, dense=not self.sync_free)
        maps = maps_all if (keep is None or self.sync_free) else maps_all[keep.bool()]
        F_ = nb * self.BNF
        # (a view of the full table: its row 0 announces all n_blocks rows, which stay readable behind the nb rows in use)
        desc = (self._block_map_desc_all if decim == 1 else self._block_map_desc_dec)[:nb]
        self._blocks_decimated = decim > 1
        nbins = nb * self._bins_per_block
        # full-resolution maps: texel gradients go through the texture-space bins (coarse phase 8.6 -> 4.2 ms/step, fine phase
        # 3.1 -> 2.8 ms/step on the bench config); decimated maps use the in-tile LDS hash
        texbins = None if decim > 1 else (self._block_bin_base[:nb], self._block_bin_info[:nbins], nbins)
        return PackedScene(verts.reshape(-1, 3), self._block_faces_all[:F_], self._block_face_uvs_all[:F_],
                           self._block_face_map_all[:F_], desc, maps.reshape(-1), texbins)

    def build_scene(self, filter_transparent=False):
        """Background + ground + blocks as ONE scene (dbw.py:250-266), for the non-decoupled hard / SSAA evaluation render."""
        env = self.build_env_scene()
        blocks = self.build_blocks_scene(filter_transparent=filter_transparent)
        return env if blocks is None else PackedScene.join([env, blocks])

    def _shared_randn_like(self, t):
        """Opacity noise (and the overlap samples) must be identical on every data-parallel rank: they are
        drawn from the device's default generator, which ShardedTrainStep seeds identically on all ranks and which every
        rank advances in lockstep (the default generator is also the one hipGraph capture knows how to replay)."""
        return torch.randn(t.shape, device=t.device, dtype=t.dtype)

    # ------------------------------------------------------------------------------------------------ rendering
    def _ensure_cameras(self, inp):
        if 'K' in inp and self.renderer.cameras.K is None:          # intrinsics frozen from the first sample (dbw.py:204-208)
            for r in (self.renderer, self.renderer_fine, self.renderer_env):
                r.update_cameras(device=inp['imgs'].device, K=inp['K'][0:1])

    def render_joined(self, inp, filter_transparent=False):
        """-> (B,4,H,W): the NON-decoupled rendering (dbw.py:225-232, `decouple_rendering: False`; no shipped config uses it): sky dome,
        ground and blocks joined into one scene and rendered in ONE soft pass of the phase's renderer, the env faces fully opaque
        (alpha_env = 1) next to the blocks' learned opacities.  Same kernels as the fg pass of the decoupled path."""
        self._ensure_cameras(inp)
        fine = not self.is_live('coarse_learning')
        filter_tsp = filter_transparent or fine
        renderer = self.renderer_fine if fine else self.renderer
        scene = self.build_scene(filter_transparent=filter_tsp)
        alpha = None
        if not filter_tsp:
            n_blk = scene.faces.shape[0] - self.env_n_faces
            alpha = torch.cat([torch.ones(self.env_n_faces, device=scene.verts.device), self._alpha.repeat_interleave(self.BNF)[:n_blk]])
        return renderer.render_packed(scene, inp['R'], inp['T'], faces_alpha=alpha)

    def render_layers(self, inp, filter_transparent=False):
        """-> fg (B,4,H,W), env (B,4,H,W): the two passes of the decoupled rendering (dbw.py:213-222)."""
        if not self.decouple_rendering:
            raise NotImplementedError('render_layers is the decoupled rendering; decouple_rendering=False renders one joined scene: render_joined')
        self._ensure_cameras(inp)
        R, T = inp['R'], inp['T']
        fine = not self.is_live('coarse_learning')
        filter_tsp = filter_transparent or fine
        renderer = self.renderer_fine if fine else self.renderer
        if self.overlap_passes:
            # the env pass (hard, 1 face/pixel) and the fg pass are independent until the composite: issue the env pass on a
            # side stream so its latency-bound kernels overlap the fg kernels (autograd replays the same streams in backward)
            cur = torch.cuda.current_stream()
            side = self._side_stream = getattr(self, '_side_stream', None) or torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                env = self.renderer_env.render_packed(self.build_env_scene(), R, T, lds_aggregate=True)
            env.record_stream(cur)
        else:
            env = self.renderer_env.render_packed(self.build_env_scene(), R, T, lds_aggregate=True)     # magnified textures
        blocks = self.build_blocks_scene(filter_transparent=filter_tsp)
        if blocks is not None:
            alpha = None if filter_tsp else self._alpha.repeat_interleave(self.BNF)   # shared by all views (== .repeat(B))
            fg = renderer.render_packed(blocks, R, T, faces_alpha=alpha, lds_aggregate=self._blocks_decimated)
        else:
            fg = torch.zeros_like(env)
        if self.overlap_passes:
            torch.cuda.current_stream().wait_stream(self._side_stream)
        return fg, env

    def _host_packed_rebuild(self):
        """Context of a visualisation-only rebuild of the scene with the kept blocks packed on the host (sync_free off): the per-step state
        the last forward cached (`_alpha`, `_alpha_full`, the maps, the keep mask -- a later compute_losses reads them) is put back
        afterwards, and the opacity noise of the rebuild is the forward's or none: the shared default generator, which the data-parallel
        ranks advance in lock step, is not touched."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            names = ('_alpha', '_alpha_full', '_blocks_maps', '_bkg_maps', '_ground_maps', '_keep_mask', '_blocks_decimated')
            saved = {n: getattr(self, n) for n in names if hasattr(self, n)}
            sf, nz, on = self.sync_free, self._noise_override, self.opacity_noise
            self.sync_free = False
            if self._noise_override is None:
                self.opacity_noise = False
            try:
                yield
            finally:
                self.sync_free, self._noise_override, self.opacity_noise = sf, nz, on
                for n, v in saved.items():
                    setattr(self, n, v)
        return ctx()

    def predict(self, inp, labels=None, w_edges=False, filter_transparent=False):
        if self.decouple_rendering:
            fg, env = self.render_layers(inp, filter_transparent)
            rec = ops.composite(fg, env)                               # rec = rec_fg*mask + (1-mask)*rec_env (dbw.py:223)
        else:
            rec = self.render_joined(inp, filter_transparent)[:, :3]   # rec, mask = renderer(scene).split([3, 1]) (dbw.py:232)
        if w_edges:                                                    # dbw.py:234-238: wireframe of the joined scene, coloured per block
            fine = not self.is_live('coarse_learning')
            filter_tsp = filter_transparent or fine
            with torch.no_grad():
                # host-packed scene (only the kept blocks): the colour table below has one row per KEPT face, while the sync-free
                # scene of the training path keeps all n_blocks blocks (culled ones collapsed to a point) and its face ids run over all
                with self._host_packed_rebuild():
                    scene = self.build_scene(filter_transparent=filter_tsp)
                colors = self.get_scene_face_colors(filter_transparent=filter_tsp).repeat(len(inp['R']), 1)
                renderer = self.renderer_fine if fine else self.renderer
                rec = renderer.draw_edges(rec, scene, inp['R'], inp['T'], colors=colors)
        return rec

    @torch.no_grad()
    def predict_synthetic(self, inp, labels=None):
        """Blocks alone in their synthetic colours (dbw.py:240-248), exact 4x anti-aliased hard render.  The reference shades them
        with a directional light (`renderer_light`: phong shading, visualisation-only); here the colours are flat."""
        self._ensure_cameras(inp)
        was_training = self.training
        self.eval()
        with self._host_packed_rebuild():                  # host-packed: `colors` / `desc` below hold the kept blocks only
            blocks = self.build_blocks_scene(filter_transparent=True)
        self.train(was_training)
        if blocks is None:
            return torch.ones_like(inp['imgs'])
        keep = (self.get_opacities() > 0.5).nonzero().flatten()
        values = torch.linspace(0, 1, self.n_blocks + 1)[1:][keep.cpu()]
        colors = torch.from_numpy(M.get_fancy_cmap()(values.numpy())).float().to(blocks.verts.device)
        desc = PackedScene.describe_maps([(1, 1)] * len(colors), [(0, 0)] * len(colors), blocks.verts.device)[0]
        flat = PackedScene(blocks.verts, blocks.faces, blocks.face_uvs, blocks.face_map, desc, colors.reshape(-1).contiguous())
        bg_renderer = Renderer(self.img_size, **{**self.renderer.init_kwargs, 'background_color': (1, 1, 1)})
        bg_renderer.update_cameras(device=blocks.verts.device, K=self.renderer.cameras.K)
        return bg_renderer.render_packed(flat, inp['R'], inp['T'], viz_purpose=True)[:, :3]

    @torch.no_grad()
    def get_scene_face_colors(self, filter_transparent=False, w_env=True):       # dbw.py:420-431
        val_blocks = torch.linspace(0, 1, self.n_blocks + 1)[1:]
        if filter_transparent:
            val_blocks = val_blocks[self.get_opacities().cpu() > 0.5]
        elif self.kill_blocks:
            val_blocks = val_blocks[self.get_opacities().cpu() > 0.01]
        NFE = self.env_n_faces if w_env else 0
        values = torch.cat([torch.zeros(NFE), val_blocks.repeat_interleave(self.BNF)])
        return torch.from_numpy(M.get_fancy_cmap()(values.numpy())).float().to(self.sq_eps.device)

    @torch.no_grad()
    def get_arranged_block_txt(self):                                            # dbw.py:433-438
        maps = torch.sigmoid(self.textures).permute(0, 3, 1, 2)
        ncol, nrow = 5, len(maps) // 5
        rows = [torch.cat([maps[k] for k in range(ncol * i, ncol * (i + 1))], dim=2) for i in range(nrow)]
        return torch.cat(rows, dim=1)[None]

    def forward(self, inp, labels=None):
        self._view_ids = inp.get('view_ids')          # (for a perceptual criterion with cached targets: _perceptual_term)
        try:
            return self._forward(inp)
        finally:
            self._view_ids = None

    def _forward(self, inp):
        if inp['imgs'].shape[0] == 0:
            # a data-parallel rank whose shard is exhausted (parallel.py / trainer.py: uneven shards): nothing to render, the step
            # only carries this rank's share of the view-independent regularisers
            self.build_env_scene()
            self.build_blocks_scene(filter_transparent=not self.is_live('coarse_learning'))
            return self.compute_losses(inp['imgs'], None, layers=None)
        if not self.decouple_rendering:                                # one joined scene, one soft pass; losses through the general path
            return self.compute_losses(inp['imgs'], self.render_joined(inp)[:, :3])
        fused = self._forward_fused(inp) if self.fused_loss_epilogue else None
        if fused is not None:
            return fused
        fg, env = self.render_layers(inp)
        return self.compute_losses(inp['imgs'], None, layers=(fg, env))

    def _forward_fused(self, inp):
        """The training iteration with the reconstruction loss as the epilogue of the fg pass (ops.render_decoupled_mse): env pass,
        then fg pass + decoupled composite + MSE in one kernel -- no fg image, no composite kernel, no image-sized gradient round
        trips.  Returns None when the configuration needs the general path (non-decoupled, perceptual term, no block left, layered
        fallbacks switched off)."""
        w = self.loss_weights
        if (not self.decouple_rendering or 'rgb' not in w or 'perceptual' in w or not self.default_criteria
                or not (ops.FUSED_FORWARD and ops.FUSED_BACKWARD and ops.TILED_FRAGMENTS and ops.UV_FRAGMENTS)):
            return None
        self._ensure_cameras(inp)
        fine = not self.is_live('coarse_learning')
        renderer = self.renderer_fine if fine else self.renderer
        if not renderer.detach_bary or renderer.faces_per_pixel < 2 or renderer.cam_name != 'perspective' or renderer.cameras.K is None:
            return None
        if not renderer.clip_inside:          # the sigmoid opacity (renderer.py:257-258): generic shading kernels only, the loss epilogue is exp-only
            return None
        blocks = self.build_blocks_scene(filter_transparent=fine)
        if blocks is None or blocks.faces.shape[0] >= (1 << 20) or blocks.map_desc.shape[0] >= (1 << 11):
            return None
        env = self.build_env_scene()
        R, T = inp['R'].float().contiguous(), inp['T'].float().contiguous()
        Kmat = renderer.cameras.K[0].to(R.device).contiguous()
        alpha = None if fine else self._alpha.repeat_interleave(self.BNF)
        cfg_e = self.renderer_env._cfg(env.faces.shape[0], lds_aggregate=True, const_faces=env.const_faces)
        cfg_f = renderer._cfg(blocks.faces.shape[0], lds_aggregate=self._blocks_decimated, texbins=blocks.texbins)
        imgs = inp['imgs']
        count = imgs.numel() if getattr(self, '_global_count', None) is None else self._global_count
        rgb = ops.render_decoupled_mse(env, blocks, alpha, imgs, float(w['rgb']) / float(count), R, T, Kmat, cfg_e, cfg_f,
                                       self.renderer_env._bg, renderer._bg)
        return self.compute_losses(imgs, None, layers=None, rgb_value=rgb)

    # ------------------------------------------------------------------------------------------------ evaluation (dbw.py:464-493)
    @torch.no_grad()
    def quantitative_eval(self, loader, device, hard_inference=True):
        """PSNR / SSIM (/ LPIPS when a perceptual network was supplied with set_perceptual) of the reconstruction over `loader`
        (batches of (inp, labels) like the reference's datasets).  hard_inference: exact anti-aliased render of the joined scene
        (4x resolution, sigma 0, one face per pixel, 4x4 average pooling; renderer.py:56-60,178-183), else the soft decoupled
        prediction.  -> OrderedDict like the reference's; LPIPS is NaN when no network is installed."""
        from collections import OrderedDict
        from .metrics import AverageMeter, mse2psnr, ssim
        was_training = self.training
        self.eval()
        opacities = self.get_opacities()
        n_blocks = int((opacities > 0.5).sum().item())
        meters = {k: AverageMeter() for k in ('L_tot', 'L_rec', 'PSNR', 'SSIM', 'LPIPS')}
        scene = self.build_scene(filter_transparent=True) if hard_inference else None
        for inp, labels in loader:
            inp = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in inp.items()}
            imgs, N = inp['imgs'], len(inp['imgs'])
            if hard_inference:
                self._ensure_cameras(inp)
                rec = self.renderer.render_packed(scene, inp['R'], inp['T'], viz_purpose=True)[:, :3]
            else:
                rec = self.predict(inp, labels, filter_transparent=True)
            losses = self.compute_losses(imgs, rec)
            meters['L_tot'].update(losses['total'], N=N)
            meters['L_rec'].update(sum(losses[k] for k in ('rgb', 'perceptual') if k in losses), N=N)
            meters['PSNR'].update(mse2psnr(F.mse_loss(imgs, rec)), N=N)
            meters['SSIM'].update(ssim(imgs, rec, padding=False).mean(), N=N)
            meters['LPIPS'].update(self.perceptual_fn(imgs, rec) if self.perceptual_fn is not None else float('nan'), N=N)
        self.train(was_training)
        return OrderedDict([('n_blocks', n_blocks)] + [(k, m.avg) for k, m in meters.items()]
                           + [(f'alpha{k}', a.item()) for k, a in enumerate(opacities)])

    # ------------------------------------------------------------------------------------------------ losses (dbw.py:361-408)
    def _perceptual_term(self, imgs, rec, coarse, view_ids=None):
        """view_ids: the batch's `view_ids` entry, if the caller's loader provides one (trainer.py does) -- handed to a criterion that keeps
        the constant half of its work per training view (lpips_vgg.LPIPSVGG.cache_targets); never needed for the value."""
        if self.perceptual_fn is None:
            raise RuntimeError('perceptual_weight > 0 needs model.set_perceptual(fn): lpips is a third-party network '
                               'outside the HIP path (SURVEY.md 8a A10)')
        # the perceptual criterion is a mean over the views it is given: under view-sharded data parallelism the gradients of all ranks
        # are SUMMED, so a rank's term is weighted by its share of the global batch (like the MSE, which is normalised by the
        # global element count) -- the sum over ranks is then the reference's mean over the whole batch
        share = 1.0
        if self.world_size > 1 and getattr(self, '_global_count', None):
            share = imgs.numel() / float(self._global_count)
        if view_ids is not None and getattr(self.perceptual_fn, 'target_cache', None) is not None:
            value = self.perceptual_fn(imgs, rec, view_ids=view_ids)
        else:
            value = self.perceptual_fn(imgs, rec)
        return self.loss_weights['perceptual'] * (1 if coarse else 0.1) * share * value

    def compute_losses(self, imgs, rec, layers=None, rgb_value=None):
        w = self.loss_weights
        dev = imgs.device
        coarse = self.is_live('coarse_learning')
        ws = self.world_size
        # view-independent regularisers: every rank computes them identically; scaled by 1/world_size so that the
        # sum all-reduce of gradients counts them once
        rs = 1.0 / ws
        if (layers is not None or rgb_value is not None) and 'rgb' in w and self.default_criteria:
            # training path: composite + MSE and the regularisers as ONE autograd node (ops.fused_losses); factors of
            # dbw.py:373-405: parsimony and overlap only act in the coarse phase, tv is scaled by 0.1 afterwards (and the
            # ground map once more)
            count = imgs.numel() if getattr(self, '_global_count', None) is None else self._global_count
            tv_f = 1 if coarse else 0.1
            cfg = {'rgb': float(w['rgb']), 'count': float(count),
                   'parsimony': float(w['parsimony']) * rs if ('parsimony' in w and coarse) else None,
                   'tv': float(w['tv']) * tv_f * rs if 'tv' in w else None, 'tv_ground_factor': tv_f,
                   'overlap': float(w['overlap']) * rs if ('overlap' in w and coarse) else None,
                   'overlap_consts': (float(self.ratio_block_scene), float(self.scale_min), OVERLAP_TEMPERATURE, OVERLAP_N_BLOCKS)}
            u = None
            if cfg['overlap']:
                u = self._overlap_u_override
                if u is None:
                    u = torch.rand(self.n_blocks, OVERLAP_N_POINTS, 3, device=dev)
            fg_l, env_l = layers if layers is not None else (None, None)
            vals = ops.fused_losses(fg_l, env_l, imgs, self._alpha_full, self._bkg_maps, self._blocks_maps, self._ground_maps,
                                    self.sq_eps, self.S, self.R_6d, self.T, u, cfg)
            losses = {}
            for k in w:
                if k in _FUSED_SLOT:
                    losses[k] = vals[_FUSED_SLOT[k]]
            total = vals.sum()
            if rgb_value is not None:       # reconstruction term already reduced by the fg pass's epilogue (slot 0 of vals is 0)
                losses['rgb'] = rgb_value
                total = total + rgb_value
            if 'perceptual' in w:
                losses['perceptual'] = self._perceptual_term(imgs, ops.composite(*layers), coarse, getattr(self, '_view_ids', None))
                total = total + losses['perceptual']
            losses = {k: losses[k] for k in w}
            losses['total'] = total
            return losses
        losses = {k: torch.zeros((), device=dev) for k in w}          # (fill kernel: hipGraph-capturable, unlike an H2D copy)
        empty = imgs.shape[0] == 0                                     # no views on this rank in this step: regularisers only
        if 'rgb' in losses and not empty:
            # a mean over the GLOBAL batch under view-sharded data parallelism (the ranks' gradients are summed)
            share = imgs.numel() / float(self._global_count) if (ws > 1 and getattr(self, '_global_count', None)) else 1.0
            losses['rgb'] = w['rgb'] * share * _CRITERIA[self.criterion_name](imgs, rec if rec is not None else ops.composite(*layers))
        if 'perceptual' in losses and not empty:
            losses['perceptual'] = self._perceptual_term(imgs, rec if rec is not None else ops.composite(*layers), coarse, getattr(self, '_view_ids', None))
        if 'parsimony' in losses:
            factor = 1 if coarse else 0
            alpha = self._alpha_full if coarse else (self._alpha_full > 0.5).float()
            losses['parsimony'] = w['parsimony'] * factor * rs * safe_pow(alpha, 0.5).mean()
        if 'tv' in losses:
            factor = 1 if coarse else 0.1
            if self.tv_type == 'l2sq':
                tv = ops.tv_l2sq(self._bkg_maps) + ops.tv_l2sq(self._blocks_maps, wrap_x=True) + ops.tv_l2sq(self._ground_maps) * factor
            else:       # dbw.py:378-387 with tv_norm_funcs['l1' | 'l2'], in torch on the prepared maps
                norm, bm = _TV_NORMS[self.tv_type], self._blocks_maps
                tv = sum(norm(torch.diff(self._bkg_maps, dim=k)).mean() for k in (1, 2))
                tv = tv + norm(torch.diff(bm, dim=2, append=bm[:, :, 0:1])).sum(0).mean() + norm(torch.diff(bm, dim=1)).sum(0).mean()
                tv = tv + sum(norm(torch.diff(self._ground_maps, dim=k)).mean() for k in (1, 2)) * factor
            losses['tv'] = w['tv'] * factor * rs * tv
        if 'overlap' in losses and coarse:
            # (the term is switched off after the coarse phase, dbw.py:390: no samples are drawn then -- the fused and the native path
            # do not draw either, and the ranks' default generators have to stay in lock step)
            u = self._overlap_u_override
            if u is None:
                u = torch.rand(self.n_blocks, OVERLAP_N_POINTS, 3, device=dev)
            ov = ops.overlap_loss(self.sq_eps, self.S, self.R_6d, self.T, self._alpha_full, u, self.ratio_block_scene, self.scale_min,
                                  OVERLAP_TEMPERATURE, OVERLAP_N_BLOCKS)
            losses['overlap'] = w['overlap'] * rs * ov
        losses['total'] = sum(losses.values())
        return losses
