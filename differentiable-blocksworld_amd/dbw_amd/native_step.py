"""One optimisation iteration of the training path WITHOUT autograd: the forward kernels, then the backward kernels in a fixed order,
every parameter gradient accumulated straight into the flat gradient buffer (parallel.FlatParams) that the all-reduce and the fused
Adam consume.

Why: `model(inp); loss.backward()` drives the same kernels through ~25 autograd nodes.  Each node costs Python time, and wherever two
nodes feed one parameter (pose: blocks + overlap; textures: render + TV; opacities: render + parsimony + overlap) autograd inserts an
elementwise add, plus one more per parameter to accumulate into the preallocated `.grad` -- 18 five-microsecond launches per step, next
to `cat`s, fills, a `repeat_interleave` pair and three reductions that only exist because tensors travel between nodes
(66 launches per step when measured).  Here every kernel writes where its result is needed: 31 launches
(profiles/r02_step_sequence_epoch0.txt).

Same mathematics as DifferentiableBlocksWorld.forward (src/model/dbw.py:198-408) + backward, checked against it
(tests/test_gpu_model.py::test_native_step_equals_autograd_step).  Scope: the decoupled training render with MSE + parsimony + TV +
overlap (every shipped config minus LPIPS); anything else -> `supported()` is False and the caller uses the autograd path."""
import weakref

import torch

from . import _lib, ops
from .dbw import OVERLAP_N_BLOCKS, OVERLAP_N_POINTS, OVERLAP_TEMPERATURE

_p = ops._ptr
_SIDE_STREAMS = {}


class NativeStep:
    def __init__(self, model, params):
        self.m, self.params = model, params
        self.grad = {n: model.get_parameter(n).grad for n, _, _ in params.names}       # views of the flat gradient buffer
        self._env_verts = None
        self._target, self._target_key = None, None
        self._side, self.overlap_regularisers, self.side_priority = None, True, True
        self.regularisers_behind_fg = True
        self.binned_concurrent = None         # full-resolution phases: env backward chain next to the fg backward kernel instead of
                                              # behind it (None: on one GPU; 1.655 -> 1.635 and 1.017 -> 1.007 ms per step)
        # None: by configuration -- one after the other when the blocks' gradients are worth announcing early (data parallel: their
        # all-reduce then runs next to the env backward) or when a bin reduction follows the fg kernel (full-resolution phases: measured
        # 2 % faster), both at once otherwise (1 % faster on one GPU with decimated maps); True / False force one or the other
        self.sequential_backward = None
        self.on_block_grads_ready = None      # callback: every gradient that does not depend on the env pass is final (on the current stream)

    def _tiled_target(self, imgs):
        """The target images in the tile-planar layout, tiled ONCE per tensor (the training views are static: the bench and a trainer
        that keeps its views resident hand over the same tensor every step; a fresh mini-batch costs one permute kernel)."""
        # keyed on the tensor OBJECT (and its version counter), not on its address: the allocator hands a freed mini-batch's address
        # to the next one
        src = self._target_key[0]() if self._target_key is not None else None
        if src is not imgs or self._target_key[1] != imgs._version:
            self._target, self._target_key = ops.tile_image(imgs), (weakref.ref(imgs), imgs._version)
        return self._target

    def supported(self):
        m, w = self.m, self.m.loss_weights
        r = self.m.renderer
        # (clip_inside = False -- the sigmoid opacity -- is implemented by the generic shading / backward kernels of the autograd path only: the
        # specialised uv kernels and the loss epilogue are exp-only, sigma >= 0)
        return (m.decouple_rendering and m.sync_free and 'rgb' in w and 'perceptual' not in w and r.detach_bary and r.faces_per_pixel > 1
                and r.clip_inside and getattr(m.renderer_fine, 'clip_inside', True) and getattr(m, 'default_criteria', True)
                and r.cam_name == 'perspective' and m.blocks_n_faces < (1 << 20) and m.n_blocks + 2 < (1 << 11)
                and ops.FUSED_FORWARD and ops.FUSED_BACKWARD and ops.TILED_FRAGMENTS and ops.UV_FRAGMENTS)

    def __call__(self, inp, global_count=None, zero_grad=None):
        """Forward + backward of one iteration on this rank's views.  The caller has opened the zero arena (ops.ARENA.begin_step) and
        either zeroed the flat gradient buffer or passes `zero_grad` (called here, on the side stream, off the critical path).
        -> {'rgb', 'parsimony', 'tv', 'overlap', 'total'} as 0-dim device tensors.

        Schedule (two HIP streams; `overlap_regularisers = False` runs the same calls on one):
          main: ground mesh -> env projection -> env per-face set-up | env pass | fg pass + MSE | env backward + tail [| fg textures]
          side: texture prep, zero grads, opacities | blocks' vertices, projection, fg per-face set-up | regularisers | fg backward + tail
        (the backward kernels one after the other or both at once: see `sequential_backward`)
        """
        m, g = self.m, self.grad
        w = m.loss_weights
        imgs = inp['imgs']
        dev = imgs.device
        m._ensure_cameras(inp)
        B = imgs.shape[0]
        coarse = m.is_live('coarse_learning')              # (training mode)
        fine = not coarse
        decim = int(m.decim_factor) if m.is_live('decimate_txt') else 1
        decim_blocks = decim if coarse else 1              # dbw.py:329-334: blocks are only decimated in the coarse phase
        rs = 1.0 / m.world_size
        S_w, R_w, T_w = m._world_consts()
        nb, nv, TS, u_ = m.n_blocks, m._block_nv, m.txt_size, m.txt_bkg_upscale
        renderer = m.renderer_fine if fine else m.renderer
        R, T = inp['R'].float().contiguous(), inp['T'].float().contiguous()
        Kmat = renderer.cameras.K[0].to(dev).contiguous()
        Fe, Ff = m._env_faces.shape[0], nb * m.BNF
        cur = torch.cuda.current_stream(dev)
        side = cur
        if self.overlap_regularisers:
            if self._side is None:
                # high priority: its small kernels overtake the big render kernels of the main stream instead of queueing behind them, and
                # the fg backward finishes before the (lighter) env backward it shares the GPU with, so that the LONGER tail hides.
                # ONE side stream per process, device and priority: torch hands streams out of a pool round-robin and HIP multiplexes
                # them onto a few hardware queues -- the fourth / fifth NativeStep of a process used to get a stream that shares its
                # queue with the main stream, and its steps took 2.1 ms instead of 1.2 (tools/diag/degrade.py)
                key = (dev.index, bool(self.side_priority))
                if key not in _SIDE_STREAMS:
                    _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=-1 if self.side_priority else 0)
                self._side = _SIDE_STREAMS[key]
            side = self._side
            side.wait_stream(cur)                          # the previous step's Adam, the zero arena
        st_main, st_side = cur.cuda_stream, side.cuda_stream

        # ---- main: ground mesh, env projection + clipping, per-face set-up of the env pass (needs no texture) ----
        st = st_main
        nbv = m._bkg_verts.shape[0]
        ngv = m._ground_base.shape[0]
        key = (float(S_w), m.R_world._version, m.T_world._version, m.R_world.data_ptr())
        if self._env_verts is None or self._env_key != key:          # the sky dome is constant: written once
            self._env_verts = torch.empty(nbv + ngv, 3, device=dev)
            self._env_verts[:nbv] = ((m._bkg_verts * S_w) @ R_w + T_w)
            self._env_key = key
        env_verts = self._env_verts
        _lib.call('dbw_posed_mesh_fwd', _p(m._ground_base), ngv, _p(m.R_6d_ground), _p(m.T_ground), float(S_w), _p(R_w), _p(T_w),
                  env_verts.data_ptr() + nbv * 12, st)
        Te = TS * u_
        ce = (Te // decim) ** 2 * 3
        env_maps = torch.empty(2 * ce, device=dev)                    # [sky | ground]
        blk_maps = torch.empty(nb * (TS // decim_blocks) ** 2 * 3, device=dev)
        cfg_e = m.renderer_env._cfg(Fe, lds_aggregate=True, const_faces=m._n_bkg_faces)   # the sky dome's vertices are constants
        desc_e = m._env_map_desc if decim == 1 else m._env_map_desc_dec
        cl_e = ops.project_clip(env_verts, m._env_faces, R, T, Kmat, cfg_e.eps, cfg_e.z_clip, cfg_e.persp)
        lay_e = ops.hard_layout(cfg_e, None, desc_e)                  # 3: hard uv-fragments
        # image-shaped buffers of the step (env image, the two gradient images) live in the 8x8-tile planar layout of the fragments
        # (include/dbw_hip.h: image_layout 1) and the targets are tiled once: every wave access to them is one 256 B line
        env_state = ops._render_fwd_fused(cl_e['face_verts'].view(-1, 3, 3), cl_e, B, cfg_e, m._env_face_uvs, m._env_face_map, desc_e, env_maps,
                                          None, m.renderer_env._bg, lay_e, stage=1, img_tiled=True)
        target = self._tiled_target(imgs)

        # ---- side: textures first -- sigmoid (+ decimation to cell resolution); `sig` = the undecimated maps of the TV term -- because
        # the env pass on the main stream waits for them; then zero the gradients and the opacities (dbw.py:297-311) ----
        torch.cuda.set_stream(side)
        st = st_side
        tv_f = 1.0 if coarse else 0.1
        tv = float(w['tv']) * tv_f * rs if 'tv' in w else 0.0
        sets = []                                                      # the three texture tensors: one launch per pass over them
        for tex, d, out, wrap, sc, name in ((m.texture_bkg, decim, env_maps[:ce], 0, tv, 'texture_bkg'),
                                            (m.textures, decim_blocks, blk_maps, 1, tv, 'textures'),
                                            (m.texture_ground, decim, env_maps[ce:], 0, tv * tv_f, 'texture_ground')):
            n, h, ww, _ = tex.shape
            sig = torch.empty_like(tex) if d > 1 else out.view(tex.shape)
            sets.append(dict(texture=_p(tex), n=n, h=h, w=ww, decim=d, maps=_p(out), sig=_p(sig) if d > 1 else 0, wrap_x=wrap, tv_scale=sc,
                             grad_texture=_p(g[name]), _sig=sig))
        def launch(fn, which, *more):
            arr, k = _lib.texture_sets([{a: b for a, b in sets[i].items() if a[0] != '_'} for i in which])
            _lib.call(fn, arr, k, *more)
        launch('dbw_texture_prep_fwd_sets', (0, 1, 2), st)
        maps_ready = None
        if side is not cur:
            maps_ready = torch.cuda.Event()
            maps_ready.record(side)
        if zero_grad is not None:
            zero_grad()
        vals = torch.zeros(8, device=dev)                  # 0 rgb (filled on demand), 1 parsimony, 2 tv, 3 overlap; outlives the step
        noise, noise_scale = None, 0.0
        if m.opacity_noise and coarse:
            noise = m._noise_override if m._noise_override is not None else m._shared_randn_like(m.alpha_logit)
            noise_scale = float(m.opacity_noise)
        masked = fine or m.kill_blocks
        thresh = (0.5 if fine else 0.01) if masked else -1.0
        alpha, alpha_full = torch.empty(nb, device=dev), torch.empty(nb, device=dev)
        keep = torch.empty(nb, dtype=torch.int32, device=dev)
        _lib.call('dbw_block_alpha_fwd', _p(m.alpha_logit), _p(noise), noise_scale, thresh, nb, _p(alpha), _p(alpha_full), _p(keep), st)
        keep_p = _p(keep) if masked else 0

        # ---- main: the env pass ----
        torch.cuda.set_stream(cur)
        st = st_main
        if maps_ready is not None:
            cur.wait_event(maps_ready)
        p2f_e, bary_e, dists_e, img_e = ops._render_fwd_fused(cl_e['face_verts'].view(-1, 3, 3), cl_e, B, cfg_e, m._env_face_uvs, m._env_face_map,
                                                              desc_e, env_maps, None, m.renderer_env._bg, lay_e, stage=2, state=env_state,
                                                              img_tiled=True)

        # ---- side, next to the env pass: the blocks' vertices, their projection and the per-face set-up of the fg pass (boxes, face +
        # shading records, bins); the regularisers: value + gradient in one pass, weights folded into the kernels' scales
        # (dbw.py:373-405) -- they only need the parameters, opacities and maps prepared above ----
        torch.cuda.set_stream(side)
        st = st_side
        desc_f = m._block_map_desc_all if decim_blocks == 1 else m._block_map_desc_dec
        texbins = None if decim_blocks > 1 else (m._block_bin_base, m._block_bin_info, nb * m._bins_per_block)
        cfg_f = renderer._cfg(Ff, lds_aggregate=decim_blocks > 1, texbins=texbins)
        blk_verts = torch.empty(nb * nv, 3, device=dev)
        _lib.call('dbw_sq_blocks_fwd', _p(m.sq_eps), _p(m.S), _p(m.R_6d), _p(m.T), _p(m._trig), keep_p, 0, nb, nv, float(m.ratio_block_scene),
                  float(m.scale_min), float(S_w), _p(R_w), _p(T_w), _p(blk_verts), st)
        cl_f = ops.project_clip(blk_verts, m._block_faces_all, R, T, Kmat, cfg_f.eps, cfg_f.z_clip, cfg_f.persp)
        fa = None if fine else alpha                                   # one opacity per block = per texture map (alpha_len < 0)
        fg_state = ops.render_fwd_fused_mse(cl_f, B, cfg_f, m._block_face_uvs_all, m._block_face_map_all, desc_f, blk_maps, fa, renderer._bg,
                                            None, None, 0.0, stage=1, img_tiled=True)
        fg_ready = None
        if side is not cur:                                            # the fg pass only waits for its set-up, not for the regularisers
            fg_ready = torch.cuda.Event()                              # behind it (they run next to the fg forward)
            fg_ready.record(side)
        count = float(imgs.numel() if global_count is None else global_count)
        scale = float(w['rgb']) / count

        def fg_pass():
            # ---- main: the fg pass, ending in the composite + MSE ----
            torch.cuda.set_stream(cur)
            st = st_main
            if fg_ready is not None:
                cur.wait_event(fg_ready)                                   # blocks projected, per-face records, tile lists
            p2f, bary, dists, part, g_fg, g_env = ops.render_fwd_fused_mse(cl_f, B, cfg_f, m._block_face_uvs_all, m._block_face_map_all, desc_f,
                                                                          blk_maps, fa, renderer._bg, img_e, target, scale, stage=2,
                                                                          state=fg_state, img_tiled=True)
            return p2f, bary, dists, part, g_fg, g_env

        def regularisers():
            # ---- side: the regularisers, enqueued AFTER the main stream's fg pass: a stream that waits for an event of another stream was
            # observed to wait for everything that stream had been given by then (the fg pass started when the last regulariser ended, 15 us
            # after the env pass it really depends on) ----
            torch.cuda.set_stream(side)
            st = st_side
            if cfg_f.texbins is not None:          # (full-resolution maps: the record sub-ranges of this step's backward, from the last step's demand)
                cfg_f.bin_demand.prepare(cfg_f.texbins[2], ops.texbin_capacity(B, cfg_f.H, cfg_f.W, cfg_f.K, cfg_f.texbins[2]), dev)
            g_alpha_full = ops.ARENA.zeros(nb, torch.float32, dev)                                  # d / d alpha_full (parsimony, overlap)
            if 'parsimony' in w and coarse:
                _lib.call('dbw_sqrt_mean', _p(alpha_full), nb, 1e-6, float(w['parsimony']) * rs, vals.data_ptr() + 4, _p(g_alpha_full), st)
            if 'tv' in w:
                for t in sets:
                    t['_g_sig'] = torch.empty_like(t['_sig'])
                    t['sig'], t['grad_sig_out'], t['grad_sig'] = _p(t['_sig']), _p(t['_g_sig']), _p(t['_g_sig'])
                launch('dbw_tv_l2sq_sets', (0, 1, 2), vals.data_ptr() + 8, st)
            if 'overlap' in w and coarse:
                u = m._overlap_u_override if m._overlap_u_override is not None else torch.rand(nb, OVERLAP_N_POINTS, 3, device=dev)
                ws = ops.ARENA.zeros(nb * 18, torch.float32, dev)
                _lib.call('dbw_overlap_loss', _p(u), u.shape[1], _p(m.sq_eps), _p(m.S), _p(m.R_6d), _p(m.T), _p(alpha_full), nb,
                          float(m.ratio_block_scene), float(m.scale_min), OVERLAP_TEMPERATURE, OVERLAP_N_BLOCKS, float(w['overlap']) * rs,
                          vals.data_ptr() + 12, _p(g['sq_eps']), _p(g['S']), _p(g['R_6d']), _p(g['T']), _p(g_alpha_full), _p(ws), st)

            torch.cuda.set_stream(cur)
            st = st_main
            return g_alpha_full

        if self.regularisers_behind_fg:
            p2f, bary, dists, part, g_fg, g_env = fg_pass()
            g_alpha_full = regularisers()
        else:
            g_alpha_full = regularisers()
            p2f, bary, dists, part, g_fg, g_env = fg_pass()
        # ---- backward of the two passes (upstream gradient 1: nothing sits above this step), each followed by its tail of small
        # kernels (projection backward, pose / shape, textures, opacities) ----
        fg_out = {}

        def fg_backward(st, after_kernel=None, with_textures=True):
            g_blk_maps, g_fa, g_fvc = ops._fused_bwd(p2f, bary, dists, cl_f, m._block_face_uvs_all, m._block_face_map_all, desc_f, blk_maps, fa,
                                                     cfg_f, renderer._bg, 2, g_fg, B, None, after_kernel, img_tiled=True)
            fg_out.update(g_blk_maps=g_blk_maps, g_fa=g_fa)
            g_blk_verts = ops.project_clip_bwd(blk_verts, m._block_faces_all, R, T, Kmat, cl_f, g_fvc, cfg_f.eps, cfg_f.z_clip, cfg_f.persp)
            _lib.call('dbw_sq_blocks_bwd', _p(m.sq_eps), _p(m.S), _p(m.R_6d), _p(m.T), _p(m._trig), keep_p, 0, nb, nv, float(m.ratio_block_scene),
                      float(m.scale_min), float(S_w), _p(R_w), _p(g_blk_verts), _p(g['sq_eps']), _p(g['S']), _p(g['R_6d']), _p(g['T']), st)
            if with_textures:
                fg_textures(st)
            return g_blk_maps, g_fa, g_fvc, g_blk_verts

        def fg_textures(st):
            """The half of the fg tail that does not depend on the geometry half: block textures and opacities."""
            sets[1]['grad_maps'] = _p(fg_out['g_blk_maps'])
            launch('dbw_texture_prep_bwd_sets', (1,), st)
            _lib.call('dbw_block_alpha_bwd', _p(alpha), keep_p, _p(fg_out['g_fa']), ops.ALPHA_SPREAD, _p(g_alpha_full), nb, _p(g['alpha_logit']), st)
            if self.on_block_grads_ready is not None:
                self.on_block_grads_ready()                            # e.g. the data-parallel driver starts reducing the blocks' textures

        def env_backward(st):
            g_env_maps, _, g_fvc_e = ops._fused_bwd(p2f_e, bary_e, dists_e, cl_e, m._env_face_uvs, m._env_face_map, desc_e, env_maps, None, cfg_e,
                                                    m.renderer_env._bg, lay_e, g_env, B, None, img_tiled=True)
            g_env_verts = ops.project_clip_bwd(env_verts, m._env_faces, R, T, Kmat, cl_e, g_fvc_e, cfg_e.eps, cfg_e.z_clip, cfg_e.persp)
            _lib.call('dbw_posed_mesh_bwd', _p(m._ground_base), ngv, _p(m.R_6d_ground), _p(m.T_ground), float(S_w), _p(R_w),
                      g_env_verts.data_ptr() + nbv * 12, _p(g['R_6d_ground']), _p(g['T_ground']), st)
            sets[0]['grad_maps'], sets[2]['grad_maps'] = _p(g_env_maps[:ce]), _p(g_env_maps[ce:])
            launch('dbw_texture_prep_bwd_sets', (0, 2), st)
            return g_env_maps, g_fvc_e, g_env_verts

        if side is not cur:
            # Two kernels that each fill the GPU gain nothing from sharing it (together they take the sum of their solo times), so the
            # order only decides what hides behind what.  `seq`: the fg backward kernel first, alone; when it is done the env chain
            # starts on the main stream, and next to it run, on the side stream, the fg tail (and the bin reduction of the
            # full-resolution phases, a low-occupancy kernel) and -- data parallel -- the all-reduce of the blocks' textures, 83 % of
            # the gradient bytes; Adam follows the env tail on the main stream.  Otherwise both chains are enqueued at once
            side.wait_stream(cur)                                      # g_fg, g_env written
            kernel_done = torch.cuda.Event()
            seq = self.sequential_backward
            if seq is None:
                seq = m.world_size > 1 or decim_blocks == 1
            both = False
            if cfg_f.texbins is not None:
                # full-resolution maps: the texel gradients of the blocks leave the fg kernel as binned records and only reach
                # g_blk_maps in the bin reduction that follows it on the side stream -- `kernel_done` is recorded in front of that, so the
                # texture half of the tail must not move to the main stream (the concurrent order would read an incomplete gradient)
                both = (m.world_size == 1) if self.binned_concurrent is None else bool(self.binned_concurrent)
                seq = True
            torch.cuda.set_stream(side)
            keep_f = fg_backward(side.cuda_stream, lambda: kernel_done.record(side), with_textures=seq)
            torch.cuda.set_stream(cur)
            if seq and not both:
                cur.wait_event(kernel_done)
            keep_e = env_backward(st_main)
            if not seq:
                # both kernels were enqueued at once; the (lighter) env chain is done long before the fg kernel, so the main stream takes
                # the texture / opacity half of the fg tail while the side stream runs the geometry half
                cur.wait_event(kernel_done)
                fg_textures(st_main)
            cur.wait_stream(side)
        else:
            keep_f = fg_backward(st_main)
            keep_e = env_backward(st_main)
        m._alpha, m._alpha_full = alpha, alpha_full
        return LazyLosses(vals, part, scale, [k for k in w if k in _SLOT])


_SLOT = {'rgb': 0, 'parsimony': 1, 'tv': 2, 'overlap': 3}


class LazyLosses(dict):
    """The loss values of a native step.  Nothing is differentiated through them, so the reductions that produce them (the sum of the
    per-tile squared differences, the total) are only launched when somebody reads a value -- a logging tick, not every step."""

    def __init__(self, vals, part, scale, names):
        super().__init__()
        self._vals, self._part, self._scale, self._names, self._done = vals, part, scale, names, False

    def _finish(self):
        if not self._done:
            v = self._vals
            v[0] = self._part.sum() * self._scale
            v[4] = v[:4].sum()
            for k in self._names:
                dict.__setitem__(self, k, v[_SLOT[k]])
            dict.__setitem__(self, 'total', v[4])
            self._part, self._done = None, True

    def __getitem__(self, k):
        self._finish()
        return dict.__getitem__(self, k)

    def items(self):
        self._finish()
        return dict.items(self)

    def keys(self):
        self._finish()
        return dict.keys(self)

    def values(self):
        self._finish()
        return dict.values(self)

    def __iter__(self):
        self._finish()
        return dict.__iter__(self)

    def __len__(self):
        self._finish()
        return dict.__len__(self)

    def __contains__(self, k):
        self._finish()
        return dict.__contains__(self, k)

    def get(self, k, default=None):
        self._finish()
        return dict.get(self, k, default)

    def copy(self):
        self._finish()
        return dict(self)

    def __repr__(self):
        self._finish()
        return dict.__repr__(self)

    def __reduce__(self):                       # pickling / deepcopy: a plain dict of the finished values
        self._finish()
        return (dict, (dict(self),))
