"""The optimisation iteration behind ONE call into libdbw_hip.so (include/dbw_hip.h: dbw_train_step_*).

`native_step.NativeStep` issues the ~33 launches of an iteration one by one through ctypes: 0.48 ms of host time for ~0.1 ms of kernels at
the reference's batch size (4 views, configs/dtu/default.yml:28; src/trainer.py:137-147).  `CStep` hands the library a description of
the model once (pointers into the flat parameter / gradient buffers, the constant tables, the phase's constants: a *plan*, one per
training phase) and then makes one call per iteration; the library enqueues ~16 launches on two streams (csrc/train_step.hip).

Same mathematics as NativeStep (which is checked against the autograd iteration, which is checked against the oracle):
tests/test_gpu_model.py::test_c_step_equals_native_step holds them to each other for every fuse mask.  Scope = NativeStep's: the decoupled
training render with MSE + parsimony + TV + overlap on a sync-free model; anything else -> `supported()` is False."""
import ctypes
import os
import warnings
import weakref

import torch

from . import _lib, ops
from .dbw import OVERLAP_N_BLOCKS, OVERLAP_N_POINTS, OVERLAP_TEMPERATURE

_p = ops._ptr
FUSE_ALL = 127          # include/dbw_hip.h: dbw_step_desc.fuse
_SIDE_STREAMS = {}
_OFF = {'alpha': 0, 'alpha_full': 1, 'keep': 2, 'losses': 3, 'arena_begin': 4, 'arena_end': 5, 'g_fg': 6, 'g_env': 7, 'env_img': 8, 'blk_verts': 9,
        'loss_part': 10, 'p2f_env': 11, 'uvj_env': 12}


def side_stream(dev, priority=True):
    """ONE side stream per process, device and priority (torch hands streams out of a pool round-robin and HIP multiplexes them onto a few
    hardware queues: see native_step.py)."""
    key = (dev.index, bool(priority))
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=-1 if priority else 0)
    return _SIDE_STREAMS[key]


class StepLosses(dict):
    """Loss values of a C step: five floats the step left on the device (and, with read_losses, copied to host memory by the step itself:
    ONE device -> host copy and one wait instead of six `.item()`, src/trainer.py:143).  Reads like the dict of 0-dim tensors the model
    returns; `host()` gives plain floats.  perceptual: the value of the caller's network term (a device scalar), added to the total."""
    NAMES = ('rgb', 'parsimony', 'tv', 'overlap', 'total')

    def __init__(self, step, dev_vals, names, pending_host, perceptual=None):
        super().__init__()
        self._step, self._pending, self._perceptual = step, pending_host, perceptual
        for k in names:                     # (the model's order: rgb, perceptual, parsimony, tv, overlap, total)
            if k == 'perceptual':
                dict.__setitem__(self, k, perceptual)
            elif k in self.NAMES:
                dict.__setitem__(self, k, dev_vals[self.NAMES.index(k)])
        dict.__setitem__(self, 'total', dev_vals[4] if perceptual is None else dev_vals[4] + perceptual)

    def host(self):
        """-> {name: float}; waits for the step's own copy when it made one, else reads the device values."""
        if self._pending:
            out = (ctypes.c_float * 5)()
            _lib.call('dbw_train_step_losses', self._step._plan_handle(), ctypes.cast(out, ctypes.c_void_p))
            res = {k: float(out[self.NAMES.index(k)]) for k in self if k in self.NAMES}
            if self._perceptual is not None:
                res['perceptual'] = float(self._perceptual)
                res['total'] += res['perceptual']
            return {k: res[k] for k in self}
        v = torch.stack([dict.__getitem__(self, k) for k in self]).tolist()
        return dict(zip(self.keys(), v))


class CStep:
    def __init__(self, model, params, opt_state, fuse=FUSE_ALL, max_views=None):
        """params: parallel.FlatParams; opt_state: (exp_avg, exp_avg_sq) flat tensors."""
        self.m, self.params, self.opt = model, params, opt_state
        self.fuse = int(fuse)
        self.max_views = max_views
        self.side_priority = True
        self.backward_order = None          # None: by configuration (see _plan_for), 0 / 1 force
        self.binned_concurrent = None
        self.use_side_stream = True         # False: everything in order on the caller's stream
        self.own_side_stream = False        # True: the env chain on a (high-priority) torch stream of the caller instead of the plan's own
        self.serial_setup_max_views = 12    # include/dbw_hip.h: up to that many views the blocks' set-up stays on the main stream
        # how the step's streams wait for each other: polled words in device memory (include/dbw_hip.h: sync_events), or HIP events
        self.sync_events = os.environ.get('DBW_STEP_EVENTS', '0') not in ('', '0')
        self.read_losses = False            # copy the five loss values to host memory in every step (StepLosses.host())
        self._plans = {}                    # key -> (handle, workspace tensor, keep-alive list)
        self._pid = os.getpid()
        self._cur = None
        self._target, self._target_key = None, None
        self._env_key = None
        self._inp = _lib.StepInputs()
        self._clean_plan = None             # handle of the plan whose arena a caller cleared itself (arena_cleaned_by_caller)
        self._calls = 0                     # default counter of the step's random numbers (callers with a step count of their own pass it)
        self._voided_seen = {}              # plan handle -> voided runs already reported

    # ---- what the plan covers -----------------------------------------------------------------------------------------------------------
    def supported(self):
        m, w = self.m, self.m.loss_weights
        r = m.renderer
        # (the perceptual term: a network of the caller's evaluated between two phases of the step -- needs the env layer inside the fg pass)
        perceptual_ok = 'perceptual' not in w or (m.perceptual_fn is not None and (self.fuse & 18) == 18)
        # (clip_inside = False, the sigmoid opacity, travels as a negative sigma that dbw_train_step_create refuses: the autograd path has it)
        ok = (m.decouple_rendering and m.sync_free and 'rgb' in w and perceptual_ok and r.detach_bary and r.faces_per_pixel > 1
              and r.clip_inside and getattr(m.renderer_fine, 'clip_inside', True) and getattr(m, 'default_criteria', True)
              and r.cam_name == 'perspective' and m.blocks_n_faces < (1 << 20) and m.n_blocks + 2 < (1 << 11) and m.n_blocks <= 64
              and ops.FUSED_FORWARD and ops.FUSED_BACKWARD and ops.TILED_FRAGMENTS and ops.UV_FRAGMENTS and ops.HARD_UV_FRAGMENTS
              and ops.COARSE_BINS and ops.TEXTURE_BINS)
        lib = _lib.load()
        return bool(ok and hasattr(lib, 'dbw_train_step_run'))

    def _plan_handle(self):
        return self._cur[0]

    def close(self):
        # (a fork()ed child -- multiprocessing's manager / resource-sharer processes -- inherits these objects without the GPU context they
        # belong to: only the process that made the plans destroys them)
        if os.getpid() == self._pid:
            for handle, *_ in self._plans.values():
                _lib.load().dbw_train_step_destroy(handle)
        self._plans, self._cur = {}, None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _plan_for(self, inp, B, defer=False):
        m = self.m
        coarse = m.is_live('coarse_learning')
        decim = int(m.decim_factor) if m.is_live('decimate_txt') else 1
        decim_blocks = decim if coarse else 1
        renderer = m.renderer if coarse else m.renderer_fine
        dev = inp['imgs'].device
        Kt = renderer.cameras.K
        seq = self.backward_order
        if seq is None:
            # data parallel: the fg kernel first and alone (its texture gradient is then final early: the all-reduce of that slice runs next to
            # the env chain); one GPU: both backward kernels at once.  (Texture bins: the library adds the order of its own, see binned_concurrent)
            seq = 1 if (m.world_size > 1 and not defer) else 0
        both = self.binned_concurrent
        if both is None:
            both = m.world_size == 1
        # ONE plan (and one workspace) per phase / configuration, whatever the batch size: a plan runs any B up to its max_views, so a
        # ragged last mini-batch or a smaller shard reuses the plan of the full batch; only a LARGER batch replaces it
        key = (coarse, decim, decim_blocks, m.world_size, self.fuse, int(seq), int(bool(both)), Kt.data_ptr(), m.R_world.data_ptr(),
               m.R_world._version, m.T_world._version, float(m.S_world),
               self.params.flat.data_ptr(), tuple(sorted(m.loss_weights.items())), float(m.opacity_noise or 0.0), bool(m.kill_blocks),
               int(self.serial_setup_max_views), bool(self.sync_events), bool(defer))
        have = self._plans.get(key)
        if have is not None and have[3] >= B:
            self._cur = have
            return self._cur
        max_views = max(int(self.max_views or 0), B)
        if have is not None:
            if os.getpid() == self._pid:
                with torch.cuda.device(dev):
                    torch.cuda.synchronize(dev)
                    _lib.load().dbw_train_step_destroy(have[0])
            if self._clean_plan == have[0]:
                self._clean_plan = None
            del self._plans[key]
        w = m.loss_weights
        rs = 1.0 / m.world_size
        fine = not coarse
        S_w, R_w, T_w = m._world_consts()
        TS, u_, nb = m.txt_size, m.txt_bkg_upscale, m.n_blocks
        keep = []                                   # tensors the plan points into
        d = _lib.StepDesc()
        H, W = m.img_size
        d.H, d.W, d.faces_per_pixel, d.max_views = H, W, int(renderer.faces_per_pixel), max_views
        d.n_blocks, d.block_nv, d.block_nf = nb, m._block_nv, m.BNF
        d.n_sky_verts, d.n_ground_verts = m._bkg_verts.shape[0], m._ground_base.shape[0]
        d.n_sky_faces, d.n_ground_faces = m._n_bkg_faces, m._n_ground_faces
        d.txt_size, d.env_txt_size, d.decim_env, d.decim_blocks, d.coarse = TS, TS * u_, decim, decim_blocks, int(coarse)
        cfg = renderer._cfg(nb * m.BNF)
        d.sigma, d.blur_radius = float(cfg.sigma), float(cfg.blur)
        d.z_clip, d.cam_eps, d.perspective_correct = float(cfg.z_clip or 0.0), float(cfg.eps), int(cfg.persp)
        for i in range(3):
            d.bg_fg[i], d.bg_env[i] = float(renderer.background_color[i]), float(m.renderer_env.background_color[i])
        d.S_world, d.ratio_block_scene, d.scale_min = float(S_w), float(m.ratio_block_scene), float(m.scale_min)
        d.opacity_noise = float(m.opacity_noise) if (m.opacity_noise and coarse) else 0.0
        masked = fine or m.kill_blocks
        d.mask_threshold = (0.5 if fine else 0.01) if masked else -1.0
        tv_f = 1.0 if coarse else 0.1
        # (deferred texture gradients: the TV gradient is added on every rank BEHIND the all-reduce -- full weight in the kernels, the reported
        # value scaled instead, include/dbw_hip.h: tv_value_scale)
        tv = float(w['tv']) * tv_f * (1.0 if defer else rs) if 'tv' in w else 0.0
        d.tv_value_scale = rs if defer else 1.0
        d.w_rgb = float(w['rgb'])
        d.w_parsimony = float(w['parsimony']) * rs if ('parsimony' in w and coarse) else 0.0
        d.w_tv_bkg, d.w_tv_blocks, d.w_tv_ground = tv, tv, tv * tv_f
        d.w_overlap = float(w['overlap']) * rs if ('overlap' in w and coarse) else 0.0
        d.overlap_points, d.overlap_temperature, d.overlap_n_blocks = OVERLAP_N_POINTS, OVERLAP_TEMPERATURE, OVERLAP_N_BLOCKS
        # constant tables
        Kmat = Kt[0].to(dev).contiguous()
        nbv = m._bkg_verts.shape[0]
        env_verts = torch.empty(nbv + m._ground_base.shape[0], 3, device=dev)
        env_verts[:nbv] = ((m._bkg_verts * S_w) @ R_w + T_w)
        desc_e = m._env_map_desc if decim == 1 else m._env_map_desc_dec
        desc_f = m._block_map_desc_all if decim_blocks == 1 else m._block_map_desc_dec
        keep += [R_w, T_w, Kmat, env_verts]
        d.R_world, d.T_world, d.Kmat = _p(R_w), _p(T_w), _p(Kmat)
        d.ground_base, d.env_verts = _p(m._ground_base), _p(env_verts)
        d.env_faces, d.env_face_uvs, d.env_face_map, d.env_map_desc = _p(m._env_faces), _p(m._env_face_uvs), _p(m._env_face_map), _p(desc_e)
        d.trig, d.block_faces, d.block_face_uvs = _p(m._trig), _p(m._block_faces_all), _p(m._block_face_uvs_all)
        d.block_face_map, d.block_map_desc = _p(m._block_face_map_all), _p(desc_f)
        if decim_blocks == 1:
            d.block_bin_base, d.block_bin_info, d.n_bins = _p(m._block_bin_base), _p(m._block_bin_info), nb * m._bins_per_block
        # parameters and gradients (views of the flat buffers)
        g = {n: m.get_parameter(n).grad for n, _, _ in self.params.names}
        for f, n in (('sq_eps', 'sq_eps'), ('S', 'S'), ('R6', 'R_6d'), ('T', 'T'), ('alpha_logit', 'alpha_logit'), ('R6_ground', 'R_6d_ground'),
                     ('T_ground', 'T_ground'), ('texture_bkg', 'texture_bkg'), ('texture_ground', 'texture_ground'), ('textures', 'textures')):
            setattr(d, f, _p(m.get_parameter(n)))
            setattr(d, 'g_' + f, _p(g[n]))
        P = self.params
        d.flat_param, d.flat_grad, d.exp_avg, d.exp_avg_sq = _p(P.flat), _p(P.grad), _p(self.opt[0]), _p(self.opt[1])
        d.group_end[0], d.group_end[1] = P.bounds[0][1], P.bounds[1][1]
        d.small_grads, d.n_small_grads = _p(P.grad), P.bounds[0][1]
        d.fuse, d.backward_order, d.binned_concurrent = self.fuse, int(seq), int(bool(both))
        d.serial_setup_max_views = int(self.serial_setup_max_views)
        d.seed = int(getattr(m, '_rng_seed', 227391)) & 0xffffffffffffffff
        d.sync_events = int(bool(self.sync_events))
        lib = _lib.load()
        nbytes = lib.dbw_train_step_workspace_bytes(ctypes.byref(d))
        if nbytes == 0:
            raise RuntimeError(f'dbw_train_step_workspace_bytes: {lib.dbw_last_error().decode()}')
        wsb = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        off = (-wsb.data_ptr()) % 256
        wsb = wsb[off:off + nbytes]
        with torch.cuda.device(dev):
            handle = lib.dbw_train_step_create(ctypes.byref(d), wsb.data_ptr(), nbytes)
        if not handle:
            raise RuntimeError(f'dbw_train_step_create: {lib.dbw_last_error().decode()}')
        self._plans[key] = self._cur = (handle, wsb, keep + [d], max_views)
        return self._cur

    def view(self, name, dtype=torch.float32, numel=None):
        """A buffer of the current plan's workspace as a tensor (tests, diagnostics, the opacities a logging tick reads)."""
        handle, wsb = self._cur[:2]
        off = _lib.load().dbw_train_step_offset(handle, _OFF[name])
        t = wsb[off:].view(dtype)
        return t if numel is None else t[:numel]

    def arena(self):
        handle, wsb = self._cur[:2]
        lib = _lib.load()
        return wsb[lib.dbw_train_step_offset(handle, 4):lib.dbw_train_step_offset(handle, 5)]

    def arena_cleaned_by_caller(self):
        """The caller has just enqueued a clear of the CURRENT plan's zero arena (it ran Adam itself with zero = self.arena()): the next run
        of THAT plan skips its own fill.  Tracked per plan -- another plan's arena is none the cleaner for it."""
        self._clean_plan = self._cur[0]

    def void_flag(self):
        """One float of the current plan's workspace: != 0 behind a run whose cross-stream wait gave up (include/dbw_hip.h:
        dbw_train_step_void_flag_offset).  The plan's own Adam launches read it; a data-parallel caller sums it over the ranks with the
        gradients and hands it to ops.adam_step_groups_(skip=...), so that every rank skips the update of a step one of them voided."""
        handle, wsb = self._cur[:2]
        off = _lib.load().dbw_train_step_void_flag_offset(handle)
        return wsb[off:off + 4].view(torch.float32)

    def voided_runs(self):
        """Runs of the current plan that voided themselves (host-side counter, no synchronisation); > 0: the plan now runs on events."""
        return int(_lib.load().dbw_train_step_voided_runs(self._cur[0]))

    _WAITS = ('prologue', 'tile order', 'fg forward', 'regularisers', 'bin layout', 'fg backward kernel', "blocks' textures", 'env chain', 'texture preparation')

    def last_timeout(self):
        """(name of the cross-stream counter the first poll that gave up was waiting on, value wanted, value last seen), or None.  Synchronises."""
        out = (ctypes.c_int * 3)()
        _lib.call('dbw_debug_train_step_last_timeout', self._cur[0], out)
        if out[0] < 0:
            return None
        seen, asked = (ctypes.c_uint * 12)(), (ctypes.c_uint * 12)()
        _lib.call('dbw_debug_train_step_counters', self._cur[0], seen, asked)
        return (self._WAITS[out[0]] if out[0] < len(self._WAITS) else str(out[0]), out[1], out[2],
                {n: (int(seen[i]), int(asked[i])) for i, n in enumerate(self._WAITS)})          # counter: (seen then, enqueued so far)

    def _report_voided(self):
        h = self._cur[0]
        n = self.voided_runs()
        if n > self._voided_seen.get(h, 0):
            self._voided_seen[h] = n
            warnings.warn(f'dbw C step: a cross-stream wait gave up ({n} run(s) so far; first: {self.last_timeout()}): the step voided itself -- no '
                          'parameter was updated -- and the plan orders its streams through HIP events from now on', RuntimeWarning)

    def kernel_times(self, inp, global_count=None, reps=5, alone=False):
        """ms of the four big kernels INSIDE a step (env pass, fg pass, fg backward, env backward), everything that shares the GPU with them
        in a real step running next to them: HIP events recorded by the step itself on the streams the kernels run on; averaged over
        `reps` steps without Adam (the parameters do not move).  alone: the same launches in order on ONE stream -- every kernel by
        itself on the GPU, in exactly the form the step launches it.  -> {'env_fwd', 'fg_fwd', 'fg_bwd', 'env_bwd'} (env_fwd is 0 when
        the env layer is folded into the fg pass)"""
        was = self.use_side_stream
        if alone:
            self.use_side_stream = False
        acc = [0.0] * 4
        out = (ctypes.c_float * 4)()
        try:
            self(inp, global_count)
            _lib.call('dbw_train_step_profile', self._cur[0], 1)
            for _ in range(reps):
                self(inp, global_count)
                _lib.call('dbw_train_step_kernel_times', self._cur[0], ctypes.cast(out, ctypes.c_void_p))
                acc = [a + float(o) for a, o in zip(acc, out)]
        finally:
            _lib.call('dbw_train_step_profile', self._cur[0], 0)
            self.use_side_stream = was
        return dict(zip(('env_fwd', 'fg_fwd', 'fg_bwd', 'env_bwd'), [a / reps for a in acc]))

    def map_grads(self):
        """The gradient of the prepared texture maps (blocks, sky, ground) of the current plan: one range of its zero arena.  With
        `defer_textures` this -- not the texture gradient: 1 / 64 of its bytes while the maps are decimated 8 x 8 -- is what data-parallel
        ranks sum, next to the small gradients at the head of the flat buffer."""
        lib = _lib.load()
        a, b = lib.dbw_train_step_offset(self._cur[0], 13), lib.dbw_train_step_offset(self._cur[0], 14)
        return self._cur[1][a:b].view(torch.float32)

    def finish(self, adam=None):
        """Behind a step with `defer_textures` and the caller's all-reduce: the backward of the texture preparation (+ the TV gradient), then
        Adam (`adam` as in __call__), which clears the zero arena."""
        a = self._inp
        if adam is not None:
            step, lrs, betas, eps = adam
            a.with_adam, a.adam_step = 1, int(step)
            a.lr[0], a.lr[1], a.beta1, a.beta2, a.adam_eps = float(lrs[0]), float(lrs[1]), float(betas[0]), float(betas[1]), float(eps)
        else:
            a.with_adam = 0
        dev = self.params.flat.device
        with torch.cuda.device(dev):
            _lib.call('dbw_train_step_finish', self._cur[0], ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream)

    def sync_timeouts(self):
        """Cross-stream waits of the current plan that gave up (never, in a healthy process); synchronises the device."""
        with torch.cuda.device(self.params.flat.device):
            return _lib.load().dbw_train_step_sync_timeouts(self._cur[0])

    def wait_blocks_ready(self, stream):
        """`stream` (a torch stream) waits until the blocks' texture gradient of the last step is final (data parallel: the early slice of
        the all-reduce)."""
        with torch.cuda.device(self.params.flat.device):
            _lib.call('dbw_train_step_wait_blocks_ready', self._cur[0], stream.cuda_stream)

    # ---- one iteration ------------------------------------------------------------------------------------------------------------------
    def __call__(self, inp, global_count=None, adam=None, tiled_target=True, defer_textures=False, rng_step=None):
        """Enqueue forward + backward (+ Adam when `adam` = (step, (lr, lr_texture), (beta1, beta2), eps)) of one iteration on this rank's
        views.  -> StepLosses.  tiled_target: keep the targets in the tile-planar layout across steps while the SAME tensor comes back
        (resident training views); a fresh mini-batch is tiled by the step itself.  rng_step: the counter the step's random numbers are
        keyed on -- the optimisation-step count of the caller (ShardedTrainStep.n_steps: identical on every rank, checkpointed); default:
        the number of calls made through this object."""
        m = self.m
        imgs = inp['imgs']
        dev = imgs.device
        m._ensure_cameras(inp)
        B = imgs.shape[0]
        handle, wsb = self._plan_for(inp, B, defer_textures)[:2]
        R, T = inp['R'].float().contiguous(), inp['T'].float().contiguous()
        imgs = ops._chk(imgs, torch.float32, 'imgs')
        a = self._inp
        tiled = 0
        if tiled_target:
            src = self._target_key[0]() if self._target_key is not None else None
            if src is inp['imgs'] and self._target_key[1] == inp['imgs']._version:
                imgs, tiled = self._target, 1
            elif getattr(self, '_seen', None) is inp['imgs']:          # the second step on the same tensor: worth keeping tiled
                self._target, self._target_key = ops.tile_image(imgs), (weakref.ref(inp['imgs']), inp['imgs']._version)
                imgs, tiled = self._target, 1
            self._seen = inp['imgs']
        a.imgs, a.imgs_tiled, a.R, a.T, a.B = imgs.data_ptr(), tiled, R.data_ptr(), T.data_ptr(), B
        a.global_count = float(inp['imgs'].numel() if global_count is None else global_count)
        nz, u = m._noise_override, m._overlap_u_override
        a.noise_override, a.overlap_u_override = _p(nz), _p(u)
        if adam is not None:
            step, lrs, betas, eps = adam
            a.with_adam, a.adam_step = 1, int(step)
            a.lr[0], a.lr[1], a.beta1, a.beta2, a.adam_eps = float(lrs[0]), float(lrs[1]), float(betas[0]), float(betas[1]), float(eps)
        else:
            a.with_adam = 0
        a.defer_textures = int(bool(defer_textures))
        a.read_losses = int(self.read_losses)
        a.arena_is_clean = int(self._clean_plan is not None and self._clean_plan == handle)
        if a.arena_is_clean:
            self._clean_plan = None
        a.rng_step = int(self._calls if rng_step is None else rng_step) & 0xffffffffffffffff
        self._calls += 1
        cur = torch.cuda.current_stream(dev)
        # the env chain and the regularisers: streams of the plan (NULL), a torch stream of this process, or the caller's own stream
        side = side_stream(dev, self.side_priority).cuda_stream if (self.use_side_stream and self.own_side_stream) else 0
        a.single_stream = int(not self.use_side_stream)
        w = m.loss_weights
        perceptual = None
        a.phase, a.rec_out, a.grad_rec = 0, 0, 0
        with torch.cuda.device(dev):
            if 'perceptual' in w:
                # dbw.py:369-371: weight * (1 | 0.1 after the coarse phase) * LPIPS(imgs, rec); under view-sharded data parallelism times this
                # rank's share of the global batch (the gradients of the ranks are summed).  Phase 1 leaves the composite, the caller's
                # network runs on it (torch autograd, MIOpen convolutions), phase 2 takes d term / d rec
                rec = torch.empty(B, 3, m.img_size[0], m.img_size[1], device=dev)       # (a fresh leaf every step: autograd owns it)
                a.phase, a.rec_out = 1, rec.data_ptr()
                _lib.call('dbw_train_step_run', handle, ctypes.byref(a), cur.cuda_stream, side)
                with torch.enable_grad():
                    leaf = rec.requires_grad_(True)
                    perceptual = m._perceptual_term(inp['imgs'], leaf, m.is_live('coarse_learning'), inp.get('view_ids'))
                    g_rec, = torch.autograd.grad(perceptual, leaf)
                perceptual, g_rec = perceptual.detach(), g_rec.contiguous()
                a.phase, a.rec_out, a.grad_rec = 2, 0, g_rec.data_ptr()
                self._keep_rec = (rec, g_rec)
            _lib.call('dbw_train_step_run', handle, ctypes.byref(a), cur.cuda_stream, side)
        self._keep = (imgs, R, T, nz, u)            # inputs stay referenced until the next call has been enqueued behind this one
        self._report_voided()
        nb = m.n_blocks
        m._alpha, m._alpha_full = self.view('alpha', numel=nb), self.view('alpha_full', numel=nb)
        return StepLosses(self, self.view('losses', numel=5), list(w), bool(self.read_losses), perceptual)
