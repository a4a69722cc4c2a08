"""View-sharded data parallelism: one process per GPU, identical replicated parameters, each rank renders
its own views, ONE in-place RCCL all-reduce (sum, fp32) of a single flat gradient buffer per optimisation step, then the
same fused Adam update on every rank.  No other collective exists on the path (views are independent units).

The reference has no live distributed code (dead `DDPCust`, src/model/__init__.py:44-53); this is the MI355X design for
what `trainer.py:137-147` does on one GPU: zero_grad -> model(inp) -> total.backward() -> optimizer.step()
(Adam, two groups: names starting with 'texture' use their own lr, optimizer.py:9-15)."""
import torch
import torch.distributed as dist

from . import ops


def shard_views(n_views, world_size, rank):
    """Contiguous ceil-split of view indices: 49 views over 8 ranks -> 7,6,6,6,6,6,6,6."""
    base, extra = divmod(n_views, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class FlatParams:
    """Re-homes the model's parameters into two contiguous fp32 buffers (group 0: pose/shape/opacity, group 1: names
    starting with 'texture') and pre-binds each `.grad` to a view of one flat gradient buffer, so that backward
    accumulates straight into the buffer the all-reduce and the fused Adam consume (no gather/scatter copies)."""

    def __init__(self, model):
        named = list(model.named_parameters())
        groups = [[(n, p) for n, p in named if not n.startswith('texture')], [(n, p) for n, p in named if n.startswith('texture')]]
        dev = named[0][1].device
        sizes = [sum(p.numel() for _, p in g) for g in groups]
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.flat)
        self.bounds = [(0, sizes[0]), (sizes[0], sizes[0] + sizes[1])]
        self.names = []
        off = 0
        for g in groups:
            for n, p in g:
                k = p.numel()
                self.flat[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                self.names.append((n, off, k))
                off += k

    def zero_grad(self):
        self.grad.zero_()


class ShardedTrainStep:
    """One optimisation step per call: the C step (c_step.py: one call into the library per iteration) where it covers the model's
    configuration, else the launch-by-launch native step, else the autograd iteration; the gradient sum over the ranks; fused Adam.
    (Replaying an iteration from a hipGraph was built in rounds 2-3 and measured slower than the C step's plain enqueues -- a graph node
    costs what a launch costs once the host is out of the way, profiles/r03_experiments.md -- and is gone.)"""

    def __init__(self, model, lr=5e-3, lr_texture=5e-2, betas=(0.9, 0.999), eps=1e-8, process_group=None, adam_fn=None,
                 seed=None, use_native=True, overlap_allreduce=None, early_param='textures', use_c_step=True, fuse=None):
        self.model, self.pg = model, process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        model.world_size, model.rank = self.world_size, self.rank
        self.params = FlatParams(model)
        self.lrs, self.betas, self.eps = (lr, lr_texture), betas, eps
        self.exp_avg = torch.zeros_like(self.params.flat)
        self.exp_avg_sq = torch.zeros_like(self.params.flat)
        self.n_steps = 0
        self.adam_fn = adam_fn or ops.adam_step_
        # the iteration without autograd (native_step.py: same kernels, 31 launches instead of 66) whenever the model is the HIP
        # DifferentiableBlocksWorld in a configuration it covers; the autograd path otherwise
        self.native = None
        if use_native and hasattr(model, 'loss_weights') and hasattr(model, 'renderer_env'):
            from .native_step import NativeStep
            self.native = NativeStep(model, self.params)
            self.native.on_block_grads_ready = self.start_early_allreduce
        # ... and the same iteration behind ONE call into the library (c_step.py: ~16 launches enqueued from C, fused set-up / tail
        # kernels, the random numbers drawn inside them, Adam inside the call on one GPU) where the library has it; the launch-by-launch
        # native step stays as the form the C step is tested against
        self.cstep = None
        if use_c_step and self.native is not None and self.params.flat.is_cuda:
            from .c_step import CStep, FUSE_ALL
            self.cstep = CStep(model, self.params, (self.exp_avg, self.exp_avg_sq), fuse=FUSE_ALL if fuse is None else fuse)
        # Overlapped all-reduce: the blocks' texture gradient (`early_param`: 83 % of the buffer at config 2) is final long before the
        # step ends -- the env chain finishes last -- so its slice is reduced on the communicator's own stream while the env backward
        # still runs, and only the rest (1.6 MB) after the step.  Same collectives in the same order on every rank, whatever path a
        # rank's step took.  Default: on with RCCL ('nccl'), off elsewhere (the gloo tests switch it on explicitly).
        self._early_range = next(((off, off + k) for n, off, k in self.params.names if n == early_param), None)
        if overlap_allreduce is None:
            overlap_allreduce = dist.is_initialized() and dist.get_backend(process_group) == 'nccl' and self.world_size > 1
        self.overlap_allreduce = bool(overlap_allreduce) and dist.is_initialized() and self._early_range is not None
        self._early_work, self._early_done = None, False
        self._coalesce, self._coalesce_verified, self._coalesce_error = True, False, None
        self._coalesce_backends = ('nccl',)        # (tests add 'gloo' to run the trial logic on CPU tensors)
        self.allreduce_path = None
        self.defer_textures = True        # data parallel through the C step: sum the prepared maps' gradient, not the textures' (_c_iteration)
        if dist.is_initialized() or seed is not None:
            # identical noise / overlap samples on every rank: same seed for the default generator everywhere
            s = torch.tensor([seed if seed is not None else 0], dtype=torch.int64)
            if seed is None:
                s.random_()
            if dist.is_initialized():
                s = s.to(self.params.flat.device if dist.get_backend(process_group) == 'nccl' else 'cpu')
                dist.broadcast(s, src=0, group=process_group)
            torch.manual_seed(int(s.item()))
            model._rng_seed = int(s.item())           # the C step's counter-based generator: the same key on every rank
        else:
            # one process, no seed given: the key follows torch's own seed (the trainer's torch.manual_seed(cfg seed)), not a constant
            model._rng_seed = int(torch.initial_seed()) & 0xffffffffffffffff

    def _global_count(self, imgs, global_count):
        """Number of image elements in the GLOBAL batch (MSE is a mean over all views of all ranks, dbw.py:367).  Callers that
        know how the batch is split pass it (Trainer: from shard_views and the batch index, identical on every rank; bench: local x
        world_size) -- no collective, no host sync.  Without it the local counts are all-reduced on EVERY call: nothing is cached,
        because ranks whose shards differ in size must issue the same sequence of collectives."""
        if global_count is not None:
            return float(global_count)
        if self.world_size == 1:
            return float(imgs.numel())
        gloo = dist.get_backend(self.pg) != 'nccl'
        t = torch.tensor([float(imgs.numel())], device='cpu' if gloo else imgs.device)
        dist.all_reduce(t, group=self.pg)
        return float(t.item())

    def __call__(self, inp, labels=None, global_count=None):
        """One optimisation step on this rank's shard of views (possibly EMPTY: the rank then only contributes its share of the
        view-independent regularisers and still takes part in the all-reduce); returns the (local) loss dict (device tensors).
        `n_steps` -- Adam's bias-correction count, the counter of the step's random numbers, part of a checkpoint -- advances with every
        CALL: a step the C plan voided (a cross-stream wait gave up, c_step.py: RuntimeWarning; never in a healthy process) moved no
        parameter and no moment but still counts, on every rank alike -- the host learns of it a run later and cannot know how many runs
        it had enqueued in between, and a bias correction that is one step ahead is harmless where replicas that disagree are not."""
        self.model._global_count = self._global_count(inp['imgs'], global_count)
        dirty = None
        if self.cstep is not None and self.model.training and inp['imgs'].shape[0] > 0 and self._fused_adam() and self.cstep.supported():
            return self._c_iteration(inp)
        # gradients accumulate into the preallocated flat buffer, so nothing carved out of the zero arena outlives the
        # iteration: all zero-initialised scratch of the step comes from one buffer cleared by one launch
        ops.ARENA.enabled = True
        try:
            ops.ARENA.begin_step(self.params.flat.device)
            native = self.native is not None and self.model.training and inp['imgs'].shape[0] > 0 and self.native.supported()
            if native:
                with torch.no_grad():         # (zeroes the gradient buffer itself: on its side stream, off the critical path)
                    losses = self.native(inp, self.model._global_count, zero_grad=self.params.zero_grad)
            else:
                self.params.zero_grad()
                losses = self.model(inp, labels)
                losses['total'].backward()
            dirty = ops.ARENA.end_step(self.params.flat.device) if (native and self._fused_adam()) else None
        finally:
            ops.ARENA.enabled = False
        if not native:
            losses = {k: v.detach() for k, v in losses.items()}    # logging values only: do not keep the autograd graph alive
        if self.world_size > 1 or self.overlap_allreduce:     # (a one-rank group with the overlap forced on: the API smoke test)
            self.allreduce_gradients()
        self.n_steps += 1
        if self._fused_adam():
            # both learning-rate groups in one launch, which also clears the zero arena for the next step
            ops.adam_step_groups_(self.params.flat, self.params.grad, self.exp_avg, self.exp_avg_sq, [b for _, b in self.params.bounds],
                                  self.lrs, self.n_steps, self.betas, self.eps, zero=dirty)
            return losses
        for (a, b), lr in zip(self.params.bounds, self.lrs):
            if b > a:
                self.adam_fn(self.params.flat[a:b], self.params.grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], lr, self.n_steps,
                             self.betas, self.eps)
        return losses

    def _c_iteration(self, inp):
        """The iteration through the C step.  One GPU: everything, Adam included, in the one call.  Data parallel: the call stops in front of
        the backward of the texture preparation; the ranks sum the gradient of the PREPARED maps (sigmoid and decimation are linear behind it:
        with 8x-decimated maps 1 / 64 of the texture gradient's bytes, 0.15 instead of 9.4 MB at config 2) and the 159 small gradients; a
        second call runs that backward -- adding the TV gradient, which every rank holds in full -- and Adam, which clears the step's zero
        arena.  A mini-batch with fewer views than ranks (some rank's shard is empty and takes the launch-by-launch path) is reduced the old
        way on every rank -- the whole flat buffer, the blocks' texture slice early -- so that all ranks issue the same collectives."""
        cs = self.cstep
        distributed = self.world_size > 1 or self.overlap_allreduce
        # the step's random numbers are keyed on the optimisation-step count: identical on every rank whatever plan / batch size / phase a
        # rank's run uses, never replayed by a ragged batch or a new phase, restored with a checkpoint (n_steps is part of it)
        rng = self.n_steps
        with torch.no_grad():
            if not distributed:
                losses = cs(inp, self.model._global_count, adam=(self.n_steps + 1, self.lrs, self.betas, self.eps), rng_step=rng)
                self.n_steps += 1
                return losses
            # (cs.void_flag(): != 0 where a rank's cross-stream wait gave up -- summed with the gradients, so that EVERY rank skips the
            # update of a step one of them voided and the replicas stay identical)
            if self._defer_textures(inp):
                losses = cs(inp, self.model._global_count, adam=None, defer_textures=True, rng_step=rng)
                small = self.params.grad[:self.params.bounds[0][1]]
                self._allreduce_all([cs.map_grads(), small, cs.void_flag()])
                self.n_steps += 1
                cs.finish(adam=(self.n_steps, self.lrs, self.betas, self.eps))
                return losses
            losses = cs(inp, self.model._global_count, adam=None, rng_step=rng)
            if self.overlap_allreduce:
                # the blocks' texture gradient is final long before the step ends (the step says when): its slice is reduced from a
                # stream of its own next to the rest of the fg tail and the env chain
                dev = self.params.flat.device
                from .c_step import side_stream
                side = side_stream(dev, cs.side_priority)
                if cs.use_side_stream:
                    cs.wait_blocks_ready(side)
                else:
                    side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self.start_early_allreduce()
                torch.cuda.current_stream(dev).wait_stream(side)      # (a synchronous backend -- gloo through a host copy -- wrote on `side`)
            self.allreduce_gradients()
            void = cs.void_flag()
            if self.world_size > 1:
                self._allreduce(void)
            self.n_steps += 1
            ops.adam_step_groups_(self.params.flat, self.params.grad, self.exp_avg, self.exp_avg_sq, [b for _, b in self.params.bounds],
                                  self.lrs, self.n_steps, self.betas, self.eps, zero=cs.arena(), skip=void)
            cs.arena_cleaned_by_caller()
        return losses

    def _defer_textures(self, inp):
        """Whether this step sums the map gradients instead of the texture gradients: every rank must decide the same, from what every rank
        knows -- the size of the GLOBAL batch (a batch with fewer views than ranks leaves some rank without views, and that rank's step is
        not the C step)."""
        if not self.defer_textures:
            return False
        per_view = float(inp['imgs'][0].numel()) if inp['imgs'].shape[0] > 0 else 0.0
        return per_view > 0 and self.model._global_count / per_view >= self.world_size

    def _allreduce_all(self, tensors):
        """Sum all-reduce of several small buffers: one coalesced collective where the backend has it (RCCL: one launch), else one each.
        torch's coalescing context is NOT public API, so its first use is a trial: the buffers (0.15 MB at config 2) are copied first, and
        whatever goes wrong -- building the context, a collective inside it, the launch at its exit -- puts the copies back and reduces
        them one by one; the ranks then agree through one PUBLIC all-reduce of a flag that the trial held everywhere, and if it did not
        on any rank, all of them drop the coalesced form for the rest of the run.  Only a trial that every rank passed switches the
        copies off.  `allreduce_path` says which form the last call took ('coalesced' / 'per-tensor'; bench.py prints it)."""
        usable = (self._coalesce and dist.get_backend(self.pg) in self._coalesce_backends and all(t.is_cuda or 'gloo' in self._coalesce_backends for t in tensors)
                  and hasattr(dist, '_coalescing_manager'))
        if not usable:
            for t in tensors:
                self._allreduce(t)
            self.allreduce_path = 'per-tensor'
            return
        trial = not self._coalesce_verified
        backup = [t.clone() for t in tensors] if trial else None
        ok = True
        try:
            with dist._coalescing_manager(group=self.pg, device=tensors[0].device, async_ops=False):
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        except Exception as e:          # noqa: BLE001  (private API: anything may come out of it)
            if not trial:
                raise                   # (it has worked on every rank before: this is a failure of the collective itself, not of the API)
            ok, self._coalesce_error = False, repr(e)
        if trial:
            flag = torch.tensor([1.0 if ok else 0.0], device=tensors[0].device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            if float(flag.item()) == 1.0:
                self._coalesce_verified = True
            else:
                self._coalesce = False
                if ok:
                    # this rank's trial went through but another's did not: what it summed is not what the others will sum now
                    pass
                for t, c in zip(tensors, backup):
                    t.copy_(c)
                for t in tensors:
                    self._allreduce(t)
                self.allreduce_path = 'per-tensor'
                return
        self.allreduce_path = 'coalesced'

    def _fused_adam(self):
        return self.adam_fn is ops.adam_step_ and self.params.flat.is_cuda

    def _allreduce(self, t, async_op=False):
        """In-place sum all-reduce of (a slice of) the flat gradient buffer: RCCL over xGMI ('nccl'); gloo on CPU tensors; ranks
        sharing one GPU over gloo (tests) go through a host copy."""
        if t.is_cuda and dist.get_backend(self.pg) != 'nccl':
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
            t.copy_(h)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)

    def start_early_allreduce(self):
        """Called (by the native step, on the stream that has just finished them) once the gradients of the early slice are final."""
        if not self.overlap_allreduce or self._early_done:
            return
        a, b = self._early_range
        self._early_work, self._early_done = self._allreduce(self.params.grad[a:b], async_op=True), True

    def allreduce_gradients(self):
        """The gradient sum over the ranks: ONE all-reduce of the flat buffer, or -- overlapped mode -- the early slice (possibly
        already in flight) followed by the rest."""
        g = self.params.grad
        if not self.overlap_allreduce:
            self._allreduce(g)
            return
        self.start_early_allreduce()                 # ranks whose step did not announce it (empty batch, autograd path): same order
        a, b = self._early_range
        for lo, hi in ((0, a), (b, g.numel())):
            if hi > lo:
                self._allreduce(g[lo:hi])
        if self._early_work is not None:
            self._early_work.wait()                  # (the current stream waits, not the host)
        self._early_work, self._early_done = None, False
