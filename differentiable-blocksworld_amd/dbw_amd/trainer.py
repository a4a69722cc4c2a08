"""The step either side of the render path: learning-rate schedule, optimisation loop and
checkpointing with the semantics of the reference's src/scheduler.py:26-69, src/optimizer.py:6-18 and
src/trainer.py:109-169,201-209, driving `ShardedTrainStep` (flat parameter buffer, fused Adam, optional RCCL all-reduce).

Host-side scalars only; nothing here touches pixels.  No logging / visualisation / evaluation plumbing (out of scope)."""
import time
from collections import Counter

import torch

from .parallel import ShardedTrainStep, shard_views


class MultiStepLR:
    """Per-group learning rates, stepped once per EPOCH (trainer.py:127,163-169).  Reproduces the reference's chainable
    recurrence including its warm-up quirk (the constructor divides the freshly set warm-up lr by `warmup` once more,
    scheduler.py:41-46): pinned by tests/golden/lr_schedule.npz."""

    def __init__(self, base_lrs, milestones=None, gamma=0.1, warmup=0):
        self.base_lrs = [float(x) for x in base_lrs]
        self.milestones = Counter(milestones or [])
        self.gamma = [gamma] * len(self.base_lrs) if isinstance(gamma, float) else list(gamma)
        self.warmup = warmup
        self.last_epoch = 0
        self.lrs = self._next(list(self.base_lrs))
        if warmup > 0:
            self.lrs = [lr / warmup for lr in self.lrs]

    def _next(self, lrs):
        if self.warmup > self.last_epoch:
            return [lr / self.warmup * (self.last_epoch + 1) for lr in self.base_lrs]
        if self.last_epoch not in self.milestones:
            return lrs
        return [lr * g ** self.milestones[self.last_epoch] for lr, g in zip(lrs, self.gamma)]

    def step(self):
        self.last_epoch += 1
        self.lrs = self._next(self.lrs)
        return self.lrs

    def get_last_lr(self):
        return list(self.lrs)

    def state_dict(self):
        return {'last_epoch': self.last_epoch, 'lrs': list(self.lrs)}

    def load_state_dict(self, state):
        """Own state ({'last_epoch', 'lrs'}) or the state_dict of the reference's torch scheduler (src/scheduler.py: 'last_epoch',
        '_last_lr', '_step_count', ...): there the decayed rates are read from '_last_lr', or replayed from the base rates and the
        milestones when that key is missing too -- never left at the base values past a milestone."""
        self.last_epoch = int(state['last_epoch'])
        if 'lrs' in state:
            self.lrs = [float(x) for x in state['lrs']]
        elif '_last_lr' in state and len(state['_last_lr']) == len(self.base_lrs):
            self.lrs = [float(x) for x in state['_last_lr']]
        else:
            target, self.last_epoch = self.last_epoch, 0
            self.lrs = self._next(list(self.base_lrs))
            if self.warmup > 0:
                self.lrs = [lr / self.warmup for lr in self.lrs]
            while self.last_epoch < target:
                self.step()


class Trainer:
    """cfg: the reference's YAML dict (`model`, `training.{batch_size, optimizer, scheduler, n_epoches, seed}`).
    views: dict of tensors {'imgs' (V,3,H,W), 'R' (V,3,3), 'T' (V,3), 'K' (V,4,4)} resident on the device (the reference's
    DataLoader yields the same dict batch by batch, trainer.py:118,141)."""

    def __init__(self, cfg, model, views, process_group=None, sync_free=True, cache_perceptual_targets=True):
        tr = cfg['training']
        opt = dict(tr.get('optimizer') or {})
        if opt.pop('name', 'adam') != 'adam':
            raise NotImplementedError('only Adam (the optimiser of every shipped config, default.yml:31) is fused')
        txt = dict(opt.pop('texture', {}) or {})
        lr = opt.pop('lr', 1e-3)
        lr_txt = txt.pop('lr', lr)
        sch = dict(tr.get('scheduler') or {})
        if sch.pop('name', 'multi_step') != 'multi_step':
            raise NotImplementedError('only the multi_step scheduler')
        self.model, self.views = model, views
        model.sync_free = sync_free
        self.step_fn = ShardedTrainStep(model, lr=lr, lr_texture=lr_txt, betas=opt.pop('betas', (0.9, 0.999)), eps=opt.pop('eps', 1e-8),
                                        process_group=process_group, seed=tr.get('seed'))
        if opt or txt:       # the reference forwards these to torch.optim.Adam (optimizer.py:9-18); the fused kernel implements plain Adam
            raise NotImplementedError(f'optimizer options not supported by the fused Adam: {sorted(opt) + ["texture." + k for k in sorted(txt)]}')
        self.scheduler = MultiStepLR([lr, lr_txt], **sch)
        self.step_fn.lrs = tuple(self.scheduler.get_last_lr())
        self.batch_size = tr.get('batch_size', 4)
        self.n_epoches = tr.get('n_epoches', 1)
        self.epoch, self.n_iters, self.time_per_img = 1, 0, 0.0
        V, ws = views['imgs'].shape[0], self.step_fn.world_size
        a, b = shard_views(V, ws, self.step_fn.rank)
        self.local = {k: v[a:b] for k, v in views.items()}
        self.shard_sizes = [shard_views(V, ws, r)[1] - shard_views(V, ws, r)[0] for r in range(ws)]
        self.n_batches = -(-max(self.shard_sizes) // self.batch_size)          # every rank runs this many steps per epoch
        self.per_view = views['imgs'][0].numel() if V else 0
        self._perm_gen = torch.Generator().manual_seed(int(tr.get('seed') or 0))
        # The perceptual criterion's target half is a constant per training view (frozen network, fixed images): a criterion that can keep
        # it (lpips_vgg.LPIPSVGG.cache_targets) computes it once for this rank's views, if that fits in a quarter of the free memory, and
        # every batch then carries the local indices of its views.  A third of the criterion's FLOPs; the values do not change.
        self.view_ids = False
        fn = getattr(model, 'perceptual_fn', None)
        if (cache_perceptual_targets and 'perceptual' in getattr(model, 'loss_weights', {}) and hasattr(fn, 'cache_targets') and self.local['imgs'].is_cuda
                and self.local['imgs'].shape[0] > 0):
            need = fn.target_bytes(*self.local['imgs'].shape[2:], n_views=self.local['imgs'].shape[0])
            if need <= torch.cuda.mem_get_info(self.local['imgs'].device)[0] // 4:
                fn.cache_targets(self.local['imgs'])
                self.view_ids = True

    # trainer.py:137-147
    def run_single_batch_train(self, inp, global_count=None):
        t0 = time.time()
        self.model.train()
        losses = self.step_fn(inp, global_count=global_count)
        self.n_iters += 1
        return losses, t0

    def global_count(self, batch):
        """Image elements of mini-batch `batch` over ALL ranks -- every rank derives it from the shard sizes, no collective."""
        bs = self.batch_size
        return self.per_view * sum(min(bs, max(0, v - batch * bs)) for v in self.shard_sizes)

    def run_epoch(self, shuffle=True):
        """One pass over this rank's views in mini-batches (DataLoader(shuffle=True) equivalent), then the per-epoch
        scheduler / model step (trainer.py:127,163-169)."""
        V = self.local['imgs'].shape[0]
        order = torch.randperm(V, generator=self._perm_gen) if shuffle else torch.arange(V)
        if self.view_ids:
            # the criterion is shared by whoever holds the model: if its cache is not (or no longer) the one of THESE views -- new weights,
            # another Trainer on other views -- the ids would index foreign features; built again (a forward pass over the views, once)
            fn = self.model.perceptual_fn
            if not (hasattr(fn, 'cache_matches') and fn.cache_matches(self.local['imgs'])):
                fn.cache_targets(self.local['imgs'])
        last = None
        t_start, n_img = time.time(), 0
        # Uneven shards (49 views over 8 ranks: 7,6,...,6; batch 4 -> two steps everywhere, the second of sizes 3,2,...,2): all ranks
        # run n_batches steps -- a rank that has run out of views steps on an EMPTY batch (regularisers only) -- so that the
        # sequence of collectives is the same everywhere, and the MSE normalisation of a step is the size of its global batch
        for b in range(self.n_batches):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size].to(self.local['imgs'].device)
            batch = {k: v[idx] for k, v in self.local.items()}
            if self.view_ids:
                batch['view_ids'] = idx
            last, _ = self.run_single_batch_train(batch, self.global_count(b))
            n_img += idx.numel()
        if self.local['imgs'].is_cuda:
            torch.cuda.synchronize()
        self.time_per_img = (time.time() - t_start) / max(n_img, 1)
        self.step_fn.lrs = tuple(self.scheduler.step())
        self.model.step()
        self.epoch += 1
        opac = self.model.get_opacities()
        if (opac > 0.01).sum() == 0:
            raise RuntimeError('No more blocks....')                   # trainer.py:152-154
        return last

    def run(self, n_epoches=None):
        last = None
        for _ in range(self.epoch, (n_epoches or self.n_epoches) + 1):
            last = self.run_epoch()
        return last

    # trainer.py:201-209 / 84-107
    def state_dict(self):
        """Same keys as trainer.py:201-209.  'epoch' / 'batch' = the last COMPLETED epoch and its number of batches, like the
        reference's end-of-epoch save, so either trainer resumes the other's checkpoint at the same epoch.  'model_state'
        interchanges; 'scheduler_state' is read in either form (MultiStepLR.load_state_dict takes the torch scheduler's keys too,
        this trainer writes {'last_epoch', 'lrs'}).  'optimizer_state' does NOT interchange: it holds the fused Adam's flat moment
        buffers ({'exp_avg', 'exp_avg_sq', 'n_steps'} in FlatParams order), not a torch.optim.Adam state_dict.  Under data
        parallelism 'batch' counts this trainer's batches per epoch, ceil(largest shard / batch_size), not the reference's
        len(loader) over all views: the reference would take such a checkpoint for a mid-epoch one."""
        return {'epoch': self.epoch - 1, 'batch': self.n_batches, 'model_name': self.model.name, 'model_kwargs': self.model.init_kwargs,
                'model_state': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                'optimizer_state': {'exp_avg': self.step_fn.exp_avg.clone(), 'exp_avg_sq': self.step_fn.exp_avg_sq.clone(),
                                    'n_steps': self.step_fn.n_steps},
                'scheduler_state': self.scheduler.state_dict()}

    def load_state_dict(self, ckpt):
        self.model.load_state_dict(ckpt['model_state'])
        o = ckpt['optimizer_state']
        if 'exp_avg' not in o:
            raise KeyError("optimizer_state is not a fused-Adam state ({'exp_avg', 'exp_avg_sq', 'n_steps'}): a reference checkpoint's "
                           'torch.optim.Adam state_dict cannot be loaded; load its model_state / scheduler_state and restart the moments')
        self.step_fn.exp_avg.copy_(o['exp_avg'])
        self.step_fn.exp_avg_sq.copy_(o['exp_avg_sq'])
        self.step_fn.n_steps = o['n_steps']
        self.scheduler.load_state_dict(ckpt['scheduler_state'])
        self.step_fn.lrs = tuple(self.scheduler.get_last_lr())
        self.epoch = ckpt['epoch'] + 1                         # trainer.py:94-97: resume with the epoch after the saved one
        self.model.set_cur_epoch(ckpt['epoch'])
