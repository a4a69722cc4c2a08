"""CPU, gloo, world_size 2: the view-sharded data-parallel step (flat gradient buffer, one all-reduce, regularisers counted
once, global MSE normalisation) reproduces the single-process full-batch step.  The kernels are not involved: a small
torch model with the same loss structure stands in for the renderer."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from dbw_amd.parallel import FlatParams, ShardedTrainStep, shard_views


def test_shard_views_matches_survey_split():
    assert [shard_views(49, 8, r)[1] - shard_views(49, 8, r)[0] for r in range(8)] == [7, 6, 6, 6, 6, 6, 6, 6]
    covered = [i for r in range(8) for i in range(*shard_views(49, 8, r))]
    assert covered == list(range(49))
    assert shard_views(3, 4, 3) == (3, 3)                      # ragged: an empty shard
    assert shard_views(5, 1, 0) == (0, 5)


class ToyModel(nn.Module):
    """MSE over views (mean over the GLOBAL batch) + a view-independent regulariser scaled by 1/world_size."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.S = nn.Parameter(torch.randn(4, 3))
        self.texture_bkg = nn.Parameter(torch.randn(2, 5))
        self.world_size, self._global_count = 1, None

    # the few members of DifferentiableBlocksWorld that Trainer touches
    name, init_kwargs, sync_free, cur_epoch = 'toy', {}, True, 0

    def step(self):
        self.cur_epoch += 1

    def set_cur_epoch(self, e):
        self.cur_epoch = e

    def get_opacities(self):
        return torch.ones(1)

    def forward(self, inp, labels=None):
        pred = (inp['imgs'] * self.S.sum() + self.texture_bkg.sum())
        count = self._global_count or inp['imgs'].numel()
        rgb = ((pred - 1.0) ** 2).sum() / count
        reg = (self.S ** 2).mean() / self.world_size + (self.texture_bkg ** 2).mean() / self.world_size
        return {'rgb': rgb, 'reg': reg, 'total': rgb + reg}


def torch_adam(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8):
    m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
    v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    p.addcdiv_(m, v.sqrt() / bc2 ** 0.5 + eps, value=-lr / bc1)


def _worker(rank, world, port, views, out, overlap=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = ToyModel()
    step = ShardedTrainStep(model, adam_fn=torch_adam, overlap_allreduce=overlap, early_param='texture_bkg')
    assert step.overlap_allreduce == overlap
    if overlap and rank == 0:            # one rank announces the early slice itself (as the native step does), the other does not
        rest = step.allreduce_gradients
        def early_then_rest():
            step.start_early_allreduce()
            assert step._early_done
            rest()
        step.allreduce_gradients = early_then_rest
    a, b = shard_views(views.shape[0], world, rank)
    for _ in range(3):
        step({'imgs': views[a:b]})
    out[rank] = step.params.flat.clone()
    dist.destroy_process_group()


def test_two_rank_gloo_step_equals_single_process():
    views = torch.rand(6, 3, 4, 4, generator=torch.Generator().manual_seed(1))
    ref_model = ToyModel()
    ref = ShardedTrainStep(ref_model, adam_fn=torch_adam)
    for _ in range(3):
        ref({'imgs': views})
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, 29512, views, out), nprocs=2, join=True)
    assert torch.allclose(out[0], out[1], rtol=0, atol=0)                   # replicas stay bit-identical
    assert torch.allclose(out[0], ref.params.flat, rtol=1e-5, atol=1e-6)   # and equal the full-batch run


def test_overlapped_allreduce_two_collectives_equal_the_single_one():
    """overlap_allreduce: the early slice (announced by the step on one rank, not on the other: an empty / autograd step) and then the
    rest -- same collectives in the same order on both ranks, same result as the single all-reduce."""
    views = torch.rand(6, 3, 4, 4, generator=torch.Generator().manual_seed(1))
    mgr = mp.Manager()
    out, ref = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(2, 29531, views, ref, False), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, 29532, views, out, True), nprocs=2, join=True)
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], ref[0])


def test_flat_params_bind_grads_in_place_and_group_textures_last():
    m = ToyModel()
    fp = FlatParams(m)
    assert [n for n, _, _ in fp.names] == ['S', 'texture_bkg'] and fp.bounds == [(0, 12), (12, 22)]
    m({'imgs': torch.ones(1, 2)})['total'].backward()
    assert fp.grad.abs().sum() > 0 and m.S.grad.data_ptr() == fp.grad.data_ptr()
    fp.zero_grad()
    assert m.S.grad.abs().sum() == 0


def _trainer_worker(rank, world, port, views, out):
    from dbw_amd.trainer import Trainer
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cfg = {'training': {'batch_size': 2, 'n_epoches': 2, 'seed': 7, 'optimizer': {'name': 'adam', 'lr': 1e-2, 'texture': {'lr': 3e-2}},
                        'scheduler': {'name': 'multi_step', 'gamma': [0.1, 0.1], 'milestones': [50]}}}
    tr = Trainer(cfg, ToyModel(), {'imgs': views})
    tr.step_fn.adam_fn = torch_adam
    steps = 0
    for _ in range(2):
        tr.run_epoch(shuffle=False)
        steps += tr.n_batches
    out[rank] = (tr.step_fn.params.flat.clone(), tr.n_iters, tr.shard_sizes, steps)
    dist.destroy_process_group()


def test_uneven_shards_every_rank_runs_the_same_steps_and_matches_the_union_batches():
    """5 views on 2 ranks, batch 2: shards 3 / 2 -> both ranks run 2 steps per epoch, the second one on batches of 1 and 0 views (the
    empty one only carries its share of the regulariser).  Must not hang (same collectives everywhere: the advisor's finding on
    the cached element count) and must equal a single process stepping on the union of the ranks' batches, whose MSE is normalised
    by the size of THAT global batch."""
    views = torch.rand(5, 3, 4, 4, generator=torch.Generator().manual_seed(3))
    ref = ShardedTrainStep(ToyModel(), lr=1e-2, lr_texture=3e-2, adam_fn=torch_adam)
    for _ in range(2):
        ref({'imgs': views[[0, 1, 3, 4]]})           # step 1: rank 0 holds views 0-2, rank 1 views 3-4
        ref({'imgs': views[[2]]})                    # step 2: rank 0's last view, rank 1 idle
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trainer_worker, args=(2, 29513, views, out), nprocs=2, join=True)
    assert out[0][2] == [3, 2] and out[0][1] == out[1][1] == 4 and out[0][3] == 4
    assert torch.equal(out[0][0], out[1][0])
    assert torch.allclose(out[0][0], ref.params.flat, rtol=1e-5, atol=1e-6)


def test_global_count_is_never_cached_across_differently_sized_batches():
    step = ShardedTrainStep(ToyModel(), adam_fn=torch_adam)
    a, b = torch.rand(4, 3, 2, 2), torch.rand(3, 3, 2, 2)
    assert step._global_count(a, None) == a.numel() and step._global_count(b, None) == b.numel()
    assert step._global_count(b, 123) == 123.0


def test_perceptual_term_is_weighted_by_the_rank_share_of_the_global_batch():
    """Advisor finding: rgb uses the global count, the regularisers 1/world_size -- the perceptual mean must be weighted by
    local / global views, otherwise the summed gradient counts it world_size times."""
    import dbw_amd
    cfg = {'model': {'name': 'dbw', 'mesh': {'n_blocks': 2, 'txt_size': 8}, 'renderer': {'cameras': {'name': 'perspective'}},
                     'rend_optim': {'decouple_rendering': True}, 'loss': {'rgb_weight': 1, 'perceptual_weight': 0.1}}}
    m = dbw_amd.create_model(cfg, (8, 8))
    m.set_perceptual(lambda a, b: ((a - b) ** 2).mean())
    imgs, rec = torch.rand(3, 3, 8, 8), torch.rand(3, 3, 8, 8)
    single = m._perceptual_term(imgs, rec, True)
    m.world_size, m._global_count = 2, float(5 * 3 * 8 * 8)          # this rank holds 3 of the 5 views of the step
    assert torch.allclose(m._perceptual_term(imgs, rec, True), single * 3 / 5)
    assert torch.allclose(m._perceptual_term(imgs, rec, False), single * 0.1 * 3 / 5)


def _coalesce_worker(rank, world, port, out, fails):
    """_allreduce_all with torch's private coalescing context replaced by one that runs the collectives of its block and then (fails)
    raises at its exit -- the worst case for the caller: the buffers ARE summed when the error surfaces."""
    import contextlib
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)

    @contextlib.contextmanager
    def fake_manager(group=None, device=None, async_ops=False):
        yield
        if fails:
            raise RuntimeError('coalescing is not what it used to be')
    dist._coalescing_manager = fake_manager
    step = ShardedTrainStep(ToyModel(), adam_fn=torch_adam)
    step._coalesce_backends = ('nccl', 'gloo')
    res = []
    for call in range(2):
        a, b = torch.full((5,), float(rank + 1 + call)), torch.full((3,), 10.0 * (rank + 1))
        step._allreduce_all([a, b])
        res.append((a.clone(), b.clone(), step.allreduce_path, step._coalesce, step._coalesce_verified))
    out[rank] = res
    dist.destroy_process_group()


@pytest.mark.parametrize('fails', [True, False])
def test_coalesced_allreduce_survives_a_private_api_that_raises_inside_its_block(fails):
    """The first coalesced all-reduce is a trial: if torch's private coalescing context raises -- here at its exit, AFTER the collectives
    of the block ran -- the buffers are put back, reduced one by one, every rank learns of it through a public all-reduce of a flag and
    drops the coalesced form for good; the sums are the plain sums either way (nothing is reduced twice), on every rank alike."""
    world, port = 2, 29500 + (os.getpid() * 7 + int(fails)) % 1000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_coalesce_worker, args=(world, port, out, fails), nprocs=world, join=True)
    for rank in range(world):
        for call, (a, b, path, coalesce, verified) in enumerate(out[rank]):
            assert torch.equal(a, torch.full((5,), float(1 + call) + float(2 + call))) and torch.equal(b, torch.full((3,), 30.0))
            assert path == ('per-tensor' if fails else 'coalesced')
            assert coalesce == (not fails) and verified == (not fails)
