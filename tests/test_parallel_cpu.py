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


def _worker(rank, world, port, views, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = ToyModel()
    step = ShardedTrainStep(model, adam_fn=torch_adam)
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


def test_flat_params_bind_grads_in_place_and_group_textures_last():
    m = ToyModel()
    fp = FlatParams(m)
    assert [n for n, _, _ in fp.names] == ['S', 'texture_bkg'] and fp.bounds == [(0, 12), (12, 22)]
    m({'imgs': torch.ones(1, 2)})['total'].backward()
    assert fp.grad.abs().sum() > 0 and m.S.grad.data_ptr() == fp.grad.data_ptr()
    fp.zero_grad()
    assert m.S.grad.abs().sum() == 0
