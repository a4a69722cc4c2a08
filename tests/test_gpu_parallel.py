"""Multi-GPU readiness on a one-GPU box (`-m gpu`): two ranks of the REAL model (HIP kernels, opacity noise and overlap sampling
on) share cuda:0 and all-reduce over gloo -- the same ShardedTrainStep code path a node runs with one rank per GPU over RCCL
(backend 'nccl'); only the transport differs.  The driver measures the 8-GPU scaling itself (bench.py --gpus N)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

H, W, NB, TS, FPP, V = 48, 64, 4, 32, 6, 5


def _cfg():
    return {'model': {'name': 'dbw', 'mesh': {'n_blocks': NB, 'S_world': 0.5, 'R_world': [115, 0, 0], 'txt_size': TS},
                      'renderer': {'faces_per_pixel': FPP, 'cameras': {'name': 'perspective'}, 'detach_bary': True, 'z_clip': 0.001},
                      'rend_optim': {'coarse_learning': 1500, 'decimate_txt': 750, 'decimate_factor': 8, 'kill_blocks': True,
                                     'decouple_rendering': True, 'opacity_noise': True},
                      'loss': {'rgb_weight': 1, 'perceptual_weight': 0, 'parsimony_weight': 0.01, 'tv_weight': 0.1, 'overlap_weight': 1}}}


def _views(n=V):
    import oracle as O                                          # camera rig + targets only (checker-side helper)
    R, T, Km = O.synthetic_cameras(n, R_world=O.world_rotation(115, 0, 0))
    imgs = torch.rand(n, 3, H, W, generator=torch.Generator().manual_seed(2))
    return dict(imgs=imgs, R=R, T=T, K=Km)


def _run_nccl_single(rank, port, out):
    """A ONE-rank RCCL group with the overlapped all-reduce forced on: the async collective on the communicator's stream, launched from
    the native step's side stream and waited for before Adam, must leave the step unchanged (sum over one rank)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'differentiable-blocksworld_amd'), os.path.join(root, 'oracle')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dbw_amd
    from dbw_amd.parallel import ShardedTrainStep
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    res = []
    for overlap, defer in ((False, True), (True, False), (True, True)):
        # (no group; the whole gradient buffer reduced with its early slice overlapped; the prepared maps' gradient reduced instead)
        if overlap and not dist.is_initialized():
            dist.init_process_group('nccl', rank=0, world_size=1)
        torch.manual_seed(227391)
        model = dbw_amd.create_model(_cfg(), (H, W)).to(dev).train()
        model.sync_free = True
        step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99, overlap_allreduce=overlap)
        step.defer_textures = defer
        assert step.overlap_allreduce == overlap
        views = {k: v.to(dev) for k, v in _views().items()}
        for _ in range(3):
            step(views)
        torch.cuda.synchronize()
        res.append(step.params.flat.detach().cpu().clone())
    # (the deferred flow went through ONE coalesced RCCL call for the map gradients + the small gradients + the void flag: torch's
    # coalescing context was accepted, not replaced by the per-tensor fallback)
    out[1] = bool(step._coalesce and step._coalesce_verified and step.allreduce_path == 'coalesced') and dist.get_backend() == 'nccl'
    dist.destroy_process_group()
    out[0] = max(float((res[0] - res[1]).abs().max()), float((res[0] - res[2]).abs().max()))



def _manager():
    """The manager's server process is SPAWNED, not forked: a fork of this process would inherit the GPU objects earlier tests left to the
    garbage collector (plans, graphs, streams) without the context they live in, and abort when it collects them."""
    import gc
    gc.collect()
    return mp.get_context('spawn').Manager()

def test_overlapped_allreduce_on_a_one_rank_rccl_group_leaves_the_step_unchanged():
    mgr = _manager()
    out = mgr.dict()
    mp.spawn(_run_nccl_single, args=(29541, out), nprocs=1, join=True)
    assert out[0] < 1e-4, out[0]          # (atomics order: not bit-identical from run to run)
    assert out[1] is True


def _run(rank, world, port, out, n_views=V, n_steps=3, defer=True):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'differentiable-blocksworld_amd'), os.path.join(root, 'oracle')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dbw_amd
    from dbw_amd.parallel import ShardedTrainStep, shard_views
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    torch.manual_seed(227391)
    model = dbw_amd.create_model(_cfg(), (H, W)).to(dev).train()
    model.sync_free = True
    # overlapped all-reduce: the native step announces the blocks' texture gradient from its side stream, the rest follows the step
    step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=99, overlap_allreduce=world > 1)
    step.defer_textures = defer          # True: the ranks sum the gradient of the prepared maps; False: the whole flat buffer, its early slice overlapped
    assert step.overlap_allreduce == (world > 1) and step.native is not None and step.native.on_block_grads_ready is not None
    views = {k: v.to(dev) for k, v in _views(n_views).items()}
    a, b = shard_views(n_views, world, rank)
    local = {k: v[a:b] for k, v in views.items()}
    count = views['imgs'].numel()                              # the global batch of every step: all V views
    grads, params = [], []
    for _ in range(n_steps):
        step(local, global_count=count)
        grads.append(step.params.grad.detach().cpu().clone())   # after the all-reduce: the gradient of the global batch
        params.append(step.params.flat.detach().cpu().clone())
    out[rank] = (grads, params, step.params.names)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('defer', [True, False])
def test_two_ranks_sharing_one_gpu_reproduce_the_full_batch_step(defer):
    """Two ranks with shards of 3 and 2 views against one process with all 5: whether the ranks sum the gradient of the prepared maps
    (defer: the C step's data-parallel flow) or the whole flat gradient buffer with the blocks' texture slice overlapped."""
    mgr = _manager()
    ref, out = mgr.dict(), mgr.dict()
    mp.spawn(_run, args=(1, 0, ref), nprocs=1, join=True)                     # single process, all 5 views (its own CUDA context)
    mp.spawn(_run, args=(2, 29517 + int(defer), out, V, 3, defer), nprocs=2, join=True)                 # shards of 3 and 2 views
    g_ref, p_ref, names = ref[0]
    for s in range(3):
        assert torch.equal(out[0][1][s], out[1][1][s]), f'replicas diverged at step {s}'           # (i) bit-identical replicas
        assert torch.equal(out[0][0][s], out[1][0][s])
    for n, off, k in names:                                                                        # (ii) == the full-batch gradient
        a, b = out[0][0][0][off:off + k], g_ref[0][off:off + k]
        err = float((a - b).abs().max() / b.abs().max().clamp(min=1e-20))
        assert err < 1e-5, (n, err)
    assert float((out[0][1][2] - p_ref[2]).abs().max()) < 1e-4                                     # and the same parameters after 3 steps


def test_config3_split_49_views_over_8_ranks_reproduces_the_full_batch_step():
    """the 49 views of config 2 sharded 8 ways (7,6,6,6,6,6,6,6), here at a small resolution with the eight ranks
    sharing cuda:0 over gloo: replicas bit-identical, gradient == the single-process gradient of all 49 views."""
    from dbw_amd.parallel import shard_views
    assert [shard_views(49, 8, r)[1] - shard_views(49, 8, r)[0] for r in range(8)] == [7, 6, 6, 6, 6, 6, 6, 6]
    mgr = _manager()
    ref, out = mgr.dict(), mgr.dict()
    mp.spawn(_run, args=(1, 0, ref, 49, 2), nprocs=1, join=True)
    mp.spawn(_run, args=(8, 29561, out, 49, 2), nprocs=8, join=True)
    g_ref, p_ref, names = ref[0]
    for r in range(1, 8):
        assert torch.equal(out[0][1][-1], out[r][1][-1]), f'replica {r} diverged'
    for n, off, k in names:
        a, b = out[0][0][0][off:off + k], g_ref[0][off:off + k]
        err = float((a - b).abs().max() / b.abs().max().clamp(min=1e-20))
        assert err < 1e-5, (n, err)
