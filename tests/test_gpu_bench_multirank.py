"""`bench.py --gpus N` (N > 1) as the driver launches it -- python -m torch.distributed.run, one rank per process -- exercised on a
single-GPU box: the two ranks share cuda:0 and all-reduce over gloo (DBW_BENCH_BACKEND / DBW_BENCH_SHARE_GPU, bench.py).  Everything
but the transport is the code path of the 8-GPU run: rendezvous on 127.0.0.1, the view split of `--scaling strong`, the global-count
MSE normalisation, the overlapped all-reduce, MAX-over-ranks timing, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bench_two_ranks_through_torchrun(scaling):
    env = dict(os.environ, DBW_BENCH_BACKEND='gloo', DBW_BENCH_SHARE_GPU='1', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 29500 + (os.getpid() % 400) + (0 if scaling == 'weak' else 450)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--scaling', scaling,
           '--views', '6', '--H', '96', '--W', '128', '--txt', '64', '--no-phases', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['nranks'] == 2 and d['scaling'] == scaling
    assert d['allreduce_ms'] is not None and d['allreduce_ms'] > 0
    assert d['final_loss'] == d['final_loss'] and 0 < d['final_loss'] < 10           # finite
    assert d['value'] > 0 and d['ms_per_step'] > 0
    if scaling == 'weak':
        assert d['config']['views_per_gpu'] == 6 and d['config']['views_per_step'] == 12
    else:
        assert d['config']['views_per_gpu'] == 3 and d['config']['views_per_step'] == 6
    assert abs(d['value'] - d['config']['views_per_step'] * d['steps'] / (d['ms_per_step'] * 1e-3 * d['steps'])) < 1e-6 * d['value']
    # the other scaling form measured in the same run, and the step without the early slice of the all-reduce
    o = d['other_scaling']
    assert o['scaling'] == ('strong' if scaling == 'weak' else 'weak') and o['value'] > 0 and o['ms_per_step'] > 0
    assert o['views_per_step'] == (6 if scaling == 'weak' else 12) and o['views_on_rank0'] == (3 if scaling == 'weak' else 6)
    assert d['ms_per_step_overlap_allreduce_off'] is None or d['ms_per_step_overlap_allreduce_off'] > 0
    # the ranks sum the gradient of the prepared maps + the small gradients; the step with the whole flat buffer reduced is timed next to it
    assert 0 < d['allreduce_bytes'] < 2.0e6 and d['ms_per_step_flat_allreduce'] > 0 and 'prepared texture maps' in d['config']['parallelism']
    assert 'C-ABI call per iteration' in d['config']['launch'] and d['sync_timeouts'] == 0


def test_bench_defaults_to_config_3_as_written_when_it_gets_more_than_one_gpu():
    """`bench.py --gpus 2` without --scaling: the SAME views split over the ranks (BASELINE configs[2], strong scaling), the weak form next
    to it in `other_scaling`."""
    env = dict(os.environ, DBW_BENCH_BACKEND='gloo', DBW_BENCH_SHARE_GPU='1', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 29500 + (os.getpid() % 400) + 900
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
           '--views', '7', '--H', '96', '--W', '128', '--txt', '64', '--no-phases', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['scaling'] == 'strong' and d['config']['views_per_step'] == 7 and d['config']['views_per_gpu'] == 4       # 4 + 3
    assert d['other_scaling']['scaling'] == 'weak' and d['other_scaling']['views_per_step'] == 14
