"""GPU, N > 1: the sharded log-likelihood against the single-GPU sum (SURVEY 8(e) parity row: invariance of the result to the
number of ranks within 1e-9 rel, fp64 accumulate), and `bench.py --gpus N` starting its own ranks.

Over RCCL (backend nccl) the tests need N GPUs and skip on a smaller box.  The SAME control path -- rank spawn through
torch.distributed.run, rendezvous on 127.0.0.1, the sharded sum, bench.py's ring of asynchronous all-reduces, the MAX over ranks
of the elapsed time, rank 0's JSON line -- also runs over gloo with the ranks sharing cuda:0, on any box (the N > 1
code must not meet hardware for the first time on the 8-GPU node).  Ranks are children of tests/launcher.py, never of the
GPU-initialised pytest process."""
import json
import os
import socket
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sharded_rank import N_ROWS

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


# (('nccl', 1): one rank over RCCL on the one GPU of the test box -- communicator set-up and the fp64 all-reduce through the real
#  backend, which the gloo runs cannot show; the multi-rank RCCL cases need their GPUs)
@pytest.mark.parametrize('backend,world', [('gloo', 2), ('gloo', 3), ('nccl', 1), ('nccl', 2), ('nccl', 8)])
def test_sharded_sum_matches_single_gpu(backend, world, run_child):
    if backend == 'nccl' and torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs')
    r = run_child([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr',
                   '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'sharded_rank.py'),
                   '--backend', backend], env={'HSA_ENABLE_IPC_MODE_LEGACY': '0'},
                  unset=('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'), timeout=900)
    assert r['rc'] == 0, r['stderr'][-3000:]
    import re
    res = [json.loads(m) for m in re.findall(r'RESULT (\{[^{}]*\})', r['stdout'])]     # (ranks share stdout: lines may run together)
    assert len(res) == world, r['stdout'][-2000:]
    single = [x['single'] for x in res if x['single'] is not None][0]
    covered = sorted((x['lo'], x['hi']) for x in res)
    assert covered[0][0] == 0 and covered[-1][1] == N_ROWS
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    for x in res:
        assert x['world'] == world and x['async_agree']
        assert abs(x['total'] - single) <= 1e-9 * abs(single), (x['total'], single)


@pytest.mark.parametrize('backend', ['gloo', 'nccl'])
def test_bench_starts_its_own_ranks(backend, run_child):
    """`bench.py --gpus 2 [--backend gloo]`: spawn_ranks -> torch.distributed.run -> init_process_group -> the 4-deep PendingSum ring
    -> all_reduce(MAX) of the elapsed time -> rank 0's JSON line (gloo: both ranks on cuda:0)."""
    if backend == 'nccl' and torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline']
    if backend == 'gloo':
        cmd += ['--backend', 'gloo']
    out = run_child(cmd, env={'HSA_ENABLE_IPC_MODE_LEGACY': '0'},
                    unset=('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'), timeout=1500)
    assert out['rc'] == 0, out['stderr'][-3000:]
    lines = [l for l in out['stdout'].splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines                    # rank 0 only
    r = json.loads(lines[-1])
    assert r['n_gpus'] == 2 and r['collective_ranks'] == 2 and r['collective_backend'] == backend and r['scaling'] == 'weak'
    assert r['rccl_ranks'] == (2 if backend == 'nccl' else 0)
    assert r['steps'] == 3 and r['warmup'] == 1 and r['value'] > 0 and r['ms_per_step'] > 0
    assert 'configs' not in r and 'training' not in r            # the single-GPU extras stay out of an N > 1 line
    # the all-reduced sum is that of two independent shards (seeds 1234 + rank): of the magnitude of 2 x 2^20 rows
    assert r['log_prob_sum'] < 0 and abs(r['log_prob_sum']) > 1e7
