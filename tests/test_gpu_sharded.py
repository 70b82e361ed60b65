"""GPU, N > 1 (skipped on a single-GPU box): the sharded log-likelihood over RCCL against the single-GPU sum
(SURVEY 8(e) parity row: invariance of the result to the number of ranks within 1e-9 rel, fp64 accumulate), and
`bench.py --gpus N` starting its own ranks."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

N_ROWS, DIM = 4099, 64          # ragged on purpose: shard sizes differ by one row


def _flow_and_batch():
    import stribor_amd as st
    from stribor_amd.util import flowdesc as fd
    torch.manual_seed(0)
    flow = fd.build_flow(st, fd.cfg2_desc(), DIM)
    x = torch.randn(N_ROWS, DIM, generator=torch.Generator().manual_seed(11))
    return flow, x


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    try:
        from stribor_amd.sharded import ShardedLogProb
        flow, x = _flow_and_batch()
        flow = flow.to(f'cuda:{rank}')
        sh = ShardedLogProb(flow)
        lo, hi = sh.my_rows(N_ROWS)
        with torch.no_grad():
            total = sh.log_prob_sum(x[lo:hi].to(f'cuda:{rank}'))
            outs = [torch.zeros(1, dtype=torch.float64, device=f'cuda:{rank}') for _ in range(3)]
            pend = [sh.log_prob_sum_async(x[lo:hi].to(f'cuda:{rank}'), o) for o in outs]
            for p in pend:
                assert abs(p.wait().item() - total.item()) <= 1e-12 * abs(total.item())
        q.put((rank, lo, hi, total.item(), dist.get_world_size()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_sum_over_rccl_matches_single_gpu(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs')
    flow, x = _flow_and_batch()
    with torch.no_grad():
        single = flow.to('cuda:0').log_prob_sum(x.to('cuda:0')).item()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    covered = sorted((lo, hi) for _, lo, hi, _, _ in res)
    assert covered[0][0] == 0 and covered[-1][1] == N_ROWS
    for _, _, _, tot, ws in res:
        assert ws == world
        assert abs(tot - single) <= 1e-9 * abs(single), (tot, single)


def test_bench_starts_its_own_ranks():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    r = json.loads(line)
    assert r['n_gpus'] == 2 and r['rccl_ranks'] == 2 and r['scaling'] == 'weak'
