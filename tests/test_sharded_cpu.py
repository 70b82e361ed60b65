"""CPU, world_size 2 over gloo: the N>1 path (row sharding + the single all-reduce) with the oracle
standing in for the per-shard HIP evaluation."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import flowdesc as fd
from goldens import Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_shard_rows_partition():
    from stribor_amd.sharded import shard_rows
    for n in (0, 1, 7, 256, 1 << 20, (1 << 23) + 5):
        for w in (1, 2, 3, 8):
            blocks = [shard_rows(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from oracle import stribor_oracle as orc
        from stribor_amd.sharded import ShardedLogProb
        g = Golden('f4_cfg2')
        spec = fd.flow_spec(g.meta['cfg2']['desc'], g.state('cfg2'))
        x = g.t('cfg2/x')

        def local_sum(y, out):
            out += orc.flow_log_prob(spec, y).double().sum()

        sh = ShardedLogProb(group=None, local_sum=local_sum)
        lo, hi = sh.my_rows(x.shape[0])
        total = sh.log_prob_sum(x[lo:hi])
        # pipelined form (bench.py): several batches in flight, each with its own result buffer
        outs = [torch.zeros(1, dtype=torch.float64) for _ in range(3)]
        pend = [sh.log_prob_sum_async(x[lo:hi], o) for o in outs]
        for p in pend:
            assert abs(p.wait().item() - total.item()) <= 1e-12 * abs(total.item())
        q.put((rank, lo, hi, total.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_sharded_sum_matches_unsharded(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = Golden('f4_cfg2')
    want = g.t('cfg2/log_prob').double().sum().item()
    want64 = g.t('cfg2/log_prob_f64').sum().item()
    covered = sorted((lo, hi) for _, lo, hi, _ in res)
    assert covered[0][0] == 0 and covered[-1][1] == 256
    for _, _, _, tot in res:                      # every rank holds the same global sum
        assert abs(tot - want) <= 1e-9 * abs(want)
        assert abs(tot - want64) <= 1e-6 * abs(want64)          # SURVEY 8(e) parity row
