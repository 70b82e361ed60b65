"""One rank of the sharded log-likelihood check (tests/test_gpu_sharded.py), run under torch.distributed.run:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P \
        tests/sharded_rank.py --backend nccl|gloo

nccl (RCCL): rank r on cuda:r.  gloo: ranks share the GPUs round-robin (a 1-GPU box runs every rank on cuda:0).  Every rank prints
one `RESULT {json}` line: its row block, the all-reduced sum (blocking form) and that the three async handles agreed with it.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_ROWS, DIM = 4099, 64          # ragged on purpose: shard sizes differ by one row
BIG_ROWS = (1 << 23) + 5        # cfg 5's batch (BASELINE.json configs[4]) plus five rows: `--rows big`


def flow_and_batch():
    import torch
    import stribor_amd as st
    from stribor_amd.util import flowdesc as fd
    torch.manual_seed(0)
    flow = fd.build_flow(st, fd.cfg2_desc(), DIM)
    x = torch.randn(N_ROWS, DIM, generator=torch.Generator().manual_seed(11))
    return flow, x


def device_rows(lo, hi, dev):
    """Rows [lo, hi) of the big batch, generated ON the device in 2^16-row pieces seeded by the piece index: any rank (and the
    single-process reference on rank 0) gets the same rows for the same range without a 2 GiB host tensor per rank."""
    import torch
    P = 1 << 16
    out = torch.empty(hi - lo, DIM, device=dev)
    g = torch.Generator(device=dev)
    for piece in range(lo // P, (hi + P - 1) // P):
        g.manual_seed(1000 + piece)
        blk = torch.randn(P, DIM, device=dev, generator=g)
        a, b = max(lo, piece * P), min(hi, (piece + 1) * P)
        out[a - lo:b - lo] = blk[a - piece * P:b - piece * P]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl')
    ap.add_argument('--rows', choices=['small', 'big'], default='small')
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.backend == 'nccl':
        dev = torch.device('cuda', local)
        torch.cuda.set_device(dev)
        dist.init_process_group('nccl', device_id=dev)
    else:
        dev = torch.device('cuda', local % max(1, torch.cuda.device_count()))
        torch.cuda.set_device(dev)
        dist.init_process_group('gloo')
    try:
        from stribor_amd.sharded import ShardedLogProb
        flow, x = flow_and_batch()
        flow = flow.to(dev)
        sh = ShardedLogProb(flow)
        n_rows = BIG_ROWS if args.rows == 'big' else N_ROWS
        lo, hi = sh.my_rows(n_rows)
        with torch.no_grad():
            mine = device_rows(lo, hi, dev) if args.rows == 'big' else x[lo:hi].to(dev)
            total = sh.log_prob_sum(mine)
            outs = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(3)]
            pend = [sh.log_prob_sum_async(mine, o) for o in outs]
            agree = all(abs(p.wait().item() - total.item()) <= 1e-12 * abs(total.item()) for p in pend)
            single = None
            if rank == 0:
                del mine
                single = flow.log_prob_sum(device_rows(0, n_rows, dev) if args.rows == 'big' else x.to(dev)).item()
        print('RESULT ' + json.dumps({'rank': rank, 'lo': lo, 'hi': hi, 'n_rows': n_rows, 'total': total.item(), 'world': dist.get_world_size(),
                                      'async_agree': bool(agree), 'single': single, 'device': str(dev)}), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
