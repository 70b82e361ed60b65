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


def flow_and_batch():
    import torch
    import stribor_amd as st
    from stribor_amd.util import flowdesc as fd
    torch.manual_seed(0)
    flow = fd.build_flow(st, fd.cfg2_desc(), DIM)
    x = torch.randn(N_ROWS, DIM, generator=torch.Generator().manual_seed(11))
    return flow, x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl')
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
        lo, hi = sh.my_rows(N_ROWS)
        with torch.no_grad():
            total = sh.log_prob_sum(x[lo:hi].to(dev))
            outs = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(3)]
            pend = [sh.log_prob_sum_async(x[lo:hi].to(dev), o) for o in outs]
            agree = all(abs(p.wait().item() - total.item()) <= 1e-12 * abs(total.item()) for p in pend)
            single = flow.log_prob_sum(x.to(dev)).item() if rank == 0 else None
        print('RESULT ' + json.dumps({'rank': rank, 'lo': lo, 'hi': hi, 'total': total.item(), 'world': dist.get_world_size(),
                                      'async_agree': bool(agree), 'single': single, 'device': str(dev)}), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
