"""log_prob of 4-layer rational-quadratic coupling flows with conditioners wider than the one-launch tier's 128 hidden units
(2^18 rows; the hidden-64 neighbour first): python tools/experiments/wide_spline.py [hidden ...] -> one JSON line per case, and a
check of every case against the oracle on 512 rows."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import stribor_amd as st
from stribor_amd.util import flowdesc as fd

ROWS = 1 << 18


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    from oracle import stribor_oracle as orc
    dev = torch.device('cuda', 0)
    hiddens = [int(a) for a in sys.argv[1:]] or [64, 160, 256]
    with torch.no_grad():
        for hidden in hiddens:
            torch.manual_seed(0)
            masks = ['ordered_right_half', 'ordered_left_half'] * 2
            desc = [{'kind': 'coupling_rqs', 'dim': 64, 'hidden': [hidden], 'mask': m, 'latent_dim': 0, 'n_bins': 16, 'lower': -3,
                     'upper': 3} for m in masks]
            flow = fd.build_flow(st, desc, 64)
            spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
            flow = flow.to(dev)
            x = torch.randn(ROWS, 64, device=dev)
            want = orc.flow_log_prob(spec, x[:512].cpu())
            got = flow.log_prob(x[:512]).cpu()
            err = ((got - want).abs() / want.abs().clamp_min(1.0)).max().item()
            fused = flow._fused_program(True, 64, 0, dev) is not None
            ms = timed(lambda: flow.log_prob(x))
            print(json.dumps({'coupling': 'rqs', 'hidden': hidden, 'rows': ROWS, 'layers': 4, 'one_fused_launch': fused, 'ms': ms,
                              'max_rel_err_vs_oracle': err}), flush=True)
            del flow, x


if __name__ == '__main__':
    main()
