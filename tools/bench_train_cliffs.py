"""Training-step cliffs of spline couplings: forward + backward of -log_prob.mean() per 2^18 rows of 4-layer
rational-quadratic coupling flows (64 columns, 16 bins) at the cfg-3 hidden width and beyond it, with the path that answered;
STRIBOR_SPLINE_UNFUSED=1 python tools/bench_train_cliffs.py  times the per-row parameter path.  -> one JSON line per case."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stribor_amd as st
from stribor_amd.util import flowdesc as fd

ROWS = 1 << 18


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device('cuda', 0)
    hiddens = [int(a) for a in sys.argv[1:]] or [64, 96, 128, 160]
    for hidden in hiddens:
        torch.manual_seed(0)
        masks = ['ordered_right_half', 'ordered_left_half'] * 2
        desc = [{'kind': 'coupling_rqs', 'dim': 64, 'hidden': [hidden], 'mask': m, 'latent_dim': 0, 'n_bins': 16, 'lower': -3,
                 'upper': 3} for m in masks]
        flow = fd.build_flow(st, desc, 64).to(dev)
        x = torch.randn(ROWS, 64, device=dev)

        def step():
            for p in flow.parameters():
                p.grad = None
            (-flow.log_prob(x).mean()).backward()
        ms = timed(step)
        with torch.no_grad():
            fwd = timed(lambda: flow.log_prob(x))
        cpl = flow.transforms[0]
        print(json.dumps({'coupling': 'rqs', 'hidden': hidden, 'rows': ROWS, 'layers': 4, 'train_step_ms': ms, 'forward_ms': fwd,
                          'slab_backward': bool(cpl._slab_backward_ok(cpl.transform.latent_net, cpl.transform)),
                          'unfused_switch': os.environ.get('STRIBOR_SPLINE_UNFUSED', '0')}), flush=True)
        del flow, x


if __name__ == '__main__':
    main()
