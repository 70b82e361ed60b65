"""Quantifies the fused tier's cliffs: log_prob time per 2^18 rows of 4-layer flows just inside and just beyond
each limit of the one-launch tier -- hidden width 128 | 160, columns 128 | 160, spline bins 16 | 24 -- with the tier
that answered.  python tools/bench_cliffs.py  -> one JSON line per case."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    dev = torch.device('cuda', 0)
    cases = []
    for dim, hidden in ((64, 64), (64, 128), (64, 160), (64, 256), (128, 64), (160, 64), (200, 64)):
        cases.append(('affine', dim, hidden, 0))
    cases.append(('affine', 128, 160, 0))
    for K in (4, 8, 12, 16, 17, 24, 32):      # (round 6: every K <= 16 on the straight-line phases; 17 .. 32 on their two-tile form)
        cases.append(('rqs', 64, 64, K))
    cases.append(('rqs', 64, 160, 16))
    cases.append(('rqs', 64, 256, 16))
    cases.append(('cubic', 64, 64, 8))
    cases.append(('cubic', 64, 64, 16))
    cases.append(('cubic', 64, 160, 16))
    with torch.no_grad():
        for kind, dim, hidden, K in cases:
            torch.manual_seed(0)
            masks = ['ordered_right_half', 'ordered_left_half'] * 2
            if kind == 'affine':
                desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': m, 'latent_dim': 0} for m in masks]
            else:
                desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'mask': m, 'latent_dim': 0, 'n_bins': K, 'lower': -3,
                         'upper': 3, 'spline_type': 'cubic' if kind == 'cubic' else 'quadratic'} for m in masks]
            flow = fd.build_flow(st, desc, dim).to(dev)
            x = torch.randn(ROWS, dim, device=dev)
            fused = flow._fused_program(True, dim, 0, dev) is not None
            tier = 'one launch' if fused else 'layer by layer'
            if not fused and kind in ('rqs', 'cubic'):
                try:                            # the slab forward tier (hidden layers beyond 128 units): two launches per layer
                    flow.transforms[0]._spline_slab_plan(dim, 0, dev)
                    tier = 'slab forward'
                except NotImplementedError:
                    pass
            ms = timed(lambda: flow.log_prob(x))
            print(json.dumps({'coupling': kind, 'dim': dim, 'hidden': hidden, 'n_bins': K, 'layers': 4, 'rows': ROWS,
                              'one_fused_launch': fused, 'tier': tier, 'ms': ms, 'rows_per_s': ROWS / ms * 1e3}), flush=True)
            del flow, x


if __name__ == '__main__':
    main()
