"""log_prob time of 4-layer rational-quadratic spline coupling flows (D = 64, hidden 64, 2^18 rows) by bin count: the K = 16 bounded-logit
path against the K-generic sweeps (K != 16) and their two-tile form (17 .. 32 bins)."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
import torch
import stribor_amd as st
from stribor_amd.util import flowdesc as fd
from bench_cliffs import timed
dev = torch.device('cuda', 0)
rows = 1 << 18
with torch.no_grad():
    for K in [int(a) for a in sys.argv[1:]] or [8, 12, 16, 24, 32]:
        torch.manual_seed(0)
        masks = ['ordered_right_half', 'ordered_left_half'] * 2
        desc = [{'kind': 'coupling_rqs', 'dim': 64, 'hidden': [64], 'mask': m, 'latent_dim': 0, 'n_bins': K, 'lower': -3, 'upper': 3} for m in masks]
        flow = fd.build_flow(st, desc, 64).to(dev)
        x = torch.randn(rows, 64, device=dev)
        ms = timed(lambda: flow.log_prob(x))
        print(json.dumps({'n_bins': K, 'ms': round(ms, 4), 'one_fused_launch': flow._fused_program(True, 64, 0, dev) is not None}), flush=True)
