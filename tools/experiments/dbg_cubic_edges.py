#!/usr/bin/env python3
"""Round 6: rows exactly on / one ulp around the bounds through fused spline couplings (K <= 16 straight-line phases) -- the fused
tier, the layer-by-layer tier (STRIBOR_CUBIC_UNFUSED=1 for cubic) and the oracle (fp32 and fp64) side by side, row by row."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

import flowdesc as fd
import stribor_amd as st
from oracle import stribor_oracle as orc

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
stype = sys.argv[2] if len(sys.argv) > 2 else 'cubic'
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 1
torch.manual_seed(100 * K + len(stype))
dim, lo, hi = 24, -2.0, 2.0
desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [32], 'n_bins': K, 'lower': lo, 'upper': hi, 'spline_type': stype,
         'mask': ('ordered_right_half', 'ordered_left_half', 'parity_odd')[i % 3], 'latent_dim': 0} for i in range(layers)]
flow = fd.build_flow(st, desc, dim)
spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
spec64 = fd.flow_spec(desc, {k: v.clone().double() for k, v in flow.state_dict().items()})
flow = flow.to('cuda')
edge = torch.tensor([lo, hi, float(np.nextafter(np.float32(lo), np.float32(0))), float(np.nextafter(np.float32(hi), np.float32(0))),
                     float(np.nextafter(np.float32(lo), np.float32(-9))), float(np.nextafter(np.float32(hi), np.float32(9))),
                     hi - 1e-3, hi - 0.05, lo + 1e-3, 2.5, -2.5, 0.0])
x = torch.randn(16, dim) * 1.2
for r in range(len(edge)):
    x[r] = edge[r]
want = orc.flow_log_prob(spec, x)
want64 = orc.flow_log_prob(spec64, x.double())
got = flow.log_prob(x.cuda()).cpu()
os.environ['STRIBOR_CUBIC_UNFUSED'] = '1'
flow2 = fd.build_flow(st, desc, dim)
flow2.load_state_dict({k: v.cpu() for k, v in flow.state_dict().items()})
flow2 = flow2.cuda()
got2 = flow2.log_prob(x.cuda()).cpu()
print(f'K={K} {stype} layers={layers}: row, x, fused, unfused, oracle fp32, oracle fp64')
for r in range(16):
    print(r, f'{x[r, 0].item():+.8f}', f'{got[r].item():.5f}', f'{got2[r].item():.5f}', f'{want[r].item():.5f}', f'{want64[r].item():.5f}',
          '' if abs(got[r].item() - want[r].item()) < 1e-3 else '  <-- fused off', '' if abs(got2[r].item() - want[r].item()) < 1e-3 else '  <-- unfused off')
