"""Training step of a cfg-3-shaped flow with CUBIC spline couplings (the reference's default spline_type), slab path vs the
per-row parameter path:  python tools/bench_cubic_train.py [rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stribor_amd as st  # noqa: E402
from stribor_amd.util import flowdesc as fd  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
dev = torch.device('cuda:0')
torch.manual_seed(0)
desc = [dict(d, spline_type='cubic') for d in fd.cfg3_desc()]
flow = fd.build_flow(st, desc, 64).to(dev)
x = torch.randn(rows, 64, device=dev)


def step():
    for p in flow.parameters():
        p.grad = None
    (-flow.log_prob(x).mean()).backward()


for mode in ('slab', 'unfused'):
    if mode == 'unfused':
        os.environ['STRIBOR_SPLINE_UNFUSED'] = '1'
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step(); step()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 2)
    ts.sort()
    print(json.dumps({'config': 'cfg3 shape with cubic splines, forward+backward', 'path': mode, 'rows': rows, 'ms_per_step': ts[2],
                      'rows_per_s': rows / ts[2] * 1e3}))
