#!/usr/bin/env python3
"""Experiment: time of the fused kernel vs number of coupling layers (fixed cost vs per-step cost)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import flowdesc as fd
import stribor_amd as st
from tools.bench_configs import timed

dev = torch.device('cuda', 0)
x = torch.randn(1 << 20, 64, device=dev)
for L in (0, 1, 2, 4, 8, 16):
    torch.manual_seed(0)
    desc = fd.cfg2_desc(L) if L else [{'kind': 'flip'}]
    flow = fd.build_flow(st, desc, 64).to(dev)
    ms = timed(lambda: flow.log_prob(x))
    print(f'L={L} {ms:.4f} ms')
