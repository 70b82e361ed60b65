"""cProfile of the eager launch path of a small log_prob call (cfg 1: 1024 rows, dim 2): where the host time of a
launch-bound call goes.  Run on an MI355X."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import flowdesc as fd
import stribor_amd as st

dev = torch.device('cuda', 0)
desc = [{'kind': 'coupling_affine', 'dim': 2, 'hidden': [64], 'mask': 'ordered_right_half', 'latent_dim': 0}]
flow = fd.build_flow(st, desc, 2).to(dev)
x = torch.randn(1024, 2, device=dev)
with torch.no_grad():
    for _ in range(100):
        flow.log_prob(x)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        flow.log_prob(x)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
