"""log_prob time of 4-layer affine coupling flows at 128 / 160 / 200 / 256 columns (hidden 64, 2^18 rows): the eight-tile programs (MODE 20)."""
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
    for dim in [int(a) for a in sys.argv[1:]] or [128, 160, 200, 256]:
        torch.manual_seed(0)
        masks = ['ordered_right_half', 'ordered_left_half'] * 2
        desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [64], 'mask': m, 'latent_dim': 0} for m in masks]
        flow = fd.build_flow(st, desc, dim).to(dev)
        x = torch.randn(rows, dim, device=dev)
        ms = timed(lambda: flow.log_prob(x))
        print(json.dumps({'dim': dim, 'ms': round(ms, 4), 'one_fused_launch': flow._fused_program(True, dim, 0, dev) is not None}), flush=True)
