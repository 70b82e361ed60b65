"""Times the reference's flagship stack (affine coupling -> Flip -> Sigmoid -> cubic-spline coupling -> Logit, D = 64, K = 16) as one
fused launch (kernel MODE 14) and layer by layer; 2^20 rows.  python tools/bench_stack.py"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stribor_amd as st
import bench


def main():
    dev = torch.device('cuda', 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    from stribor_amd.util import flowdesc as fd
    with torch.no_grad():
        print(json.dumps(bench.extra_flow_entries(st, fd, dev, gen, 20)))


if __name__ == '__main__':
    main()
