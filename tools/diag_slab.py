"""Diagnostic: gradients of a 2-layer quadratic-spline flow through (a) the fused slab backward, (b) the per-row parameter
path, (c) fp64 autograd of the oracle -- max errors relative to each tensor's scale."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stribor_amd as st  # noqa: E402
from stribor_amd.util import flowdesc as fd  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import stribor_oracle as orc  # noqa: E402

DEV = 'cuda:0'


def main():
    n, dim, hidden, K = (int(sys.argv[2]) if len(sys.argv) > 2 else 1000), 64, 64, 16
    pert = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
    torch.manual_seed(21)
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -2.5, 'upper': 2.5,
             'mask': 'ordered_right_half' if i % 2 == 0 else 'parity_even', 'latent_dim': 0} for i in range(2)]
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(torch.randn_like(p) * pert)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(n, dim) * 1.5
    leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items()}
    xin = x.double().clone().requires_grad_(True)
    (-orc.flow_log_prob(fd.flow_spec(desc, leaves), xin).mean()).backward()
    ref = [xin.grad] + [leaves[k].grad for k, _ in flow.named_parameters()]

    def grads():
        for p in flow.parameters():
            p.grad = None
        xg = x.to(DEV).requires_grad_(True)
        (-flow.log_prob(xg).mean()).backward()
        return [xg.grad.cpu().double()] + [p.grad.cpu().double() for p in flow.parameters()]
    fused = grads()
    os.environ['STRIBOR_SPLINE_UNFUSED'] = '1'
    unfused = grads()
    names = ['x'] + [k for k, _ in flow.named_parameters()]
    for nm, a, b, r in zip(names, fused, unfused, ref):
        sc = r.abs().max().item() + 1e-30
        print(f'{nm:28s} scale {sc:9.3e}  slab-f64 {(a - r).abs().max().item() / sc:9.2e}  unfused-f64 '
              f'{(b - r).abs().max().item() / sc:9.2e}  slab-unfused {(a - b).abs().max().item() / sc:9.2e}'
              + (f'  elements off by > 3e-4 scale: slab {((a - r).abs() > 3e-4 * sc).sum().item()}, unfused {((b - r).abs() > 3e-4 * sc).sum().item()}' if nm == 'x' else ''))


if __name__ == '__main__':
    main()
