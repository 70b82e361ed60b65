"""Timing of sx_tri_inverse_f64 on the cfg-4 training shape: 8 matrices of 128 x 128 (4 AffineLU layers: L and U)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stribor_amd.flows.linear import TriInverse

dev = 'cuda'
torch.manual_seed(0)
for k, D in ((8, 128), (8, 64), (1, 128)):
    W = torch.randn(k, D, D, dtype=torch.float64, device=dev) * 0.1
    L = torch.tril(W, -1) + torch.eye(D, dtype=torch.float64, device=dev)
    U = torch.triu(W, 1) + 1.5 * torch.eye(D, dtype=torch.float64, device=dev)
    for name, T, lower, unit in (('unit lower', L, True, True), ('upper', U, False, False)):
        for _ in range(3):
            X = TriInverse.apply(T, lower, unit)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            X = TriInverse.apply(T, lower, unit)
        e1.record()
        torch.cuda.synchronize()
        err = (X @ T - torch.eye(D, dtype=torch.float64, device=dev)).abs().max().item()
        print(f'{k} x {D} x {D} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch, |X T - I| max {err:.1e}')
