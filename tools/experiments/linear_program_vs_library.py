"""BatchLinear's program launches (net/mlp.py, SX_STEP_MLP_INPUT) against the library GEMM of the same product, forward and dL/dx:
    python tools/experiments/linear_program_vs_library.py
ms per call at 2^18 rows, median of 10 x 4 calls."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
from bench_configs import timed  # noqa: E402
from stribor_amd.net.mlp import BatchLinear  # noqa: E402

n = 1 << 18
for in_dim, out_dim in [(64, 64), (128, 256), (64, 1504), (200, 64), (256, 256)]:
    x = torch.randn(n, in_dim, device='cuda')
    gy = torch.randn(n, out_dim, device='cuda')
    W, b = torch.randn(out_dim, in_dim, device='cuda'), torch.randn(out_dim, device='cuda')
    with torch.no_grad():
        row = {'in': in_dim, 'out': out_dim,
               'forward program ms': timed(lambda: BatchLinear._program_linear(x, W, b, False)),
               'forward library ms': timed(lambda: torch.nn.functional.linear(x, W, b)),
               'dL/dx program ms': timed(lambda: BatchLinear._program_linear(gy, W, None, True)),
               'dL/dx library ms': timed(lambda: gy @ W)}
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()})
