#!/usr/bin/env python3
"""The stand-alone spline kernels alone (rqs_kernel / cubic_kernel, both directions + the cubic's reference-mode inverse) at
the cfg-3 shape: python tools/bench_spline_kernels.py [rows] [reps].  Driver of tools/pmc_spline_kernels.sh."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from stribor_amd import _hip
from stribor_amd.flows.spline import run_rqs_kernel, run_cubic_kernel
from tools.bench_configs import timed

dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
D, K, nl = 64, 16, 32
torch.manual_seed(0)
x = torch.randn(N, D, device=dev)


def rec(name, ms, nbytes):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print(json.dumps({'kernel': name, 'ms': ms, 'algorithmic_bytes': nbytes, 'GB/s': gbs, 'frac_of_8TBs': gbs / 8000.0,
                      'elements_per_s': N * nl / (ms * 1e-3)}), flush=True)


P = 3 * K - 1
params = torch.randn(N, nl * P, device=dev)
for rev in (False, True):
    ms = timed(lambda: run_rqs_kernel(x, params, nl * P, None, 32, nl, K, -3., 3., -3., 3., rev, True, False))
    rec(f'rqs_kernel reverse={rev}', ms, N * (nl * P * 4 + 2 * D * 4 + 4))
Pc = 2 * K + 2
params_c = torch.randn(N, nl * Pc, device=dev)
y = torch.empty_like(x)
ldj = torch.empty(N, device=dev)
for rev in (0, 1, 2):       # 2 = the inverse with the reference's log-det (what a coupling's log_prob runs)
    def go():
        _hip.call('sx_cubic_coupling', x, x.data_ptr(), y.data_ptr(), ldj.data_ptr(), None, params_c.data_ptr(), nl * Pc, None,
                  32, nl, K, -3.0, 3.0, N, D, _hip.dtype_code(x), rev, 0, 1.0)
    ms = timed(go)
    rec(f'cubic_kernel reverse={rev}', ms, N * (nl * Pc * 4 + 2 * D * 4 + 4))
