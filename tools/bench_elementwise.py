#!/usr/bin/env python3
"""HBM-roofline fractions of the stand-alone element-wise kernels (the unfused path), 2^20 rows."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import stribor_amd as st
from stribor_amd import _hip
from stribor_amd.flows.affine import run_affine_kernel
from stribor_amd.flows.spline import run_rqs_kernel
from tools.bench_configs import timed

dev = torch.device('cuda', 0)
N, D = 1 << 20, 64
PEAK = 8000.0
out = []
def rec(name, ms, nbytes):
    gbs = nbytes / (ms * 1e-3) / 1e9
    out.append({'kernel': name, 'ms': ms, 'algorithmic_bytes': nbytes, 'GB/s': gbs, 'frac_of_8TBs': gbs / PEAK})
    print(json.dumps(out[-1]))

for dt, sx in ((torch.bfloat16, 2), (torch.float32, 4)):
    x = torch.randn(N, D, device=dev).to(dt)
    params = torch.randn(N, D, device=dev) * 0.1
    ms = timed(lambda: run_affine_kernel(x, params, D, None, 0, D // 2, True, True, True, -1.0))
    rec(f'affine_coupling_vec x={dt}', ms, N * (2 * D * sx + D * 4 + 4))
    p = st.Permute(D).to(dev)
    ms = timed(lambda: p(x))
    rec(f'permute x={dt}', ms, N * 2 * D * sx)
    un = st.UnitNormal(D).to(dev)
    ms = timed(lambda: un.log_prob(x))
    rec(f'unit_normal_logprob x={dt}', ms, N * (D * sx + 4))
x = torch.randn(N, D, device=dev)
K = 16; P = 3 * K - 1; nl = 32
N2 = 1 << 18
x2 = torch.randn(N2, D, device=dev)
params = torch.randn(N2, nl * P, device=dev)
for rev in (False, True):
    ms = timed(lambda: run_rqs_kernel(x2, params, nl * P, None, 32, nl, K, -3., 3., -3., 3., rev, True, False))
    rec(f'rqs_kernel K=16 live=32 reverse={rev} (2^18 rows)', ms, N2 * (nl * P * 4 + 2 * D * 4 + 4))
from stribor_amd.flows.spline import run_cubic_kernel
Pc = 2 * K + 2
params_c = torch.randn(N2, nl * Pc, device=dev)
for rev in (False, True):
    ms = timed(lambda: run_cubic_kernel(x2, params_c, nl * Pc, None, 32, nl, K, -3., 3., rev, True, False))
    rec(f'cubic_kernel K=16 live=32 reverse={rev} (2^18 rows)', ms, N2 * (nl * Pc * 4 + 2 * D * 4 + 4))
from stribor_amd.flows.pointwise import run_pointwise, PW_SIGMOID, PW_CUMSUM
xs = torch.randn(N, D, device=dev)
ms = timed(lambda: run_pointwise(xs, PW_SIGMOID, want_ldj=True))
rec('pointwise sigmoid (+ldj) x=float32', ms, N * (2 * D * 4 + 4))
ms = timed(lambda: run_pointwise(xs, PW_CUMSUM))
rec('cumsum x=float32', ms, N * 2 * D * 4)
v = torch.randn(N * 64, device=dev)
o = torch.zeros(1, dtype=torch.float64, device=dev)
ms = timed(lambda: _hip.check(_hip.lib().sx_sum_f64(v.data_ptr(), v.numel(), o.data_ptr(), _hip.stream()), 'sum'))
rec('sum_f64 (2^26 floats)', ms, v.numel() * 4)
