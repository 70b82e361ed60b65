#!/usr/bin/env python3
"""Sweeps the experiment knobs of the stand-alone affine coupling kernel (SX_AFFINE_VARIANT = 10 * rows-in-flight + nt,
SX_AFFINE_GRID = workgroups per CU): one child process per setting (the library reads the knobs once)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
from stribor_amd.flows.affine import run_affine_kernel
from tools.bench_configs import timed
dev = torch.device('cuda', 0); N, D = 1 << 20, 64
for dt, sx in ((torch.bfloat16, 2), (torch.float32, 4)):
    x = torch.randn(N, D, device=dev).to(dt); params = torch.randn(N, D, device=dev) * 0.1
    ms = timed(lambda: run_affine_kernel(x, params, D, None, 0, D // 2, True, True, True, -1.0))
    b = N * (2 * D * sx + D * 4 + 4)
    print(json.dumps({'variant': os.environ.get('SX_AFFINE_VARIANT'), 'grid': os.environ.get('SX_AFFINE_GRID'), 'x': str(dt), 'ms': ms, 'frac_of_8TBs': b / (ms * 1e-3) / 8e12}))
''' % (ROOT, ROOT)
for variant in ('10', '11', '20', '21', '40', '41'):
    for grid in ('8', '16', '32'):
        env = dict(os.environ, SX_AFFINE_VARIANT=variant, SX_AFFINE_GRID=grid)
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
        for l in out.stdout.splitlines():
            if l.startswith('{'):
                print(l, flush=True)
        if out.returncode:
            print(out.stderr[-500:], flush=True)
