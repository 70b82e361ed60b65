#!/usr/bin/env python3
"""Throughput of every BASELINE.json config on one MI355X (not the driver's contract bench: see bench.py).

    python tools/bench_configs.py [cfg1 cfg2 cfg2_f32 cfg3 cfg4 ...] [--rows N] [--train] [--train-only] [--optim] [--graph]

Prints one JSON line per config: rows/s of log_prob, ms per batch, launches per batch.  --graph adds a line with the
same call captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed: what a launch-bound small batch costs
without the Python / ctypes launch path.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch

import flowdesc as fd
import stribor_amd as st


def timed(fn, reps=10, inner=4):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    return ts[len(ts) // 2]


CONFIGS = {
    'cfg1': (lambda: [{'kind': 'coupling_affine', 'dim': 2, 'hidden': [64], 'mask': 'ordered_right_half', 'latent_dim': 0}], 2, 1024, torch.float32),
    'cfg2': (lambda: fd.cfg2_desc(), 64, 1 << 20, torch.bfloat16),
    'cfg2_f32': (lambda: fd.cfg2_desc(), 64, 1 << 20, torch.float32),
    'cfg3': (lambda: fd.cfg3_desc(), 64, 1 << 20, torch.float32),
    'cfg4': (lambda: fd.cfg4_desc(), 128, 1 << 20, torch.float32),
    'cfg2_deep': (lambda: [dict(d, hidden=[64, 64]) for d in fd.cfg2_desc()], 64, 1 << 20, torch.float32),
    'cfg3_deep': (lambda: [dict(d, hidden=[64, 64]) for d in fd.cfg3_desc()], 64, 1 << 20, torch.float32),
    'cfg3_cubic': (lambda: [dict(d, spline_type='cubic') for d in fd.cfg3_desc()], 64, 1 << 20, torch.float32),
    'cfg3_k24': (lambda: fd.cfg3_desc(n_layers=4, n_bins=24), 64, 1 << 18, torch.float32),
    'cfg3_k8': (lambda: fd.cfg3_desc(n_layers=4, n_bins=8), 64, 1 << 18, torch.float32),
    'cfg2_parity': (lambda: [dict(d, mask='parity_even' if i % 2 == 0 else 'parity_odd') for i, d in enumerate(fd.cfg2_desc())], 64, 1 << 20, torch.float32),
}


def main():
    names = [a for a in sys.argv[1:] if not a.startswith('--') and not a.isdigit()] or list(CONFIGS)
    rows_override = None
    if '--rows' in sys.argv:
        rows_override = int(sys.argv[sys.argv.index('--rows') + 1])
    dev = torch.device('cuda', 0)
    for name in names:
        mk, dim, rows, dt = CONFIGS[name]
        rows = rows_override or rows
        torch.manual_seed(0)
        flow = fd.build_flow(st, mk(), dim).to(dev)
        x = torch.randn(rows, dim, device=dev).to(dt)
        fused = flow._fused_program(True, dim, 0, dev) is not None
        with torch.no_grad():
            ms = float('nan') if '--train-only' in sys.argv else timed(lambda: flow.log_prob(x))
            lp = flow.log_prob(x)
        if '--graph' in sys.argv:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(3):
                    flow.log_prob(x)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side), torch.no_grad():
                static_out = flow.log_prob(x)
            gms = timed(graph.replay, reps=10, inner=50)
            same = bool(torch.equal(static_out, lp))
            print(json.dumps({'config': name + ' HIP-graph replay', 'rows': rows, 'ms_per_batch': gms,
                              'rows_per_s': rows / (gms * 1e-3), 'equals_eager': same}))
        if name in ('cfg2_f32', 'cfg3', 'cfg4') and '--train' in sys.argv:
            # --optim: with an optimizer step (SGD, lr = 0: the weights keep their values, their versions move -- every Linear of
            # the flow is re-laid into fragments before the next forward, as in a real training loop)
            opt = torch.optim.SGD(flow.parameters(), lr=0.0) if '--optim' in sys.argv else None

            def train_step():
                for p_ in flow.parameters():
                    p_.grad = None
                loss = -flow.log_prob(x).mean()
                loss.backward()
                if opt is not None:
                    opt.step()
            tms = timed(train_step, reps=5, inner=2)
            print(json.dumps({'config': name + (' forward+backward+SGD step' if opt is not None else ' forward+backward (loss = -mean log_prob)'), 'rows': rows,
                              'ms_per_batch': tms, 'rows_per_s': rows / (tms * 1e-3)}))
        print(json.dumps({'config': name, 'rows': rows, 'dim': dim, 'x_dtype': str(dt), 'fused_single_launch': fused,
                          'ms_per_batch': ms, 'rows_per_s': rows / (ms * 1e-3), 'finite': bool(torch.isfinite(lp).all()),
                          'mean_log_prob': lp.mean().item()}))


if __name__ == '__main__':
    main()
