"""The PCIe-inclusive rate of the headline workload: cfg 2's 2^20 x 64 bf16 rows start in PINNED HOST memory, are copied to the device,
evaluated (log_prob, one fused launch) and the 2^20 fp32 results copied back -- what a caller that holds host buffers pays.  Never
bench.py's `value` (its inputs are resident in HBM when the timed region starts); quoted in DESIGN.md section 6.
    python tools/bench_pcie.py [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stribor_amd as st  # noqa: E402
from stribor_amd.util import flowdesc as fd  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    flow = fd.build_flow(st, fd.cfg2_desc(), 64).to(dev)
    n, d = 1 << 20, 64
    host = (torch.randn(n, d) * 1.2).bfloat16().pin_memory()
    out_host = torch.empty(n, 1, dtype=torch.float32).pin_memory()
    x = torch.empty(n, d, dtype=torch.bfloat16, device=dev)

    def once(overlap_chunks=1):
        rows = n // overlap_chunks
        for c in range(overlap_chunks):
            sl = slice(c * rows, (c + 1) * rows)
            x[sl].copy_(host[sl], non_blocking=True)
            with torch.no_grad():
                lp = flow.log_prob(x[sl])
            out_host[sl].copy_(lp, non_blocking=True)
    res = {}
    for name, chunks in (('one copy + one launch', 1), ('8 row blocks on one stream', 8)):
        for _ in range(3):
            once(chunks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            once(chunks)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res[name] = {'ms': dt * 1e3, 'rows_per_s': n / dt, 'h2d_GBps_equiv': n * d * 2 / dt / 1e9}
    # the copy alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        x.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res['h2d copy alone'] = {'ms': dt * 1e3, 'GBps': n * d * 2 / dt / 1e9}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
