"""Timing of sx_wgrad (fp16 x 3 row-group layout) on the cfg-4 training shapes: (M, Nc) = (128, 128) dense layers,
(128, 64) dW2 and (64, 64) dW1 of a coupling, group stride 320 features as in the backward program's side buffer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stribor_amd import _hip

dev = 'cuda'
lib = _hip.lib()
width = 320
for rows in (1 << 14, 1 << 16, 1 << 18, 1 << 20):
    ng = (rows + 31) // 32
    side = torch.randn(ng, width, 32, device=dev) * 0.1
    ld = width * 32
    for M, Nc, offA, offB in ((128, 128, 0, 128), (128, 64, 192, 64), (64, 64, 128, 0)):
        dW = torch.zeros(M, Nc, device=dev)
        db = torch.zeros(M, device=dev)
        sc = _hip.scratch(dev, lib.sx_wgrad_scratch_floats(M, Nc, _hip.WGRAD_ROW_GROUPS))
        p0 = side.data_ptr()
        def run():
            _hip.check(lib.sx_wgrad(p0 + 128 * offA, ld, M, p0 + 128 * offB, ld, Nc, rows, _hip.WGRAD_ROW_GROUPS_F16X3, dW.data_ptr(),
                                    dW.stride(0), db.data_ptr(), None, None, sc.data_ptr(), _hip.stream()), 'sx_wgrad')
        for _ in range(3):
            run()
        dW.zero_(); run()
        A = side[:, offA:offA + M, :].permute(0, 2, 1).reshape(-1, M)[:rows].double()
        Bm = side[:, offB:offB + Nc, :].permute(0, 2, 1).reshape(-1, Nc)[:rows].double()
        want = A.T @ Bm
        err = ((dW.double() - want).abs().max() / want.abs().max()).item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        mb = rows * (M + Nc) * 4 / 1e6
        print(f'rows 2^{rows.bit_length() - 1} {M:3d} x {Nc:3d}: {us:7.1f} us (contraction + reduce kernels), {mb / us:5.2f} TB/s of operand bytes, rel err {err:.1e}')
