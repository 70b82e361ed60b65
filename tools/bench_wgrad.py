"""Times sx_wgrad on the shapes the cfg-2 backward uses (224 features per row, 32-row groups): prints us per call
and the HBM rate of the operand features it reads.  Run on an MI355X."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from stribor_amd import _hip


def time_call(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    lib = _hip.lib()
    dev = torch.device('cuda:0')
    width = 224
    for n in (1 << 14, 1 << 16, 1 << 18, 1 << 20):
        side = torch.randn((n + 31) // 32, width, 32, device=dev)
        for (M, Nc, a0, b0) in ((64, 64, 160, 32), (64, 32, 96, 0), (128, 64, 96, 32), (128, 128, 0, 96)):
            dW = torch.zeros(M, Nc, device=dev)
            db = torch.zeros(M, device=dev)
            A, B = side[0, a0], side[0, b0]                     # feature a0 / b0 of group 0; ld = floats per group

            def fn():
                _hip.check(lib.sx_wgrad(A.data_ptr(), width * 32, M, B.data_ptr(), width * 32, Nc, n, _hip.WGRAD_ROW_GROUPS, dW.data_ptr(), Nc,
                                        db.data_ptr(), None, None, _hip.scratch(dev, lib.sx_wgrad_scratch_floats(M, Nc, _hip.WGRAD_ROW_GROUPS)).data_ptr(), _hip.stream()), 'sx_wgrad')
            us = time_call(fn)
            gb = n * (M + Nc) * 4 / 1e9
            print(f'n={n:8d} M={M:3d} Nc={Nc:3d}: {us:8.1f} us  {gb / (us * 1e-6) / 1e3:6.2f} TB/s (operand features)')


if __name__ == '__main__':
    main()
