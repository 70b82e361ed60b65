"""Times sx_rqs_slab_bwd alone (one spline-coupling layer's fused backward) on synthetic inputs.
    python tools/bench_slab.py [--rows N] [--dim D] [--hidden H] [--bins K]
Prints one JSON line: ms per launch (main kernel + the two reduce kernels), rows/s."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stribor_amd import _hip  # noqa: E402
from stribor_amd.flows.spline import slab_slot_rows  # noqa: E402


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    n, d, H, K = arg('--rows', 1 << 18), arg('--dim', 64), arg('--hidden', 64), arg('--bins', 16)
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    n_live, P = d // 2, 3 * K - 1
    x = torch.rand(n, d, device=dev) * 1.2 - 0.1
    gy = torch.randn(n, d, device=dev)
    gl = torch.randn(n, device=dev)
    h = torch.tanh(torch.randn(n, H, device=dev))
    W2 = torch.randn(n_live * P, H, device=dev) * 0.3
    b2 = torch.randn(n_live * P, device=dev) * 0.1
    lib = _hip.lib()
    slots, ht = lib.sx_rqs_slab_slots(n_live), (H + 31) // 32
    mt = slots // 32
    slot_rows = torch.from_numpy(slab_slot_rows(n_live, K)).to(dev)
    hid = np.full(ht * 32, -1, dtype=np.int32)
    hid[:H] = np.arange(H)
    hid = torch.from_numpy(hid).to(dev)
    n_fwd = _hip.packed_linear_floats(mt, ht)
    packs = torch.empty(n_fwd + ht * mt * 1024, dtype=torch.float32, device=dev)
    flag = _hip.err_flag(dev)
    _hip.call('sx_pack_linear', x, W2.data_ptr(), b2.data_ptr(), W2.shape[0], H, slot_rows.data_ptr(), hid.data_ptr(), mt, ht,
              None, None, 0.0, 0, _hip.GEMM_F16X3, flag, packs.data_ptr())
    _hip.call('sx_pack_linear', x, W2.data_ptr(), None, W2.shape[0], H, hid.data_ptr(), slot_rows.data_ptr(), ht, mt, None, None,
              0.0, 1, _hip.GEMM_F16X3, flag, packs.data_ptr() + 4 * n_fwd)
    gx, gh = gy.clone(), torch.empty(n, H, device=dev)
    gW, gb = torch.zeros_like(W2), torch.zeros_like(b2)
    sc = torch.empty(lib.sx_rqs_slab_scratch_floats(n, n_live, H), dtype=torch.float32, device=dev)

    def run():
        _hip.call('sx_rqs_slab_bwd', x, x.data_ptr(), gy.data_ptr(), gl.data_ptr(), None, h.data_ptr(), h.stride(0), H, packs.data_ptr(),
                  packs.data_ptr() + 4 * n_fwd, slot_rows.data_ptr(), gx.data_ptr(), gh.data_ptr(), gh.stride(0), gW.data_ptr(),
                  gW.stride(0), gb.data_ptr(), None, n_live, n_live, K, 0.0, 1.0, 0.0, 1.0, n, d, 1.0, 0, None, sc.data_ptr(), flag)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    print(json.dumps({'kernel': 'sx_rqs_slab_bwd', 'lib': os.path.basename(_hip.LIB_PATH), 'rows': n, 'dim': d, 'hidden': H,
                      'bins': K, 'ms': best, 'rows_per_s': n / best * 1e3, 'finite': bool(torch.isfinite(gW).all())}))


if __name__ == '__main__':
    main()
