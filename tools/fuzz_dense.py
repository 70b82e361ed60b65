"""Randomised differential check of the 128-column kernel family (round 5: k-major dense layers, the re-pipelined split coupling,
the per-sample power-of-two rescale): random stacks of AffineLU / MatrixExponential / affine couplings / Flip at 33 .. 128 columns,
hidden 4 .. 64, split or parity masks (the split-mask stacks are the pure MODE 5 / 7 programs, the others the general ones), 1 .. 700
rows, optionally rows scaled to 1e4 .. 1e6 (beyond fp16's range: the rescale paths) and bf16 storage -- log_prob, inverse + log-det and
forward + log-det of the HIP path against the fp64 oracle.  Values that went through dense layers are held relative to the row's
largest entry; an out-of-range row to 16 x the fp32 oracle's own error (tests/test_gpu_precision.py).
    python tools/fuzz_dense.py [n_cases] [seed] [--big] [--mid] [--bf16] [--exact] [--detail]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import stribor_amd as st  # noqa: E402
from stribor_amd.util import flowdesc as fd  # noqa: E402
import stribor_oracle as orc  # noqa: E402

DEV = 'cuda:0'
# allowance in units of the fp32 oracle's own error against fp64 (per row): 16 on ordinary data; rows of 1e4 .. 1e6 (--big) make the
# conditioners' pre-activations cancelling sums of huge terms.  Where such a sum lands inside tanh's linear range (|pre| < 3: a few rows
# per hundred cases) the fp16 x 3 weights' ABSOLUTE resolution (3e-8: the low half of a weight below 0.125 is an fp16 subnormal) times a
# state entry of 1e5 shows: 20 .. 170 x fp32's error on that row (tools/experiments/dbg_big_layers.py); 'exact' stays at fp32's
# (round 6: rows beyond fp16's range are evaluated by the exact-fp32 kernel in 'fast' as well -- the redo pass of sx_flow_run2 --, so --big holds
#  them to the same 16 x as --exact; round 5's 64 x was the allowance for the fp16 weights' low halves against entries of 1e5)
KREF = 16.0
BIG = '--big' in sys.argv
MID = '--mid' in sys.argv      # rows of 30 .. 6e4: INSIDE fp16's range, either side of SX_REDO_ABOVE = 2048 (round 6: the fp16 weights' absolute
                               # resolution times a large entry -- 27 x fp32's error on a row of 6.4e4, seed 914 case 85 -- is why rows are
                               # named from 2048 on, not from 65504)
BF16 = '--bf16' in sys.argv


def case(rng):
    full = rng.random() < 0.6                                  # whole tiles (64 / 128 columns: vector loads, k-major dense arm) or ragged
    dim = int(rng.choice([64, 128, 128, 96])) if full else int(rng.integers(33, 129))
    split = rng.random() < 0.7
    masks = ['ordered_right_half', 'ordered_left_half'] if split else ['parity_even', 'parity_odd', 'ordered_right_half']
    hidden = int(rng.choice([64, 64, 32, 16, 48, 5]))
    desc = []
    for i in range(int(rng.integers(2, 9))):
        kind = str(rng.choice(['affine_lu', 'matrix_exp', 'coupling_affine', 'coupling_affine', 'flip' if not split else 'coupling_affine']))
        if kind == 'coupling_affine':
            desc.append({'kind': kind, 'dim': dim, 'hidden': [hidden], 'mask': str(masks[i % len(masks)]), 'latent_dim': 0})
        elif kind == 'matrix_exp':
            desc.append({'kind': kind, 'dim': dim, 'bias': bool(rng.random() < 0.3), 'log_time': False})
        elif kind == 'affine_lu':
            desc.append({'kind': kind, 'dim': dim})
        else:
            desc.append({'kind': 'flip'})
    if not any(d['kind'] == 'coupling_affine' for d in desc):
        desc.append({'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': masks[0], 'latent_dim': 0})
    return desc, dim, int(rng.integers(1, 700))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n_cases = int(args[0]) if args else 40
    seed = int(args[1]) if len(args) > 1 else 0
    rng = np.random.default_rng(seed)
    if '--exact' in sys.argv:
        st.set_gemm_precision('exact')
    worst = 0.0
    fails = 0
    for i in range(n_cases):
        desc, dim, n = case(rng)
        torch.manual_seed(seed * 1000 + i)
        flow = fd.build_flow(st, desc, dim)
        with torch.no_grad():
            for name, p in flow.named_parameters():
                if p.dim() == 2 and p.shape == (dim, dim):
                    p.mul_(0.08)                               # dense layers near the identity: deep random stacks stay conditioned
                else:
                    p.add_(torch.randn_like(p) * 0.03)
        state = {k: v.clone() for k, v in flow.state_dict().items()}
        flow = flow.to(DEV)
        x = torch.randn(n, dim) * 1.3
        big_rows = torch.zeros(n, dtype=torch.bool)
        if MID:
            for r in rng.choice(n, size=min(n, int(rng.integers(1, 9))), replace=False):
                if rng.random() < 0.5:
                    x[r] *= float(10.0 ** rng.uniform(1.5, 4.8)) / x[r].abs().max()
                else:
                    x[r, int(rng.integers(0, dim))] = float(10.0 ** rng.uniform(1.5, 4.8)) * (1 if rng.random() < 0.5 else -1)
                big_rows[r] = True
        if BIG:
            for r in rng.choice(n, size=min(n, int(rng.integers(1, 5))), replace=False):
                if rng.random() < 0.5:
                    x[r] *= float(10.0 ** rng.uniform(4.0, 6.0)) / x[r].abs().max()
                else:
                    x[r, int(rng.integers(0, dim))] = float(10.0 ** rng.uniform(4.8, 6.0)) * (1 if rng.random() < 0.5 else -1)
                big_rows[r] = True
        if BF16:
            x = x.bfloat16().float()
        spec32 = fd.flow_spec(desc, state)
        spec64 = orc.spec_to(spec32, torch.float64)
        xd = x.to(DEV).bfloat16() if BF16 else x.to(DEV)
        with torch.no_grad():
            lp = flow.log_prob(xd).cpu().double().reshape(n)
            z, li = flow.inverse_and_log_det_jacobian(xd)
            y, lf = flow.forward_and_log_det_jacobian(xd)
        st.check_errors()
        z, y = z.float().cpu().double(), y.float().cpu().double()
        li, lf = li.cpu().double().reshape(n), lf.cpu().double().reshape(n)
        w_lp = orc.flow_log_prob(spec64, x.double()).reshape(n)
        wz, wli = orc.flow_inverse_and_ldj(spec64, x.double())
        wy, wlf = orc.flow_forward_and_ldj(spec64, x.double())
        r_lp = orc.flow_log_prob(spec32, x).double().reshape(n)
        rz, rli = orc.flow_inverse_and_ldj(spec32, x)
        ry, rlf = orc.flow_forward_and_ldj(spec32, x)
        rz, ry, rli, rlf = rz.double(), ry.double(), rli.double().reshape(n), rlf.double().reshape(n)

        def rowerr(g, r, t, rtol):
            """max over rows of err / (16 x the fp32 oracle's row error + rtol x row max)"""
            g, r, t = g.reshape(n, -1), r.reshape(n, -1), t.reshape(n, -1)
            bound = KREF * (r - t).abs().amax(1) + rtol * t.abs().amax(1).clamp_min(1.0)
            return ((g - t).abs().amax(1) / bound).max().item()
        vt = 6e-3 if BF16 else 2e-5                            # bf16 outputs carry 8 bits
        e = {'lp': rowerr(lp, r_lp, w_lp, 2e-5), 'z': rowerr(z, rz, wz, vt), 'y': rowerr(y, ry, wy, vt),
             'ldj': ((li - wli.reshape(n)).abs() / (KREF * (rli - wli.reshape(n)).abs() + 3e-4 + 3e-5 * wli.reshape(n).abs())).max().item(),
             'ldj_f': ((lf - wlf.reshape(n)).abs() / (KREF * (rlf - wlf.reshape(n)).abs() + 3e-4 + 3e-5 * wlf.reshape(n).abs())).max().item()}
        m = max(e.values())
        worst = max(worst, m)
        fused = flow._fused_program(True, dim, 0, torch.device(DEV)) is not None
        kinds = ''.join({'affine_lu': 'L', 'matrix_exp': 'M', 'coupling_affine': 'c', 'flip': 'f'}[d['kind']] for d in desc)
        bad = m > 1.0 or not bool(torch.isfinite(lp).all())
        if bad and '--detail' in sys.argv:
            for key, g, r, t in (('lp', lp, r_lp, w_lp), ('ldj', li, rli, wli.reshape(n)), ('ldj_f', lf, rlf, wlf.reshape(n)),
                                 ('z', z, rz, wz), ('y', y, ry, wy)):
                g2, r2, t2 = g.reshape(n, -1), r.reshape(n, -1), t.reshape(n, -1)
                err = (g2 - t2).abs().amax(1)
                rr = int(err.argmax())
                print(f'      {key}: worst row {rr} (big {bool(big_rows[rr])}) ours {err[rr]:.3e} fp32 oracle {(r2 - t2).abs().amax(1)[rr]:.3e} '
                      f'row max {t2[rr].abs().max():.3e} x row max {x[rr].abs().max():.3e}')
        fails += bad
        print(f'case {i:3d} dim {dim:3d} n {n:3d} {kinds:10s} fused {int(fused)} big {int(big_rows.sum())} ' +
              ' '.join(f'{k} {v:.2f}' for k, v in e.items()) + ('  FAIL' if bad else ''), flush=True)
    print(f'worst {worst:.2f} (1.0 = the bound), {fails} failures of {n_cases}')
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
