"""Randomised differential check of the training paths: random spline / affine coupling flows (dims, widths, masks, bin counts,
spline types, latent inputs, batch sizes) -- gradients of -log_prob.mean() from the HIP paths against fp64 autograd of the oracle.
    python tools/fuzz_train.py [n_cases] [seed] [--forward] [--infer] [--mix] [--time] [--wide] [--xwide] [--long] [--fat] [--bf16] [--sets] [--k16] [--poison] [--only=i ...]
(--forward: forward_and_log_det_jacobian instead of log_prob; --infer: the no-graph paths; --mix: every transform kind)"""
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
MASKS = ['ordered_right_half', 'ordered_left_half', 'parity_even', 'parity_odd']


WIDE = '--wide' in sys.argv          # 65 .. 125 columns: the four-tile kernel variants (one wave per SIMD)
FAT = '--fat' in sys.argv            # hidden widths 65 .. 200 and up to 32 bins: the four-hidden-tile variants and the tiers beyond them
XWIDE = '--xwide' in sys.argv        # 129 .. 256 columns: affine couplings on eight state tiles (kernel MODE 20), everything else layer by layer
LONG = '--long' in sys.argv          # 10 .. 40 layers: flows beyond one 128-step program run as segments


def case(rng):
    dim = int(rng.integers(129, 257)) if XWIDE else int(rng.integers(65, 126)) if WIDE else int(rng.integers(2, 71))
    latent = 0 if XWIDE else int(rng.choice([0, 0, 0, 3]))
    layers = int(rng.integers(10, 41)) if LONG else int(rng.integers(1, 4))
    desc = []
    for _ in range(layers):
        kind = rng.choice(['affine', 'affine', 'affine', 'rqs'] if XWIDE else ['rqs', 'rqs', 'cubic', 'affine'])
        hidden = [int(rng.integers(65, 201) if FAT else rng.integers(4, 65)) for _ in range(int(rng.integers(1, 3)))]
        d = {'dim': dim, 'hidden': hidden, 'mask': str(rng.choice(MASKS)), 'latent_dim': latent}
        if kind == 'affine':
            d['kind'] = 'coupling_affine'
        else:
            d.update(kind='coupling_rqs', n_bins=16 if '--k16' in sys.argv else int(rng.integers(1, 33 if FAT else 17)), lower=-3.0, upper=3.0,
                     spline_type='quadratic' if kind == 'rqs' else 'cubic')
        desc.append(d)
        if rng.random() < 0.3:
            desc.append({'kind': 'flip'})
    return desc, dim, latent, int(rng.integers(1, 700))


def case_mix(rng):
    """Every transform kind of the path in one flow: couplings beside element-wise affine / spline layers, dense linear
    layers (AffineLU, MatrixExponential), Permute and Flip -- the mixes decide which fused program (or tier) a flow runs on."""
    dim = int(rng.integers(2, 71))
    latent = int(rng.choice([0, 0, 3]))
    desc = []
    for _ in range(int(rng.integers(1, 6))):
        kind = str(rng.choice(['coupling_affine', 'coupling_affine', 'coupling_rqs', 'affine', 'affine_latent', 'rqs', 'affine_lu',
                               'matrix_exp', 'permute', 'flip', 'leaky_relu', 'cumsum', 'diff', 'identity']))
        # (bijections of the whole real line only: ELU^-1 is undefined below -1 and logit(sigmoid(x)) saturates in fp32 beyond
        #  |x| ~ 17 -- in the reference as much as here; those flows have their own domain-aware tests)
        if kind in ('cumsum', 'diff', 'identity'):
            desc.append({'kind': kind})
            continue
        if kind == 'leaky_relu':
            desc.append({'kind': kind, 'negative_slope': float(rng.choice([0.01, 0.2]))})
            continue
        hidden = [int(rng.integers(4, 65)) for _ in range(int(rng.integers(1, 3)))]
        if kind == 'affine_latent' and not latent:
            kind = 'affine'
        if kind == 'coupling_affine':
            d = {'dim': dim, 'hidden': hidden, 'mask': str(rng.choice(MASKS)), 'latent_dim': latent}
        elif kind == 'coupling_rqs':
            d = {'dim': dim, 'hidden': hidden, 'mask': str(rng.choice(MASKS)), 'latent_dim': latent,
                 'n_bins': int(rng.integers(1, 17)), 'lower': -3.0, 'upper': 3.0,
                 'spline_type': str(rng.choice(['quadratic', 'quadratic', 'cubic']))}
        elif kind == 'affine':
            d = {'dim': dim}
        elif kind == 'affine_latent':
            d = {'dim': dim, 'hidden': hidden, 'latent_dim': latent}
        elif kind == 'rqs':
            d = {'dim': dim, 'n_bins': int(rng.integers(1, 33 if FAT else 9)), 'lower': -3.0, 'upper': 3.0,
                 'hidden': hidden if latent else None, 'latent_dim': latent,
                 'spline_type': str(rng.choice(['quadratic', 'cubic']))}
        elif kind == 'matrix_exp':
            d = {'dim': dim, 'bias': bool(rng.integers(0, 2)), 'log_time': False}
        else:
            d = {'dim': dim}
        d['kind'] = kind
        desc.append(d)
    return desc, dim, latent, int(rng.integers(1, 700))


def case_time(rng):
    """Time-conditioned stacks: ContinuousAffineCoupling with every time net (coupling.py:98-213, net/time_net.py:6-91) inside
    NeuralFlow (flow.py:155-184), evaluated at per-row times t (and t0)."""
    dim = int(rng.integers(1, 41))
    latent = int(rng.choice([0, 0, 3]))
    desc = []
    for _ in range(int(rng.integers(1, 4))):
        desc.append({'kind': 'continuous_affine_coupling', 'dim': dim,
                     'hidden': [int(rng.integers(4, 65)) for _ in range(int(rng.integers(1, 3)))],
                     'mask': str(rng.choice(MASKS + ['none'])) if dim > 1 else 'none', 'latent_dim': latent,
                     'time_kind': str(rng.choice(['identity', 'linear', 'tanh', 'log', 'fourier', 'fourier_bounded'])),
                     'concatenate_time': bool(rng.integers(0, 2))})
    return desc, dim, latent, int(rng.integers(1, 500))


def _poison_empty():
    """--poison: torch.empty / empty_like hand out NaN-filled float tensors: an op that reads what it never wrote shows up as NaN
    instead of depending on what the caching allocator happens to return."""
    e0, el0 = torch.empty, torch.empty_like

    def fill(t):
        return t.fill_(float('nan')) if t.is_floating_point() and t.is_cuda else t
    torch.empty = lambda *a, **k: fill(e0(*a, **k))
    torch.empty_like = lambda *a, **k: fill(el0(*a, **k))


def main():
    if '--poison' in sys.argv:
        _poison_empty()
    fwd = '--forward' in sys.argv
    timed = '--time' in sys.argv
    infer = '--infer' in sys.argv
    mix = '--mix' in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n_cases = int(args[0]) if len(args) > 0 else 40
    seed = int(args[1]) if len(args) > 1 else 0
    rng = np.random.default_rng(seed)
    worst = 0.0
    only = [int(a.split('=')[1]) for a in sys.argv if a.startswith('--only=')]
    for i in range(n_cases):
        desc, dim, latent, n = (case_time if timed else case_mix if mix else case)(rng)
        if only and i not in only:
            continue
        if mix and n % 3 == 0 and '--sets' in sys.argv:
            # the second batch axis read as sets of 3 elements: couplings mask over the set axis (coupling.py:48-53)
            desc = [dict(d, set_data=True) if d['kind'].startswith('coupling_') else d for d in desc]
        for a in sys.argv:                         # --shift=k: k live 8-byte tensors ahead of the case (moves the small-block layout)
            if a.startswith('--shift='):
                _keep = [torch.zeros(2, dtype=torch.int32, device=DEV) for _ in range(int(a.split('=')[1]))]
        for a in sys.argv:                         # --reset=scratch|work|alloc|te|flags: drop one piece of process-wide state before each case
            if a.startswith('--reset='):
                from stribor_amd import _hip as _h
                what = a.split('=')[1]
                if what == 'scratch': _h._scratch.clear()
                if what == 'work': _h._work.clear()
                if what == 'flags': _h._flag_words.clear()
                if what == 'alloc': torch.cuda.empty_cache()
                if what == 'te':
                    from stribor_amd.flows import linear as _l
                    _l._TE_CACHE.clear()
        if '--poison' in sys.argv:
            # the caching allocator hands freed blocks to the next torch.empty: fill a few with NaN first, so that a kernel
            # reading memory it (or an earlier kernel of the step) never wrote shows up as NaN instead of depending on history
            junk = [torch.full((1 << k,), float('nan'), device=DEV) for k in range(10, 27)]
            del junk
            from stribor_amd import _hip as _h
            for t_ in list(_h._scratch.values()):          # ... and the library's persistent scratch (partials of earlier launches)
                (t_[0] if isinstance(t_, (tuple, list)) else t_).fill_(float('nan'))
        torch.manual_seed(seed * 1000 + i)
        flow = fd.build_flow(st, desc, dim)
        with torch.no_grad():
            for p in flow.parameters():
                p.add_(torch.randn_like(p) * 0.03)
        state = {k: v.clone() for k, v in flow.state_dict().items()}
        flow = flow.to(DEV)
        lead = (n,)
        if mix and n % 3 == 0:                 # a second batch axis (the product flattens leading axes itself)
            lead = (n // 3, 3)
        x = torch.randn(*lead, dim) * 1.4
        lat = torch.randn(*lead, latent) if latent else None
        if timed:
            nf = st.NeuralFlow([fd.build_transform(st, d) for d in desc])
            with torch.no_grad():
                for p in nf.parameters():
                    p.add_(torch.randn_like(p) * 0.05)
            state = {k: v.clone() for k, v in nf.state_dict().items()}
            nf = nf.to(DEV)
            spec = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
            tt, t0 = torch.rand(*lead, 1) * 2.0, torch.rand(*lead, 1)
            kw = {} if lat is None else {'latent': lat.to(DEV)}
            l64 = None if lat is None else lat.double()
            with torch.no_grad():
                y1 = nf(x.to(DEV), t=tt.to(DEV), **kw).cpu().double()
                y2 = nf(x.to(DEV), t=tt.to(DEV), t0=t0.to(DEV), **kw).cpu().double()
                f0 = nf.transforms[0]
                yf, lf = f0.forward_and_log_det_jacobian(x.to(DEV), tt.to(DEV), **kw)
                xb, lb = f0.inverse_and_log_det_jacobian(yf, tt.to(DEV), **kw)
            st.check_errors()
            w1 = orc.neural_flow_forward(spec, x.double(), tt.double(), None, l64)
            w2 = orc.neural_flow_forward(spec, x.double(), tt.double(), t0.double(), l64)
            wyf, wlf = orc.continuous_affine_coupling(spec[0], x.double(), tt.double(), l64, False)
            rel = lambda a, b: ((a - b).abs() / (1.0 + b.abs())).max().item()
            e = [rel(y1, w1), rel(y2, w2), rel(yf.cpu().double(), wyf), rel(lf.cpu().double(), wlf),
                 (xb.cpu().double() - x.double()).abs().max().item(), rel(-lb.cpu().double(), wlf)]
            worst = max(worst, max(e))
            kinds = [d['time_kind'] + ('+cat' if d['concatenate_time'] else '') + ':' + d['mask'][:8] for d in desc]
            print(f'case {i:3d} dim {dim:2d} lat {latent} n {n:3d} {kinds} y(t) {e[0]:.1e} y(t,t0) {e[1]:.1e} layer y {e[2]:.1e} ldj {e[3]:.1e} '
                  f'round trip {e[4]:.1e} inverse ldj {e[5]:.1e}' + ('  FAIL' if max(e) > 2e-4 else ''), flush=True)
            continue
        if infer and '--bf16' in sys.argv:   # bf16 storage: the product reads / writes bf16, computes in fp32; the oracle gets the rounded values
            x = x.bfloat16().float()
            # (bf16's grid contains the spline domain's bounds +-3: an input ON the bound has its inverse within an ulp of it, and
            #  whether the reference's forward log-det at that point is the tail's 0 is a coin flip of its own rounding -- fp64 too)
            x = torch.where(x.abs() == 3.0, x.sign() * 2.984375, x)          # the next bf16 value inside the domain
        if infer:         # no-graph paths (fused programs / tiers): log_prob, forward + log-det, inverse round trip vs fp64
            spec = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
            l64 = None if lat is None else lat.double()
            kw = {} if lat is None else {'latent': lat.to(DEV)}
            okw = {}
            xdev = x.to(DEV).bfloat16() if '--bf16' in sys.argv else x.to(DEV)
            with torch.no_grad():
                lp = flow.log_prob(xdev, **kw).cpu().double()
                y, ldj = flow.forward_and_log_det_jacobian(xdev, **kw)
                xr = flow.inverse(y, **kw).cpu().double()
            st.check_errors()
            want_lp = orc.flow_log_prob(spec, x.double(), l64, **okw)
            wy, wl = orc.flow_forward_and_ldj(spec, x.double(), l64, **okw)
            e1 = ((lp - want_lp).abs() / (1.0 + want_lp.abs())).max().item()
            e2 = ((y.cpu().double() - wy).abs() / (1.0 + wy.abs())).max().item()
            e3 = ((ldj.cpu().double() - wl).abs() / (1.0 + wl.abs())).max().item()
            # (relative to the largest transformed value: Cumsum / Diff stacks make |y| ~ 1e3 and the way back subtracts those)
            e4 = (xr - x.double()).abs().max().item() / (1.0 + 1e-2 * wy.abs().max().item())
            if '--bf16' in sys.argv:            # y / the round trip are stored in bf16 (8 mantissa bits)
                e2, e4 = e2 / 64.0, 0.0            # (the round trip goes through bf16-stored y: its error follows |y|, e.g. after Cumsum)
            m = max(e1, e2, e3)
            worst = max(worst, m)
            kinds = [d['kind'] + ('/' + d['spline_type'][0] if 'spline_type' in d else '') + (f":K{d['n_bins']}" if 'n_bins' in d else '') for d in desc]
            # the reference's closed-form cubic root in fp32 is itself 3e-3 .. 6e-3 off on round trips near a bin whose cubic
            # degenerates (the fp32 oracle shows the same numbers as the kernel on these cases): wider bound for cubic flows
            rt_tol = 1e-2 if any(d.get('spline_type') == 'cubic' for d in desc) else 2e-3
            print(f'case {i:3d} dim {dim:2d} lat {latent} n {n:3d} {kinds} log_prob {e1:.1e} y {e2:.1e} ldj {e3:.1e} round trip {e4:.1e}' + ('  FAIL' if m > 2e-4 or e4 > rt_tol else ''), flush=True)
            continue
        tol = 1e-3

        def compare(x, lat):
            leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items()}
            xin = x.double().clone().requires_grad_(True)
            nn_ = x.shape[0]
            if fwd:
                y64, l64 = orc.flow_forward_and_ldj(fd.flow_spec(desc, leaves), xin, None if lat is None else lat.double())
                want = ((y64 ** 2).sum() * 0.1 + l64.sum()) / nn_
            else:
                want = -orc.flow_log_prob(fd.flow_spec(desc, leaves), xin, None if lat is None else lat.double()).mean()
            want.backward()
            for p in flow.parameters():
                p.grad = None
            xg = x.to(DEV).requires_grad_(True)
            if fwd:
                yg, lg = flow.forward_and_log_det_jacobian(xg, latent=None if lat is None else lat.to(DEV))
                loss = ((yg ** 2).sum() * 0.1 + lg.sum()) / nn_
            else:
                loss = -flow.log_prob(xg, latent=None if lat is None else lat.to(DEV)).mean()
            loss.backward()
            torch.cuda.synchronize()
            st.check_errors()
            errs = {'loss': abs(loss.item() - want.item()) / (abs(want.item()) + 1e-9)}
            ref = xin.grad.float()
            errs['x'] = ((xg.grad.cpu() - ref).abs().max() / (ref.abs().max() + 1e-12)).item()
            row_err = (xg.grad.cpu() - ref).abs().flatten(1).max(1).values / (ref.abs().max() + 1e-12)
            for name, p in flow.named_parameters():
                if p.numel() == 0:
                    continue
                ref = leaves[name].grad.float()
                errs[name] = ((p.grad.cpu() - ref).abs().max() / (ref.abs().max() + 1e-12)).item()
            return errs, row_err

        def reference_gradient_jumps(xrow, latrow):
            """Is the fp64 reference's OWN gradient discontinuous within fp32 rounding of this row?  (Splines are C1: the second
            derivative -- the log-det's gradient -- jumps at every knot; an input within an ulp of a knot has no fp32 answer.)"""
            spec = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
            l1 = None if latrow is None else latrow.double()[None]

            def g(xr):
                xr = xr.clone().requires_grad_(True)
                if fwd:
                    y64, l64 = orc.flow_forward_and_ldj(spec, xr[None], l1)
                    ((y64 ** 2).sum() * 0.1 + l64.sum()).backward()
                else:
                    (-orc.flow_log_prob(spec, xr[None], l1)).sum().backward()
                return xr.grad
            x0 = xrow.double()
            g0 = g(x0).abs().max().item()
            worst_jump = 0.0
            for j in range(x0.numel()):
                e = torch.zeros_like(x0)
                e.view(-1)[j] = 1e-6 * max(1.0, abs(x0.view(-1)[j].item()))
                worst_jump = max(worst_jump, (g(x0 + e) - g(x0 - e)).abs().max().item() / (g0 + 1e-12))
            return worst_jump

        errs, row_err = compare(x, lat)
        note = ''
        if errs['x'] > tol and x.dim() == 3 and '--sets' not in sys.argv:
            # a second batch axis without set semantics: rows are independent, classify on the flattened rows
            x, lat = x.reshape(-1, x.shape[-1]), (None if lat is None else lat.reshape(-1, lat.shape[-1]))
            errs, row_err = compare(x, lat)
        if errs['x'] > tol and x.dim() == 2:
            suspects = torch.nonzero(row_err > tol).flatten().tolist()
            if 0 < len(suspects) <= 4:
                jumps = {r: reference_gradient_jumps(x[r], None if lat is None else lat[r]) for r in suspects}
                # (a jump of size J explains an error of up to J on that row)
                if all(v > 0.5 * row_err[r].item() for r, v in jumps.items()):
                    keep = torch.ones(x.shape[0], dtype=torch.bool)
                    keep[suspects] = False
                    errs, row_err = compare(x[keep], None if lat is None else lat[keep])
                    note = '  [rows ' + ', '.join(f'{r} (reference gradient jumps {v:.1e} of its size within 1e-6 of x)' for r, v in jumps.items()) + ' set aside]'
                else:
                    note = '  [rows ' + ', '.join(f'{r}: err {row_err[r].item():.1e}, reference jump {v:.1e}' for r, v in jumps.items()) + ']'
            else:
                note = f'  [{len(suspects)} rows over tolerance]'
        bad = {k: v for k, v in errs.items() if not (v <= (1e-4 if k == 'loss' else tol))}
        m = max(v for k, v in errs.items() if k != 'loss')
        worst = max(worst, m)
        kinds = [d['kind'] + ('/' + d['spline_type'][0] if 'spline_type' in d else '') + (f":K{d['n_bins']}" if 'n_bins' in d else '') + (f":H{d['hidden']}" if 'hidden' in d else '') for d in desc]
        print(f'case {i:3d} dim {dim:2d} lat {latent} n {n:3d} {kinds}  max grad err {m:.2e}' + (f'  FAIL {bad}' if bad else '') + note, flush=True)
        if '--show-work' in sys.argv:
            from stribor_amd import _hip as _h
            torch.cuda.synchronize()
            print('   work counters after the case:', {k: v.tolist() for k, v in _h._work.items()}, flush=True)
    print('worst', worst)


if __name__ == '__main__':
    main()
