#!/usr/bin/env python3
"""Capture golden vectors from the UNMODIFIED reference (mbilos/stribor @ /root/reference).

Runs in the build container only (the reference never travels to the GPU box); its outputs,
``tests/golden/*.npz``, are committed.  Recipe = SURVEY.md Appendix C: two annotation-only
stub modules (``torchtyping``, ``torchdiffeq``; neither touches hot-path arithmetic) are put on
``sys.path`` ahead of ``/root/reference`` so that ``import stribor`` succeeds with the
reference files untouched.  Bytecode writing is disabled so nothing lands in the reference tree.

    python tests/golden/make_golden.py            # rewrites every fixture

Fixtures:  F1 doc known-answer, F2 masks, F3 cfg 1, F4 cfg 2 (N=256, + bf16
rounded inputs), F5 cfg 3 (RQ-spline couplings, N=128, incl. tails / on-bound / on-knot rows),
F6 cfg 4 (AffineLU + MatrixExponential + couplings), F7 Permute/Flip, F8 the reference test-suite
shapes with autograd log|det J|, F9 cubic splines (suite shapes + a D=64 coupling flow), F10 parameter-free element-wise flows + the on-path part of
test_normalizing_flow.py's stack, F11 ContinuousAffineCoupling / NeuralFlow, F12 Coupling(set_data=True), hand-written
conditioners and widths beyond the fused kernel's tiles (round 2), F13 cfg 3 and cfg 4 at their FULL depth (8 spline couplings at N=64,
16 layers at N=256; round 5: pins the oracle to the reference at the depth the benchmark runs).
"""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/ for flowdesc
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))          # repo root: flowdesc lives in stribor_amd.util

REF = '/root/reference'


def import_reference():
    stub = tempfile.mkdtemp(prefix='stribor_stubs_')
    os.makedirs(os.path.join(stub, 'torchtyping'))
    os.makedirs(os.path.join(stub, 'torchdiffeq'))
    with open(os.path.join(stub, 'torchtyping', '__init__.py'), 'w') as f:
        f.write('class TensorType:\n    def __class_getitem__(cls, item):\n        return cls\n')
    with open(os.path.join(stub, 'torchdiffeq', '__init__.py'), 'w') as f:
        f.write('def odeint(*a, **k):\n    raise NotImplementedError\nodeint_adjoint = odeint\n')
    sys.path.insert(0, REF)
    sys.path.insert(0, stub)
    import stribor  # noqa
    assert stribor.__file__.startswith(REF), stribor.__file__
    return stribor


import numpy as np
import torch

import flowdesc as fd

st = import_reference()
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def full_state(flow, desc):
    """state_dict + the plain-attribute tensors the reference forgets to register (quirks Q6/Q7)."""
    s = {k: v.clone() for k, v in flow.state_dict().items()}
    for i, (d, f) in enumerate(zip(desc, flow.transforms)):
        if d['kind'] == 'permute':
            s[f'transforms.{i}.permutation'] = f.permutation.clone()
    return s


def save(name, arrays, meta):
    arrays = {k: npy(v) for k, v in arrays.items()}
    arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + '.npz')
    np.savez(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays')


def trace_flow(flow, y, **kw):
    """Per-layer (x_out, ldj) of inverse_and_log_det_jacobian in the order the flow visits them."""
    out = {}
    with torch.no_grad():
        cur = y
        for i, f in reversed(list(enumerate(flow.transforms))):
            cur, ldj = f.inverse_and_log_det_jacobian(cur, **kw)
            out[f'inv_x.{i}'] = cur
            out[f'inv_ldj.{i}'] = ldj
    return out


def tensor_sha(t):
    import hashlib
    a = np.ascontiguousarray(npy(t))
    return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()


def flow_case(prefix, desc, dim, seed, x, arrays, extra_inputs=None, state_hashes=None, trace_every=1, **kw):
    """state_hashes: a dict that receives {key: sha256} of every state tensor INSTEAD of the tensors (large flows: the weights are
    the reference's default init under `seed`, which the host classes reproduce draw for draw -- the test rebuilds them and holds
    every tensor to the hash captured here, so the fixture stays small and the state is still the reference's, bit for bit)."""
    torch.manual_seed(seed)
    flow = fd.build_flow(st, desc, dim)
    state = full_state(flow, desc)
    for k, v in state.items():
        if state_hashes is None:
            arrays[f'{prefix}/state/{k}'] = v
        else:
            state_hashes[k] = tensor_sha(v)
    arrays[f'{prefix}/x'] = x
    with torch.no_grad():
        arrays[f'{prefix}/log_prob'] = flow.log_prob(x, **kw)
        for k, v in trace_flow(flow, x, **kw).items():
            if int(k.rsplit('.', 1)[1]) % trace_every == 0 or k.startswith('inv_ldj'):
                arrays[f'{prefix}/{k}'] = v
        z, ldj_inv = flow.inverse_and_log_det_jacobian(x, **kw)
        arrays[f'{prefix}/inverse'] = z
        arrays[f'{prefix}/inverse_ldj'] = ldj_inv
        yf, ldj_f = flow.forward_and_log_det_jacobian(x, **kw)
        arrays[f'{prefix}/forward'] = yf
        arrays[f'{prefix}/forward_ldj'] = ldj_f
        # fp64 "truth" of the same flow (reference cast to double)
        f64 = fd.build_flow(st, desc, dim).double()
        f64.load_state_dict({k: v.double() for k, v in flow.state_dict().items()})
        for i, (d, f) in enumerate(zip(desc, flow.transforms)):
            if d['kind'] == 'permute':
                f64.transforms[i].permutation = f.permutation
                f64.transforms[i].inverse_permutation = f.inverse_permutation
        try:
            f64.base_dist = st.Normal(torch.zeros(dim).double(), torch.ones(dim).double())
            for m in f64.transforms:
                if hasattr(m, 'diag_ones'):
                    m.diag_ones = m.diag_ones.double()
            kw64 = {k: (v.double() if torch.is_tensor(v) else v) for k, v in kw.items()}
            arrays[f'{prefix}/log_prob_f64'] = f64.log_prob(x.double(), **kw64)
        except Exception as e:  # CPU-pinned fp32 constants in the reference (quirk Q6)
            print(f'  [{prefix}] fp64 run skipped: {type(e).__name__}: {e}')
        if extra_inputs:
            for nm, xe in extra_inputs.items():
                arrays[f'{prefix}/{nm}/x'] = xe
                arrays[f'{prefix}/{nm}/log_prob'] = flow.log_prob(xe, **kw)
    return flow


# ------------------------------------------------------------------------------------------ F1
def f1_doc_example():
    """stribor/test/test_normalizing_flow.py:45-55 (the suite's only known-answer test)."""
    torch.manual_seed(123)
    dim = 2
    f = st.NormalizingFlow(st.UnitNormal(dim), [st.Affine(dim)])
    x = torch.randn(3, 2)
    with torch.no_grad():
        lp = f.log_prob(x)
        rng = torch.get_rng_state()
        z = f.base_dist.sample((1,))
        torch.set_rng_state(rng)
        s = f.sample(1)
        assert torch.equal(s, f.forward(z))
    assert torch.allclose(lp, torch.Tensor([[-1.7560], [-1.7434], [-2.1792]]), atol=1e-4)
    assert torch.allclose(s, torch.Tensor([[-0.5204, 0.4196]]), atol=1e-4)
    arrays = {'x': x, 'log_prob': lp, 'base_sample': z, 'sample': s}
    for k, v in f.state_dict().items():
        arrays['state/' + k] = v
    save('f1_doc_example', arrays, {'desc': [{'kind': 'affine', 'dim': 2}], 'dim': 2, 'seed': 123})


# ------------------------------------------------------------------------------------------ F2
MASK_NAMES = ['none', 'ordered_right_half', 'ordered_0', 'ordered_left_half', 'ordered_1',
              'parity_even', 'parity_odd']


def f2_masks():
    arrays = {}
    for name in MASK_NAMES:
        gen = st.util.get_mask(name)
        for d in (1, 2, 3, 5, 8, 10, 64, 127, 128):
            arrays[f'{name}/{d}'] = gen(d)
    save('f2_masks', arrays, {'names': MASK_NAMES})


# ------------------------------------------------------------------------------------------ F3
def f3_cfg1():
    desc = [{'kind': 'coupling_affine', 'dim': 2, 'hidden': [64], 'mask': 'ordered_right_half', 'latent_dim': 0}]
    arrays = {}
    torch.manual_seed(1000)
    x = torch.randn(1024, 2)
    flow_case('cfg1', desc, 2, 0, x, arrays)
    save('f3_cfg1', arrays, {'cfg1': {'desc': desc, 'dim': 2, 'seed': 0}})


# ------------------------------------------------------------------------------------------ F4
def f4_cfg2():
    desc = fd.cfg2_desc()
    arrays = {}
    torch.manual_seed(1001)
    x = torch.randn(256, 64)
    xb = x.bfloat16().float()               # bf16 storage variant: oracle sees rounded x
    x_wide = torch.randn(64, 64) * 4.0      # heavier tails
    flow_case('cfg2', desc, 64, 0, x, arrays, extra_inputs={'bf16': xb, 'wide': x_wide})
    save('f4_cfg2', arrays, {'cfg2': {'desc': desc, 'dim': 64, 'seed': 0}})


# ------------------------------------------------------------------------------------------ F5
def f5_cfg3():
    """RQ-spline couplings.  2 layers at the full cfg-3 widths (D=64, K=16, H=64) keep the file small;
    N=128 keeps the reference's O(M^2) domain check (quirk Q1) affordable."""
    desc = fd.cfg3_desc(n_layers=2)
    arrays = {}
    torch.manual_seed(1002)
    x = torch.randn(128, 64)
    x[0, :] = 3.0            # exactly on upper
    x[1, :] = -3.0           # exactly on lower
    x[2, :] = 3.5            # all in the upper tail
    x[3, :] = -7.0           # all in the lower tail
    x[4, ::2] = 3.0000002    # just outside
    x[5, :] = 0.0
    x[6, :] = torch.linspace(-3, 3, 64)
    x[7, :] = torch.linspace(-2.999999, 2.999999, 64)
    flow_case('cfg3', desc, 64, 0, x, arrays)
    save('f5_cfg3', arrays, {'cfg3': {'desc': desc, 'dim': 64, 'seed': 0}})


# ------------------------------------------------------------------------------------------ F6
def f6_cfg4():
    desc = fd.cfg4_desc(n_blocks=2)
    arrays = {}
    torch.manual_seed(1003)
    x = torch.randn(256, 128)
    flow_case('cfg4', desc, 128, 0, x, arrays)
    meta = {'cfg4': {'desc': desc, 'dim': 128, 'seed': 0}}
    # MatrixExponential alone: scalar t, per-row t, bias, log_time
    for bias in (False, True):
        for log_time in (False, True):
            name = f'matexp_b{int(bias)}_l{int(log_time)}'
            d = [{'kind': 'matrix_exp', 'dim': 16, 'bias': bias, 'log_time': log_time}]
            torch.manual_seed(7)
            f = fd.build_flow(st, d, 16)
            xm = torch.randn(32, 16)
            t = torch.randn(32, 1)
            with torch.no_grad():
                for k, v in f.state_dict().items():
                    arrays[f'{name}/state/{k}'] = v
                arrays[f'{name}/x'] = xm
                arrays[f'{name}/t'] = t
                m = f.transforms[0]
                arrays[f'{name}/fwd_t'] = m(xm, t=t)
                arrays[f'{name}/inv_t'] = m.inverse(xm, t=t)
                arrays[f'{name}/ldj_t'] = m.log_det_jacobian(xm, None, t=t)
                arrays[f'{name}/fwd_s'] = m(xm, t=0.7)
                arrays[f'{name}/inv_s'] = m.inverse(xm, t=0.7)
                arrays[f'{name}/ldj_s'] = m.log_det_jacobian(xm, None, t=0.7)
                arrays[f'{name}/fwd_default'] = m(xm)
            meta[name] = {'desc': d, 'dim': 16, 'seed': 7}
    save('f6_cfg4', arrays, meta)


# ------------------------------------------------------------------------------------------ F7
def f7_permute():
    arrays = {}
    torch.manual_seed(123)
    p = st.Permute(64)
    x = torch.randn(33, 64)
    arrays['perm64/permutation'] = p.permutation
    arrays['perm64/inverse_permutation'] = p.inverse_permutation
    arrays['perm64/x'] = x
    arrays['perm64/fwd'] = p(x)
    arrays['perm64/inv'] = p.inverse(x)
    xb = x.bfloat16()
    arrays['perm64/x_bf16_bits'] = xb.view(torch.int16)
    arrays['perm64/fwd_bf16_bits'] = p(xb).view(torch.int16)
    fl = st.Flip([-1])
    arrays['flip/x'] = x
    arrays['flip/fwd'] = fl(x)
    arrays['flip/inv'] = fl.inverse(x)
    # a flow mixing permutations with couplings
    desc = [fd.cfg2_desc(1, 10, 13)[0], {'kind': 'permute', 'dim': 10},
            {'kind': 'coupling_affine', 'dim': 10, 'hidden': [13], 'mask': 'parity_even', 'latent_dim': 0},
            {'kind': 'flip'},
            {'kind': 'coupling_affine', 'dim': 10, 'hidden': [13], 'mask': 'ordered_left_half', 'latent_dim': 0}]
    torch.manual_seed(1004)
    xm = torch.randn(50, 10)
    flow_case('mixed', desc, 10, 5, xm, arrays)
    save('f7_permute', arrays, {'mixed': {'desc': desc, 'dim': 10, 'seed': 5}})


# ------------------------------------------------------------------------------------------ F8
SHAPES = [(1, 1), (2, 10), (10, 2), (7, 4, 5)]          # test_coupling.py:7 and friends


def autograd_logdet(f, x, **kw):
    """stribor/test/base.py:24-44: log|det| of the autograd Jacobian of f and of f.inverse."""
    from stribor.test.base import _get_full_jacobian
    xf, kwf, jac, jac_inv = _get_full_jacobian(f, x, **kw)
    return torch.det(jac).abs().log(), torch.det(jac_inv).abs().log(), torch.diagonal(jac, dim1=-2, dim2=-1)


def suite_case(prefix, desc_one, dim, x, arrays, meta, latent=None, t=None):
    torch.manual_seed(123)
    f = fd.build_transform(st, desc_one)
    kw = {}
    if latent is not None:
        kw['latent'] = latent
    if t is not None:
        kw['t'] = t
    for k, v in f.state_dict().items():
        arrays[f'{prefix}/state/transforms.0.{k}'] = v
    if desc_one['kind'] == 'permute':
        arrays[f'{prefix}/state/transforms.0.permutation'] = f.permutation
    arrays[f'{prefix}/x'] = x
    if latent is not None:
        arrays[f'{prefix}/latent'] = latent
    if t is not None:
        arrays[f'{prefix}/t'] = t
    with torch.no_grad():
        y = f(x, **kw)
        arrays[f'{prefix}/y'] = y
        arrays[f'{prefix}/x_back'] = f.inverse(y, **kw)
        arrays[f'{prefix}/ldj'] = f.log_det_jacobian(x, y, **kw)
        _, l1 = f.forward_and_log_det_jacobian(x, **kw)
        _, l2 = f.inverse_and_log_det_jacobian(y, **kw)
        arrays[f'{prefix}/ldj_fwd'] = l1
        arrays[f'{prefix}/ldj_inv'] = l2
        if hasattr(f, 'log_diag_jacobian') and desc_one['kind'] not in ('permute', 'flip'):
            arrays[f'{prefix}/ldiag'] = f.log_diag_jacobian(x, y, **kw)
    ld, ld_inv, jd = autograd_logdet(f, x, **kw)
    arrays[f'{prefix}/autograd_logdet'] = ld
    arrays[f'{prefix}/autograd_logdet_inv'] = ld_inv
    meta[prefix] = {'desc': [desc_one], 'dim': dim}


def f8_suite():
    arrays, meta = {}, {}
    for shp in SHAPES:
        dim = shp[-1]
        tag = 'x'.join(map(str, shp))
        # test_coupling.py:7-26 (affine coupling, ordered_left_half, hidden [13], latent 0/1/13)
        for ld in (0, 1, 13):
            torch.manual_seed(123)
            x = torch.randn(*shp)
            latent = torch.randn(*shp[:-1], ld) if ld else None
            d = {'kind': 'coupling_affine', 'dim': dim, 'hidden': [13], 'mask': 'ordered_left_half', 'latent_dim': ld}
            suite_case(f'coupling_affine/{tag}/l{ld}', d, dim, x, arrays, meta, latent=latent)
        # RQ-spline coupling (same protocol; 8(a) a9-a11 through Coupling)
        for K in (1, 3, 10):
            torch.manual_seed(123)
            x = torch.rand(*shp) * 2
            d = {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [12], 'mask': 'ordered_right_half', 'latent_dim': 0,
                 'n_bins': K, 'lower': 0, 'upper': 2}
            suite_case(f'coupling_rqs/{tag}/k{K}', d, dim, x, arrays, meta)
        # test_spline.py:8-33 quadratic rows
        for K in (1, 3, 10):
            for ld in (0, 1, 13):
                np.random.seed(123)
                torch.manual_seed(123)
                x = torch.rand(*shp) * 2
                latent = torch.randn(*shp[:-1], ld) if ld else None
                d = {'kind': 'rqs', 'dim': dim, 'n_bins': K, 'lower': 0, 'upper': 2, 'hidden': [12], 'latent_dim': ld}
                suite_case(f'rqs/{tag}/k{K}/l{ld}', d, dim, x, arrays, meta, latent=latent)
        # test_affine.py:27-40 latent affine
        for ld in (1, 13):
            torch.manual_seed(123)
            x = torch.randn(*shp)
            latent = torch.randn(*shp[:-1], ld)
            d = {'kind': 'affine_latent', 'dim': dim, 'hidden': [32], 'latent_dim': ld}
            suite_case(f'affine_latent/{tag}/l{ld}', d, dim, x, arrays, meta, latent=latent)
        # test_affine.py:44-55
        torch.manual_seed(123)
        x = torch.randn(*shp)
        suite_case(f'affine_lu/{tag}', {'kind': 'affine_lu', 'dim': dim}, dim, x, arrays, meta)
        # test_affine.py:58-80
        for bias in (True, False):
            for log_time in (True, False):
                torch.manual_seed(123)
                x = torch.randn(*shp)
                t = torch.randn(*shp[:-1], 1)
                d = {'kind': 'matrix_exp', 'dim': dim, 'bias': bias, 'log_time': log_time}
                suite_case(f'matrix_exp/{tag}/b{int(bias)}l{int(log_time)}/tvec', d, dim, x, arrays, meta, t=t)
                suite_case(f'matrix_exp/{tag}/b{int(bias)}l{int(log_time)}/tdef', d, dim, x, arrays, meta)
        # test_permute.py
        torch.manual_seed(123)
        x = torch.randn(*shp)
        suite_case(f'permute/{tag}', {'kind': 'permute', 'dim': dim}, dim, x, arrays, meta)
        suite_case(f'flip/{tag}', {'kind': 'flip'}, dim, x, arrays, meta)
    save('f8_suite', arrays, meta)


# ------------------------------------------------------------------------------------------ F9
def f9_cubic():
    """Cubic splines (the reference's default spline_type): test_spline.py:8-33 cubic rows, cubic couplings on the
    suite shapes, and a two-layer D=64, K=16 coupling flow with tail / on-bound rows (as F5)."""
    arrays, meta = {}, {}
    for shp in SHAPES:
        dim = shp[-1]
        tag = 'x'.join(map(str, shp))
        for K in (1, 3, 10):
            for ld in (0, 1, 13):
                np.random.seed(123)
                torch.manual_seed(123)
                x = torch.rand(*shp) * 2
                latent = torch.randn(*shp[:-1], ld) if ld else None
                d = {'kind': 'rqs', 'dim': dim, 'n_bins': K, 'lower': 0, 'upper': 2, 'hidden': [12], 'latent_dim': ld,
                     'spline_type': 'cubic'}
                suite_case(f'cubic/{tag}/k{K}/l{ld}', d, dim, x, arrays, meta, latent=latent)
        for K in (3, 10):
            torch.manual_seed(123)
            x = torch.rand(*shp) * 2.4 - 0.2          # some elements in the linear tails
            d = {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [12], 'mask': 'ordered_right_half', 'latent_dim': 0,
                 'n_bins': K, 'lower': 0, 'upper': 2, 'spline_type': 'cubic'}
            suite_case(f'coupling_cubic/{tag}/k{K}', d, dim, x, arrays, meta)
    desc = [dict(d, spline_type='cubic') for d in fd.cfg3_desc(n_layers=2)]
    torch.manual_seed(1009)
    x = torch.randn(128, 64)
    x[0, :] = 3.0
    x[1, :] = -3.0
    x[2, :] = 3.5
    x[3, :] = -7.0
    x[4, ::2] = 3.0000002
    x[5, :] = 0.0
    x[6, :] = torch.linspace(-3, 3, 64)
    x[7, :] = torch.linspace(-2.999999, 2.999999, 64)
    flow_case('cubic_flow', desc, 64, 0, x, arrays)
    meta['cubic_flow'] = {'desc': desc, 'dim': 64, 'seed': 0}
    save('f9_cubic', arrays, meta)


# ------------------------------------------------------------------------------------------ F10
def f10_pointwise():
    """Parameter-free element-wise flows on the suite shapes (test_sigmoid.py, test_activations.py, test_cumsum.py)
    and the on-path part of test_normalizing_flow.py's stack: Coupling(Affine) -> Flip -> Sigmoid ->
    Coupling(cubic Spline) -> Logit."""
    arrays, meta = {}, {}
    for shp in SHAPES + [(3, 4, 5), (2, 3, 4, 5)]:
        dim = shp[-1]
        tag = 'x'.join(map(str, shp))
        for kind in ('sigmoid', 'logit', 'elu', 'leaky_relu', 'cumsum', 'diff', 'identity'):
            np.random.seed(123)
            torch.manual_seed(123)
            x = torch.rand(*shp) * 0.5 + 0.25 if kind == 'logit' else torch.randn(*shp)        # test_sigmoid.py:17
            if kind == 'sigmoid' and x.numel() > 4:
                x.view(-1)[0], x.view(-1)[1] = 30.0, -120.0                  # saturating values (clamps)
            if kind == 'logit' and x.numel() > 4:
                x.view(-1)[0], x.view(-1)[1] = 0.0, 1.0                      # clamped ends
            d = {'kind': kind}
            if kind == 'leaky_relu':
                d['negative_slope'] = 0.01 if len(shp) == 2 else 0.3
            suite_case(f'{kind}/{tag}', d, dim, x, arrays, meta)
    dim, K = 2, 5
    desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [32, 64], 'mask': 'ordered_1', 'latent_dim': 0},
            {'kind': 'flip'}, {'kind': 'sigmoid'},
            {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [32, 64], 'mask': 'ordered_0', 'latent_dim': 0, 'n_bins': K,
             'lower': 0, 'upper': 1, 'spline_type': 'cubic'},
            {'kind': 'logit'}]
    torch.manual_seed(77)
    x = torch.randn(3, 4, dim)
    flow_case('stack', desc, dim, 5, x, arrays)
    meta['stack'] = {'desc': desc, 'dim': dim, 'seed': 5}
    save('f10_pointwise', arrays, meta)


# ------------------------------------------------------------------------------------------ F11
def f11_continuous():
    """ContinuousAffineCoupling (test_coupling.py:29-52: ordered_left_half, hidden [13], latent 0/1/13, TimeLinear) plus
    the other time nets, and the on-path layer of test_neural_flow.py through NeuralFlow (t, and t = t0 round trip)."""
    arrays, meta = {}, {}
    for shp in SHAPES:
        dim = shp[-1]
        tag = 'x'.join(map(str, shp))
        for ld in (0, 1, 13):
            for tk in (('linear',) if ld else ('linear', 'identity', 'tanh', 'log')):
                torch.manual_seed(123)
                x = torch.randn(*shp)
                latent = torch.randn(*shp[:-1], ld) if ld else None
                t = torch.randn_like(x[..., :1]) if tk != 'log' else torch.rand_like(x[..., :1])
                d = {'kind': 'continuous_affine_coupling', 'dim': dim, 'hidden': [13], 'mask': 'ordered_left_half',
                     'latent_dim': ld, 'time_kind': tk}
                case = f'cac/{tag}/l{ld}/{tk}'
                torch.manual_seed(321)
                f = fd.build_transform(st, d)
                for k, v in f.state_dict().items():
                    arrays[f'{case}/state/transforms.0.{k}'] = v
                kw = {} if latent is None else {'latent': latent}
                with torch.no_grad():
                    y, ldj = f.forward_and_log_det_jacobian(x, t, **kw)
                    xb, ldj_i = f.inverse_and_log_det_jacobian(y, t, **kw)
                arrays[f'{case}/x'], arrays[f'{case}/t'], arrays[f'{case}/y'], arrays[f'{case}/ldj'] = x, t, y, ldj
                arrays[f'{case}/x_back'], arrays[f'{case}/ldj_inv'] = xb, ldj_i
                if latent is not None:
                    arrays[f'{case}/latent'] = latent
                meta[case] = {'desc': [d], 'dim': dim}
    # test_neural_flow.py:9-16, the ContinuousAffineCoupling layer (concatenate_time=False, TimeLinear(dim)), stacked twice
    dim = 2
    desc = [{'kind': 'continuous_affine_coupling', 'dim': dim, 'hidden': [32], 'mask': m, 'latent_dim': 0,
             'time_kind': 'linear', 'time_out': dim, 'concatenate_time': False} for m in ('ordered_0', 'ordered_1')]
    torch.manual_seed(123)
    nf = st.NeuralFlow([fd.build_transform(st, d) for d in desc])
    for k, v in nf.state_dict().items():
        arrays[f'neural_flow/state/{k}'] = v
    x = torch.randn(10, 4, 2)
    t, t0 = torch.randn_like(x[..., :1]), torch.randn_like(x[..., :1])
    with torch.no_grad():
        arrays['neural_flow/x'], arrays['neural_flow/t'], arrays['neural_flow/t0'] = x, t, t0
        arrays['neural_flow/y_t'] = nf(x, t=t)
        arrays['neural_flow/y_t_t0'] = nf(x, t=t, t0=t0)
        arrays['neural_flow/y_zero'] = nf(x, t=torch.zeros_like(t))
    meta['neural_flow'] = {'desc': desc, 'dim': dim}
    save('f11_continuous', arrays, meta)


# ----------------------------------------------------------------------------------------- F12
def f12_wide_and_set():
    """Round 2: the branches of Coupling the fused kernel does not cover --
    set_data=True (coupling.py:48-51: mask over the set axis, test-suite shape (7,4,5) and a 2-D (4,5) set),
    conditioners that are not a stribor MLP, widths beyond the fused kernel's tiles (D=200 / H=256), n_bins > 16."""
    arrays, meta = {}, {}
    # set_data: affine, quadratic and cubic spline couplings; with and without latent
    for shp in [(7, 4, 5), (4, 5), (3, 2, 6, 1)]:
        dim = shp[-1]
        tag = 'x'.join(map(str, shp))
        for mask in ('ordered_left_half', 'parity_even'):
            for ld in (0, 3):
                torch.manual_seed(321)
                x = torch.randn(*shp)
                latent = torch.randn(*shp[:-1], ld) if ld else None
                d = {'kind': 'coupling_affine', 'dim': dim, 'hidden': [13], 'mask': mask, 'latent_dim': ld, 'set_data': True}
                suite_case(f'set_affine/{tag}/{mask}/l{ld}', d, dim, x, arrays, meta, latent=latent)
            for stype, K in (('quadratic', 5), ('cubic', 4)):
                torch.manual_seed(321)
                x = torch.rand(*shp) * 2
                d = {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [12], 'mask': mask, 'latent_dim': 0, 'n_bins': K,
                     'lower': 0, 'upper': 2, 'spline_type': stype, 'set_data': True}
                suite_case(f'set_spline/{tag}/{mask}/{stype}', d, dim, x, arrays, meta)
    # hand-written conditioners (not a stribor MLP)
    for shp in [(10, 6), (7, 4, 5)]:
        dim = shp[-1]
        tag = 'x'.join(map(str, shp))
        for ld in (0, 2):
            torch.manual_seed(322)
            x = torch.randn(*shp)
            latent = torch.randn(*shp[:-1], ld) if ld else None
            d = {'kind': 'coupling_affine', 'dim': dim, 'hidden': [24], 'mask': 'ordered_right_half', 'latent_dim': ld, 'net': 'hand'}
            suite_case(f'hand_affine/{tag}/l{ld}', d, dim, x, arrays, meta, latent=latent)
        torch.manual_seed(322)
        x = torch.rand(*shp) * 2
        d = {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [24], 'mask': 'parity_odd', 'latent_dim': 0, 'n_bins': 6,
             'lower': 0, 'upper': 2, 'net': 'hand'}
        suite_case(f'hand_rqs/{tag}', d, dim, x, arrays, meta)
    # ContinuousAffineCoupling with time nets that have no in-kernel form, and a hand-written conditioner
    for shp in [(10, 4), (7, 4, 5)]:
        dim = shp[-1]
        tag = 'x'.join(map(str, shp))
        for tk, net in (('fourier', 'mlp'), ('fourier_bounded', 'mlp'), ('tanh', 'hand'), ('fourier', 'hand')):
            torch.manual_seed(323)
            x = torch.randn(*shp)
            t = torch.rand(*shp[:-1], 1) * 2
            d = {'kind': 'continuous_affine_coupling', 'dim': dim, 'hidden': [13], 'mask': 'ordered_left_half', 'latent_dim': 0,
                 'time_kind': tk, 'time_hidden': 5, 'net': net}
            torch.manual_seed(123)
            f = fd.build_transform(st, d)
            prefix = f'cac/{tag}/{tk}_{net}'
            for k, v in f.state_dict().items():
                arrays[f'{prefix}/state/transforms.0.{k}'] = v
            arrays[f'{prefix}/x'], arrays[f'{prefix}/t'] = x, t
            with torch.no_grad():
                y, ldj = f.forward_and_log_det_jacobian(x, t)
                xb, ldj_inv = f.inverse_and_log_det_jacobian(y, t)
            arrays[f'{prefix}/y'], arrays[f'{prefix}/ldj'] = y, ldj
            arrays[f'{prefix}/x_back'], arrays[f'{prefix}/ldj_inv'] = xb, ldj_inv
            meta[prefix] = {'desc': [d], 'dim': dim}
    save('f12_set_and_hand', arrays, meta)
    # wide flows (kept in their own file: the weights dominate)
    arrays, meta = {}, {}
    desc = [{'kind': 'coupling_affine', 'dim': 200, 'hidden': [256], 'mask': 'ordered_right_half', 'latent_dim': 0},
            {'kind': 'coupling_affine', 'dim': 200, 'hidden': [256], 'mask': 'parity_even', 'latent_dim': 0}]
    torch.manual_seed(1200)
    x = torch.randn(48, 200)
    flow_case('wide_affine', desc, 200, 12, x, arrays)
    meta['wide_affine'] = {'desc': desc, 'dim': 200, 'seed': 12}
    desc = [{'kind': 'coupling_rqs', 'dim': 12, 'hidden': [160], 'mask': 'ordered_left_half', 'latent_dim': 0, 'n_bins': 24,
             'lower': -3, 'upper': 3},
            {'kind': 'coupling_rqs', 'dim': 12, 'hidden': [160], 'mask': 'ordered_right_half', 'latent_dim': 0, 'n_bins': 24,
             'lower': -3, 'upper': 3}]
    torch.manual_seed(1201)
    x = torch.randn(40, 12)
    flow_case('wide_rqs', desc, 12, 13, x, arrays)
    meta['wide_rqs'] = {'desc': desc, 'dim': 12, 'seed': 13}
    save('f12_wide', arrays, meta)


# ------------------------------------------------------------------------------------------ F13
def f13_full_depth():
    """BASELINE cfg 3 and cfg 4 at the depth bench.py times them (flow.py:127-130 walks every layer): all 8 spline couplings at
    N = 64 (the reference's O(M^2) domain check, quirk Q1, allows no more on this container) and all 16 layers of cfg 4 at N = 256,
    with per-layer values and the fp64 log_prob of the same flow."""
    arrays, meta = {}, {}
    desc3 = fd.cfg3_desc()
    torch.manual_seed(1013)
    x3 = torch.randn(64, 64)
    x3[0, :] = torch.linspace(-3.5, 3.5, 64)      # both tails and the interior in one row
    h3 = {}
    flow_case('cfg3_full', desc3, 64, 0, x3, arrays, state_hashes=h3)
    meta['cfg3_full'] = {'desc': desc3, 'dim': 64, 'seed': 0, 'state_sha256': h3}
    desc4 = fd.cfg4_desc()
    torch.manual_seed(1014)
    x4 = torch.randn(256, 128)
    h4 = {}
    flow_case('cfg4_full', desc4, 128, 0, x4, arrays, state_hashes=h4, trace_every=4)
    meta['cfg4_full'] = {'desc': desc4, 'dim': 128, 'seed': 0, 'state_sha256': h4}
    save('f13_full_depth', arrays, meta)


if __name__ == '__main__':
    which = sys.argv[1:] or ['f1', 'f2', 'f3', 'f4', 'f5', 'f6', 'f7', 'f8', 'f9', 'f10', 'f11', 'f12', 'f13']
    table = {'f1': f1_doc_example, 'f2': f2_masks, 'f3': f3_cfg1, 'f4': f4_cfg2, 'f5': f5_cfg3,
             'f6': f6_cfg4, 'f7': f7_permute, 'f8': f8_suite, 'f9': f9_cubic, 'f10': f10_pointwise, 'f11': f11_continuous, 'f12': f12_wide_and_set, 'f13': f13_full_depth}
    for w in which:
        table[w]()
