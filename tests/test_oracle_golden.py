"""CPU: pin the oracle against vectors captured from the unmodified reference."""
import os
import sys

import pytest
import torch

import flowdesc as fd
from goldens import Golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import stribor_oracle as orc

TIGHT = dict(rtol=1e-6, atol=1e-6)


def test_f1_doc_known_answer():
    """stribor/test/test_normalizing_flow.py:45-55."""
    g = Golden('f1_doc_example')
    spec = fd.flow_spec(g.meta['desc'], {'transforms.' + k[len('transforms.'):]: v for k, v in g.state('').items()})
    lp = orc.flow_log_prob(spec, g.t('x'))
    assert torch.allclose(lp, torch.tensor([[-1.7560], [-1.7434], [-2.1792]]), atol=1e-4)
    assert torch.equal(lp, g.t('log_prob'))
    s = orc.flow_forward(spec, g.t('base_sample'))
    assert torch.allclose(s, torch.tensor([[-0.5204, 0.4196]]), atol=1e-4)
    assert torch.equal(s, g.t('sample'))


def test_f2_masks_exact():
    """Exact vectors of stribor/test/test_mask.py:4-30 plus captured ones for more dims."""
    assert orc.mask_vector('ordered_right_half', 5).tolist() == [0, 0, 1, 1, 1]
    assert orc.mask_vector('ordered_left_half', 5).tolist() == [1, 1, 0, 0, 0]
    assert orc.mask_vector('parity_even', 5).tolist() == [0, 1, 0, 1, 0]
    assert orc.mask_vector('parity_odd', 5).tolist() == [1, 0, 1, 0, 1]
    for n in ('ordered_right_half', 'ordered_left_half', 'parity_even', 'parity_odd'):
        assert orc.mask_vector(n, 1).tolist() == [1]
    g = Golden('f2_masks')
    for key in g.arrays:
        name, d = key.split('/')
        assert torch.equal(orc.mask_vector(name, int(d)), g.t(key)), key


@pytest.mark.parametrize('fixture,case', [('f3_cfg1', 'cfg1'), ('f4_cfg2', 'cfg2'), ('f5_cfg3', 'cfg3'),
                                          ('f6_cfg4', 'cfg4'), ('f7_permute', 'mixed'), ('f9_cubic', 'cubic_flow')])
def test_flow_fixtures(fixture, case):
    g = Golden(fixture)
    m = g.meta[case]
    spec = fd.flow_spec(m['desc'], g.state(case))
    x = g.t(case + '/x')
    trace = []
    z, acc = orc.flow_inverse_and_ldj(spec, x, trace=trace)
    n = len(spec)
    loose = case == 'cfg4'      # tri-solve ordering differs run to run at 1e-6; fp32 self error is 3.7e-6
    tol = dict(rtol=2e-5, atol=2e-5) if loose else TIGHT
    for step, (xo, ldj) in enumerate(trace):
        i = n - 1 - step
        assert torch.allclose(xo, g.t(f'{case}/inv_x.{i}'), **tol), (case, i)
        assert torch.allclose(ldj, g.t(f'{case}/inv_ldj.{i}'), **tol), (case, i)
    lp = orc.flow_log_prob(spec, x)
    ref = g.t(case + '/log_prob')
    assert lp.shape == ref.shape
    assert torch.allclose(lp, ref, rtol=1e-6 if not loose else 1e-5, atol=1e-5)
    yf, lf = orc.flow_forward_and_ldj(spec, x)
    assert torch.allclose(yf, g.t(case + '/forward'), **(tol if loose else dict(rtol=1e-5, atol=1e-5)))
    assert torch.allclose(lf, g.t(case + '/forward_ldj'), rtol=1e-5, atol=1e-4)
    if g.has(case + '/log_prob_f64'):
        lp64 = orc.flow_log_prob(orc.spec_to(spec, torch.float64), x.double())
        assert torch.allclose(lp64, g.t(case + '/log_prob_f64'), rtol=1e-10, atol=1e-9)
    for extra in ('bf16', 'wide'):
        if g.has(f'{case}/{extra}/x'):
            lpe = orc.flow_log_prob(spec, g.t(f'{case}/{extra}/x'))
            assert torch.allclose(lpe, g.t(f'{case}/{extra}/log_prob'), rtol=1e-6, atol=1e-5)


def test_f6_matrix_exponential_variants():
    g = Golden('f6_cfg4')
    for case in g.cases('matexp_'):
        m = g.meta[case]
        spec = fd.flow_spec(m['desc'], g.state(case))[0]
        x, t = g.t(case + '/x'), g.t(case + '/t')
        tol = dict(rtol=1e-5, atol=1e-5)
        assert torch.allclose(orc.transform_apply(spec, x, False, t=t), g.t(case + '/fwd_t'), **tol)
        assert torch.allclose(orc.transform_apply(spec, x, True, t=t), g.t(case + '/inv_t'), **tol)
        assert torch.allclose(orc.transform_ldj(spec, x, t=t), g.t(case + '/ldj_t'), **tol)
        assert torch.allclose(orc.transform_apply(spec, x, False, t=0.7), g.t(case + '/fwd_s'), **tol)
        assert torch.allclose(orc.transform_apply(spec, x, True, t=0.7), g.t(case + '/inv_s'), **tol)
        assert torch.allclose(orc.transform_ldj(spec, x, t=0.7), g.t(case + '/ldj_s'), **tol)
        assert torch.allclose(orc.transform_apply(spec, x, False), g.t(case + '/fwd_default'), **tol)


def test_f7_permute_flip_bit_exact():
    g = Golden('f7_permute')
    perm = g.t('perm64/permutation')
    x = g.t('perm64/x')
    spec = {'kind': 'permute', 'perm': perm}
    assert torch.equal(orc.transform_apply(spec, x, False), g.t('perm64/fwd'))
    assert torch.equal(orc.transform_apply(spec, x, True), g.t('perm64/inv'))
    assert torch.equal(orc.transform_apply({'kind': 'flip'}, x, False), g.t('flip/fwd'))
    assert torch.equal(orc.transform_apply({'kind': 'flip'}, x, True), g.t('flip/inv'))
    xb = g.t('perm64/x_bf16_bits')
    assert torch.equal(xb[..., perm], g.t('perm64/fwd_bf16_bits'))


def test_f8_suite_shapes():
    """The reference suite's shapes/protocol (stribor/test/base.py:8-44) on captured values."""
    g = Golden('f8_suite')
    n = 0
    for case, m in g.meta.items():
        d = m['desc'][0]
        spec = fd.transform_spec(d, g.state(case), 'transforms.0.')
        x = g.t(case + '/x')
        latent = g.t(case + '/latent') if g.has(case + '/latent') else None
        t = g.t(case + '/t') if g.has(case + '/t') else 1.0
        tol = dict(rtol=1e-5, atol=1e-5)
        y = orc.transform_apply(spec, x, False, latent, t)
        assert torch.allclose(y, g.t(case + '/y'), **tol), case
        xb = orc.transform_apply(spec, y, True, latent, t)
        assert torch.allclose(xb, g.t(case + '/x_back'), rtol=1e-4, atol=1e-4), case
        assert torch.allclose(xb, x, atol=1e-4), case                         # base.py:8-11
        ldj = orc.transform_ldj(spec, x, latent, t)
        assert torch.allclose(ldj, g.t(case + '/ldj'), **tol), case
        _, l2 = orc.transform_inverse_and_ldj(spec, y, latent, t)
        assert torch.allclose(ldj, -l2, atol=1e-4), case                      # base.py:21-22
        # autograd log|det J| of the reference (base.py:24-44), flattened leading dims
        assert torch.allclose(ldj.reshape(-1), g.t(case + '/autograd_logdet'), atol=1e-4), case
        assert torch.allclose(l2.reshape(-1), g.t(case + '/autograd_logdet_inv'), atol=1e-4), case
        n += 1
    assert n == 112


def test_f9_cubic_suite_shapes():
    """Cubic splines (spline_type='cubic', the reference default; stribor/test/test_spline.py:8-33 cubic rows and
    cubic couplings) on values captured from the reference: the oracle reproduces them bit for bit."""
    g = Golden('f9_cubic')
    n = 0
    for case, m in g.meta.items():
        if case == 'cubic_flow':
            continue
        d = m['desc'][0]
        spec = fd.transform_spec(d, g.state(case), 'transforms.0.')
        x = g.t(case + '/x')
        latent = g.t(case + '/latent') if g.has(case + '/latent') else None
        y = orc.transform_apply(spec, x, False, latent)
        assert torch.equal(y, g.t(case + '/y')), case
        xb = orc.transform_apply(spec, y, True, latent)
        assert torch.equal(xb, g.t(case + '/x_back')), case
        assert torch.allclose(xb, x, atol=1e-4), case                         # base.py:8-11
        ldj = orc.transform_ldj(spec, x, latent)
        assert torch.equal(ldj, g.t(case + '/ldj')), case
        _, l2 = orc.transform_inverse_and_ldj(spec, y, latent)
        assert torch.equal(l2, g.t(case + '/ldj_inv')), case
        assert torch.allclose(ldj, -l2, atol=1e-4), case                      # base.py:21-22
        assert torch.allclose(ldj.reshape(-1), g.t(case + '/autograd_logdet'), atol=1e-4), case
        n += 1
    assert n == 4 * (9 + 2)


def test_f10_pointwise_flows_and_stack():
    """Sigmoid / Logit / ELU / LeakyReLU / Cumsum / Diff / Identity on the reference suite's shapes, and the on-path part
    of test_normalizing_flow.py's stack (affine coupling -> Flip -> Sigmoid -> cubic-spline coupling -> Logit)."""
    g = Golden('f10_pointwise')
    n = 0
    for case, m in g.meta.items():
        if case == 'stack':
            continue
        spec = fd.transform_spec(m['desc'][0], {}, 'transforms.0.')
        x = g.t(case + '/x')
        y = orc.transform_apply(spec, x, False)
        assert torch.equal(y, g.t(case + '/y')), case
        assert torch.equal(orc.transform_apply(spec, y, True), g.t(case + '/x_back')), case
        ldj = orc.transform_ldj(spec, x)
        assert torch.equal(ldj, g.t(case + '/ldj')), case
        _, l2 = orc.transform_inverse_and_ldj(spec, y)
        assert torch.equal(l2, g.t(case + '/ldj_inv')), case
        assert torch.equal(orc.pointwise_log_diag(spec, x), g.t(case + '/ldiag')), case
        n += 1
    assert n == 6 * 7
    m = g.meta['stack']
    spec = fd.flow_spec(m['desc'], g.state('stack'))
    x = g.t('stack/x')
    assert torch.equal(orc.flow_log_prob(spec, x), g.t('stack/log_prob'))
    assert torch.equal(orc.flow_forward(spec, x), g.t('stack/forward'))
    assert torch.equal(orc.flow_inverse(spec, x), g.t('stack/inverse'))


def test_f11_continuous_affine_coupling_and_neural_flow():
    """ContinuousAffineCoupling on the reference suite's shapes (test_coupling.py:29-52) with every time net on the path,
    and the NeuralFlow container (test_neural_flow.py): bit-exact against the reference's values."""
    g = Golden('f11_continuous')
    n = 0
    for case, m in g.meta.items():
        if case == 'neural_flow':
            continue
        spec = fd.transform_spec(m['desc'][0], g.state(case), 'transforms.0.')
        x, t = g.t(case + '/x'), g.t(case + '/t')
        latent = g.t(case + '/latent') if g.has(case + '/latent') else None
        y, ldj = orc.continuous_affine_coupling(spec, x, t, latent, False)
        assert torch.equal(y, g.t(case + '/y')) and torch.equal(ldj, g.t(case + '/ldj')), case
        xb, li = orc.continuous_affine_coupling(spec, y, t, latent, True)
        assert torch.equal(xb, g.t(case + '/x_back')) and torch.equal(-li, g.t(case + '/ldj_inv')), case
        n += 1
    assert n == 4 * (4 + 1 + 1)
    m = g.meta['neural_flow']
    spec = [fd.transform_spec(d, g.state('neural_flow'), f'transforms.{i}.') for i, d in enumerate(m['desc'])]
    x, t, t0 = g.t('neural_flow/x'), g.t('neural_flow/t'), g.t('neural_flow/t0')
    assert torch.equal(orc.neural_flow_forward(spec, x, t), g.t('neural_flow/y_t'))
    assert torch.equal(orc.neural_flow_forward(spec, x, t, t0), g.t('neural_flow/y_t_t0'))
    assert torch.equal(orc.neural_flow_forward(spec, x, torch.zeros_like(t)), x)          # identity at t = 0


def test_f12_set_data_hand_nets_and_wide_flows():
    """Round 2 fixtures: Coupling(set_data=True) (coupling.py:48-51), conditioners that are not a stribor MLP, Fourier time
    nets, widths beyond the fused kernel's tiles (D=200 / H=256, n_bins=24) -- the oracle against the reference's values."""
    g = Golden('f12_set_and_hand')
    n = 0
    for case, m in g.meta.items():
        d = m['desc'][0]
        spec = fd.transform_spec(d, g.state(case), 'transforms.0.')
        x = g.t(case + '/x')
        if case.startswith('cac/'):
            t = g.t(case + '/t')
            y, ldj = orc.continuous_affine_coupling(spec, x, t, None, False)
            assert torch.allclose(y, g.t(case + '/y'), rtol=1e-6, atol=1e-6), case
            assert torch.allclose(ldj, g.t(case + '/ldj'), rtol=1e-6, atol=1e-6), case
            xb, li = orc.continuous_affine_coupling(spec, g.t(case + '/y'), t, None, True)
            assert torch.allclose(xb, g.t(case + '/x_back'), rtol=1e-5, atol=1e-5), case
            assert torch.allclose(-li, g.t(case + '/ldj_inv'), rtol=1e-6, atol=1e-6), case
            n += 1
            continue
        latent = g.t(case + '/latent') if g.has(case + '/latent') else None
        tol = dict(rtol=1e-5, atol=1e-5)
        y = orc.transform_apply(spec, x, False, latent)
        assert torch.allclose(y, g.t(case + '/y'), **tol), case
        assert torch.allclose(orc.transform_apply(spec, y, True, latent), g.t(case + '/x_back'), rtol=1e-4, atol=1e-4), case
        assert torch.allclose(orc.transform_ldj(spec, x, latent, y=y), g.t(case + '/ldj'), rtol=1e-5, atol=1e-4), case
        _, l2 = orc.transform_inverse_and_ldj(spec, y, latent)
        assert torch.allclose(l2, g.t(case + '/ldj_inv'), rtol=1e-5, atol=1e-4), case
        n += 1
    assert n == 38
    g = Golden('f12_wide')
    for case in ('wide_affine', 'wide_rqs'):
        m = g.meta[case]
        spec = fd.flow_spec(m['desc'], g.state(case))
        x = g.t(case + '/x')
        assert torch.allclose(orc.flow_log_prob(spec, x), g.t(case + '/log_prob'), rtol=1e-5, atol=1e-4), case
        assert torch.allclose(orc.flow_inverse(spec, x), g.t(case + '/inverse'), rtol=1e-5, atol=1e-5), case
        assert torch.allclose(orc.flow_forward(spec, x), g.t(case + '/forward'), rtol=1e-5, atol=1e-5), case


@pytest.mark.parametrize('case', ['cfg3_full', 'cfg4_full'])
def test_f13_full_depth_flows(case):
    """BASELINE cfg 3 (8 spline couplings) and cfg 4 (16 layers) at the depth bench.py times, against the reference's per-layer
    values, forward pass and fp64 log_prob (flow.py:118-130); weights = the reference's seeded default init, held to its hashes."""
    g = Golden('f13_full_depth')
    m = g.meta[case]
    spec = fd.flow_spec(m['desc'], g.seeded_state(case))
    x = g.t(case + '/x')
    trace = []
    z, acc = orc.flow_inverse_and_ldj(spec, x, trace=trace)
    n = len(spec)
    assert n == (8 if case == 'cfg3_full' else 16)
    loose = case == 'cfg4_full'
    tol = dict(rtol=2e-5, atol=2e-5) if loose else TIGHT
    seen = 0
    for step, (xo, ldj) in enumerate(trace):
        i = n - 1 - step
        if g.has(f'{case}/inv_x.{i}'):
            assert torch.allclose(xo, g.t(f'{case}/inv_x.{i}'), **tol), (case, i)
            seen += 1
        assert torch.allclose(ldj, g.t(f'{case}/inv_ldj.{i}'), **tol), (case, i)
    assert seen >= 4
    assert torch.allclose(z, g.t(case + '/inverse'), **tol)
    lp = orc.flow_log_prob(spec, x)
    assert torch.allclose(lp, g.t(case + '/log_prob'), rtol=1e-5 if loose else 1e-6, atol=1e-5)
    yf, lf = orc.flow_forward_and_ldj(spec, x)
    assert torch.allclose(yf, g.t(case + '/forward'), **(tol if loose else dict(rtol=1e-5, atol=1e-5)))
    assert torch.allclose(lf, g.t(case + '/forward_ldj'), rtol=1e-5, atol=1e-4)
    lp64 = orc.flow_log_prob(orc.spec_to(spec, torch.float64), x.double())
    assert torch.allclose(lp64, g.t(case + '/log_prob_f64'), rtol=1e-10, atol=1e-9)
