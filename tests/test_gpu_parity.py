"""GPU: the HIP path (through the C ABI) against the golden vectors and the oracle.

Tolerance: the north_star bound "<= 1e-5 rel for fp32 transforms" is applied as
|got - want| <= 1e-5 + 1e-5*|want|; Permute/Flip are bit-exact.
"""
import os
import sys

import numpy as np
import pytest
import torch

import flowdesc as fd
from goldens import Golden
from producthelp import close, close_vs_f64, product_flow, product_transform

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stribor_oracle as orc

import stribor_amd as st

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_mode():
    """Parity of the inference kernels: with a graph (parameters require grad by default) log_prob of spline / mixed flows
    and ContinuousAffineCoupling would take the layer-wise training path instead, which tests/test_gpu_backward.py covers."""
    with torch.no_grad():
        yield

DEV = 'cuda'


def test_f1_doc_known_answer():
    """stribor/test/test_normalizing_flow.py:45-55 on the GPU."""
    g = Golden('f1_doc_example')
    flow = fd.build_flow(st, g.meta['desc'], 2)
    flow.load_state_dict(g.state(''))
    flow = flow.to(DEV)
    lp = flow.log_prob(g.t('x').to(DEV))
    assert lp.shape == (3, 1)
    close(lp, torch.tensor([[-1.7560], [-1.7434], [-2.1792]]), rtol=0, atol=1e-4)
    close(lp, g.t('log_prob'))
    s = flow.forward(g.t('base_sample').to(DEV))
    close(s, torch.tensor([[-0.5204, 0.4196]]), rtol=0, atol=1e-4)
    close(s, g.t('sample'))
    assert flow.sample(5).shape == (5, 2) and flow.sample(5).is_cuda


@pytest.mark.parametrize('fixture,case', [('f3_cfg1', 'cfg1'), ('f4_cfg2', 'cfg2'), ('f7_permute', 'mixed')])
def test_fused_flow_against_golden(fixture, case):
    g = Golden(fixture)
    flow = product_flow(g, case)
    x = g.t(case + '/x').to(DEV)
    close(flow.log_prob(x), g.t(case + '/log_prob'))
    z, ldj = flow.inverse_and_log_det_jacobian(x)
    close(z, g.t(case + '/inverse'))
    close(ldj, g.t(case + '/inverse_ldj'), atol=1e-4)
    close(flow.inverse(x), g.t(case + '/inverse'))
    y, ldf = flow.forward_and_log_det_jacobian(x)
    close(y, g.t(case + '/forward'))
    close(ldf, g.t(case + '/forward_ldj'), atol=1e-4)
    close(flow.forward(x), g.t(case + '/forward'))
    close(flow.log_det_jacobian(x, None), g.t(case + '/forward_ldj'), atol=1e-4)
    # vs fp64 truth: not worse than the reference's own fp32 error by more than 1e-5 rel
    if g.has(case + '/log_prob_f64'):
        close(flow.log_prob(x).double(), g.t(case + '/log_prob_f64'), rtol=1e-5, atol=1e-5)
    for extra in ('bf16', 'wide'):
        if g.has(f'{case}/{extra}/x'):
            close(flow.log_prob(g.t(f'{case}/{extra}/x').to(DEV)), g.t(f'{case}/{extra}/log_prob'))


@pytest.mark.parametrize('fixture,case', [('f3_cfg1', 'cfg1'), ('f4_cfg2', 'cfg2'), ('f7_permute', 'mixed')])
def test_per_layer_api_against_golden(fixture, case):
    """The Transform plugin surface layer by layer (flow.py:42-47 called on every transform)."""
    g = Golden(fixture)
    flow = product_flow(g, case)
    cur = g.t(case + '/x').to(DEV)
    for i in reversed(range(len(flow.transforms))):
        f = flow.transforms[i]
        nxt, ldj = f.inverse_and_log_det_jacobian(cur)
        close(nxt, g.t(f'{case}/inv_x.{i}'))
        close(ldj, g.t(f'{case}/inv_ldj.{i}'), atol=2e-5)
        close(f.inverse(cur), g.t(f'{case}/inv_x.{i}'))
        close(f.log_det_jacobian(nxt, cur), -g.t(f'{case}/inv_ldj.{i}'), atol=2e-5)
        close(f.forward(nxt), cur.cpu(), atol=1e-4)                # base.py:8-11 round trip
        cur = nxt


def test_bf16_storage_matches_oracle_on_rounded_input():
    """x stored bf16, arithmetic fp32, log_prob fp32 == oracle fed x.bfloat16().float()."""
    g = Golden('f4_cfg2')
    flow = product_flow(g, 'cfg2')
    xb = g.t('cfg2/bf16/x')
    lp = flow.log_prob(xb.to(DEV).bfloat16())
    assert lp.dtype == torch.float32
    close(lp, g.t('cfg2/bf16/log_prob'))
    z = flow.inverse(xb.to(DEV).bfloat16())
    assert z.dtype == torch.bfloat16
    spec = fd.flow_spec(g.meta['cfg2']['desc'], g.state('cfg2'))
    close(z.float(), orc.flow_inverse(spec, xb).bfloat16().float(), rtol=1e-2, atol=1e-2)


def test_suite_shapes_coupling_affine_and_affine():
    """stribor/test/test_coupling.py:7-26, test_affine.py:27-40 shapes incl. latent 0/1/13 and 3-D inputs."""
    g = Golden('f8_suite')
    n = 0
    for case in g.cases('coupling_affine/') + g.cases('affine_latent/'):
        f = product_transform(g, case)
        x = g.t(case + '/x').to(DEV)
        kw = {'latent': g.t(case + '/latent').to(DEV)} if g.has(case + '/latent') else {}
        y = f(x, **kw)
        close(y, g.t(case + '/y'))
        close(f.inverse(y, **kw), g.t(case + '/x'), atol=1e-4)
        ldj = f.log_det_jacobian(x, y, **kw)
        close(ldj, g.t(case + '/ldj'), atol=2e-5)
        _, l1 = f.forward_and_log_det_jacobian(x, **kw)
        _, l2 = f.inverse_and_log_det_jacobian(y, **kw)
        close(l1, g.t(case + '/ldj'), atol=2e-5)
        close(-l2, g.t(case + '/ldj'), atol=1e-4)                              # base.py:14-22
        close(ldj.reshape(-1), g.t(case + '/autograd_logdet'), atol=1e-4)      # base.py:35-44
        n += 1
    assert n == 4 * 3 + 4 * 2


def test_permute_flip_bit_exact():
    g = Golden('f7_permute')
    p = st.Permute(64)
    p.load_state_dict({'permutation': g.t('perm64/permutation')})
    p = p.to(DEV)
    x = g.t('perm64/x').to(DEV)
    assert torch.equal(p(x).cpu(), g.t('perm64/fwd'))
    assert torch.equal(p.inverse(x).cpu(), g.t('perm64/inv'))
    assert torch.equal(p.inverse(p(x)), x)
    xb = g.t('perm64/x_bf16_bits').to(DEV).view(torch.bfloat16)
    assert torch.equal(p(xb).view(torch.int16).cpu(), g.t('perm64/fwd_bf16_bits'))
    fl = st.Flip([-1])
    assert torch.equal(fl(x).cpu(), g.t('flip/fwd'))
    assert torch.equal(fl.inverse(x).cpu(), g.t('flip/inv'))
    assert torch.equal(p.log_det_jacobian(x, x), torch.zeros(33, 1, device=DEV))
    g8 = Golden('f8_suite')
    for case in g8.cases('permute/') + g8.cases('flip/'):
        f = product_transform(g8, case)
        x = g8.t(case + '/x').to(DEV)
        assert torch.equal(f(x).cpu(), g8.t(case + '/y'))
        assert torch.equal(f.inverse(f(x)), x)


def test_unit_normal_log_prob():
    torch.manual_seed(0)
    for shape in [(5, 64), (3, 7, 5), (1, 1), (1000, 128)]:
        x = torch.randn(*shape)
        got = st.UnitNormal(shape[-1]).to(DEV).log_prob(x.to(DEV))
        close(got, orc.unit_normal_log_prob(x))


def test_mlp_standalone_matches_oracle():
    """net/mlp.py:65 through the fused MFMA kernel: 1 and 2 hidden layers, odd widths, every activation."""
    torch.manual_seed(3)
    for (i, hs, o, act) in [(64, [64], 128, 'Tanh'), (10, [13], 20, 'Tanh'), (5, [12, 7], 9, 'ReLU'),
                            (32, [64, 64], 3008, 'Tanh'), (2, [64], 4, 'Sigmoid'), (3, [8], 5, 'ELU'),
                            (3, [8], 5, 'Softplus'), (3, [8], 5, 'LeakyReLU'), (3, [8], 5, 'SiLU'), (3, [8], 5, 'GELU')]:
        net = st.net.MLP(i, hs, o, activation=act)
        x = torch.randn(77, i)
        spec = {'weights': [w for (w, _) in net.linears()], 'biases': [b for (_, b) in net.linears()], 'activation': act}
        with torch.no_grad():
            want = orc.mlp_forward(spec, x)
        got = net.to(DEV)(x.to(DEV))
        close(got, want, rtol=1e-5, atol=2e-6)


def test_elementwise_affine_kernel_against_oracle():
    """sx_affine_coupling standalone (params in HBM): vector path, generic path, bf16, broadcast params."""
    from stribor_amd.flows.affine import run_affine_kernel
    torch.manual_seed(1)
    for (n, d, l0, nl, bf) in [(1000, 64, 0, 32, False), (1000, 64, 32, 32, True), (33, 10, 5, 5, False),
                               (257, 128, 64, 64, False), (5, 7, 0, 7, False)]:
        x = torch.randn(n, d)
        if bf:
            x = x.bfloat16().float()
        params = torch.randn(n, 2 * nl) * 0.5
        ls, sh = params[:, :nl], params[:, nl:]
        for reverse in (False, True):
            want = x.clone()
            want[:, l0:l0 + nl] = orc.affine_apply(x[:, l0:l0 + nl], ls, sh, reverse)
            xin = x.to(DEV).bfloat16() if bf else x.to(DEV)
            y, ldj = run_affine_kernel(xin, params.to(DEV), 2 * nl, None, l0, nl, reverse, True, True, -1.0)
            if bf:
                close(y.float(), want.bfloat16().float(), rtol=1e-2, atol=1e-2)
            else:
                close(y, want)
            close(ldj, -ls.sum(-1), atol=2e-5)
    # scattered live set -> generic kernel with live_idx
    x = torch.randn(50, 10)
    live = torch.tensor([1, 4, 6, 9], dtype=torch.int32)
    params = torch.randn(50, 8) * 0.3
    want = x.clone()
    want[:, live.long()] = orc.affine_apply(x[:, live.long()], params[:, :4], params[:, 4:], False)
    y, ldj = run_affine_kernel(x.to(DEV), params.to(DEV), 8, live.to(DEV), 0, 4, False, True, True)
    close(y, want)
    close(ldj, params[:, :4].sum(-1), atol=2e-5)


def test_empty_and_ragged_batches():
    g = Golden('f4_cfg2')
    flow = product_flow(g, 'cfg2')
    x = g.t('cfg2/x')
    want = g.t('cfg2/log_prob')
    assert flow.log_prob(x[:0].to(DEV)).shape == (0, 1)
    for n in (1, 31, 33, 127, 129, 255):
        close(flow.log_prob(x[:n].to(DEV)), want[:n])
    # 3-D leading shape (test_coupling.py:7 uses (7, 4, 5))
    close(flow.log_prob(x[:252].reshape(7, 36, 64).to(DEV)), want[:252].reshape(7, 36, 1))


def test_log_prob_sum_is_sum_of_log_probs():
    g = Golden('f4_cfg2')
    flow = product_flow(g, 'cfg2')
    x = g.t('cfg2/x').to(DEV)
    s = flow.log_prob_sum(x)
    assert s.dtype == torch.float64
    want = g.t('cfg2/log_prob').double().sum()
    assert abs(s.item() - want.item()) <= 1e-6 * abs(want.item())


def test_full_size_properties():
    """BASELINE cfg 2 at N = 2^20: inverse(forward(x)) == x, three-way log-det consistency, batch-split
    invariance, agreement with the oracle on a slice."""
    torch.manual_seed(0)
    flow = fd.build_flow(st, fd.cfg2_desc(), 64).to(DEV)
    n = 1 << 20
    x = torch.randn(n, 64, device=DEV)
    y, ldj_f = flow.forward_and_log_det_jacobian(x)
    xb, ldj_i = flow.inverse_and_log_det_jacobian(y)
    assert (xb - x).abs().max().item() < 1e-4                                  # base.py:8-11
    assert (ldj_f + ldj_i).abs().max().item() < 1e-3                           # base.py:21-22
    lp = flow.log_prob(x)
    assert torch.isfinite(lp).all()
    lp2 = torch.cat([flow.log_prob(x[:300_001]), flow.log_prob(x[300_001:])])
    assert torch.equal(lp, lp2)                                                # rows are independent
    spec = fd.flow_spec(fd.cfg2_desc(), {k: v.cpu() for k, v in flow.state_dict().items()})
    sl = slice(777_000, 777_512)
    close(lp[sl], orc.flow_log_prob(spec, x[sl].cpu()))
    s = flow.log_prob_sum(x)
    assert abs(s.item() - lp.double().sum().item()) <= 1e-9 * abs(s.item())


# ------------------------------------------------------------------------------------------------
# rational-quadratic spline
# ------------------------------------------------------------------------------------------------
def test_suite_shapes_rqs_and_coupling_rqs():
    """stribor/test/test_spline.py:8-33 (quadratic rows: n_bins 1/3/10, latent 0/1/13, box [0,2]) plus the
    same protocol through Coupling."""
    g = Golden('f8_suite')
    n = 0
    for case in g.cases('rqs/') + g.cases('coupling_rqs/'):
        f = product_transform(g, case)
        x = g.t(case + '/x').to(DEV)
        kw = {'latent': g.t(case + '/latent').to(DEV)} if g.has(case + '/latent') else {}
        y = f(x, **kw)
        close(y, g.t(case + '/y'))
        close(f.inverse(y, **kw), g.t(case + '/x'), atol=1e-4)                 # base.py:8-11
        ldj = f.log_det_jacobian(x, y, **kw)
        close(ldj, g.t(case + '/ldj'), atol=1e-4)                              # the suite's own atol (base.py:22)
        _, l1 = f.forward_and_log_det_jacobian(x, **kw)
        _, l2 = f.inverse_and_log_det_jacobian(y, **kw)
        close(l1, g.t(case + '/ldj'), atol=1e-4)
        close(-l2, g.t(case + '/ldj'), atol=1e-4)                              # base.py:14-22
        close(ldj.reshape(-1), g.t(case + '/autograd_logdet'), atol=1e-4)      # base.py:35-44
        if g.has(case + '/ldiag'):
            close(f.log_diag_jacobian(x, y, **kw), g.t(case + '/ldiag'), atol=1e-4)
        n += 1
    assert n == 4 * 9 + 4 * 3
    from stribor_amd.flows.spline import check_errors
    check_errors()


def test_cfg3_spline_flow_against_golden():
    """2 RQ-spline couplings at the cfg-3 widths (D=64, K=16, H=64, box [-3,3]) incl. rows in the tails,
    exactly on the bounds and on a grid through the knots (fixture F5)."""
    g = Golden('f5_cfg3')
    flow = product_flow(g, 'cfg3')
    x = g.t('cfg3/x').to(DEV)
    assert flow._fused_program(True, 64, 0, x.device) is not None            # whole spline flow = one launch
    assert flow._fused_program(False, 64, 0, x.device) is not None
    cur = x
    for i in reversed(range(len(flow.transforms))):
        nxt, ldj = flow.transforms[i].inverse_and_log_det_jacobian(cur)
        close(nxt, g.t(f'cfg3/inv_x.{i}'))
        close(ldj, g.t(f'cfg3/inv_ldj.{i}'), rtol=1e-5, atol=1e-4)
        cur = nxt
    close(flow.log_prob(x), g.t('cfg3/log_prob'), rtol=1e-5, atol=1e-4)
    close(flow.forward(x), g.t('cfg3/forward'))
    close(flow.inverse(x), g.t('cfg3/inverse'))
    y, ldf = flow.forward_and_log_det_jacobian(x)
    close(ldf, g.t('cfg3/forward_ldj'), rtol=1e-5, atol=1e-4)
    close(flow.log_prob(x).double(), g.t('cfg3/log_prob_f64'), rtol=1e-5, atol=1e-4)


def test_rqs_kernel_against_oracle_random_params():
    """sx_rqs_coupling standalone: random parameters, both directions, asymmetric box (test_spline.py:36-52),
    bf16 storage, scattered live columns, K from 1 to 32."""
    from stribor_amd.flows.spline import run_rqs_kernel
    torch.manual_seed(5)
    for (n, d, K, box) in [(300, 8, 5, (-1.0, 1.0, -3.0, 2.0)), (64, 64, 16, (-3.0, 3.0, -3.0, 3.0)),
                           (17, 3, 1, (0.0, 2.0, 0.0, 2.0)), (50, 6, 32, (-2.0, 2.0, -2.0, 2.0))]:
        left, right, bottom, top = box
        P = 3 * K - 1
        params = torch.randn(n, d * P)
        pv = params.view(n, d, P)
        uw, uh, ud = pv[..., :K], pv[..., K:2 * K], pv[..., 2 * K:]
        x = torch.rand(n, d) * (right - left) * 1.2 + left - 0.1 * (right - left)      # some rows in the tails
        yo, lo = orc.rqs_unconstrained(x, uw, uh, ud, False, None, None, left, right, bottom, top)
        y64, l64 = orc.rqs_unconstrained(x.double(), uw.double(), uh.double(), ud.double(), False, None, None,
                                         left, right, bottom, top)
        y, ldj, ldiag = run_rqs_kernel(x.to(DEV), params.to(DEV), d * P, None, 0, d, K, left, right, bottom, top,
                                       False, True, True)
        close(y, yo)
        # The log-derivative divides by a bin width that the reference forms as a DIFFERENCE of two cumsum
        # knots (rational_quadratic_spline.py:185,192): for narrow bins that cancellation amplifies fp32
        # rounding to ~1e-4, in the reference as much as here.  Bar: 2e-4 against the reference's fp32 values
        # AND no further from the fp64 truth than the reference's own fp32 path (x2 + 1e-5).
        close(ldiag, lo, atol=2e-4)
        ref_err = (lo.double() - l64).abs().max().item()
        our_err = (ldiag.cpu().double() - l64).abs().max().item()
        assert our_err <= 2 * ref_err + 1e-5, (our_err, ref_err)
        close(ldj, lo.sum(-1), atol=5e-4)
        xi = torch.rand(n, d) * (top - bottom) * 1.2 + bottom - 0.1 * (top - bottom)
        xo, li = orc.rqs_unconstrained(xi, uw, uh, ud, True, None, None, left, right, bottom, top)
        xg, ldj_i, ldiag_i = run_rqs_kernel(xi.to(DEV), params.to(DEV), d * P, None, 0, d, K, left, right, bottom,
                                            top, True, True, True)
        x64, li64 = orc.rqs_unconstrained(xi.double(), uw.double(), uh.double(), ud.double(), True, None, None,
                                          left, right, bottom, top)
        close(xg, xo, atol=3e-4)                       # quadratic-root cancellation, same conditioning argument
        assert (xg.cpu().double() - x64).abs().max().item() <= 2 * (xo.double() - x64).abs().max().item() + 1e-5
        close(ldiag_i, li, atol=5e-4)
        assert (ldiag_i.cpu().double() - li64).abs().max().item() <= 2 * (li.double() - li64).abs().max().item() + 1e-5
    # scattered live columns + pass-through copy
    n, d, K = 40, 10, 4
    P = 3 * K - 1
    live = torch.tensor([0, 3, 4, 9], dtype=torch.int32)
    params = torch.randn(n, 4 * P)
    x = torch.rand(n, d) * 2 - 1
    pv = params.view(n, 4, P)
    yo, lo = orc.rqs_unconstrained(x[:, live.long()], pv[..., :K], pv[..., K:2 * K], pv[..., 2 * K:], False, -1., 1.)
    want = x.clone()
    want[:, live.long()] = yo
    y, ldj, ldiag = run_rqs_kernel(x.to(DEV), params.to(DEV), 4 * P, live.to(DEV), 0, 4, K, -1, 1, -1, 1, False, True, True)
    close(y, want)
    close(ldj, lo.sum(-1), atol=3e-4)
    wd = torch.zeros(n, d)
    wd[:, live.long()] = lo
    close(ldiag, wd, atol=2e-4)


def test_spline_error_behaviour():
    with pytest.raises(ValueError, match='Minimal bin width too large'):       # rational_quadratic_spline.py:96-97
        st.Spline(2, 1001, spline_type='quadratic').to(DEV)(torch.rand(3, 2, device=DEV))


# ------------------------------------------------------------------------------------------------
# dense linear layers
# ------------------------------------------------------------------------------------------------
def test_suite_shapes_affine_lu_and_matrix_exponential():
    """stribor/test/test_affine.py:44-80: AffineLU; MatrixExponential with bias / log_time, default t and a
    per-row t tensor; identity at t = 0 (:77-80)."""
    g = Golden('f8_suite')
    n = 0
    for case in g.cases('affine_lu/') + g.cases('matrix_exp/'):
        f = product_transform(g, case)
        x = g.t(case + '/x').to(DEV)
        kw = {'t': g.t(case + '/t').to(DEV)} if g.has(case + '/t') else {}
        y = f(x, **kw)
        close(y, g.t(case + '/y'), rtol=2e-5, atol=2e-5)
        close(f.inverse(y, **kw), g.t(case + '/x'), atol=1e-4)                 # base.py:8-11
        ldj = f.log_det_jacobian(x, y, **kw)
        close(ldj, g.t(case + '/ldj'), atol=2e-5)
        _, l1 = f.forward_and_log_det_jacobian(x, **kw)
        _, l2 = f.inverse_and_log_det_jacobian(y, **kw)
        close(l1, g.t(case + '/ldj'), atol=2e-5)
        close(-l2, g.t(case + '/ldj'), atol=1e-4)
        close(ldj.reshape(-1), g.t(case + '/autograd_logdet'), atol=1e-4)      # base.py:35-44
        if case.startswith('matrix_exp/') and '/b0' in case:
            t0 = torch.zeros(*x.shape[:-1], 1, device=DEV)
            close(f(x, t=t0), x.cpu(), atol=1e-6)                              # test_affine.py:77-80
        n += 1
    assert n == 4 + 4 * 4 * 2


def test_matrix_exponential_variants():
    g = Golden('f6_cfg4')
    for case in g.cases('matexp_'):
        f = product_flow(g, case).transforms[0]
        x, t = g.t(case + '/x').to(DEV), g.t(case + '/t').to(DEV)
        tol = dict(rtol=2e-5, atol=2e-5)
        close(f(x, t=t), g.t(case + '/fwd_t'), **tol)
        close(f.inverse(x, t=t), g.t(case + '/inv_t'), **tol)
        close(f.log_det_jacobian(x, None, t=t), g.t(case + '/ldj_t'), **tol)
        close(f(x, t=0.7), g.t(case + '/fwd_s'), **tol)
        close(f.inverse(x, t=0.7), g.t(case + '/inv_s'), **tol)
        close(f.log_det_jacobian(x, None, t=0.7), g.t(case + '/ldj_s'), **tol)
        close(f(x), g.t(case + '/fwd_default'), **tol)


def test_cfg4_flow_against_golden():
    """D=128: [AffineLU, Coupling(Affine), MatrixExponential, Coupling(Affine)] x 2, fully fused (fixture F6).
    log_prob and every log-det: 1e-5 rel against the reference's fp32 values and against the fp64 truth.
    Transformed values: this flow's default-init matrices amplify by ~1e3 per block and the reference's own fp32 result
    (substitution order inside the triangular solves) is only 1e-6..4e-6 rel accurate, so the values are held to the
    north_star's bound MEASURED AGAINST FP64: never more than twice the reference's own fp32 error + 1e-5 rel
    (close_vs_f64), layer by layer and end to end."""
    g = Golden('f6_cfg4')
    flow = product_flow(g, 'cfg4')
    x = g.t('cfg4/x').to(DEV)
    spec64 = orc.spec_to(fd.flow_spec(g.meta['cfg4']['desc'], g.state('cfg4')), torch.float64)
    x64 = g.t('cfg4/x').double()
    assert flow._fused_program(True, 128, 0, x.device) is not None             # one launch, no per-layer fallback
    lp = flow.log_prob(x)
    close(lp, g.t('cfg4/log_prob'), rtol=1e-5, atol=1e-4)
    close(lp.double(), g.t('cfg4/log_prob_f64'), rtol=1e-5, atol=1e-4)
    z, ldj = flow.inverse_and_log_det_jacobian(x)
    close_vs_f64(z, g.t('cfg4/inverse'), orc.flow_inverse(spec64, x64))
    close(ldj, g.t('cfg4/inverse_ldj'), rtol=1e-5, atol=1e-4)
    y, ldf = flow.forward_and_log_det_jacobian(x)
    close_vs_f64(y, g.t('cfg4/forward'), orc.flow_forward(spec64, x64))
    close(ldf, g.t('cfg4/forward_ldj'), rtol=1e-5, atol=1e-4)
    # layer by layer, each layer fed the REFERENCE's fp32 input of that layer (so errors do not compound across layers)
    n = len(flow.transforms)
    for i in reversed(range(n)):
        cur32 = g.t('cfg4/x') if i == n - 1 else g.t(f'cfg4/inv_x.{i + 1}')
        nxt, l = flow.transforms[i].inverse_and_log_det_jacobian(cur32.to(DEV))
        want64 = orc.transform_apply(spec64[i], cur32.double(), True)
        close_vs_f64(nxt, g.t(f'cfg4/inv_x.{i}'), want64)
        close(l, g.t(f'cfg4/inv_ldj.{i}'), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('case,dim', [('cfg3_full', 64), ('cfg4_full', 128)])
def test_f13_full_depth_flows_against_golden(case, dim):
    """BASELINE cfg 3 (all 8 spline couplings, N=64) and cfg 4 (all 16 layers, N=256) as ONE fused launch against the reference's own
    values at full depth (fixture F13, flow.py:118-130): log_prob, inverse / forward with log-dets, and the per-layer log-dets.
    cfg 3: 1e-5 rel on values, 1e-4 abs on the log-dets of 8 x 32 spline elements; cfg 4: values against fp64 (close_vs_f64, as
    test_cfg4_flow_against_golden: the reference's own fp32 error at this init is 1e-6..4e-6 rel per block)."""
    g = Golden('f13_full_depth')
    m = g.meta[case]
    state = g.seeded_state(case)
    flow = fd.build_flow(st, m['desc'], dim)
    flow.load_state_dict(state)
    flow = flow.to(DEV)
    x = g.t(case + '/x').to(DEV)
    assert flow._fused_program(True, dim, 0, x.device) is not None
    assert flow._fused_program(False, dim, 0, x.device) is not None
    spec64 = orc.spec_to(fd.flow_spec(m['desc'], state), torch.float64)
    x64 = g.t(case + '/x').double()
    lp = flow.log_prob(x)
    close(lp, g.t(case + '/log_prob'), rtol=1e-5, atol=2e-4)
    close(lp.double(), g.t(case + '/log_prob_f64'), rtol=1e-5, atol=2e-4)
    z, ldj = flow.inverse_and_log_det_jacobian(x)
    y, ldf = flow.forward_and_log_det_jacobian(x)
    close(ldj, g.t(case + '/inverse_ldj'), rtol=1e-5, atol=2e-4)
    close(ldf, g.t(case + '/forward_ldj'), rtol=1e-5, atol=2e-4)
    # cfg 4: sixteen layers, eight of them dense 128 x 128 maps whose default init amplifies by ~1e3 per block: an element's error is
    # set by the row's largest entries (the reference's own fp32 result is 1e-5 .. 1e-4 of max |row| off its fp64 self on small
    # elements), so the 1e-5 allowance is taken relative to max |row| there; cfg 3 (element-wise splines): element by element
    dense = case == 'cfg4_full'
    close_vs_f64(z, g.t(case + '/inverse'), orc.flow_inverse(spec64, x64), row_scale=dense)
    close_vs_f64(y, g.t(case + '/forward'), orc.flow_forward(spec64, x64), row_scale=dense)
    # per-layer log-dets along the reference's own trajectory (each layer fed the reference's fp32 input where the fixture holds it)
    n = len(flow.transforms)
    cur = x
    for i in reversed(range(n)):
        nxt, l = flow.transforms[i].inverse_and_log_det_jacobian(cur)
        if g.has(f'{case}/inv_x.{i}'):
            cur = g.t(f'{case}/inv_x.{i}').to(DEV)            # re-anchor on the reference's trajectory
        else:
            cur = nxt
        if g.has(f'{case}/inv_x.{i + 1}') or i == n - 1:
            close(l, g.t(f'{case}/inv_ldj.{i}'), rtol=1e-5, atol=1e-4)


def test_mixed_spline_affine_flow_falls_back_per_layer_and_matches_oracle():
    """A flow the fused kernel cannot take whole (spline + affine couplings + AffineLU) runs layer by layer on
    HIP kernels; checked against the oracle built from the same state_dict."""
    torch.manual_seed(11)
    desc = [fd.cfg3_desc(1, 16, 24, 5)[0], fd.cfg2_desc(2, 16, 24)[1], {'kind': 'affine_lu', 'dim': 16},
            {'kind': 'permute', 'dim': 16}, fd.cfg3_desc(2, 16, 24, 5)[1]]
    flow = fd.build_flow(st, desc, 16)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    x = torch.randn(300, 16)
    assert flow._fused_program(True, 16, 0, torch.device(DEV)) is None
    with torch.no_grad():
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=1e-4)
        close(flow.forward(x.to(DEV)), orc.flow_forward(spec, x), rtol=2e-5, atol=2e-5)
        close(flow.inverse(x.to(DEV)), orc.flow_inverse(spec, x), rtol=2e-5, atol=2e-5)


def test_coupling_with_deep_conditioner_and_other_activations():
    """Conditioners beyond Linear-Tanh-Linear: two hidden layers (fused as CPL_HIDDEN -> COUPLING_AFFINE_DEEP steps,
    hidden state in registers) and other activations (run-time activation path)."""
    torch.manual_seed(21)
    for hidden, act, dim, mask in [([24, 16], 'Tanh', 10, 'ordered_left_half'), ([40], 'ReLU', 64, 'ordered_right_half'),
                                   ([32, 32], 'ELU', 7, 'parity_even')]:
        net = st.net.MLP(dim, hidden, 2 * dim, activation=act)
        f = st.Coupling(st.Affine(dim, latent_net=net), mask=mask)
        lin = net.linears()
        spec = {'kind': 'coupling_affine', 'mask': mask,
                'net': {'weights': [w.detach().clone() for (w, _) in lin], 'biases': [b.detach().clone() for (_, b) in lin],
                        'activation': act}}
        f = f.to(DEV)
        x = torch.randn(130, dim)
        with torch.no_grad():
            y = f(x.to(DEV))
            close(y, orc.transform_apply(spec, x, False))
            close(f.inverse(y), x, atol=1e-4)
            yy, ldj = f.forward_and_log_det_jacobian(x.to(DEV))
            close(ldj, orc.transform_ldj(spec, x), atol=2e-5)
            xb, ldi = f.inverse_and_log_det_jacobian(y)
            close(ldi, -orc.transform_ldj(spec, x), atol=1e-4)
        # a flow of them: not fusable as one program -> per-layer loop
        flow = st.NormalizingFlow(st.UnitNormal(dim), [f]).to(DEV)
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob([spec], x), rtol=1e-5, atol=1e-4)


def test_full_size_properties_spline_and_linear_flows():
    """BASELINE cfg 3 and cfg 4 at their full N = 2^20 rows: round trips, forward/inverse log-det antisymmetry,
    batch-split invariance, agreement with the oracle on a slice."""
    torch.set_grad_enabled(False)        # inference properties: with a graph, spline flows take the layer-wise training path
    try:
        _full_size_properties_spline_and_linear()
    finally:
        torch.set_grad_enabled(True)


def _full_size_properties_spline_and_linear():
    for name, desc, dim, tol_x, tol_l in [('cfg3', fd.cfg3_desc(), 64, 2e-3, 1e-3), ('cfg4', fd.cfg4_desc(), 128, 2e-2, 5e-3)]:
        torch.manual_seed(0)
        flow = fd.build_flow(st, desc, dim).to(DEV)
        assert flow._fused_program(True, dim, 0, torch.device(DEV)) is not None
        n = 1 << 20
        x = torch.randn(n, dim, device=DEV)
        y, ldj_f = flow.forward_and_log_det_jacobian(x)
        xb, ldj_i = flow.inverse_and_log_det_jacobian(y)
        assert torch.isfinite(y).all() and torch.isfinite(ldj_f).all()
        # cfg 4's default-init matrices amplify by ~1e3 per block (mean log_prob ~ -4e4): compare relatively
        scale = 1.0 + x.abs().max().item()
        assert (xb - x).abs().max().item() < tol_x * scale, name
        # cfg 4: per-layer log-dets of magnitude ~1e2..1e3 cancel to O(1) totals and the inverse pass re-derives them
        # from a round-tripped state that is itself only 1e-3 accurate at this (default) init;
        # splines: 256 log-derivatives per row, each carrying the ~1e-4 knot-difference conditioning noise
        assert ((ldj_f + ldj_i).abs() / (1.0 + ldj_f.abs())).max().item() < tol_l, name
        lp = flow.log_prob(x)
        lp2 = torch.cat([flow.log_prob(x[:100_001]), flow.log_prob(x[100_001:])])
        assert torch.equal(lp, lp2), name
        spec = fd.flow_spec(desc, {k: v.cpu() for k, v in flow.state_dict().items()})
        sl = slice(900_000, 900_256)
        close(lp[sl], orc.flow_log_prob(spec, x[sl].cpu()), rtol=1e-5, atol=1e-4)
        s = flow.log_prob_sum(x)
        assert abs(s.item() - lp.double().sum().item()) <= 1e-9 * abs(s.item()), name


def test_cfg5_shard_size_batch_int64_offsets():
    """2^23 rows (the whole cfg-5 batch on one GPU, 1 GiB of bf16 x): 64-bit row offsets, persistent grid-stride;
    bit-identical to evaluating the eight 2^20-row shards separately, and the fp64 batch sum matches."""
    torch.manual_seed(0)
    flow = fd.build_flow(st, fd.cfg2_desc(), 64).to(DEV)
    n = 1 << 23
    x = torch.randn(n, 64, device=DEV).bfloat16()
    with torch.no_grad():
        lp = flow.log_prob(x)
        assert lp.shape == (n, 1) and torch.isfinite(lp).all()
        parts = torch.cat([flow.log_prob(x[i << 20:(i + 1) << 20]) for i in range(8)])
        assert torch.equal(lp, parts)
        s = flow.log_prob_sum(x)
        assert abs(s.item() - lp.double().sum().item()) <= 1e-9 * abs(s.item())


def test_suite_shapes_cubic_and_coupling_cubic():
    """stribor/test/test_spline.py:8-33 cubic rows (the reference's default spline_type: n_bins 1/3/10, latent
    0/1/13, box [0,2]) and cubic couplings with some elements in the linear tails (fixture F9)."""
    g = Golden('f9_cubic')
    n = 0
    for case in g.cases('cubic/') + g.cases('coupling_cubic/'):
        f = product_transform(g, case)
        x = g.t(case + '/x').to(DEV)
        kw = {'latent': g.t(case + '/latent').to(DEV)} if g.has(case + '/latent') else {}
        y = f(x, **kw)
        close(y, g.t(case + '/y'))
        close(f.inverse(y, **kw), g.t(case + '/x'), atol=1e-4)                 # base.py:8-11
        close(f.inverse(g.t(case + '/y').to(DEV), **kw), g.t(case + '/x_back'), atol=2e-5)
        ldj = f.log_det_jacobian(x, y, **kw)
        close(ldj, g.t(case + '/ldj'), atol=1e-4)                              # the suite's own atol (base.py:22)
        _, l1 = f.forward_and_log_det_jacobian(x, **kw)
        _, l2 = f.inverse_and_log_det_jacobian(y, **kw)
        close(l1, g.t(case + '/ldj'), atol=1e-4)
        close(-l2, g.t(case + '/ldj'), atol=1e-4)                              # base.py:14-22
        close(ldj.reshape(-1), g.t(case + '/autograd_logdet'), atol=1e-4)      # base.py:35-44
        if g.has(case + '/ldiag'):
            close(f.log_diag_jacobian(x, y, **kw), g.t(case + '/ldiag'), atol=1e-4)
        n += 1
    assert n == 4 * 9 + 4 * 2


def test_cubic_spline_flow_against_golden():
    """2 cubic-spline couplings at the cfg-3 widths (D=64, K=16, H=64, box [-3,3]) incl. rows in the tails,
    exactly on the bounds and on a grid through the knots (fixture F9); runs layer by layer (MLP program + kernel)."""
    g = Golden('f9_cubic')
    flow = product_flow(g, 'cubic_flow')
    x = g.t('cubic_flow/x').to(DEV)
    # Rows 0, 1 and 6 hold elements EXACTLY on a domain bound: there the log-derivative jumps (in-domain value vs the
    # linear tails' 0) and the reference decides by re-evaluating the FORWARD spline at the inverted point
    # (Transform.inverse_and_log_det_jacobian, flow.py:42-47), i.e. by the LAST BIT of its own inverse (3.0 vs 3.0000002).
    # The kernel follows the same rule (sx_cubic_coupling reverse = 2 re-evaluates the forward log-derivative at ITS inverted
    # point), but its cubic solve differs from torch's in the last ulp, so on these measure-zero inputs individual elements
    # fall on the other side of the jump: such rows are compared on the transformed values, and their log-dets must be a
    # sum of per-element values that are each either the in-domain one or the tails' exact 0 (checked through ldiag below).
    ok = torch.ones(x.shape[0], dtype=torch.bool)
    ok[[0, 1, 6]] = False
    cur = x
    for i in reversed(range(len(flow.transforms))):
        nxt, ldj = flow.transforms[i].inverse_and_log_det_jacobian(cur)
        close(nxt, g.t(f'cubic_flow/inv_x.{i}'), atol=1e-4)      # a cubic solve: flat spots amplify fp32 rounding (as in the reference)
        close(ldj.cpu()[ok], g.t(f'cubic_flow/inv_ldj.{i}')[ok], rtol=1e-5, atol=2e-4)
        cur = nxt
    close(flow.log_prob(x).cpu()[ok], g.t('cubic_flow/log_prob')[ok], rtol=1e-5, atol=2e-4)
    close(flow.forward(x), g.t('cubic_flow/forward'), atol=2e-5)
    close(flow.inverse(x), g.t('cubic_flow/inverse'), atol=1e-4)
    close(flow.log_prob(x).double().cpu()[ok], g.t('cubic_flow/log_prob_f64')[ok], rtol=1e-5, atol=2e-4)
    yf, ldf = flow.forward_and_log_det_jacobian(x)
    close(ldf.cpu()[ok], g.t('cubic_flow/forward_ldj')[ok], rtol=1e-5, atol=2e-4)
    # on-bound rows, element by element, in the last layer the inverse pass visits first: every element's log-derivative is
    # either (minus) the forward log-derivative at the inverted point or the tails' exact 0
    from stribor_amd.flows.spline import run_cubic_kernel
    cpl = flow.transforms[-1]
    sp = cpl.transform
    progs, live_idx, live_start, n_live, width = cpl._spline_program(64, 0, x.device)
    params = torch.empty(x.shape[0], width, dtype=torch.float32, device=x.device)
    for p_ in progs:
        p_.run(x, None, mlp_out=params)
    xin, _, ld_ref_mode = run_cubic_kernel(x, params, params.stride(0), live_idx, live_start, n_live, sp.n_bins, sp.lower, sp.upper,
                                           2, False, True)
    _, _, ld_fwd = run_cubic_kernel(xin, params, params.stride(0), live_idx, live_start, n_live, sp.n_bins, sp.lower, sp.upper,
                                    False, False, True)
    a, b = ld_ref_mode.cpu(), -ld_fwd.cpu()
    assert ((a - b).abs() <= 1e-6).all()                       # reference mode = minus the forward log-derivative at the inverted point


def test_cubic_kernel_against_oracle_random_params():
    """sx_cubic_coupling vs the oracle on random parameters: both directions, bf16 storage, non-contiguous live
    columns, tails; errors against the fp64 truth bounded by small multiples of the reference's own fp32 error."""
    from stribor_amd.flows.spline import run_cubic_kernel
    torch.manual_seed(5)
    for (n, d, K, lo, hi) in [(1000, 7, 5, 0., 1.), (513, 64, 16, -3., 3.), (200, 3, 1, 0., 2.), (300, 10, 32, -1., 4.)]:
        x = torch.rand(n, d) * (hi - lo) * 1.2 + lo - 0.1 * (hi - lo)
        uw, uh, ud = torch.randn(n, d, K) * 1.5, torch.randn(n, d, K) * 1.5, torch.randn(n, d, 2) * 1.5
        params = torch.cat([uw, uh, ud], -1).reshape(n, d * (2 * K + 2)).to(DEV)
        for rev in (False, True):
            want, wl = orc.cubic_unconstrained(x, uw, uh, ud, rev, lo, hi)
            t64, tl64 = orc.cubic_unconstrained(x.double(), uw.double(), uh.double(), ud.double(), rev, lo, hi)
            y, ldj, ldiag = run_cubic_kernel(x.to(DEV), params, params.stride(0), None, 0, d, K, lo, hi, rev, True, True)
            ey, eref = (y.cpu().double() - t64).abs(), (want.double() - t64).abs()
            # (the inverse is a cubic solve: ill-conditioned near flat spots, in the reference's fp32 as much as here)
            assert ey.max().item() <= 4 * eref.max().item() + 2e-5, (n, d, K, rev, ey.max().item(), eref.max().item())
            assert torch.quantile(ey.flatten(), 0.999).item() <= 2 * torch.quantile(eref.flatten(), 0.999).item() + 1e-5
            el, elref = (ldiag.cpu().double() - tl64).abs(), (wl.double() - tl64).abs()
            # the log-derivative log(3at^2 + 2bt + c) cancels badly in steep / flat bins (random parameters make plenty):
            # bound the error distribution against the reference's own fp32 error rather than element by element
            assert el.max().item() <= 4 * elref.max().item() + 1e-4, (n, d, K, rev, el.max().item(), elref.max().item())
            assert torch.quantile(el.flatten(), 0.999).item() <= 3 * torch.quantile(elref.flatten(), 0.999).item() + 2e-5
            close(ldj, ldiag.sum(-1), atol=1e-4 * d)
    # bf16 storage + scattered live columns
    n, d, K = 257, 12, 8
    live = torch.tensor([1, 4, 5, 9, 11], dtype=torch.int32)
    xb = (torch.rand(n, d) * 1.4 - 0.2).to(torch.bfloat16)
    uw, uh, ud = torch.randn(n, 5, K), torch.randn(n, 5, K), torch.randn(n, 5, 2)
    params = torch.cat([uw, uh, ud], -1).reshape(n, 5 * (2 * K + 2)).to(DEV)
    y, ldj, _ = run_cubic_kernel(xb.to(DEV), params, params.stride(0), live.to(DEV), 0, 5, K, 0., 1., False, True, False)
    want, wl = orc.cubic_unconstrained(xb.float()[:, live.long()], uw, uh, ud, False, 0., 1.)
    full = xb.float().clone()
    full[:, live.long()] = want
    close(y.float(), full.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-2)
    close(ldj, wl.sum(-1), atol=1e-4)
    with pytest.raises(ValueError):
        st.Spline(2, 101, spline_type='cubic').to(DEV)(torch.rand(3, 2, device=DEV))


def test_pointwise_flows_suite_shapes_and_stack():
    """stribor/test/test_sigmoid.py, test_activations.py, test_cumsum.py shapes (fixture F10): Sigmoid / Logit (incl.
    saturated and clamped values), ELU, LeakyReLU, Cumsum / Diff (bit-exact: sequential order), Identity; then the
    on-path part of test_normalizing_flow.py's stack, layer by layer on the device."""
    g = Golden('f10_pointwise')
    n = 0
    for case, m in g.meta.items():
        if case == 'stack':
            continue
        kind = m['desc'][0]['kind']
        f = product_transform(g, case)
        x = g.t(case + '/x').to(DEV)
        y = f(x)
        exact = kind in ('cumsum', 'diff', 'identity', 'leaky_relu')
        if exact:
            assert torch.equal(y.cpu(), g.t(case + '/y')), case                  # test_cumsum.py:17 asserts equality
            assert torch.equal(f.inverse(y).cpu(), g.t(case + '/x_back')), case
        else:
            close(y, g.t(case + '/y'))
            close(f.inverse(g.t(case + '/y').to(DEV)), g.t(case + '/x_back'), atol=2e-5)
        ldj = f.log_det_jacobian(x, y)
        close(ldj, g.t(case + '/ldj'), atol=2e-5)
        _, l1 = f.forward_and_log_det_jacobian(x)
        _, l2 = f.inverse_and_log_det_jacobian(g.t(case + '/y').to(DEV))
        close(l1, g.t(case + '/ldj_fwd'), atol=2e-5)
        close(l2, g.t(case + '/ldj_inv'), atol=1e-4 if kind in ('sigmoid', 'logit') else 2e-5)
        close(f.log_diag_jacobian(x, y), g.t(case + '/ldiag'), atol=2e-5)
        if kind not in ('sigmoid', 'logit') or x.numel() <= 4:                   # saturated entries have no finite autograd value
            close(ldj.reshape(-1), g.t(case + '/autograd_logdet'), atol=1e-4)    # base.py:35-44
        n += 1
    assert n == 6 * 7
    # bf16 storage
    xb = torch.randn(33, 10).to(torch.bfloat16)
    yb = st.ELU().to(DEV)(xb.to(DEV))
    close(yb.float(), torch.nn.functional.elu(xb.float()).to(torch.bfloat16).float(), rtol=1e-2, atol=1e-2)
    flow = product_flow(g, 'stack')
    x = g.t('stack/x').to(DEV)
    close(flow.log_prob(x), g.t('stack/log_prob'), rtol=1e-5, atol=1e-4)
    close(flow.forward(x), g.t('stack/forward'), atol=2e-5)
    close(flow.inverse(x), g.t('stack/inverse'), atol=2e-5)
    close(flow.log_prob(x).double(), g.t('stack/log_prob_f64'), rtol=1e-5, atol=1e-4)
    with pytest.raises(AssertionError):
        st.LeakyReLU(-1.0)
    with pytest.raises(AssertionError):
        st.Cumsum(-2)


def test_pointwise_vec4_path_against_oracle():
    """Widths divisible by 4 take the 4-elements-per-lane kernel (row sums by shuffles for 64 = 16 lanes, by atomics for
    12 = 3 lanes); ragged row counts; fp32 and bf16 storage."""
    torch.manual_seed(9)
    for n, d in [(257, 64), (100, 12), (1000, 128), (5, 4)]:
        for kind, slope in [('sigmoid', None), ('logit', None), ('elu', None), ('leaky_relu', 0.2)]:
            x = torch.rand(n, d) * 0.9 + 0.05 if kind == 'logit' else torch.randn(n, d) * 2
            spec = {'kind': kind}
            f = {'sigmoid': st.Sigmoid, 'logit': st.Logit, 'elu': st.ELU}.get(kind, lambda: st.LeakyReLU(slope))().to(DEV)
            if slope:
                spec['negative_slope'] = slope
            y, ldj = f.forward_and_log_det_jacobian(x.to(DEV))
            close(y, orc.transform_apply(spec, x, False))
            close(ldj, orc.transform_ldj(spec, x), rtol=1e-5, atol=1e-5 * d)
            close(f.log_diag_jacobian(x.to(DEV), y), orc.pointwise_log_diag(spec, x), atol=2e-5)
            xb, l2 = f.inverse_and_log_det_jacobian(y)
            close(xb, x, rtol=1e-4, atol=2e-4)
            close(l2, -ldj, rtol=1e-4, atol=2e-4 * d)
    xb16 = torch.randn(129, 64).to(torch.bfloat16)
    yb = st.Sigmoid().to(DEV)(xb16.to(DEV))
    close(yb.float(), torch.sigmoid(xb16.float()).to(torch.bfloat16).float(), rtol=1e-2, atol=1e-2)


def test_pure_coupling_flow_d128_against_oracle():
    """D = 128 RealNVP (pure split-coupling program at 4 tiles: 8-wave workgroups, 2 waves per SIMD) vs the oracle,
    both directions."""
    torch.manual_seed(3)
    desc = fd.cfg2_desc(6, 128, 64)
    flow = fd.build_flow(st, desc, 128)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    x = torch.randn(1000, 128)
    with torch.no_grad():
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=1e-4)
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
        wy, wl = orc.flow_forward_and_ldj(spec, x)
        close(y, wy, rtol=1e-5, atol=2e-5)
        close(ldj, wl, rtol=1e-5, atol=1e-4)
        close(flow.inverse(y), x, rtol=1e-4, atol=1e-4)


def test_continuous_affine_coupling_and_neural_flow():
    """stribor/test/test_coupling.py:29-52 shapes (latent 0/1/13, TimeLinear) plus TimeIdentity / TimeTanh / TimeLog, and
    test_neural_flow.py's container checks (identity at t = 0, t = t0 round trip) -- fixture F11."""
    g = Golden('f11_continuous')
    n = 0
    for case, m in g.meta.items():
        if case == 'neural_flow':
            continue
        f = product_transform(g, case)
        x, t = g.t(case + '/x').to(DEV), g.t(case + '/t').to(DEV)
        kw = {'latent': g.t(case + '/latent').to(DEV)} if g.has(case + '/latent') else {}
        # round 3: conditioner + time embedding + affine map + log-det are ONE fused step (SX_STEP_COUPLING_TIME)
        assert f._fused_program(False, 1.0, x.shape[-1], kw['latent'].shape[-1] if kw else 0, x.device) is not None, case
        y, ldj = f.forward_and_log_det_jacobian(x, t, **kw)
        close(y, g.t(case + '/y'))
        close(ldj, g.t(case + '/ldj'), atol=2e-5)
        close(f(x, t, **kw), g.t(case + '/y'))
        close(f.log_det_jacobian(x, y, t=t, **kw), g.t(case + '/ldj'), atol=2e-5)
        xb, li = f.inverse_and_log_det_jacobian(g.t(case + '/y').to(DEV), t, **kw)
        close(xb, g.t(case + '/x_back'), atol=2e-5)
        close(li, g.t(case + '/ldj_inv'), atol=2e-5)
        close(f.inverse(y, t, **kw), x, atol=1e-4)                              # base.py:8-11
        n += 1
    assert n == 4 * 6
    m = g.meta['neural_flow']
    nf = st.NeuralFlow([fd.build_transform(st, d) for d in m['desc']])
    nf.load_state_dict(g.state('neural_flow'))
    nf = nf.to(DEV)
    x, t, t0 = (g.t('neural_flow/' + k).to(DEV) for k in ('x', 't', 't0'))
    assert nf._fused(x.shape[-1], 0, True, x.device) is not None and nf._fused(x.shape[-1], 0, False, x.device) is not None
    close(nf(x, t=t), g.t('neural_flow/y_t'))
    close(nf(x, t=t, t0=t0), g.t('neural_flow/y_t_t0'), atol=2e-5)
    assert torch.equal(nf(x, t=torch.zeros_like(t)), x)                          # test_neural_flow.py:24-27
    close(nf(x, t=t0, t0=t0), x, atol=1e-5)                                      # :29-32
    tf = st.net.TimeFourier(4, 8)                                                  # round 2: built (time_net.py:49-91)
    assert tf(torch.rand(5, 1)).shape == (5, 4)


def test_full_size_properties_cubic_and_pointwise():
    """Size-independent properties at BASELINE-scale batches for the widened rows: a D=64, K=16 cubic-spline coupling flow
    at 2^18 rows (round trip, forward / inverse log-det antisymmetry, batch-split invariance, oracle on a slice) and the
    point-wise kernels at 2^20 x 64 (round trips, log-det antisymmetry, Cumsum / Diff exactness against torch)."""
    torch.manual_seed(0)
    desc = [dict(d, spline_type='cubic') for d in fd.cfg3_desc(4)]
    flow = fd.build_flow(st, desc, 64).to(DEV)
    n = 1 << 18
    x = torch.randn(n, 64, device=DEV)
    y, ldj_f = flow.forward_and_log_det_jacobian(x)
    xb, ldj_i = flow.inverse_and_log_det_jacobian(y)
    assert torch.isfinite(y).all() and torch.isfinite(ldj_f).all()
    assert torch.isfinite(xb).all() and torch.isfinite(ldj_i).all()
    err = (xb - x).abs()
    # four cubic solves in a row.  The reference's fp32 closed forms lose up to 5e-3 of a bin where the cubic degenerates (and
    # f'(t) then rounds to <= 0: a handful of NaN / -1e8 rows per million in fp32); the kernel ends the solve in Newton steps on
    # the bin's cubic, so none of that is left (measured: max 1.7e-5, 99.9 % quantile 4e-6, log-det antisymmetry max 5e-5)
    assert torch.quantile(err.flatten()[:: 97], 0.999).item() < 2e-5 and err.max().item() < 2e-4
    rel = ((ldj_f + ldj_i).abs() / (1.0 + ldj_f.abs())).flatten()
    assert rel.max().item() < 5e-4
    lp = flow.log_prob(x)
    assert torch.isfinite(lp).all()                     # round 1 tolerated 1e-4 of the rows being NaN here
    lp2 = torch.cat([flow.log_prob(x[:100_001]), flow.log_prob(x[100_001:])])
    assert torch.equal(lp, lp2)
    spec = fd.flow_spec(desc, {k: v.cpu() for k, v in flow.state_dict().items()})
    sl = slice(123_000, 123_128)
    close(lp[sl], orc.flow_log_prob(spec, x[sl].cpu()), rtol=1e-5, atol=1e-4)
    # point-wise kernels
    n = 1 << 20
    x = torch.randn(n, 64, device=DEV) * 2
    for f in (st.Sigmoid(), st.ELU(), st.LeakyReLU(0.05)):
        f = f.to(DEV)
        y, l1 = f.forward_and_log_det_jacobian(x)
        xb, l2 = f.inverse_and_log_det_jacobian(y)
        ok = (y > 1e-6) & (y < 1 - 1e-6) if isinstance(f, st.Sigmoid) else torch.ones_like(y, dtype=torch.bool)
        assert ((xb - x).abs()[ok] <= 1e-3 * (1 + x.abs()[ok])).all(), type(f).__name__
        rows_ok = ok.all(-1)
        assert ((l1 + l2).abs().reshape(-1)[rows_ok] <= 1e-3 * (1 + l1.abs().reshape(-1)[rows_ok])).all(), type(f).__name__
    c = st.Cumsum(-1).to(DEV)
    yc = c(x)
    assert torch.equal(yc.cpu()[:4096], x.cpu()[:4096].cumsum(-1))               # test_cumsum.py:17 on the first rows
    close(c.inverse(yc), x, rtol=1e-5, atol=1e-4)


def test_dynamic_chunk_handout_equals_static_and_is_stream_safe():
    """The ticket-based chunk hand-out (per-stream counters re-armed by the last workgroup) gives bit-identical per-row
    results to the static stride for ragged batch sizes, launch after launch, and on two streams at once."""
    torch.manual_seed(5)
    flow = fd.build_flow(st, fd.cfg2_desc(4, 64, 64), 64).to(DEV)
    sizes = [1, 127, 128 * 2048 + 1, 300_007, 1 << 19, (1 << 19) + 255, 777_777]
    for n in sizes:
        x = torch.randn(n, 64, device=DEV)
        a = flow.log_prob(x)
        b = flow.log_prob(x)                       # second launch on the same stream: counters were re-armed
        os.environ['SX_STATIC_CHUNKS'] = '1'
        try:
            c = flow.log_prob(x)
        finally:
            del os.environ['SX_STATIC_CHUNKS']
        assert torch.equal(a, b) and torch.equal(a, c), n
        s = flow.log_prob_sum(x)
        assert abs(s.item() - a.double().sum().item()) <= 1e-9 * abs(s.item()) + 1e-6, n
    x1, x2 = torch.randn(400_000, 64, device=DEV), torch.randn(500_003, 64, device=DEV)
    want1, want2 = flow.log_prob(x1), flow.log_prob(x2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(5):
        with torch.cuda.stream(s1):
            o1 = flow.log_prob(x1)
        with torch.cuda.stream(s2):
            o2 = flow.log_prob(x2)
        outs.append((o1, o2))
    torch.cuda.synchronize()
    for o1, o2 in outs:
        assert torch.equal(o1, want1) and torch.equal(o2, want2)


@pytest.mark.parametrize('dim,hidden,masks', [
    (33, 8, ('ordered_right_half', 'ordered_left_half')), (40, 33, ('ordered_left_half', 'ordered_right_half')),
    (48, 64, ('parity_even', 'parity_odd')), (63, 100, ('ordered_right_half', 'parity_odd')),
    (64, 128, ('ordered_right_half', 'ordered_left_half')), (65, 64, ('ordered_right_half', 'ordered_left_half')),
    (96, 32, ('parity_odd', 'parity_even')), (100, 64, ('ordered_left_half', 'ordered_right_half')),
    (128, 128, ('ordered_right_half', 'ordered_left_half')),
])
def test_affine_coupling_flows_across_widths_against_oracle(dim, hidden, masks):
    """Tile-geometry sweep of the fused affine-coupling kernels (1 / 2 / 4 state tiles, 1 / 2 / 4 hidden tiles, pruned
    and dense mask layouts, pure and general modes, ragged row counts) against the oracle, both directions."""
    torch.manual_seed(dim * 131 + hidden)
    desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': masks[i % 2], 'latent_dim': 0} for i in range(5)]
    flow = fd.build_flow(st, desc, dim)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    for n in (1, 257, 3000):
        x = torch.randn(n, dim)
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=1e-4)
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
        wy, wl = orc.flow_forward_and_ldj(spec, x)
        close(y, wy, rtol=1e-5, atol=2e-5)
        close(ldj, wl, rtol=1e-5, atol=1e-4)
        close(flow.inverse(y), x, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dim,hidden,K,masks', [
    (33, 8, 3, ('ordered_right_half', 'ordered_left_half')), (48, 64, 8, ('parity_even', 'parity_odd')),
    (64, 128, 16, ('ordered_right_half', 'ordered_left_half')), (100, 64, 5, ('ordered_left_half', 'ordered_right_half')),
    (128, 32, 16, ('ordered_right_half', 'parity_odd')), (20, 16, 24, ('ordered_right_half', 'ordered_left_half')),
])
def test_spline_coupling_flows_across_widths_against_oracle(dim, hidden, K, masks):
    """Tile-geometry sweep of the spline-coupling path (fused programs where they fit -- K <= 16 --, layer by layer
    otherwise) against the oracle, both directions, inputs reaching into the tails."""
    torch.manual_seed(dim * 17 + K)
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -2.5, 'upper': 2.5,
             'mask': masks[i % 2], 'latent_dim': 0} for i in range(3)]
    flow = fd.build_flow(st, desc, dim)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    for n in (1, 300):
        x = torch.randn(n, dim) * 1.5
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=2e-4 * max(1, dim // 32))
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
        wy, wl = orc.flow_forward_and_ldj(spec, x)
        close(y, wy, rtol=1e-5, atol=2e-5)
        close(ldj, wl, rtol=1e-5, atol=2e-4 * max(1, dim // 32))
        close(flow.inverse(y), x, rtol=1e-4, atol=1e-4)
    from stribor_amd.flows.spline import check_errors
    check_errors()


@pytest.mark.parametrize('stype', ['quadratic', 'cubic'])
@pytest.mark.parametrize('K', [1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 15, 16])
def test_fused_spline_couplings_of_any_bin_count_up_to_16(K, stype):
    """Round 6: every K <= 16 runs the straight-line spline phases of the one-launch tier (sx_flow_spline.h: rqs16_c / cub16_c,
    `GEN`) -- K = 8 used to be 1.6 x slower than K = 16 because only sixteen bins had them.  The unused registers of an element's
    tile are parked by the packer (logits of bins >= K at -1e30, derivative rows >= K - 1 at the boundary constant), knots of
    index >= K are never compared and the last bin's right knot is the bound itself.  Rows exactly ON both bounds, one ulp inside
    and outside them, in the last bin and in the tails; both directions; against the oracle (rational_quadratic_spline.py:161-251,
    cubic_spline.py:71-247, search_sorted.py:3-5)."""
    torch.manual_seed(100 * K + len(stype))
    dim, lo, hi = 24, -2.0, 2.0
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [32], 'n_bins': K, 'lower': lo, 'upper': hi, 'spline_type': stype,
             'mask': ('ordered_right_half', 'ordered_left_half', 'parity_odd')[i % 3], 'latent_dim': 0} for i in range(3)]
    flow = fd.build_flow(st, desc, dim)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    assert flow._fused_program(True, dim, 0, torch.device(DEV)) is not None
    x = torch.randn(400, dim) * 1.2
    edge = torch.tensor([lo, hi, float(np.nextafter(np.float32(lo), np.float32(0))), float(np.nextafter(np.float32(hi), np.float32(0))),
                         float(np.nextafter(np.float32(lo), np.float32(-9))), float(np.nextafter(np.float32(hi), np.float32(9))),
                         hi - 1e-3, hi - 0.05, lo + 1e-3, 2.5, -2.5, 0.0])
    if stype == 'cubic':
        # exactly ON a bound and one ulp inside it the reference's own cubic evaluation is ill-conditioned (its fp32 and fp64 values are
        # 0.5 .. 1.6 apart there, and the product sits between them: `tools/experiments/dbg_cubic_edges.py`): those four values are
        # checked for finiteness only, below
        on_bound = edge[:4].clone()
        edge = edge[4:]
    for r in range(len(edge)):          # rows of one edge value, and rows that mix them with ordinary entries
        x[r] = edge[r]
        x[20 + r, ::2] = edge[r]
    x[40, :len(edge) * 2] = edge.repeat(2)
    spec64 = orc.spec_to(spec, torch.float64)
    atol_l = 2e-4 if stype == 'quadratic' else 5e-4
    atol_y = 2e-5 if stype == 'quadratic' else 2e-4

    def near(got, w32, w64, atol):
        """|got - want| <= atol + 1e-5 |want| against the reference's fp32 values OR its fp64 ones, element by element: on and one
        ulp inside the bounds the reference's own fp32 cubic evaluation is 0.5 .. 1.6 away from its fp64 one (the inverse solve at a
        domain bound; `tools/experiments/dbg_cubic_edges.py`: both product tiers sit on the fp64 value there)."""
        got = got.detach().double().cpu()
        e32, e64 = (got - w32.double()).abs(), (got - w64.double()).abs()
        ok = (e32 <= atol + 1e-5 * w32.double().abs()) | (e64 <= atol + 1e-5 * w64.abs())
        assert ok.all(), (K, stype, torch.minimum(e32, e64)[~ok].max().item(), (~ok).nonzero()[:4].tolist())

    lp = flow.log_prob(x.to(DEV))
    assert torch.isfinite(lp).all()
    near(lp, orc.flow_log_prob(spec, x), orc.flow_log_prob(spec64, x.double()), atol_l)
    y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
    wy, wl = orc.flow_forward_and_ldj(spec, x)
    wy64, wl64 = orc.flow_forward_and_ldj(spec64, x.double())
    near(y, wy, wy64, atol_y)
    near(ldj, wl, wl64, atol_l)
    xi, li = flow.inverse_and_log_det_jacobian(x.to(DEV))
    wx, wli = orc.flow_inverse_and_ldj(spec, x)
    wx64, wli64 = orc.flow_inverse_and_ldj(spec64, x.double())
    near(xi, wx, wx64, atol_y)
    near(li, wli, wli64, atol_l)
    if stype == 'cubic':
        xb = torch.randn(8, dim)
        for r in range(4):
            xb[r] = on_bound[r]
            xb[4 + r, 1::2] = on_bound[r]
        assert torch.isfinite(flow.log_prob(xb.to(DEV))).all() and torch.isfinite(flow.forward(xb.to(DEV))).all()
    from stribor_amd.flows.spline import check_errors
    check_errors()


def test_deep_conditioner_flows_fuse_into_one_launch():
    """Flows whose couplings have 2 or 3 hidden layers (different widths) plan into ONE fused program (kernel MODE 9)
    and match the oracle in both directions; mixed with single-hidden-layer couplings and a Flip."""
    torch.manual_seed(8)
    for dim, hiddens in [(64, ([64, 64], [64], [32, 64], [64, 48, 64])), (10, ([13, 7], [16, 16, 16])), (128, ([64, 64], [32]))]:
        desc = []
        for i, h in enumerate(hiddens * 2):
            # (D = 128 keeps to tile-aligned masks: a DENSE deep step there exceeds the LDS ring and would un-fuse the flow)
            masks = ('ordered_right_half', 'ordered_left_half', 'parity_odd') if dim < 128 else ('ordered_right_half', 'ordered_left_half')
            desc.append({'kind': 'coupling_affine', 'dim': dim, 'hidden': list(h), 'latent_dim': 0, 'mask': masks[i % len(masks)]})
            if i == 2:
                desc.append({'kind': 'flip'})
        flow = fd.build_flow(st, desc, dim)
        spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
        flow = flow.to(DEV)
        assert flow._fused_program(True, dim, 0, torch.device(DEV)) is not None
        assert flow._fused_program(False, dim, 0, torch.device(DEV)) is not None
        for n in (1, 1000):
            x = torch.randn(n, dim)
            close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=1e-4)
            y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
            wy, wl = orc.flow_forward_and_ldj(spec, x)
            close(y, wy, rtol=1e-5, atol=2e-5)
            close(ldj, wl, rtol=1e-5, atol=1e-4)
            close(flow.inverse(y), x, rtol=1e-4, atol=1e-4)


def test_deep_conditioner_spline_flows_fuse_into_one_launch():
    """Spline-coupling flows whose conditioners have 2 or 3 hidden layers run as ONE fused program (the earlier layers'
    activations wait in B-operand form in the registers the spline phases use anyway) and match the oracle."""
    torch.manual_seed(18)
    for dim, K, hiddens in [(64, 16, ([64, 64], [64], [32, 64])), (10, 5, ([13, 7], [16, 16, 16])), (40, 8, ([64, 64, 64],))]:
        desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': list(h), 'n_bins': K, 'lower': -3, 'upper': 3, 'latent_dim': 0,
                 'mask': ('ordered_right_half', 'ordered_left_half', 'parity_odd')[i % 3]} for i, h in enumerate(hiddens * 2)]
        flow = fd.build_flow(st, desc, dim)
        spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
        flow = flow.to(DEV)
        assert flow._fused_program(True, dim, 0, torch.device(DEV)) is not None
        assert flow._fused_program(False, dim, 0, torch.device(DEV)) is not None
        for n in (1, 500):
            x = torch.randn(n, dim) * 1.6
            close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=2e-4 * max(1, dim // 32))
            y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
            wy, wl = orc.flow_forward_and_ldj(spec, x)
            close(y, wy, rtol=1e-5, atol=2e-5)
            close(ldj, wl, rtol=1e-5, atol=2e-4 * max(1, dim // 32))
            close(flow.inverse(y), x, rtol=1e-4, atol=1e-4)
    from stribor_amd.flows.spline import check_errors
    check_errors()


@pytest.mark.parametrize('make,dim,n', [('cfg2', 64, 1024), ('cfg2', 64, 300_007), ('cfg3', 64, 2048), ('cfg4', 128, 4096),
                                         ('rqs160', 64, 1500), ('cubic160', 40, 700)])
def test_log_prob_replays_from_a_hip_graph(make, dim, n):
    """Launch-bound small batches: log_prob captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed on new
    input contents gives bit-identical results to eager launches -- no allocation, synchronisation or host-side state in
    the launch path (the chunk counters are created by the warm-up on the capture stream and re-armed by the kernel)."""
    torch.manual_seed(11)
    desc = {'cfg2': lambda: fd.cfg2_desc(8, dim, 64), 'cfg3': lambda: fd.cfg3_desc(4, dim, 64, 16),
            'cfg4': lambda: fd.cfg4_desc(2, dim, 64),
            # (the slab forward tier: two kernels + a reduction per layer, buffers from torch's allocator and the library scratch)
            'rqs160': lambda: [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [160], 'n_bins': 16, 'lower': -3, 'upper': 3, 'latent_dim': 0,
                                'mask': m} for m in ('ordered_right_half', 'ordered_left_half')],
            'cubic160': lambda: [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [200], 'n_bins': 9, 'lower': -3, 'upper': 3, 'latent_dim': 0,
                                  'mask': m, 'spline_type': 'cubic'} for m in ('parity_even', 'ordered_left_half')]}[make]()
    flow = fd.build_flow(st, desc, dim).to(DEV)
    static_x = torch.randn(n, dim, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            flow.log_prob(static_x)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        static_out = flow.log_prob(static_x)
    for seed in range(3):
        fresh = torch.randn(n, dim, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed)) * (1 + seed)
        static_x.copy_(fresh)
        graph.replay()
        torch.cuda.synchronize()
        got = static_out.clone()
        want = flow.log_prob(fresh)
        assert torch.equal(got, want), (make, seed)


@pytest.mark.gpu
@pytest.mark.parametrize('second', ['rqs', 'affine'])
def test_deep_affine_coupling_with_latent_in_one_state_tile(second):
    """Found by tools/fuzz_train.py --infer.  (1) A conditional flow (latent) whose affine coupling has a two-hidden-layer
    conditioner and fits one state tile (dim 10 + latent 3: x_tiles = 1, tiles = 2) plans the DENSE deep step over both tiles;
    the launcher used to reject it ('bad transformed tiles').  (2) With a quadratic-spline coupling in the same flow the
    planner used to put both in one program, whose spline kernel variant has no deep-affine arm: that layer was skipped
    silently (log_prob off by 0.1-0.3 relative, round trips still consistent).  Such flows now run layer by layer and the
    launcher refuses the mix.  Values against the oracle in fp64."""
    torch.manual_seed(5)
    dim, latent = 10, 3
    desc = [{'dim': dim, 'hidden': [48, 51], 'mask': 'ordered_left_half', 'latent_dim': latent, 'kind': 'coupling_affine'},
            {'kind': 'flip'}]
    if second == 'rqs':
        desc.append({'dim': dim, 'hidden': [36, 13], 'mask': 'ordered_right_half', 'latent_dim': latent, 'kind': 'coupling_rqs',
                     'n_bins': 1, 'lower': -3.0, 'upper': 3.0, 'spline_type': 'quadratic'})
    else:
        desc.append({'dim': dim, 'hidden': [36, 13], 'mask': 'ordered_right_half', 'latent_dim': latent, 'kind': 'coupling_affine'})
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x, lat = torch.randn(519, dim) * 1.4, torch.randn(519, latent)
    spec = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
    want = orc.flow_log_prob(spec, x.double(), lat.double())
    wy, wl = orc.flow_forward_and_ldj(spec, x.double(), lat.double())
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV), latent=lat.to(DEV))
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV), latent=lat.to(DEV))
        xr = flow.inverse(y, latent=lat.to(DEV))
    close(lp, want.float(), rtol=1e-5, atol=1e-4)
    close(y, wy.float(), rtol=1e-5, atol=1e-5)
    close(ldj, wl.float(), rtol=1e-5, atol=1e-4)
    close(xr, x, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dim,K,hidden,latent,masks', [
    (64, 16, [64], 0, ('ordered_right_half', 'ordered_left_half')),           # the cfg-3 shape with the reference's default spline
    (37, 7, [40, 24], 3, ('parity_even', 'ordered_left_half', 'parity_odd')),  # run-time K, deep conditioner (MODE 13), latent
    (5, 1, [16], 0, ('ordered_right_half', 'parity_odd')),                     # one bin: both knot derivatives are the boundary ones
])
def test_fused_cubic_spline_flow_matches_oracle_and_the_unfused_path(dim, K, hidden, latent, masks, monkeypatch):
    """Cubic-spline coupling flows run as ONE fused program (kernel MODE 12 / 13, round 2): values against the fp64 oracle in
    all three directions, and against the layer-by-layer path (MLP program -> [N, D(2K+2)] parameters -> cubic_kernel), which
    shares the per-element arithmetic (sx_cubic_core.h) and must agree to rounding."""
    torch.manual_seed(31)
    desc = []
    for i, m in enumerate(masks):
        desc.append({'kind': 'coupling_rqs', 'dim': dim, 'hidden': hidden, 'n_bins': K, 'lower': -3.0, 'upper': 3.0, 'mask': m,
                     'latent_dim': latent, 'spline_type': 'cubic'})
        if i == 0:
            desc.append({'kind': 'flip'})
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    n = 1337
    x = torch.randn(n, dim) * 1.5                   # ~5 % of the elements in the linear tails
    lat = torch.randn(n, latent) if latent else None
    kw = {} if lat is None else {'latent': lat.to(DEV)}
    assert flow._fused_program(True, dim, latent, torch.device(DEV, 0)) is not None          # really one program
    spec = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
    l64 = None if lat is None else lat.double()
    want_lp = orc.flow_log_prob(spec, x.double(), l64)
    wy, wl = orc.flow_forward_and_ldj(spec, x.double(), l64)
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV), **kw)
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV), **kw)
        xr = flow.inverse(y, **kw)
    close(lp, want_lp.float(), rtol=1e-5, atol=1e-4)
    close(y, wy.float(), rtol=1e-5, atol=2e-5)
    close(ldj, wl.float(), rtol=1e-5, atol=1e-4)
    close(xr, x, rtol=1e-5, atol=5e-5)
    monkeypatch.setenv('STRIBOR_CUBIC_UNFUSED', '1')
    flow2 = fd.build_flow(st, desc, dim)
    flow2.load_state_dict(state)
    flow2 = flow2.to(DEV)
    assert flow2._fused_program(True, dim, latent, torch.device(DEV, 0)) is None
    with torch.no_grad():
        lp2 = flow2.log_prob(x.to(DEV), **kw)
        y2, ldj2 = flow2.forward_and_log_det_jacobian(x.to(DEV), **kw)
    close(lp, lp2, rtol=1e-5, atol=1e-4)
    close(y, y2, rtol=1e-5, atol=2e-5)
    close(ldj, ldj2, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('cubic', [False, True])
@pytest.mark.parametrize('bf16,scattered', [(False, False), (True, False), (False, True)])
def test_spline_kernels_pipelined_dense_path_against_oracle(cubic, bf16, scattered):
    """The K = 16 dense path of rqs_kernel / cubic_kernel streams the parameter spans by LDS-DMA one group ahead (round 2): enough
    groups that every wave runs the pipeline several times (16,385 rows x 32 live columns = 8,193 groups over 3,072 waves), a
    ragged last group (leaves the pipeline), bf16 storage, scattered live columns (the prefetched input element goes through
    live_idx) -- both directions against the oracle."""
    from stribor_amd.flows.spline import run_cubic_kernel, run_rqs_kernel
    torch.manual_seed(3)
    n, d, nl, K = 16385, 64, 32, 16
    P = 2 * K + 2 if cubic else 3 * K - 1
    x = torch.randn(n, d) * 1.6
    if bf16:
        x = x.bfloat16().float()
    live = torch.sort(torch.randperm(d)[:nl]).values.to(torch.int32) if scattered else torch.arange(32, 64, dtype=torch.int32)
    params = torch.randn(n, nl * P)
    p3 = params.view(n, nl, P)
    if cubic:
        uw, uh, ud = p3[..., :K], p3[..., K:2 * K], p3[..., 2 * K:]
    else:
        uw, uh, ud = p3[..., :K], p3[..., K:2 * K], p3[..., 2 * K:]
    xin = x.to(DEV).bfloat16() if bf16 else x.to(DEV)
    for rev in (False, True):
        if cubic:
            y, ldj, _ = run_cubic_kernel(xin, params.to(DEV), nl * P, live.to(DEV) if scattered else None, 32, nl, K, -3., 3., rev, True, False)
            want, wl = orc.cubic_unconstrained(x[:, live.long()].double(), uw.double(), uh.double(), ud.double(), rev, -3., 3.)
        else:
            y, ldj, _ = run_rqs_kernel(xin, params.to(DEV), nl * P, live.to(DEV) if scattered else None, 32, nl, K, -3., 3., -3., 3.,
                                       rev, True, False)
            want, wl = orc.rqs_unconstrained(x[:, live.long()].double(), uw.double(), uh.double(), ud.double(), rev, -3., 3.)
        full = x.double().clone()
        full[:, live.long()] = want
        if bf16:
            close(y.float(), full.float().bfloat16().float(), rtol=1e-2, atol=2e-2)
        else:
            # random N(0, 1) logits make bins as narrow as the 1e-3 / 1e-2 floors: bound the distribution, not single elements
            err = (y.cpu().double() - full).abs()
            assert torch.quantile(err.flatten()[::7], 0.999).item() <= 2e-5 and err.max().item() <= 2e-3, (rev, err.max().item())
        el = (ldj.cpu().double() - wl.sum(-1)).abs() / (1.0 + wl.sum(-1).abs())
        assert torch.quantile(el[::3], 0.999).item() <= 1e-4 and el.max().item() <= 5e-3, (rev, el.max().item())


def test_bf16_storage_keeps_fp32_between_layers_of_an_unfused_flow():
    """SURVEY H5 (bf16 in, fp32 arithmetic) for flows that run layer by layer (here: Diff mixes columns, which no fused program
    does): the state stays fp32 between the layers and up to the base density, as it does in the fused kernel's
    registers -- log_prob from bf16 inputs matches the oracle fed the same rounded values at fp32 level (it was 1e-3 off when
    every layer stored bf16: tools/fuzz_train.py --bf16), outputs are rounded to bf16 once."""
    torch.manual_seed(8)
    dim = 24
    desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [32], 'mask': 'ordered_right_half', 'latent_dim': 0},
            {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [32], 'n_bins': 7, 'lower': -3.0, 'upper': 3.0, 'mask': 'parity_odd',
             'latent_dim': 0, 'spline_type': 'quadratic'},
            {'kind': 'leaky_relu', 'negative_slope': 0.2}, {'kind': 'diff'}]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    assert flow._fused_program(True, dim, 0, torch.device(DEV, 0)) is None
    x = (torch.randn(500, dim) * 1.3).bfloat16()
    spec = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV))
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
    assert lp.dtype == torch.float32 and y.dtype == torch.bfloat16
    close(lp, orc.flow_log_prob(spec, x.double()).float(), rtol=1e-5, atol=1e-4)
    wy, wl = orc.flow_forward_and_ldj(spec, x.double())
    close(ldj.float(), wl.float(), rtol=1e-5, atol=1e-4)
    close(y.float(), wy.float().bfloat16().float(), rtol=1e-2, atol=1e-2)


def _mixed_desc(dim, hidden, K, lo, hi):
    """affine coupling -> Flip -> Sigmoid -> cubic coupling on [0, 1] -> Logit -> rq-spline coupling -> LeakyReLU -> affine coupling
    -> ELU -> element-wise affine (the reference's flagship stack, test_normalizing_flow.py:13-35, and then some)."""
    return [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': 'ordered_1', 'latent_dim': 0},
            {'kind': 'flip'},
            {'kind': 'sigmoid'},
            {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'mask': 'ordered_0', 'latent_dim': 0, 'n_bins': K, 'lower': 0,
             'upper': 1, 'spline_type': 'cubic'},
            {'kind': 'logit'},
            {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'mask': 'parity_even', 'latent_dim': 0, 'n_bins': K, 'lower': lo,
             'upper': hi},
            {'kind': 'leaky_relu', 'negative_slope': 0.3},
            {'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': 'parity_odd', 'latent_dim': 0},
            {'kind': 'elu'},
            {'kind': 'affine', 'dim': dim}]


def test_reference_stack_plans_as_one_launch_and_matches_golden():
    """the on-path part of stribor/test/test_normalizing_flow.py:13-35 (affine coupling -> Flip -> Sigmoid ->
    cubic-spline coupling -> Logit; fixture f10 'stack', captured from the reference) is ONE fused program (kernel MODE 14)."""
    g = Golden('f10_pointwise')
    flow = product_flow(g, 'stack')
    x = g.t('stack/x').to(DEV)
    for reverse in (True, False):
        prog = flow._fused_program(reverse, x.shape[-1], 0, x.device)
        assert prog is not None, 'the reference stack must plan as one fused launch'
        kinds = {prog.prog.steps[i].kind for i in range(prog.prog.n_steps)}
        from stribor_amd import _hip
        assert {_hip.STEP_POINTWISE, _hip.STEP_RQS_PHASE} <= kinds and kinds & {_hip.STEP_COUPLING_AFFINE, _hip.STEP_COUPLING_AFFINE_DEEP}
    close(flow.log_prob(x), g.t('stack/log_prob'), rtol=1e-5, atol=1e-4)
    close(flow.forward(x), g.t('stack/forward'), atol=2e-5)
    close(flow.inverse(x), g.t('stack/inverse'), atol=2e-5)
    z, ldj = flow.inverse_and_log_det_jacobian(x)
    cur, acc = x, 0
    for f in reversed(flow.transforms):                     # the same flow layer by layer through HBM (flow.py:118-125)
        cur, l = f.inverse_and_log_det_jacobian(cur)
        acc = acc + l
    close(z, cur, atol=2e-5)
    close(ldj, acc, atol=1e-4)


@pytest.mark.parametrize('dim,hidden,K,n', [(64, 64, 16, 1000), (10, 12, 5, 257), (2, 13, 3, 100), (48, 40, 16, 300)])
def test_mixed_fused_programs_against_oracle(dim, hidden, K, n):
    """Affine couplings, both spline types, Sigmoid / Logit / LeakyReLU / ELU, Flip and an element-wise affine in ONE program
    (kernel MODE 14), both directions, against the oracle; and against the same flow evaluated layer by layer."""
    torch.manual_seed(61)
    desc = _mixed_desc(dim, hidden, K, -4.0, 4.0)
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(0.05 * torch.randn_like(p))
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    spec64 = orc.spec_to(spec, torch.float64)
    # inputs of the inverse pass: points of the flow's range (ELU's inverse is only defined above -1)
    x = orc.flow_forward(spec64, torch.randn(n, dim, dtype=torch.float64) * 0.8).float()
    xd = x.to(DEV)
    assert flow._fused_program(True, dim, 0, xd.device) is not None and flow._fused_program(False, dim, 0, xd.device) is not None
    want_lp, want_lp64 = orc.flow_log_prob(spec, x), orc.flow_log_prob(spec64, x.double())
    got_lp = flow.log_prob(xd)
    close_vs_f64(got_lp, want_lp, want_lp64, k=2.0, rtol=1e-5, atol=1e-4)
    zi, li = flow.inverse_and_log_det_jacobian(xd)
    wz, wl = orc.flow_inverse_and_ldj(spec64, x.double())
    close_vs_f64(zi, orc.flow_inverse(spec, x), wz, k=2.0, rtol=1e-5, atol=2e-5)
    close(li, wl.float(), rtol=1e-5, atol=2e-4)
    # forward direction from a point in the flow's range: the oracle's forward of the latent
    zf = torch.randn(n, dim) * 0.7
    wy, wlf = orc.flow_forward_and_ldj(spec64, zf.double())
    gy, glf = flow.forward_and_log_det_jacobian(zf.to(DEV))
    close_vs_f64(gy, orc.flow_forward(spec, zf), wy, k=2.0, rtol=1e-5, atol=2e-5)
    close(glf, wlf.float(), rtol=1e-5, atol=2e-4)
    # the fused launch agrees with the layer-by-layer evaluation of the same modules
    cur, acc = xd, 0
    for f in reversed(flow.transforms):
        cur, l = f.inverse_and_log_det_jacobian(cur)
        acc = acc + l
    close(zi, cur, rtol=1e-5, atol=2e-5)
    close(li, acc, rtol=1e-5, atol=2e-4)
    st.check_errors()


@pytest.mark.parametrize('dim,hidden,latent_dim,time_kind,n', [
    (64, 64, 0, 'tanh', 1000), (64, 64, 5, 'linear', 513), (10, 12, 0, 'log', 257), (33, 40, 2, 'fourier', 300),
    (8, 16, 0, 'fourier_bounded', 200), (96, 32, 0, 'identity', 129), (1, 8, 3, 'linear', 64)])
def test_fused_time_couplings_and_neural_flow_against_oracle(dim, hidden, latent_dim, time_kind, n):
    """ContinuousAffineCoupling (coupling.py:98-213) as ONE fused step and NeuralFlow.forward(x, t, t0) (flow.py:155-184) as ONE
    launch -- every time net incl. TimeFourier(Bounded) in the kernel, latent inputs, D up to 96 -- against the oracle."""
    torch.manual_seed(71)
    masks = ['ordered_0', 'ordered_1', 'parity_even', 'parity_odd'] if dim > 1 else ['none'] * 4
    desc = [{'kind': 'continuous_affine_coupling', 'dim': dim, 'hidden': [hidden], 'mask': m, 'latent_dim': latent_dim,
             'time_kind': time_kind, 'time_hidden': 7} for m in masks]
    fs = [fd.build_transform(st, d) for d in desc]
    nf = st.NeuralFlow(fs)
    with torch.no_grad():
        for p in nf.parameters():
            p.add_(0.05 * torch.randn_like(p))
    state = {k: v.clone() for k, v in nf.state_dict().items()}
    spec = [fd.transform_spec(d, state, f'transforms.{i}.') for i, d in enumerate(desc)]
    nf = nf.to(DEV)
    x, t, t0 = torch.randn(n, dim), torch.rand(n, 1) * 2, torch.rand(n, 1)
    latent = torch.randn(n, latent_dim) if latent_dim else None
    kw = {} if latent is None else {'latent': latent.to(DEV)}
    okw = {} if latent is None else {'latent': latent}
    assert nf._fused(dim, latent_dim, True, torch.device(DEV, 0)) is not None
    close(nf(x.to(DEV), t.to(DEV), **kw), orc.neural_flow_forward(spec, x, t, **okw), rtol=1e-5, atol=2e-5)
    close(nf(x.to(DEV), t.to(DEV), t0.to(DEV), **kw), orc.neural_flow_forward(spec, x, t, t0, **okw), rtol=1e-5, atol=5e-5)
    if time_kind not in ('fourier', 'fourier_bounded', 'log') or True:
        z = torch.zeros(n, 1)
        got0 = nf(x.to(DEV), z.to(DEV), **kw)
        close(got0, x, rtol=0, atol=1e-6)                                           # identity at t = 0 (test_neural_flow.py:24-27)
    f0 = nf.transforms[0]
    y, ldj = f0.forward_and_log_det_jacobian(x.to(DEV), t.to(DEV), **kw)
    wy, wl = orc.continuous_affine_coupling(spec[0], x, t, latent, False)
    close(y, wy, rtol=1e-5, atol=2e-5)
    close(ldj, wl, rtol=1e-5, atol=2e-5)
    xb, li = f0.inverse_and_log_det_jacobian(y, t.to(DEV), **kw)
    close(xb, x, rtol=1e-5, atol=5e-5)
    close(li, -wl, rtol=1e-5, atol=2e-5)
    st.check_errors()


@pytest.mark.parametrize('scale', [1.0, 20.0, 40.0])
def test_spline_k16_large_logits_take_the_guarded_sweep(scale):
    """The straight-line K = 16 spline phases run their softmax without a running maximum, which is only sound while the logits are
    bounded; the packer leaves the bound (largest |logit| the step's rows can produce from the folded tanh) behind the spline bounds
    and the kernel falls back to the sweep that keeps the maximum beyond it.  Output-layer weights scaled by 20 push the bound past
    the limit (and single logits past +-100, where exp2 without the shift would overflow the sum): both forms against the oracle.
    Scale 40 is the regime where an inverse root lands an ulp outside its bin next to a knot derivative of 1e-3: the log-determinant
    was NaN on 3 of 500 rows before the root was clamped into the bin (tools/experiments/dbg_nan40.py)
    (reference: torch.softmax, rational_quadratic_spline.py:101-105)."""
    torch.manual_seed(5)
    dim, hidden, K = 64, 64, 16
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -3.0, 'upper': 3.0,
             'mask': ('ordered_right_half', 'ordered_left_half')[i % 2], 'latent_dim': 0} for i in range(2)]
    flow = fd.build_flow(st, desc, dim)
    sd = flow.state_dict()
    for k in sd:
        if k.endswith('net.2.weight') or k.endswith('net.2.bias'):      # the conditioners' output layers
            sd[k] = sd[k] * scale
    flow.load_state_dict(sd)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    x = torch.randn(500, dim) * 1.5
    got = flow.log_prob(x.to(DEV))
    prog = flow._fused_program(True, dim, 0, torch.device(DEV, torch.cuda.current_device()))
    assert prog is not None
    # the bound slots: phase steps of the two softmax blocks carry a positive bound, below the limit only for the default init
    import numpy as np
    from stribor_amd import _hip
    blobs = prog.blobs.cpu().numpy()
    bounds = []
    for i in range(prog.prog.n_steps):
        s = prog.prog.steps[i]
        if s.kind == _hip.STEP_RQS_PHASE and s.ct == 0:          # (one slot per group: the first block's, covering both softmax blocks)
            bounds.append(blobs[s.blob_off + _hip.packed_linear_floats(4, 2) + 2])
    assert len(bounds) == 2 * 4 and min(bounds) > 0.0
    assert (max(bounds) < 96.0) == (scale == 1.0), (min(bounds), max(bounds))
    want = orc.flow_log_prob(spec, x)
    if scale == 1.0:
        close(got, want, rtol=1e-5, atol=2e-4)
    else:
        # sharply peaked bins: the evaluation is conditioned like the logits and the reference's own fp32 result is off the fp64
        # value by 1e-2 .. 1 on single rows (tools/experiments/dbg_large_logits.py: medians 2.7e-4 vs 2.8e-4, 99th percentiles
        # 3e-2 vs 7e-3 .. 2e-2, maxima 0.16 vs 1.1 at this scale) with no element-wise relation between the two error patterns:
        # hold the error DISTRIBUTION against the fp64 truth to that of the reference's op sequence in fp32
        spec64 = fd.flow_spec(desc, {k: v.cpu().double() for k, v in flow.state_dict().items()})
        f64 = orc.flow_log_prob(spec64, x.double())
        assert torch.isfinite(got).all()
        e_got = (got.cpu().double() - f64).abs().flatten()
        e_ref = (want.double() - f64).abs().flatten()
        assert e_got.median() <= 2.0 * e_ref.median() + 1e-5, (e_got.median(), e_ref.median())
        assert e_got.quantile(0.9) <= 3.0 * e_ref.quantile(0.9) + 1e-4, (e_got.quantile(0.9), e_ref.quantile(0.9))
        assert e_got.max() <= max(10.0 * e_ref.max(), 0.5), (e_got.max(), e_ref.max())
    if scale == 1.0:       # (the round trip of the guarded sweep itself: every K != 16 case of the width sweeps above)
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
        close(flow.inverse(y), x, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize('dim,hidden,K,layers', [(64, 64, 24, 3), (20, 16, 17, 2), (33, 40, 32, 2), (64, 64, 32, 4)])
def test_spline_couplings_of_17_to_32_bins_fuse(dim, hidden, K, layers):
    """VERDICT r3 missing #1 (K 16 -> 24 cost 8x: conditioner program + element-wise kernel through HBM): rational-quadratic
    couplings of 17 .. 32 bins plan into the one-launch tier -- two output tiles per element, two elements per step, the sweeps with
    the running maximum over 32 slots -- and match the oracle in both directions, inputs reaching into the tails."""
    torch.manual_seed(dim + K)
    masks = ('ordered_right_half', 'ordered_left_half')
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -2.5, 'upper': 2.5,
             'mask': masks[i % 2], 'latent_dim': 0} for i in range(layers)]
    flow = fd.build_flow(st, desc, dim)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    dev = torch.device(DEV, torch.cuda.current_device())
    assert flow._fused_program(True, dim, 0, dev) is not None and flow._fused_program(False, dim, 0, dev) is not None
    for n in (1, 300):
        x = torch.randn(n, dim) * 1.5
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=2e-4 * max(1, dim // 32))
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
        wy, wl = orc.flow_forward_and_ldj(spec, x)
        close(y, wy, rtol=1e-5, atol=2e-5)
        close(ldj, wl, rtol=1e-5, atol=2e-4 * max(1, dim // 32))
        close(flow.inverse(y), x, rtol=1e-4, atol=1e-4)
    st.check_errors()


@pytest.mark.parametrize('kind,layers', [('rqs', 16), ('cubic', 12), ('affine_deep', 70)])
def test_flows_beyond_one_program_run_as_segments(kind, layers):
    """A fused program holds 128 steps (they travel in the kernel's argument segment).  Longer flows used to fall back to the
    layer-by-layer path; they now run as a few fused launches over contiguous runs of layers, the state crossing HBM in fp32 between
    them.  Against the oracle: log_prob, both directions with log-dets, the fp64 batch sum, bf16 storage."""
    torch.manual_seed(layers)
    dim = 64
    masks = ('ordered_right_half', 'ordered_left_half')
    if kind == 'affine_deep':
        desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [32, 32], 'mask': masks[i % 2], 'latent_dim': 0} for i in range(layers)]
    else:
        desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [64], 'n_bins': 16, 'lower': -3.0, 'upper': 3.0, 'mask': masks[i % 2],
                 'latent_dim': 0, **({'spline_type': 'cubic'} if kind == 'cubic' else {})} for i in range(layers)]
    flow = fd.build_flow(st, desc, dim)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    dev = torch.device(DEV, torch.cuda.current_device())
    assert flow._fused_program(True, dim, 0, dev) is None
    segs = flow._fused_segments(True, dim, 0, dev)
    assert segs is not None and len(segs) >= 2 and sum(p.prog.n_steps for p in segs) > 128
    x = torch.randn(200, dim)
    want = orc.flow_log_prob(spec, x)
    # (cubic: the inverse's closed forms + Newton steps differ from the reference's fp32 evaluation by ~1e-4 per layer and row on
    #  single rows -- the 8-layer fixture test allows 1e-3 --; quadratic: 3e-4 per 8 layers as everywhere)
    tol = dict(rtol=2e-5, atol=1e-4) if kind == 'affine_deep' else dict(rtol=1e-5, atol=(5e-3 if kind == 'cubic' else 3e-4 * layers / 8))
    got = flow.log_prob(x.to(DEV))
    close(got, want, **tol)
    tot = flow.log_prob_sum(x.to(DEV))
    assert abs(tot.item() - want.double().sum().item()) <= 1e-5 * abs(want.double().sum().item()) + 1e-2
    z, ldj = flow.inverse_and_log_det_jacobian(x.to(DEV))
    wz, wl = orc.flow_inverse_and_ldj(spec, x) if hasattr(orc, 'flow_inverse_and_ldj') else (orc.flow_inverse(spec, x), None)
    close(z, wz, rtol=1e-4, atol=5e-3 if kind == 'cubic' else 2e-4)      # (cubic: Newton-refined inverse vs the reference's fp32 closed forms, 12 layers deep)
    y, fl = flow.forward_and_log_det_jacobian(z)
    close(y, x, rtol=1e-3, atol=1e-3)
    close(fl, -ldj, rtol=1e-4, atol=3e-3)
    # bf16 storage: one rounding on the way in, fp32 between the segments.  (Not for the cubic flow: bf16's grid puts inputs exactly
    # on domain bounds and knots, where the reference's inverse is a coin flip of its own rounding -- DESIGN_HISTORY 2.1 -- and
    # twelve layers amplify one flipped bin into O(1) of the row's log_prob.)
    if kind != 'cubic':
        xb = x.bfloat16()
        close(flow.log_prob(xb.to(DEV)), orc.flow_log_prob(spec, xb.float()), **tol)
    st.check_errors()


@pytest.mark.parametrize('dim,hidden,masks,layers', [
    (64, 160, ('ordered_right_half', 'ordered_left_half'), 4), (64, 300, ('ordered_right_half', 'ordered_left_half'), 2),
    (128, 160, ('ordered_right_half', 'ordered_left_half'), 3), (48, 200, ('parity_even', 'parity_even'), 2),
    (100, 256, ('ordered_left_half', 'ordered_right_half'), 2),
])
def test_affine_couplings_with_hidden_layers_beyond_128_fuse(dim, hidden, masks, layers):
    """VERDICT r3 missing #1 (hidden 128 -> 160 cost 6x): W2 tanh(W1 z + b1) + b2 is a sum over hidden-unit chunks, so a coupling
    with a wider hidden layer runs as a run of chunk steps of the SAME one-launch program (kernel MODE 20: the (log_scale, shift)
    accumulators stay in registers across the steps), mixed freely with narrower couplings.  Against the oracle, both directions."""
    torch.manual_seed(dim + hidden)
    desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden if i % 2 == 0 else 64], 'mask': masks[i % 2], 'latent_dim': 0}
            for i in range(layers)]
    flow = fd.build_flow(st, desc, dim)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    dev = torch.device(DEV, torch.cuda.current_device())
    prog = flow._fused_program(True, dim, 0, dev)
    assert prog is not None and flow._fused_program(False, dim, 0, dev) is not None
    from stribor_amd import _hip
    assert any(prog.prog.steps[i].kind == _hip.STEP_COUPLING_AFFINE_HC for i in range(prog.prog.n_steps))
    for n in (1, 257):
        x = torch.randn(n, dim)
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=1e-5 * max(1, dim // 32))
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
        wy, wl = orc.flow_forward_and_ldj(spec, x)
        close(y, wy, rtol=1e-5, atol=2e-5)
        close(ldj, wl, rtol=1e-5, atol=1e-5 * max(1, dim // 32))
        close(flow.inverse(y), x, rtol=1e-4, atol=1e-4)
    # a single coupling called on its own takes the same steps
    f0 = flow.transforms[0]
    x = torch.randn(65, dim)
    y0 = f0(x.to(DEV))
    close(f0.inverse(y0), x, rtol=1e-4, atol=1e-4)
    # the exact arithmetic as well
    old = st.set_gemm_precision('exact')
    try:
        x = torch.randn(100, dim)
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=1e-5 * max(1, dim // 32))
    finally:
        st.set_gemm_precision(old)
    st.check_errors()


@pytest.mark.parametrize('dim', [130, 160, 256])
def test_coupling_free_flows_of_129_to_256_columns(dim):
    """a flow of that width made ONLY of element-wise Affine / Flip / Permute layers has no eight-tile program
    (sx_flow_run: eight data tiles carry at least one coupling of kinds 22 / 23) -- the planner must refuse it at plan time so the flow
    answers layer by layer, not with a RuntimeError from the launcher.  log_prob / forward / inverse / log-dets against the oracle."""
    torch.manual_seed(dim)
    desc = [{'kind': 'affine', 'dim': dim}, {'kind': 'flip'}, {'kind': 'permute', 'dim': dim}, {'kind': 'affine', 'dim': dim}]
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    state['transforms.2.permutation'] = flow.transforms[2].permutation.clone()
    spec = fd.flow_spec(desc, state)
    flow = flow.to(DEV)
    dev = torch.device(DEV, torch.cuda.current_device())
    assert flow._fused_program(True, dim, 0, dev) is None and flow._fused_program(False, dim, 0, dev) is None
    assert flow._fused_segments(True, dim, 0, dev) is None
    assert st.NormalizingFlow(st.UnitNormal(dim), [])._fused_program(True, dim, 0, dev) is None          # the empty program
    for n in (1, 257):
        x = torch.randn(n, dim)
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=1e-5 * (dim // 32))
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
        wy, wl = orc.flow_forward_and_ldj(spec, x)
        close(y, wy)
        close(ldj, wl, rtol=1e-5, atol=1e-5 * (dim // 32))
        z, li = flow.inverse_and_log_det_jacobian(x.to(DEV))
        wz, wli = orc.flow_inverse_and_ldj(spec, x)
        close(z, wz)
        close(li, wli, rtol=1e-5, atol=1e-5 * (dim // 32))
    st.check_errors()


@pytest.mark.parametrize('dim,hidden,masks,layers', [
    (160, 64, ('ordered_right_half', 'ordered_left_half'), 4), (256, 128, ('ordered_left_half', 'ordered_right_half'), 2),
    (200, 40, ('parity_even', 'parity_odd'), 3), (132, 16, ('ordered_right_half', 'ordered_left_half'), 2),
])
def test_affine_coupling_flows_of_129_to_256_columns_fuse(dim, hidden, masks, layers):
    """VERDICT r3 missing #1 (D 128 -> 160 cost 12x): eight state tiles at one wave per SIMD (kernel MODE 20, TX = 8); a coupling is a
    hidden step + one step per transformed tile (the pieces of its weights that fit the LDS ring).  Against the oracle in both
    directions, fp32 and bf16 storage, with a Flip and an element-wise Affine in the flow, and the fp64 batch sum."""
    torch.manual_seed(dim)
    desc = []
    for i in range(layers):
        desc.append({'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': masks[i % 2], 'latent_dim': 0})
        if i == 0:
            desc.append({'kind': 'flip'})
        if i == 1:
            desc.append({'kind': 'affine', 'dim': dim})
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if name.endswith('log_scale') or name.endswith('shift'):
                p.copy_(torch.randn_like(p) * 0.2)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    dev = torch.device(DEV, torch.cuda.current_device())
    prog = flow._fused_program(True, dim, 0, dev)
    assert prog is not None and prog.prog.tiles == 8 and flow._fused_program(False, dim, 0, dev) is not None
    for n in (1, 300):
        x = torch.randn(n, dim)
        want = orc.flow_log_prob(spec, x)
        close(flow.log_prob(x.to(DEV)), want, rtol=1e-5, atol=1e-5 * (dim // 32))
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV))
        wy, wl = orc.flow_forward_and_ldj(spec, x)
        close(y, wy, rtol=1e-5, atol=2e-5)
        close(ldj, wl, rtol=1e-5, atol=1e-5 * (dim // 32))
        close(flow.inverse(y), x, rtol=1e-4, atol=1e-4)
        tot = flow.log_prob_sum(x.to(DEV))
        assert abs(tot.item() - want.double().sum().item()) <= 1e-6 * abs(want.double().sum().item()) + 1e-3
    xb = torch.randn(100, dim).bfloat16()
    close(flow.log_prob(xb.to(DEV)), orc.flow_log_prob(spec, xb.float()), rtol=1e-5, atol=1e-5 * (dim // 32))
    old = st.set_gemm_precision('exact')
    try:
        x = torch.randn(64, dim)
        close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x), rtol=1e-5, atol=1e-5 * (dim // 32))
    finally:
        st.set_gemm_precision(old)
    st.check_errors()


@pytest.mark.parametrize('dim,hidden,K,masks,latent', [
    (64, 160, 16, ('ordered_right_half', 'ordered_left_half'), 0), (64, 256, 16, ('ordered_left_half', 'ordered_right_half'), 0),
    (48, 200, 8, ('parity_even', 'parity_odd'), 0), (100, 144, 5, ('ordered_right_half', 'parity_odd'), 0),
    (33, 129, 16, ('ordered_right_half', 'ordered_left_half'), 7), (2, 130, 1, ('ordered_right_half', 'ordered_left_half'), 0),
])
def test_spline_couplings_with_hidden_layers_beyond_128_take_the_slab_forward(dim, hidden, K, masks, latent, monkeypatch):
    """VERDICT r4 missing #1 / next #4 (a spline coupling with hidden > 128 cost 8.5x its hidden-64 neighbour): the slab forward tier
    (sx_rqs_slab_fwd: hidden activation from MLP programs, parameters formed slab by slab on the matrix pipe and consumed in registers).
    Against the oracle in both directions, with inputs reaching into the tails, non-contiguous transformed columns, a latent input,
    ragged batches; and against the tier it replaces (the parameter tensor through HBM)."""
    from stribor_amd.flows.coupling import Coupling
    torch.manual_seed(dim * 13 + hidden)
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -2.5, 'upper': 2.5,
             'mask': masks[i % 2], 'latent_dim': latent} for i in range(3)]
    flow = fd.build_flow(st, desc, dim)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    calls = []
    orig = Coupling._run_spline_slab
    monkeypatch.setattr(Coupling, '_run_spline_slab', lambda self, *a: (calls.append(1), orig(self, *a))[1])
    for n in (1, 300, 2049):
        x = torch.randn(n, dim) * 1.5
        lat = torch.randn(n, latent) if latent else None
        latd = None if lat is None else lat.to(DEV)
        n0 = len(calls)
        close(flow.log_prob(x.to(DEV), latent=latd), orc.flow_log_prob(spec, x, lat), rtol=1e-5, atol=2e-4 * max(1, dim // 32))
        assert len(calls) == n0 + 3                       # every layer answered by the slab tier
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV), latent=latd)
        wy, wl = orc.flow_forward_and_ldj(spec, x, lat)
        close(y, wy, rtol=1e-5, atol=2e-5)
        close(ldj, wl, rtol=1e-5, atol=2e-4 * max(1, dim // 32))
        close(flow.inverse(y, latent=latd), x, rtol=1e-4, atol=1e-4)
        if n == 300:
            got = flow.log_prob(x.to(DEV), latent=latd)
            monkeypatch.setenv('STRIBOR_SPLINE_NO_SLAB_FWD', '1')
            ref = flow.log_prob(x.to(DEV), latent=latd)
            monkeypatch.delenv('STRIBOR_SPLINE_NO_SLAB_FWD')
            close(got, ref.cpu(), rtol=1e-5, atol=2e-4 * max(1, dim // 32))
    from stribor_amd.flows.spline import check_errors
    check_errors()


@pytest.mark.parametrize('dim,hidden,K,masks,latent', [
    (64, [160], 16, ('ordered_right_half', 'ordered_left_half'), 0), (30, [200], 7, ('parity_even', 'parity_odd'), 4),
    (12, [40, 72], 11, ('ordered_left_half', 'parity_odd'), 0),
])
def test_cubic_spline_couplings_on_the_slab_forward_tier(dim, hidden, K, masks, latent, monkeypatch):
    """The slab forward pass with MONOTONE CUBIC splines (the reference's default spline_type; cubic_kernel's arithmetic on the
    parameters in MFMA accumulators): flows whose conditioners are beyond the one-launch tier (or kept out of it by the A/B switch),
    all three directions against the fp64 oracle with ~5 % of the elements in the linear tails."""
    from stribor_amd.flows.coupling import Coupling
    torch.manual_seed(dim + K)
    monkeypatch.setenv('STRIBOR_CUBIC_UNFUSED', '1')
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': hidden, 'n_bins': K, 'lower': -3.0, 'upper': 3.0, 'mask': masks[i % 2],
             'latent_dim': latent, 'spline_type': 'cubic'} for i in range(3)]
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    calls = []
    orig = Coupling._run_spline_slab
    monkeypatch.setattr(Coupling, '_run_spline_slab', lambda self, *a: (calls.append(1), orig(self, *a))[1])
    spec = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
    for n in (1, 1337):
        x = torch.randn(n, dim) * 1.5
        lat = torch.randn(n, latent) if latent else None
        kw = {} if lat is None else {'latent': lat.to(DEV)}
        l64 = None if lat is None else lat.double()
        want_lp = orc.flow_log_prob(spec, x.double(), l64)
        wy, wl = orc.flow_forward_and_ldj(spec, x.double(), l64)
        n0 = len(calls)
        lp = flow.log_prob(x.to(DEV), **kw)
        assert len(calls) == n0 + 3
        y, ldj = flow.forward_and_log_det_jacobian(x.to(DEV), **kw)
        xr = flow.inverse(y, **kw)
        close(lp, want_lp.float(), rtol=1e-5, atol=1e-4)
        close(y, wy.float(), rtol=1e-5, atol=2e-5)
        close(ldj, wl.float(), rtol=1e-5, atol=1e-4)
        close(xr, x, rtol=1e-5, atol=5e-5)
    st.check_errors()


@pytest.mark.parametrize('hidden,act', [([40, 72], 'Tanh'), ([128, 96], 'ReLU'), ([50], 'ELU'), ([200], 'Softplus')])
def test_slab_forward_tier_behind_deep_conditioners_and_other_activations(hidden, act):
    """The slab forward pass takes the last hidden activation of ANY net.MLP conditioner (deeper ones: their MLP programs leave it
    row-major in HBM; one hidden layer: sx_rqs_slab_hidden, any activation): called directly -- also on couplings the one-launch
    tier would answer -- against the oracle, both directions, with a latent input and scattered transformed columns."""
    torch.manual_seed(len(hidden) * 7 + hidden[-1])
    dim, K, L, mask = 40, 9, 5, 'parity_odd'
    net = st.net.MLP(dim + L, hidden, dim * (3 * K - 1), activation=act)
    cpl = st.Coupling(st.Spline(dim, K, latent_net=net, lower=-2.0, upper=2.0, spline_type='quadratic'), mask=mask)
    lin = net.linears()
    spec = [{'kind': 'coupling_rqs', 'mask': mask, 'n_bins': K, 'lower': -2.0, 'upper': 2.0,
             'net': {'weights': [w.detach().clone() for (w, _) in lin], 'biases': [b.detach().clone() for (_, b) in lin],
                     'activation': act}}]
    cpl = cpl.to(DEV)
    x, lat = torch.randn(777, dim) * 1.4, torch.randn(777, L)
    xd, latd = x.to(DEV), lat.to(DEV)
    for reverse in (True, False):
        y, ldj = cpl._run_spline_slab(xd, latd, reverse, True, 1.0)
        wy, wl = orc.flow_inverse_and_ldj(spec, x, lat) if reverse else orc.flow_forward_and_ldj(spec, x, lat)
        close(y, wy, rtol=1e-5, atol=2e-5)
        close(ldj.reshape(-1, 1), wl, rtol=1e-5, atol=2e-4)
    st.check_errors()


def test_slab_forward_tier_takes_any_finite_row_in_the_default_arithmetic():
    """Round 6 (VERDICT r5 #4c): the slab forward tier's hidden-layer kernel zeroes the inputs its mask rules out
    before the fp16 x 3 split (coupling.py:61 multiplies them by mask = 0: a TRANSFORMED column of 1e6 used to turn the row into NaN)
    and rescales a sample whose CONDITIONING input leaves fp16's range by a power of two (as the fused tier's hidden layer does), so
    'fast' returns the oracle's values with no flag -- round 5 pinned NaN + GemmRangeError here.  The bound for a rescaled row is
    relative to its largest entry, like `test_inputs_beyond_fp16_range`'s."""
    torch.manual_seed(5)
    dim, hidden, K = 16, 160, 8
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -3.0, 'upper': 3.0, 'mask': 'ordered_right_half',
             'latent_dim': 0}]
    flow = fd.build_flow(st, desc, dim)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    x = torch.randn(200, dim)
    x[7, 0] = 9.0e4                                  # conditioning columns beyond 65504
    x[150, 3] = -2.0e5
    x[33, 12] = 3.0e6                                # a TRANSFORMED column (linear tail of the spline; weight 0 in the hidden layer)
    x[34, 15] = -1.0e15                             # (its square must not overflow the base density in the reference itself)
    spec64 = orc.spec_to(spec, torch.float64)
    want64 = orc.flow_log_prob(spec64, x.double())
    want = orc.flow_log_prob(spec, x)
    for mode in ('fast', 'auto'):
        old = st.set_gemm_precision(mode)
        try:
            got = flow.log_prob(x.to(DEV))
            st.check_errors()                        # no flag in either mode
        finally:
            st.set_gemm_precision(old)
        assert torch.isfinite(got).all(), mode
        plain = torch.ones(200, dtype=torch.bool); plain[[7, 150]] = False
        close(got[plain.to(got.device)], want[plain], rtol=1e-5, atol=2e-4)
        # rescaled rows: against fp64, no further off than a few times the reference's own fp32 sequence
        for r in (7, 150):
            ref_err = abs(want[r].double().item() - want64[r].item())
            assert abs(got[r].double().item() - want64[r].item()) <= 16 * ref_err + 1e-5 * abs(want64[r].item()), (mode, r, got[r].item(), want64[r].item(), ref_err)
