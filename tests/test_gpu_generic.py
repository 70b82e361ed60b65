"""GPU: the generic tier of Coupling / ContinuousAffineCoupling and the reference surface added in round 2 --
set_data=True (coupling.py:48-51), conditioners that are not a stribor MLP (affine.py:59-67, spline.py:76-87), widths
beyond the fused kernel's tiles (D = 200, H = 256, n_bins = 24), Fourier time nets (time_net.py:49-91), Flip over other
axes (permute.py:30-44), the parameter accessors of AffineLU / MatrixExponential (affine.py:148-154, 222-241, 173-179,
290-299).  Values: fixtures F12 captured from the reference (tests/golden/make_golden.py f12) and the oracle.
"""
import os
import sys

import pytest
import torch

import flowdesc as fd
from goldens import Golden
from producthelp import close, product_flow, product_transform

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stribor_oracle as orc

import stribor_amd as st

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def test_f12_set_data_and_hand_written_conditioners():
    g = Golden('f12_set_and_hand')
    n = 0
    for case, m in g.meta.items():
        if case.startswith('cac/'):
            continue
        f = product_transform(g, case)
        x = g.t(case + '/x').to(DEV)
        kw = {'latent': g.t(case + '/latent').to(DEV)} if g.has(case + '/latent') else {}
        spline = m['desc'][0]['kind'] == 'coupling_rqs'
        ltol = dict(rtol=1e-5, atol=2e-4 if spline else 1e-4)
        y = f(x, **kw)
        close(y, g.t(case + '/y'))
        close(f.inverse(g.t(case + '/y').to(DEV), **kw), g.t(case + '/x_back'), rtol=1e-4, atol=1e-4)
        close(f.log_det_jacobian(x, y, **kw), g.t(case + '/ldj'), **ltol)
        y2, l1 = f.forward_and_log_det_jacobian(x, **kw)
        close(y2, g.t(case + '/y'))
        close(l1, g.t(case + '/ldj_fwd'), **ltol)
        xb, l2 = f.inverse_and_log_det_jacobian(g.t(case + '/y').to(DEV), **kw)
        close(xb, g.t(case + '/x_back'), rtol=1e-4, atol=1e-4)
        close(l2, g.t(case + '/ldj_inv'), **ltol)
        assert y.shape == x.shape and l1.shape == x.shape[:-1] + (1,)
        n += 1
    assert n == 30


def test_f12_continuous_coupling_generic_time_nets_and_conditioners():
    g = Golden('f12_set_and_hand')
    n = 0
    for case, m in g.meta.items():
        if not case.startswith('cac/'):
            continue
        f = product_transform(g, case)
        x, t = g.t(case + '/x').to(DEV), g.t(case + '/t').to(DEV)
        y, ldj = f.forward_and_log_det_jacobian(x, t)
        close(y, g.t(case + '/y'))
        close(ldj, g.t(case + '/ldj'), rtol=1e-5, atol=1e-5)
        xb, li = f.inverse_and_log_det_jacobian(g.t(case + '/y').to(DEV), t)
        close(xb, g.t(case + '/x_back'), rtol=1e-5, atol=1e-5)
        close(li, g.t(case + '/ldj_inv'), rtol=1e-5, atol=1e-5)
        close(f(x, t), g.t(case + '/y'))
        n += 1
    assert n == 8


@pytest.mark.parametrize('case', ['wide_affine', 'wide_rqs'])
def test_f12_wide_flows(case):
    """D = 200 / H = 256 affine couplings and n_bins = 24 / H = 160 spline couplings: beyond the fused kernel's tiles,
    so every layer runs conditioner (library GEMMs) + element-wise HIP kernel."""
    g = Golden('f12_wide')
    flow = product_flow(g, case)
    x = g.t(case + '/x').to(DEV)
    lt = dict(rtol=1e-5, atol=2e-4 if case == 'wide_rqs' else 1e-4)
    close(flow.log_prob(x), g.t(case + '/log_prob'), **lt)
    close(flow.log_prob(x).double(), g.t(case + '/log_prob_f64'), **lt)
    close(flow.inverse(x), g.t(case + '/inverse'))
    close(flow.forward(x), g.t(case + '/forward'))
    z, ldj = flow.inverse_and_log_det_jacobian(x)
    close(z, g.t(case + '/inverse'))
    close(ldj, g.t(case + '/inverse_ldj'), **lt)
    cur = x
    for i in reversed(range(len(flow.transforms))):
        cur, l = flow.transforms[i].inverse_and_log_det_jacobian(cur)
        close(cur, g.t(f'{case}/inv_x.{i}'))
        close(l, g.t(f'{case}/inv_ldj.{i}'), **lt)
    # a larger batch against the oracle (round trip + log_prob)
    m = g.meta[case]
    spec = fd.flow_spec(m['desc'], g.state(case))
    xb = torch.randn(3000, m['dim'], generator=torch.Generator().manual_seed(5))
    close(flow.log_prob(xb.to(DEV)), orc.flow_log_prob(spec, xb), **lt)
    close(flow.forward(flow.inverse(xb.to(DEV))), xb, rtol=1e-4, atol=1e-4)


def test_mlp_wide_final_activation_and_unknown_activation():
    """net.MLP beyond the kernel's tiles, with a final activation (mlp.py:55-56) or an activation the kernel does not
    know: library-GEMM tier, same values."""
    torch.manual_seed(0)
    for kw in (dict(in_dim=200, hidden_dims=[256, 300], out_dim=70),
               dict(in_dim=10, hidden_dims=[16], out_dim=4, final_activation='Sigmoid'),
               dict(in_dim=10, hidden_dims=[16, 16], out_dim=4, activation='Softsign'),
               dict(in_dim=6, hidden_dims=[], out_dim=3)):
        net = st.net.MLP(**kw)
        x = torch.randn(50, kw['in_dim'])
        want = net.net(x)                                   # the module's own torch layers on the CPU = mlp.py:65
        close(net.to(DEV)(x.to(DEV)), want, rtol=1e-5, atol=1e-5)
    # inside a coupling: a final activation makes the conditioner non-fusable -> generic tier
    torch.manual_seed(1)
    net = st.net.MLP(6, [16], 12, final_activation='Tanh')
    c = st.Coupling(st.Affine(6, latent_net=net), mask='parity_even')
    x = torch.randn(40, 6)
    m = orc.mask_vector('parity_even', 6)
    p = net.net(x * m)
    ls, sh = p.chunk(2, -1)
    want = (x * ls.exp() + sh) * (1 - m) + x * m
    c = c.to(DEV)
    y, ldj = c.forward_and_log_det_jacobian(x.to(DEV))
    close(y, want)
    close(ldj, (ls * (1 - m)).sum(-1, keepdim=True), rtol=1e-5, atol=1e-5)


def test_flip_over_other_axes():
    x = torch.randn(3, 4, 5)
    for dims in ([0], [1], [0, 1], [-2], [0, 2], [-1, 0]):
        f = st.Flip(dims)
        y = f(x.to(DEV))
        assert torch.equal(y.cpu(), torch.flip(x, dims)), dims
        assert torch.equal(f.inverse(y).cpu(), x), dims
        assert torch.equal(f.log_det_jacobian(x.to(DEV), y).cpu(), torch.zeros(3, 4, 1))
        try:
            want = torch.eye(5).flip(dims).diag().log().expand_as(x)             # permute.py:44
        except IndexError:                    # the reference's own expression flips a 2-D eye: axis 2 does not exist
            with pytest.raises(IndexError):
                f.log_diag_jacobian(x.to(DEV), y)
            continue
        got = f.log_diag_jacobian(x.to(DEV), y).cpu()
        assert torch.equal(torch.isinf(got), torch.isinf(want)) and torch.equal(got[~torch.isinf(got)], want[~torch.isinf(want)])
    # inside a flow: such a Flip is not a feature relabelling -> the flow runs layer by layer
    torch.manual_seed(2)
    flow = st.NormalizingFlow(st.UnitNormal(5), [st.Affine(5), st.Flip([0]), st.Affine(5)]).to(DEV)
    xx = torch.randn(7, 5)
    a0, a1 = flow.transforms[0], flow.transforms[2]
    cur = (xx - a1.shift.cpu()) * torch.exp(-a1.log_scale.cpu())
    cur = torch.flip(cur, [0])
    cur = (cur - a0.shift.cpu()) * torch.exp(-a0.log_scale.cpu())
    close(flow.inverse(xx.to(DEV)), cur.detach())


def test_dense_layer_accessors():
    torch.manual_seed(3)
    lu = st.AffineLU(6).to(DEV)
    W, ld = lu.weight.detach().cpu(), lu.log_diag.detach().cpu()
    L = torch.tril(W, -1) + torch.eye(6)
    U = torch.triu(W, 1) + torch.eye(6) * ld.exp()
    close(lu.L, L, rtol=1e-6, atol=1e-6)
    close(lu.U, U, rtol=1e-6, atol=1e-6)
    x = torch.randn(4, 3, 6, device=DEV)
    close(lu.jacobian(x, None), ((L @ U).T).expand(4, 3, -1, -1), rtol=1e-5, atol=1e-6)
    for log_time in (False, True):
        mx = st.MatrixExponential(5, bias=True, log_time=log_time).to(DEV)
        Wm, dg = mx._weight.detach().cpu(), mx.diag.detach().cpu()
        Lm, Um = torch.tril(Wm, -1) + torch.eye(5), torch.triu(Wm) + torch.eye(5)
        gl, gu = mx.lu()
        close(gl, Lm, rtol=1e-6, atol=1e-6)
        close(gu, Um, rtol=1e-6, atol=1e-6)
        A = Lm @ Um
        Wt = (A * dg) @ torch.linalg.inv(A)
        close(mx.weight, Wt, rtol=1e-4, atol=1e-5)
        t = torch.rand(4, 1) + 0.1
        tt = torch.log1p(t.abs()) if log_time else t
        close(mx.get_time(t.to(DEV), (4, 5)), tt, rtol=1e-6, atol=1e-6)
        assert mx.get_time(0.5, (4, 5)).shape == (4, 1)
        xx = torch.randn(4, 5)
        close(mx.jacobian(xx.to(DEV), None, t=t.to(DEV)), torch.matrix_exp(Wt * tt.unsqueeze(-1)), rtol=1e-4, atol=1e-5)
        # the Jacobian really is d forward / dx (affine.py:290-299): columns of forward(e_i) - forward(0)
        e = torch.eye(5)
        y = mx(torch.cat([e, torch.zeros(1, 5)]).to(DEV), t=0.7).cpu()
        J = (y[:5] - y[5:]).T
        close(mx.jacobian(xx[:1].to(DEV), None, t=0.7)[0], J, rtol=1e-4, atol=1e-5)
    sp = st.Spline(4, 3, spline_type='quadratic')
    w0 = sp.width.detach().clone()
    sp.reset_parameters()                                                         # spline.py:71-74
    assert not torch.equal(sp.width, w0)
    st.ELU(1, 2, foo=3)                                                           # activations.py: no __init__ of its own


def test_set_data_coupling_inside_a_flow():
    """A flow holding a set_data coupling runs layer by layer; round trip + log_prob against the oracle."""
    torch.manual_seed(4)
    desc = [{'kind': 'coupling_affine', 'dim': 5, 'hidden': [16], 'mask': 'ordered_left_half', 'latent_dim': 0, 'set_data': True},
            {'kind': 'coupling_affine', 'dim': 5, 'hidden': [16], 'mask': 'ordered_right_half', 'latent_dim': 0},
            {'kind': 'coupling_affine', 'dim': 5, 'hidden': [16], 'mask': 'parity_even', 'latent_dim': 0, 'set_data': True}]
    flow = fd.build_flow(st, desc, 5)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    flow = flow.to(DEV)
    x = torch.randn(6, 4, 5)
    close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec, x))
    close(flow.forward(flow.inverse(x.to(DEV))), x, rtol=1e-4, atol=1e-4)


def test_row_sums_are_bit_reproducible_for_any_row_length():
    """Per-row log-dets of the element-wise kernels are fixed-order sums (no float atomics): rows of
    7, 33, 100 and 200 transformed columns, repeated launches bit-identical and equal to the fp64 row sums at 1e-6 rel."""
    from stribor_amd.flows.pointwise import PW_SIGMOID, run_pointwise
    from stribor_amd.flows.spline import run_cubic_kernel, run_rqs_kernel
    torch.manual_seed(9)
    for n_live, n in [(7, 5001), (33, 3000), (100, 1500), (200, 513), (64, 1000), (5, 3)]:
        d = n_live + 3
        x = (torch.rand(n, d) * 2 - 1).to(DEV)
        for K, run in ((6, 'rqs'), (5, 'cubic')):
            P = 3 * K - 1 if run == 'rqs' else 2 * K + 2
            params = torch.randn(n, n_live * P, device=DEV)
            outs = []
            for _ in range(3):
                if run == 'rqs':
                    y, ldj, ldiag = run_rqs_kernel(x, params, params.stride(0), None, 2, n_live, K, -1., 1., -1., 1., True, True, True)
                else:
                    y, ldj, ldiag = run_cubic_kernel(x, params, params.stride(0), None, 2, n_live, K, -1., 1., False, True, True)
                outs.append(ldj.clone())
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (run, n_live)
            want = ldiag.double().sum(-1)
            assert ((outs[0].double() - want).abs() <= 1e-5 + 1e-6 * ldiag.double().abs().sum(-1)).all(), (run, n_live)
        xs = torch.randn(n, d, device=DEV)
        a = run_pointwise(xs, PW_SIGMOID, want_ldj=True, want_ldiag=True)
        b = run_pointwise(xs, PW_SIGMOID, want_ldj=True, want_ldiag=True)
        assert torch.equal(a[1], b[1]), n_live
        assert ((a[1].reshape(-1).double() - a[2].double().sum(-1)).abs() <= 1e-5 + 1e-6 * a[2].double().abs().sum(-1)).all()


def test_mlp_program_with_more_than_256_output_tiles():
    """Found by tools/fuzz_train.py --fat --wide: an MLP program writes its output in tiles of 32 columns whose index the device
    step keeps in 8 bits; outputs beyond 8192 columns (e.g. 121 columns x 92 spline parameters = 348 tiles) wrapped -- the
    columns past 8192 were never written and the first ones were overwritten.  Each chunk launch now writes its own window."""
    torch.manual_seed(9)
    net = st.net.MLP(20, [48], 8192 + 2 * 4032 + 77).to(DEV)          # 511 tiles: five launches
    x = torch.randn(333, 20, device=DEV)
    with torch.no_grad():
        got = net(x)
        lin = [m for m in net.modules() if isinstance(m, torch.nn.Linear)]
        h = torch.tanh(torch.nn.functional.linear(x.double(), lin[0].weight.double(), lin[0].bias.double()))
        want = torch.nn.functional.linear(h, lin[1].weight.double(), lin[1].bias.double())
    close(got, want.float(), rtol=1e-5, atol=1e-5)


def test_wide_hidden_layers_and_wide_couplings_run_as_chunked_mfma_programs():
    """Round 3 (tier 2 widened): a single hidden layer of any width runs as one MFMA program per 128 hidden units (later chunks
    accumulate into the output), and a coupling wider than 128 columns whose conditioning columns fit the tiles reads them as a
    column subset of the wide rows -- no library GEMM; values against the oracle."""
    torch.manual_seed(33)
    net = st.net.MLP(20, [300], 70).to(DEV)
    with torch.no_grad():
        net.net[2].bias.normal_()
        x = torch.randn(517, 20, device=DEV)
        progs = net._program(torch.device(DEV, 0))
        assert len(progs) >= 3                                     # 300 hidden units = 3 chunks (x output windows)
        ws = [m.weight.detach().cpu() for m in net.net if isinstance(m, torch.nn.Linear)]
        bs = [m.bias.detach().cpu() for m in net.net if isinstance(m, torch.nn.Linear)]
        close(net(x), orc.mlp_forward({'weights': ws, 'biases': bs, 'activation': 'Tanh'}, x.cpu()), rtol=1e-5, atol=2e-5)
    for dim, hidden, kind in ((160, 200, 'coupling_affine'), (136, 64, 'coupling_rqs'), (40, 160, 'coupling_rqs')):
        d = {'kind': kind, 'dim': dim, 'hidden': [hidden], 'mask': 'ordered_left_half', 'latent_dim': 3}
        if kind == 'coupling_rqs':
            d.update(n_bins=5, lower=-3.0, upper=3.0)
        f = fd.build_transform(st, d)
        with torch.no_grad():
            for p in f.parameters():
                p.add_(0.03 * torch.randn_like(p))
        spec = fd.transform_spec(d, {'T.' + k: v.clone() for k, v in f.state_dict().items()}, 'T.')
        f = f.to(DEV)
        x, lat = torch.randn(300, dim), torch.randn(300, 3)
        with torch.no_grad():
            progs = (f._affine_unfused_program if kind == 'coupling_affine' else f._spline_program)(dim, 3, torch.device(DEV, 0))[0]
            assert len(progs) >= (2 if hidden > 128 else 1)
            y, ldj = f.forward_and_log_det_jacobian(x.to(DEV), latent=lat.to(DEV))
            wy, wl = orc.transform_forward_and_ldj(spec, x, latent=lat)
            close(y, wy, rtol=1e-5, atol=2e-5)
            close(ldj, wl, rtol=1e-5, atol=2e-4)
            xb, li = f.inverse_and_log_det_jacobian(y, latent=lat.to(DEV))
            close(xb, x, rtol=1e-4, atol=1e-4)
            close(li, -wl, rtol=1e-5, atol=2e-4)
    st.check_errors()


def _ref_coupling(transform_fwd, transform_inv, logdiag, x, mask, latent=None):
    """stribor/flows/coupling.py:55-95 + flow.py:36-47 restated in torch fp64 around three callables of the wrapped transform."""
    def z_of(v):
        z = v * mask
        if v.shape[-1] == 1:
            z = z * 0
        return z if latent is None else torch.cat([z, latent], -1)
    y = transform_fwd(x, z_of(x)) * (1 - mask) + x * mask
    ldj = (logdiag(x, y, z_of(x)) * (1 - mask)).sum(-1, keepdim=True)
    xi = transform_inv(y, z_of(y)) * (1 - mask) + y * mask
    ildj = -(logdiag(xi, y, z_of(xi)) * (1 - mask)).sum(-1, keepdim=True)
    return y, ldj, xi, ildj


@pytest.mark.parametrize('wrapped', ['sigmoid', 'elu', 'leaky', 'affine_const'])
@pytest.mark.parametrize('shape,mask', [((7, 4, 6), 'ordered_left_half'), ((10, 5), 'parity_even'), ((3, 1), 'none')])
def test_coupling_around_any_elementwise_transform(wrapped, shape, mask):
    """VERDICT r3 missing #2: the reference's Coupling calls transform(x, latent=z), transform.inverse(x, latent=z) and
    transform.log_diag_jacobian(x, y, latent=z) on WHATEVER it wraps (coupling.py:10-46,74-76,94).  Point-wise flows and an
    Affine without a conditioner inside a Coupling: every method of the reference's method set against its op sequence restated
    in fp64, and differentiable end to end."""
    torch.manual_seed(31)
    dim = shape[-1]
    if wrapped == 'sigmoid':
        t = st.Sigmoid()
        fwd, inv = (lambda v, z: torch.sigmoid(v)), (lambda v, z: torch.log(v) - torch.log1p(-v))
        ld = lambda a, b, z: torch.nn.functional.logsigmoid(a) + torch.nn.functional.logsigmoid(-a)
        x = torch.randn(*shape)
        xinv = None
    elif wrapped == 'elu':
        t = st.ELU()
        fwd = lambda v, z: torch.where(v > 0, v, torch.expm1(v))
        inv = lambda v, z: torch.where(v > 0, v, torch.log1p(v))
        ld = lambda a, b, z: torch.where(a > 0, torch.zeros_like(a), a)
        x = torch.randn(*shape)
    elif wrapped == 'leaky':
        t = st.LeakyReLU(negative_slope=0.2)
        fwd = lambda v, z: torch.where(v > 0, v, 0.2 * v)
        inv = lambda v, z: torch.where(v > 0, v, v / 0.2)
        ld = lambda a, b, z: torch.where(a > 0, torch.zeros_like(a), torch.full_like(a, float(torch.log(torch.tensor(0.2)))))
        x = torch.randn(*shape)
    else:
        t = st.Affine(dim)                                   # no latent_net: per-column constants (affine.py:63-64)
        with torch.no_grad():
            t.log_scale.copy_(torch.randn(1, dim) * 0.3)
            t.shift.copy_(torch.randn(1, dim))
        ls, sh = t.log_scale.detach().double().reshape(-1), t.shift.detach().double().reshape(-1)
        fwd, inv = (lambda v, z: v * ls.exp() + sh), (lambda v, z: (v - sh) * (-ls).exp())
        ld = lambda a, b, z: ls.expand_as(a)
        x = torch.randn(*shape)
    f = st.Coupling(t, mask=mask).to(DEV)
    m = torch.from_numpy(f.mask_vector(dim)).double().expand(*shape)
    wy, wl, wxi, wil = _ref_coupling(fwd, inv, ld, x.double(), m)
    xd = x.to(DEV)
    y = f(xd)
    close(y, wy.float(), rtol=1e-5, atol=1e-6)
    close(f.log_det_jacobian(xd, y), wl.float(), rtol=1e-5, atol=1e-5)
    y2, l2 = f.forward_and_log_det_jacobian(xd)
    close(y2, wy.float(), rtol=1e-5, atol=1e-6)
    close(l2, wl.float(), rtol=1e-5, atol=1e-5)
    # (inverse: the reference evaluates transform.inverse on the pass-through columns too and multiplies by 1 - mask = 0 afterwards:
    #  where a pass-through value lies outside the inverse's domain -- logit of a value beyond (0, 1), ELU^-1 below -1 -- that is
    #  NaN * 0 = NaN in the reference.  The point-wise kernels return a finite value there, so the product hands back the
    #  pass-through value itself: every entry the reference defines must match, the others must be the input or NaN)
    def close_where_defined(a, b, passthrough, **kw):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        ok = ~torch.isnan(b)
        assert not torch.isnan(a[ok]).any()
        close(torch.where(ok, a, torch.zeros_like(a)), torch.where(ok, b, torch.zeros_like(b)), **kw)
        if passthrough is not None:
            rest = a[~ok]
            assert (torch.isnan(rest) | (rest == passthrough.float()[~ok])).all()
    xi, il = f.inverse_and_log_det_jacobian(y)
    close_where_defined(xi, wxi.float(), wy, rtol=1e-4, atol=1e-5)
    close_where_defined(il, wil.float(), None, rtol=1e-4, atol=1e-4)
    close_where_defined(f.inverse(y), wxi.float(), wy, rtol=1e-4, atol=1e-5)
    if not torch.isnan(wxi).any():
        close(xi, x, rtol=1e-4, atol=1e-5)
    # inside a flow, with a graph: log_prob and its input gradient against autograd of the restated ops
    flow = st.NormalizingFlow(st.UnitNormal(dim), [f]).to(DEV)
    # (the input: every column inside the inverse's domain -- the reference evaluates transform.inverse on the pass-through columns
    #  too and multiplies by (1 - mask) = 0 afterwards, so a pass-through value outside the domain is NaN * 0 = NaN there)
    yin = fwd(x.double(), None)
    with torch.enable_grad():
        xg = yin.float().to(DEV).requires_grad_(True)
        lp = flow.log_prob(xg)
        lp.sum().backward()
        yr = xg.detach().cpu().double().requires_grad_(True)
        xr = inv(yr, None) * (1 - m) + yr * m
        lpr = -(ld(xr, yr, None) * (1 - m)).sum(-1, keepdim=True) + (-0.5 * xr ** 2 - 0.9189385332046727).sum(-1, keepdim=True)
        lpr.sum().backward()
    close(lp, lpr.float(), rtol=1e-4, atol=1e-4)
    close(xg.grad, yr.grad.float(), rtol=1e-3, atol=1e-4)
