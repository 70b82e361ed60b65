"""GPU: cached fused programs must follow every change made AFTER a first call.

The planner bakes the transform list, Permute index vectors, masks and the Parameter objects behind each pack job
into host tables (stribor_amd/fused.py).  Each test below evaluates once (so the program is built and cached), then
changes something the reference would pick up on its next call (it re-reads everything every call:
stribor/flow.py:99-130, flows/permute.py:62-82, flows/affine.py:148-171,222-288, flows/coupling.py:48-53) and checks
the second evaluation against the oracle.
"""
import copy
import os
import sys

import pytest
import torch
import torch.nn as nn

import flowdesc as fd
from goldens import Golden
from producthelp import close, product_flow

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stribor_oracle as orc

import stribor_amd as st

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def _oracle_logp(desc, flow, x):
    spec = fd.flow_spec(desc, {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()})
    return orc.flow_log_prob(spec, x.cpu())


def test_permutation_loaded_after_first_call():
    """flow.log_prob(x); flow.load_state_dict(other permutation); flow.log_prob(x)  (permute.py:62-82)."""
    g = Golden('f7_permute')
    m = g.meta['mixed']
    torch.manual_seed(1234)
    flow = fd.build_flow(st, m['desc'], m['dim']).to(DEV)          # random permutation != the fixture's
    x = g.t('mixed/x').to(DEV)
    first = flow.log_prob(x)
    assert first.shape == (x.shape[0], 1)
    flow.load_state_dict(g.state('mixed'))                          # in-place copy_ into the permutation buffer
    close(flow.log_prob(x), g.t('mixed/log_prob'))
    close(flow.inverse(x), g.t('mixed/inverse'))
    close(flow.forward(x), g.t('mixed/forward'))
    # and a permutation buffer replaced wholesale (new tensor object)
    perm_layers = [f for f in flow.transforms if isinstance(f, st.Permute)]
    assert perm_layers
    p = perm_layers[0]
    new = torch.randperm(p.dim, generator=torch.Generator().manual_seed(7)).to(DEV)
    p.permutation = new
    want = _oracle_logp(m['desc'], flow, x)
    close(flow.log_prob(x), want)
    # standalone layer path of the same module
    close(p.forward(x), x[:, new])
    close(p.inverse(p.forward(x)), x)


def test_transform_list_edited_after_first_call():
    """Swapping / replacing entries of flow.transforms re-plans (flow.py:99-107 loops over the live list)."""
    torch.manual_seed(3)
    desc = fd.cfg2_desc(4, 16, 32)
    flow = fd.build_flow(st, desc, 16).to(DEV)
    x = torch.randn(300, 16, device=DEV)
    close(flow.log_prob(x), _oracle_logp(desc, flow, x))
    # swap two layers (ModuleList.__setitem__ never passes through a __setattr__ of ours)
    a, b = flow.transforms[0], flow.transforms[1]
    flow.transforms[0], flow.transforms[1] = b, a
    desc2 = [desc[1], desc[0]] + desc[2:]
    close(flow.log_prob(x), _oracle_logp(desc2, flow, x))
    close(flow.inverse(x), orc.flow_inverse(fd.flow_spec(desc2, {k: v.cpu() for k, v in flow.state_dict().items()}), x.cpu()))
    # replace one layer by a new module, append another
    torch.manual_seed(4)
    flow.transforms[2] = fd.build_transform(st, desc[2]).to(DEV)
    flow.transforms.append(fd.build_transform(st, desc[0]).to(DEV))
    desc3 = desc2 + [desc[0]]
    close(flow.log_prob(x), _oracle_logp(desc3, flow, x))


def test_weight_objects_replaced_after_first_call():
    """layer.weight = nn.Parameter(...), mlp.net[i] = nn.Linear(...), coupling.transform.latent_net = MLP(...)."""
    torch.manual_seed(5)
    desc = fd.cfg2_desc(2, 8, 16)
    flow = fd.build_flow(st, desc, 8).to(DEV)
    x = torch.randn(200, 8, device=DEV)
    close(flow.log_prob(x), _oracle_logp(desc, flow, x))
    net = flow.transforms[0].transform.latent_net
    lin0 = net.net[0]
    lin0.weight = nn.Parameter(torch.randn_like(lin0.weight) * 0.3)          # new Parameter object, old one orphaned
    close(flow.log_prob(x), _oracle_logp(desc, flow, x))
    net.net[2] = nn.Linear(16, 16).to(DEV)                                   # foreign nn.Linear in the Sequential
    close(flow.log_prob(x), _oracle_logp(desc, flow, x))
    flow.transforms[1].transform.latent_net = st.net.MLP(8, [16], 16).to(DEV)
    close(flow.log_prob(x), _oracle_logp(desc, flow, x))
    # in-place updates (optimizer steps) keep working through the version counters
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(0.01 * torch.randn_like(p))
    close(flow.log_prob(x), _oracle_logp(desc, flow, x))
    # the coupling's own single-layer programs follow as well
    c = flow.transforms[0]
    want_y, want_ldj = orc.flow_forward_and_ldj(fd.flow_spec(desc[:1], {('transforms.0.' + k): v.cpu() for k, v in c.state_dict().items()}), x.cpu())
    y, ldj = c.forward_and_log_det_jacobian(x)
    close(y, want_y)
    close(ldj, want_ldj, atol=1e-4)
    net.net[0].bias = nn.Parameter(torch.randn(16, device=DEV))
    want_y, want_ldj = orc.flow_forward_and_ldj(fd.flow_spec(desc[:1], {('transforms.0.' + k): v.cpu() for k, v in c.state_dict().items()}), x.cpu())
    y, ldj = c.forward_and_log_det_jacobian(x)
    close(y, want_y)
    close(ldj, want_ldj, atol=1e-4)


def test_mask_changed_after_first_call():
    torch.manual_seed(6)
    desc = fd.cfg2_desc(2, 8, 16)
    flow = fd.build_flow(st, desc, 8).to(DEV)
    x = torch.randn(100, 8, device=DEV)
    close(flow.log_prob(x), _oracle_logp(desc, flow, x))
    flow.transforms[0].mask_func = st.util.get_mask('parity_even')
    desc2 = copy.deepcopy(desc)
    desc2[0]['mask'] = 'parity_even'
    close(flow.log_prob(x), _oracle_logp(desc2, flow, x))
    y, ldj = flow.transforms[0].forward_and_log_det_jacobian(x)
    want_y, want_ldj = orc.flow_forward_and_ldj(fd.flow_spec(desc2[:1], {k: v.cpu() for k, v in flow.state_dict().items()}), x.cpu())
    close(y, want_y)
    close(ldj, want_ldj, atol=1e-4)


@pytest.mark.parametrize('dim', [8, 128])
def test_dense_layer_logdet_follows_parameter_updates(dim):
    """AffineLU / MatrixExponential: log|det| is parameter-only (affine.py:171, 287-288) and must be refreshed together
    with the matrices -- in the flow's fused program and in the layers' own programs."""
    torch.manual_seed(7)
    desc = [{'kind': 'affine_lu', 'dim': dim},
            {'kind': 'coupling_affine', 'dim': dim, 'hidden': [32], 'mask': 'ordered_right_half', 'latent_dim': 0},
            {'kind': 'matrix_exp', 'dim': dim, 'bias': True, 'log_time': False},
            {'kind': 'coupling_affine', 'dim': dim, 'hidden': [32], 'mask': 'ordered_left_half', 'latent_dim': 0}]
    flow = fd.build_flow(st, desc, dim).to(DEV)
    x = torch.randn(257, dim, device=DEV)
    first = flow.log_prob(x)
    close(first, _oracle_logp(desc, flow, x), rtol=1e-5, atol=1e-4)
    lu, mx = flow.transforms[0], flow.transforms[2]
    y0, l0 = lu.forward_and_log_det_jacobian(x)          # builds the layers' own programs too
    y1, l1 = mx.inverse_and_log_det_jacobian(x)
    with torch.no_grad():
        lu.log_diag.add_(0.05)                            # log-det changes by 0.05 * dim
        mx.diag.mul_(1.5).add_(0.01)
    second = flow.log_prob(x)
    want = _oracle_logp(desc, flow, x)
    close(second, want, rtol=1e-5, atol=1e-4)
    assert (second - first).abs().max().item() > 1e-2     # the update is visible at all
    # standalone layers against the oracle's per-layer functions
    sd = {k: v.cpu() for k, v in flow.state_dict().items()}
    spec = fd.flow_spec(desc, sd)
    wy, wl = orc.flow_forward_and_ldj(spec[:1], x.cpu())
    y, l = lu.forward_and_log_det_jacobian(x)
    close(y, wy, rtol=1e-5, atol=1e-4)
    close(l, wl, rtol=1e-5, atol=1e-4)
    close(lu.log_det_jacobian(x, None), wl, rtol=1e-5, atol=1e-4)
    wy, wl = orc.flow_inverse_and_ldj(spec[2:3], x.cpu())
    y, l = mx.inverse_and_log_det_jacobian(x)
    close(y, wy, rtol=1e-4, atol=2e-4)
    close(l, wl, rtol=1e-5, atol=1e-4)


def test_load_state_dict_after_first_call_cfg4_slice():
    """A cfg-4 style flow evaluated, then loaded with the golden state (in-place copies), then evaluated again."""
    g = Golden('f6_cfg4')
    m = g.meta['cfg4']
    torch.manual_seed(99)
    flow = fd.build_flow(st, m['desc'], m['dim']).to(DEV)
    x = g.t('cfg4/x').to(DEV)
    flow.log_prob(x)
    flow.load_state_dict(g.state('cfg4'))
    close(flow.log_prob(x), g.t('cfg4/log_prob'))


def test_bookkeeping_attributes_do_not_replan_and_declared_ones_do():
    """`flow.step = i` in a training loop bumped the process-wide structure epoch, so every flow of the process
    re-planned on every call.  Attributes a user hangs on a module after construction are not structure; attributes the constructor
    declared (a coupling's set_data, a spline's bin count ...), sub-modules, parameters and buffers still are."""
    from stribor_amd import fused
    torch.manual_seed(0)
    flow = fd.build_flow(st, fd.cfg2_desc(2, 8, 16), 8).to(DEV)
    x = torch.randn(33, 8, device=DEV)
    with torch.no_grad():
        want = flow.log_prob(x).clone()
        prog = flow._fused_program(True, 8, 0, x.device)
        e0 = fused._STRUCT_EPOCH[0]
        for i in range(5):
            flow.step = i                                   # new public attribute on the flow
            flow.transforms[0].note = ('epoch', i)          # ... on a layer
            flow.transforms[0].transform.latent_net.tag = i
            assert flow._fused_program(True, 8, 0, x.device) is prog
        assert fused._STRUCT_EPOCH[0] == e0
        assert torch.equal(flow.log_prob(x), want)
        # a declared attribute: re-plans (and the result follows it)
        flow.transforms[0].mask_func = st.util.mask.get_mask('ordered_left_half')
        assert fused._STRUCT_EPOCH[0] > e0
        assert flow._fused_program(True, 8, 0, x.device) is not prog
        got = flow.log_prob(x)
        assert not torch.allclose(got, want)
        # a parameter replaced by a new Parameter object: re-plans
        e1 = fused._STRUCT_EPOCH[0]
        lin = flow.transforms[1].transform.latent_net.net[0]
        lin.weight = torch.nn.Parameter(lin.weight.detach() * 0.5)
        assert fused._STRUCT_EPOCH[0] > e1


def test_error_flags_are_per_stream():
    """One flag word per (device, stream): a flow that leaves the fp16 x 3 range on one stream is reported to the next call on THAT
    stream (or to check_errors()), never to work queued on another stream (one word per device let flow A's
    condition surface in flow B's next call)."""
    torch.manual_seed(1)
    from producthelp import relu_flow
    fa = relu_flow()                                        # (ReLU conditioners: see relu_flow)
    fb = relu_flow()
    xa = torch.randn(64, 16, device=DEV)
    xa[5] *= 1.0e7                                          # relu(W1 x) beyond fp16's range: flow A's row 5 comes back NaN and flags
    xb = torch.randn(64, 16, device=DEV)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    from stribor_amd import _hip
    with torch.no_grad(), _hip.no_redo():                   # (the flag path: see relu_flow)
        with torch.cuda.stream(sa):
            ya = fa.log_prob(xa)
        sa.synchronize()
        assert torch.isnan(ya[5]).all()
        with torch.cuda.stream(sb):
            for _ in range(3):
                yb = fb.log_prob(xb)                        # polls stream B's word only: nothing to report
            sb.synchronize()
        assert torch.isfinite(yb).all()
        fb.log_prob(xb)                                     # default stream: its own word, clean as well
        with torch.cuda.stream(sa):
            with pytest.raises(st.GemmRangeError):
                fa.log_prob(xa[:8])                         # the next call on stream A reports it (and clears it)
            fa.log_prob(xb)                                 # clean now
        sa.synchronize()
        # check_errors() sees every stream's word
        with torch.cuda.stream(sb):
            fb.log_prob(xa)
        with pytest.raises(st.GemmRangeError):
            st.check_errors()
        st.check_errors()


def test_training_calls_of_spline_flows_raise_inside_the_failing_call_by_default(monkeypatch):
    """Mode 'grad' (the default): a graph-building call of a flow that holds a rational-quadratic spline -- the reference op that
    asserts on its data, rational_quadratic_spline.py:175-178,223 -- synchronises once before it returns and raises what its kernels
    flagged; the same call under no_grad(), and flows without that op, stay asynchronous.  (The flag is set from the host here, the way
    a kernel's system-scope atomic would: the forward kernels of a training step have no condition a test can provoke at will.)"""
    import warnings
    from stribor_amd import _hip
    from producthelp import relu_flow
    assert _hip._sync_mode == 'grad'
    torch.manual_seed(5)
    dim, K = 8, 6
    flow = st.NormalizingFlow(st.UnitNormal(dim), [st.Coupling(st.Spline(dim, K, latent_net=st.net.MLP(dim, [16], dim * (3 * K - 1)),
                                                                         lower=-3, upper=3, spline_type='quadratic'),
                                                               mask='ordered_right_half')]).to(DEV)
    x = torch.randn(200, dim, device=DEV)
    with torch.no_grad():
        want = flow.log_prob(x)
    st.check_errors()
    calls = []
    real = _hip.end_of_flow_call

    def flagged_end(t, flag=True):
        calls.append(1)
        if flag:
            _hip._flag_entry(t.device)[1][0] |= _hip.FLAG_RQS_NEG_DISCRIMINANT      # "a kernel of this call found a negative discriminant"
        return real(t)
    monkeypatch.setattr(_hip, 'end_of_flow_call', flagged_end)
    with warnings.catch_warnings(), torch.enable_grad():    # (this file's tests run under no_grad: the training form needs grad mode)
        warnings.simplefilter('ignore')
        with pytest.raises(AssertionError, match='discriminant'):
            flow.log_prob(x)                                # parameters require grad: the training form of the call
        assert len(calls) == 1                              # log_prob -> inverse_and_log_det_jacobian: the outermost call only
        st.check_errors()                                   # raised once, nothing left behind
        monkeypatch.setattr(_hip, 'end_of_flow_call', lambda t: flagged_end(t, False))
        lp = flow.log_prob(x)                               # a clean training call: one synchronisation, the same values, a graph
        assert len(calls) == 2 and lp.requires_grad and torch.allclose(lp.detach(), want, rtol=1e-5, atol=1e-5)
        flow.inverse_and_log_det_jacobian(y=x)              # the data argument by keyword falls under the same rule
        assert len(calls) == 3
        with pytest.raises(AssertionError, match='discriminant'):
            with torch.no_grad():
                flow.log_prob(x)                            # inference: no synchronisation ...
                assert len(calls) == 3
                _hip._flag_entry(x.device)[1][0] |= _hip.FLAG_RQS_NEG_DISCRIMINANT
                flow.log_prob(x[:5])                        # ... a flag is reported by the next call on the stream
        old = _hip.set_sync_errors(False)
        try:
            assert old == 'grad'
            flow.log_prob(x)                                # mode '0': no synchronisation under grad either
            assert len(calls) == 3
        finally:
            _hip.set_sync_errors(old)
        # a flow without the asserting op stays asynchronous under grad (the rule would cost cfg 2 13 .. 24 % of a training step)
        aff = relu_flow()
        lp = aff.log_prob(torch.randn(64, 16, device=DEV))
        assert len(calls) == 3 and torch.isfinite(lp).all()
    st.check_errors()


def test_sync_errors_mode_raises_inside_the_failing_call():
    """set_sync_errors(True) / STRIBOR_SYNC_ERRORS=1: the data-dependent error leaves the call that caused it, like the reference's
    (rational_quadratic_spline.py:175-178,223) -- a script that ends right after the failing call still sees it."""
    from stribor_amd import _hip
    torch.manual_seed(2)
    from producthelp import relu_flow
    flow = relu_flow()
    x = torch.randn(64, 16, device=DEV)
    x[7] *= -1.0e7
    old = _hip.set_sync_errors(True)
    try:
        with torch.no_grad(), _hip.no_redo():               # (the flag path: see relu_flow)
            with pytest.raises(st.GemmRangeError):
                flow.log_prob(x)
            st.check_errors()                               # raised once, nothing left behind
            assert torch.isfinite(flow.log_prob(x[:5])).all()
        with torch.no_grad():
            # with the redo pass (the default for an inference call) nothing is flagged: finite values
            net = st.net.MLP(8, [16], 4, activation='ReLU').to(DEV)
            z = torch.randn(32, 8, device=DEV)
            z[3, 0] = 1.0e6
            assert torch.isfinite(net(z)).all()
            flow.log_prob(x)                                # (row 7 overflows exp() in the reference as well: no flag is what matters)
            st.check_errors()
    finally:
        _hip.set_sync_errors(old)
    with torch.no_grad(), _hip.no_redo():
        flow.log_prob(x)                                    # lazy again: no raise here ...
        torch.cuda.synchronize()
        with pytest.raises(st.GemmRangeError):
            flow.log_prob(x[:5])                            # ... the next call reports it
