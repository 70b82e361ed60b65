"""GPU: the two GEMM arithmetics of the fused kernel and the fp16 x 3 range guard.

'fast' (default): fp16 x 3 split on v_mfma_f32_32x32x16_f16.  Its operands must stay within fp16's range; a sample
whose flow state / activations leave it comes back as NaN (never as a plausible number) and GemmRangeError is raised
lazily; weights beyond the range raise the same way.  'exact': v_mfma_f32_32x32x2_f32, no range limit.  'auto':
re-runs out-of-range calls exactly.  The reference computes in plain fp32 (torch addmm): every mode must either match
it at the north_star's 1e-5 or raise.
"""
import os
import sys

import pytest
import torch

import flowdesc as fd
from goldens import Golden
from producthelp import close, product_flow

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stribor_oracle as orc

import stribor_amd as st

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(autouse=True)
def _inference_mode_and_default_precision():
    old = st.set_gemm_precision('fast')
    try:
        with torch.no_grad():
            yield
    finally:
        st.set_gemm_precision(old)
        try:
            st.check_errors()            # leave no pending flag behind for the next test
        except Exception:
            pass


def _flow_and_oracle(desc, dim, seed=0):
    torch.manual_seed(seed)
    flow = fd.build_flow(st, desc, dim)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    return flow.to(DEV), spec


def _rel_close(got, want, tol=1e-5):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = ((got - want).abs() / want.abs().clamp_min(1.0)).max().item()
    assert err <= tol, err


def test_inputs_beyond_fp16_range():
    """Inputs of 1e5: exact / auto match the oracle; fast returns NaN for exactly those rows and raises."""
    desc = fd.cfg2_desc(4, 64, 64)
    flow, spec = _flow_and_oracle(desc, 64)
    torch.manual_seed(1)
    x = torch.randn(777, 64)
    big = torch.zeros(777, dtype=torch.bool)
    big[[3, 100, 101, 500, 776]] = True
    x[3, 5] = 1.0e5                       # a conditioning column of layer 0
    x[100, 40] = -2.5e5                   # a transformed column of layer 0 (conditions layer 1)
    x[101] *= 3.0e4                       # whole row around 3e4..1e5
    x[101, 0] = 9.0e4
    x[500, 63] = 3.0e5
    x[776, 31] = 65505.0                  # just beyond the largest finite fp16
    x[10, 7] = 2.0e4                      # inside the range (also after a few layers' scaling): must stay accurate
    x[11, 50] = -2.0e4
    want = orc.flow_log_prob(spec, x)
    want_z = orc.flow_inverse(spec, x)
    xd = x.to(DEV)

    st.set_gemm_precision('exact')
    _rel_close(flow.log_prob(xd), want)
    _rel_close(flow.inverse(xd), want_z)
    st.check_errors()

    st.set_gemm_precision('auto')
    _rel_close(flow.log_prob(xd), want)
    _rel_close(flow.inverse(xd), want_z)
    tot = torch.zeros(1, dtype=torch.float64, device=DEV)
    flow.log_prob_sum(xd, tot)
    assert abs(tot.item() - want.double().sum().item()) <= 1e-6 * abs(want.double().sum().item())
    st.check_errors()                     # auto consumed the flag itself

    st.set_gemm_precision('fast')
    got = flow.log_prob(xd)
    z = flow.inverse(xd)
    torch.cuda.synchronize()
    bad = torch.isnan(got.reshape(-1)).cpu()
    assert bad[big].all(), 'out-of-range rows must come back as NaN, not as plausible numbers'
    assert not bad[~big].any()
    assert torch.isnan(z[big.to(DEV)]).all(dim=1).all() and not torch.isnan(z[~big.to(DEV)]).any()
    _rel_close(got.cpu()[~big], want[~big])
    _rel_close(z.cpu()[~big], want_z[~big])
    with pytest.raises(st.GemmRangeError):
        st.check_errors()
    st.check_errors()                     # raised once, then cleared
    # the lazy form: the NEXT call reports the previous call's condition without any explicit check
    flow.log_prob(xd)
    torch.cuda.synchronize()
    with pytest.raises(st.GemmRangeError):
        flow.log_prob(xd[:8])


def test_weights_beyond_fp16_range():
    """Weights of 1e5: flagged at pack time in fast mode; exact / auto match the oracle."""
    desc = fd.cfg2_desc(2, 16, 32)
    flow, _ = _flow_and_oracle(desc, 16, seed=2)
    with torch.no_grad():
        w = flow.transforms[0].transform.latent_net.net[0].weight
        w[3, 10] = 1.0e5
        w[7, 12] = -3.0e5
    spec = fd.flow_spec(desc, {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()})
    x = torch.randn(300, 16, generator=torch.Generator().manual_seed(3))
    want = orc.flow_log_prob(spec, x)
    xd = x.to(DEV)
    st.set_gemm_precision('exact')
    _rel_close(flow.log_prob(xd), want)
    st.set_gemm_precision('auto')
    _rel_close(flow.log_prob(xd), want)
    st.check_errors()
    st.set_gemm_precision('fast')
    got = flow.log_prob(xd)
    with pytest.raises(st.GemmRangeError):
        st.check_errors()
    assert not torch.isfinite(got).all()          # nothing plausible came back for the rows the weight touches


def test_unbounded_activations_and_mlp_program():
    """ReLU conditioners: hidden activations are unbounded B operands too (tracked); MLP.forward poisons its rows."""
    torch.manual_seed(4)
    net = st.net.MLP(8, [16, 16], 6, activation='ReLU').to(DEV)
    x = torch.randn(64, 8)
    x[5] *= 3.0e4
    x[5, 0] = 2.0e5
    x[9, 2] = 1.0e6
    ws = [m.weight.detach().cpu() for m in net.net if isinstance(m, torch.nn.Linear)]
    bs = [m.bias.detach().cpu() for m in net.net if isinstance(m, torch.nn.Linear)]
    want = orc.mlp_forward({'weights': ws, 'biases': bs, 'activation': 'ReLU'}, x)
    st.set_gemm_precision('exact')
    _rel_close(net(x.to(DEV)), want)
    st.set_gemm_precision('auto')
    _rel_close(net(x.to(DEV)), want)
    st.set_gemm_precision('fast')
    got = net(x.to(DEV)).cpu()
    bad = torch.isnan(got).any(1)
    assert bad[9] and bad[5]
    ok = ~bad
    _rel_close(got[ok], want[ok])
    with pytest.raises(st.GemmRangeError):
        st.check_errors()


@pytest.mark.parametrize('fixture,case', [('f3_cfg1', 'cfg1'), ('f4_cfg2', 'cfg2'), ('f7_permute', 'mixed'),
                                           ('f5_cfg3', 'cfg3'), ('f6_cfg4', 'cfg4')])
def test_exact_mode_against_golden(fixture, case):
    """Every kernel variant of the exact-fp32 arithmetic (couplings, splines, dense layers) against the fixtures."""
    st.set_gemm_precision('exact')
    g = Golden(fixture)
    flow = product_flow(g, case)
    x = g.t(case + '/x').to(DEV)
    close(flow.log_prob(x), g.t(case + '/log_prob'), rtol=1e-5, atol=1e-4 if case == 'cfg3' else 1e-5)
    close(flow.inverse(x), g.t(case + '/inverse'), rtol=1e-4, atol=2e-4)
    close(flow.forward(x), g.t(case + '/forward'), rtol=1e-4, atol=2e-4)
    st.check_errors()


def test_modes_agree_on_ordinary_data_and_switch_freely():
    """Switching the mode between calls re-packs the weights in the other fragment layout (one program, two blobs)."""
    desc = fd.cfg2_desc(8, 64, 64)
    flow, spec = _flow_and_oracle(desc, 64, seed=5)
    x = torch.randn(2048, 64, generator=torch.Generator().manual_seed(6))
    want = orc.flow_log_prob(spec, x)
    xd = x.to(DEV)
    for mode in ('fast', 'exact', 'fast', 'auto', 'exact'):
        st.set_gemm_precision(mode)
        _rel_close(flow.log_prob(xd), want)
    with torch.no_grad():
        for p in flow.parameters():
            p.mul_(1.01)
    spec = fd.flow_spec(desc, {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()})
    want = orc.flow_log_prob(spec, x)
    for mode in ('exact', 'fast'):
        st.set_gemm_precision(mode)
        _rel_close(flow.log_prob(xd), want)
    st.check_errors()


def test_second_device_guard():
    """Tensors on cuda:1 while cuda:0 is current: launches follow the tensors' device."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    desc = fd.cfg2_desc(2, 16, 32)
    flow, spec = _flow_and_oracle(desc, 16, seed=8)
    flow = flow.to('cuda:1')
    x = torch.randn(100, 16)
    assert torch.cuda.current_device() == 0
    _rel_close(flow.log_prob(x.to('cuda:1')), orc.flow_log_prob(spec, x))
    assert torch.cuda.current_device() == 0


def _scale_close(got, want, tol=1e-5, operands=None, atol=0.0):
    """|got - want| <= tol * (|want| + scale), scale = the largest magnitude among the result and the layer's operands: the 1e-5
    bound relative to the data's OWN scale (tiny data stays tiny; a result that is a small difference of O(scale) operands --
    the round trip back to a tiny input -- is held to the operands' scale, as fp32 itself would be)."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = want.abs().max() if operands is None else max(want.abs().max(), operands.detach().abs().max().double().cpu())
    bound = tol * (want.abs() + scale) + atol
    err = (got - want).abs()
    assert (err <= bound).all(), (err.max().item(), want.abs().max().item())


@pytest.mark.parametrize('x_exp,w_exp,all_layers', [(-12, -8, False), (-8, -12, False), (-16, -10, True), (-14, 0, False)])
def test_small_magnitude_side_of_the_fp16x3_split(x_exp, w_exp, all_layers):
    """hi + lo of a value below 2^-14 * 2^10 leaves the `lo` half an fp16 denormal (absolute floor ~3e-8).
    cfg-2-shaped flow with inputs x 2^x_exp and first-layer weights x 2^w_exp (and the transposed pairing): 'fast' against the
    fp64 oracle on EVERY per-layer output and log-det, at the 1e-5 bound relative to each tensor's own scale."""
    desc = fd.cfg2_desc(8, 64, 64)
    torch.manual_seed(12)
    flow = fd.build_flow(st, desc, 64)
    with torch.no_grad():
        for i, f in enumerate(flow.transforms):
            lin = [m for m in f.transform.latent_net.net if isinstance(m, torch.nn.Linear)]
            if i == 0 or all_layers:
                lin[0].weight.mul_(2.0 ** w_exp)
            lin[1].bias.normal_(0, 0.05)          # off the zero init (mlp.py:53): the layers do something
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    spec64 = orc.spec_to(fd.flow_spec(desc, state), torch.float64)
    flow = flow.to(DEV)
    x = torch.randn(512, 64, generator=torch.Generator().manual_seed(13)) * 2.0 ** x_exp
    cur64, cur = x.double(), x.to(DEV)
    for i in reversed(range(8)):                               # the log_prob direction, layer by layer (one launch each)
        nxt64, ldj64 = orc.transform_inverse_and_ldj(spec64[i], cur64)
        nxt, ldj = flow.transforms[i].inverse_and_log_det_jacobian(cur64.float().to(DEV))
        _scale_close(nxt, nxt64)
        # a row's log-det sums 32 log-scales, each an O(|W2|_1) GEMM result: the per-layer absolute floor the suite uses elsewhere
        # (test_per_layer_api_against_golden: 2e-5) -- it does not grow as the data shrinks, which is what this test pins
        _scale_close(ldj, ldj64, atol=2e-5)
        y, ldf = flow.transforms[i].forward_and_log_det_jacobian(nxt64.float().to(DEV))
        _scale_close(y, cur64, operands=nxt64)
        _scale_close(ldf, -ldj64, atol=2e-5)
        cur64 = nxt64
    _rel_close(flow.log_prob(x.to(DEV)), orc.flow_log_prob(spec64, x.double()))
    _scale_close(flow.inverse(x.to(DEV)), orc.flow_inverse(spec64, x.double()))
    st.check_errors()


def test_nan_inputs_propagate_to_their_own_rows_only():
    """The reference propagates NaN through torch ops: a NaN anywhere in a row makes that row's log_prob NaN and leaves every
    other row untouched.  Same here, in both arithmetics, for affine, spline and dense-linear programs."""
    for desc, dim in ((fd.cfg2_desc(4, 64, 64), 64), (fd.cfg3_desc(2, 16, 32, 8), 16), (fd.cfg4_desc(1, 32, 32), 32)):
        flow, spec = _flow_and_oracle(desc, dim, seed=21)
        x = torch.randn(200, dim, generator=torch.Generator().manual_seed(22))
        want = orc.flow_log_prob(spec, x)
        x[5, 3] = float('nan')                       # conditioning column of some layers, transformed column of others
        x[77, dim - 1] = float('nan')
        x[130] = float('nan')
        bad = torch.zeros(200, dtype=torch.bool)
        bad[[5, 77, 130]] = True
        for mode in ('fast', 'exact'):
            st.set_gemm_precision(mode)
            got = flow.log_prob(x.to(DEV)).cpu().reshape(-1)
            z = flow.inverse(x.to(DEV)).cpu()
            assert torch.isnan(got[bad]).all(), mode
            assert torch.isfinite(got[~bad]).all() and torch.isfinite(z[~bad]).all(), mode
            assert torch.isnan(z[bad]).any(dim=1).all(), mode
            _rel_close(got[~bad], want.reshape(-1)[~bad], 1e-5 if desc[0]['kind'] != 'coupling_rqs' else 1e-4)
            try:
                st.check_errors()
            except st.GemmRangeError:             # a NaN operand may be reported as out of range; it must not go unnoticed as a number
                pass


def test_auto_mode_with_wide_hidden_layers_accumulating_chunks():
    """Hidden layers wider than 128 run as one launch per 128 hidden units, the later ones ADDING into the output.  'auto' re-runs a
    flagged launch in the exact arithmetic: an accumulating launch must be re-run from the output it started from, or the rows that
    were fine in fp16 x 3 get the chunk twice and the flagged rows stay NaN.  Stand-alone MLP and a coupling
    whose conditioner has 300 hidden units, inputs with rows beyond fp16's range."""
    torch.manual_seed(9)
    net = st.net.MLP(12, [300], 20).to(DEV)
    x = torch.randn(200, 12)
    x[7, 3] = 2.0e5
    x[150] *= 4.0e4
    x[150, 1] = 1.5e5
    ws = [m.weight.detach().cpu() for m in net.net if isinstance(m, torch.nn.Linear)]
    bs = [m.bias.detach().cpu() for m in net.net if isinstance(m, torch.nn.Linear)]
    want = orc.mlp_forward({'weights': ws, 'biases': bs, 'activation': 'Tanh'}, x)
    with torch.no_grad():
        assert len(net._program(torch.device(DEV, torch.cuda.current_device()))) >= 3 and net._fits_program()
        st.set_gemm_precision('auto')
        got = net(x.to(DEV))
        assert torch.isfinite(got).all()
        _rel_close(got, want)
        st.check_errors()
        st.set_gemm_precision('fast')
        with pytest.raises(st.GemmRangeError):      # (a multi-launch call may report its first launch's condition before it returns)
            bad = torch.isnan(net(x.to(DEV))).any(1).cpu()
            assert bad[7] and bad[150] and bad.sum() == 2
            st.check_errors()
        try:
            st.check_errors()                       # (launches queued behind the one that raised may have flagged as well)
        except st.GemmRangeError:
            pass
        # the same through a coupling's conditioner programs (unfused tier: hidden > 128)
        desc = [{'kind': 'coupling_affine', 'dim': 16, 'hidden': [260], 'mask': 'ordered_right_half', 'latent_dim': 0}]
        flow, spec = _flow_and_oracle(desc, 16, seed=10)
        xx = torch.randn(150, 16)
        xx[3, 12] = 3.0e5                  # a conditioning column
        wantl = orc.flow_log_prob(spec, xx)
        st.set_gemm_precision('auto')
        _rel_close(flow.log_prob(xx.to(DEV)), wantl)
        st.check_errors()
