"""Test helper: build the product flow / transform for a golden case and move it to the GPU."""
import torch

import flowdesc as fd
import stribor_amd as st


def product_flow(g, case, device='cuda'):
    m = g.meta[case]
    flow = fd.build_flow(st, m['desc'], m['dim'])
    flow.load_state_dict(g.state(case))
    return flow.to(device)


def product_transform(g, case, device='cuda'):
    m = g.meta[case]
    d = m['desc'][0]
    f = fd.build_transform(st, d)
    state = {k[len('transforms.0.'):]: v for k, v in g.state(case).items()}
    f.load_state_dict(state)
    return f.to(device)


def close(a, b, rtol=1e-5, atol=1e-5):
    """|a-b| <= atol + rtol*|b| — the north_star's 1e-5 relative bound with a 1e-5 floor at |b| < 1."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    ok = torch.allclose(a, b, rtol=rtol, atol=atol)
    if not ok:
        err = (a - b).abs()
        i = err.argmax()
        raise AssertionError(f'max abs err {err.max().item():.3e} at {i.item()} '
                             f'(got {a.flatten()[i].item():.7g}, want {b.flatten()[i].item():.7g})')
    return True


def close_vs_f64(got, ref32, f64, k=2.0, rtol=1e-5, atol=1e-5, row_scale=False):
    """The north_star bound measured against the fp64 truth: |got - f64| <= k * |ref32 - f64| + atol + rtol * |f64|
    element-wise, i.e. never more than k times the reference's OWN fp32 error plus the 1e-5 allowance (the form the
    spline tests use).  `ref32`: the reference's fp32 values (fixture), `f64`: the same op in fp64."""
    got = got.detach().double().cpu()
    ref32, f64 = ref32.detach().double().cpu(), f64.detach().double().cpu()
    assert got.shape == f64.shape == ref32.shape, (got.shape, ref32.shape, f64.shape)
    # row_scale: values that went through DENSE layers (every output mixes all columns of the row, so its rounding error scales with
    # the row's largest entry, in the reference's fp32 arithmetic as much as here): the relative allowance is taken of max |row|
    scale = f64.abs().amax(-1, keepdim=True).expand_as(f64) if row_scale else f64.abs()
    bound = k * (ref32 - f64).abs() + atol + rtol * scale
    err = (got - f64).abs()
    bad = err > bound
    if bad.any():
        i = (err - bound).argmax()
        raise AssertionError(f'{int(bad.sum())} elements beyond {k} x the reference\'s own fp32 error + {atol}: worst '
                             f'|got - f64| = {err.flatten()[i].item():.3e}, |ref32 - f64| = '
                             f'{(ref32 - f64).abs().flatten()[i].item():.3e} at {i.item()} (f64 {f64.flatten()[i].item():.7g})')
    return True


def relu_flow(dim=16, hidden=32, layers=2, device='cuda'):
    """A coupling flow whose conditioners use ReLU: their hidden activations are UNBOUNDED fp16 x 3 operands (a row scaled by 1e7
    drives relu(W1 x) far beyond 65504).  Used by the tests of the error-flag plumbing, which run their launches under
    `stribor_amd._hip.no_redo()`: since round 6 an inference call hands such samples to the exact-fp32 kernel in a second launch
    (sx_flow_run2) and nothing is flagged; without the redo pass -- a training step, a C-ABI caller of sx_flow_run -- they come back
    as NaN + GemmRangeError."""
    masks = ('ordered_right_half', 'ordered_left_half')
    return st.NormalizingFlow(st.UnitNormal(dim), [
        st.Coupling(st.Affine(dim, latent_net=st.net.MLP(dim, [hidden], 2 * dim, activation='ReLU')), mask=masks[i % 2])
        for i in range(layers)]).to(device)
