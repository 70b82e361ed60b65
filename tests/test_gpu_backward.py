"""GPU: hand-written backward of log_prob against torch.autograd run on the oracle.

The oracle is differentiable (plain torch ops), so d(-mean log_prob)/d(parameters, input) from autograd in fp64
is the truth; the HIP backward must match it to 1e-4 relative (fp32 accumulation over the batch)."""
import os
import sys

import pytest
import torch

import flowdesc as fd
from producthelp import close

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stribor_oracle as orc

import stribor_amd as st

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def oracle_grads(desc, state, x, dtype=torch.float64):
    leaves = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in state.items()}
    spec = fd.flow_spec(desc, leaves)
    xin = x.detach().to(dtype).clone().requires_grad_(True)
    loss = -orc.flow_log_prob(spec, xin).mean()
    loss.backward()
    return loss.item(), {k: v.grad for k, v in leaves.items()}, xin.grad


@pytest.mark.parametrize('n,layers,hidden', [(257, 2, 64), (1000, 8, 64), (64, 3, 40)])
def test_log_prob_backward_matches_autograd_of_oracle(n, layers, hidden):
    torch.manual_seed(4)
    desc = fd.cfg2_desc(layers, 64, hidden)
    flow = fd.build_flow(st, desc, 64)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(n, 64)
    want_loss, want_g, want_gx = oracle_grads(desc, state, x)
    xg = x.to(DEV).requires_grad_(True)
    lp = flow.log_prob(xg)
    assert lp.requires_grad and lp.shape == (n, 1)
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss)
    close(xg.grad, want_gx.float(), rtol=1e-4, atol=1e-6)
    for name, p in flow.named_parameters():
        assert p.grad is not None, name
        ref = want_g[name].float()
        scale = ref.abs().max().item() + 1e-12
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err <= 2e-4 * scale + 1e-7, (name, err, scale)


def test_training_step_reduces_loss():
    """A few SGD steps on a toy target: loss goes down and parameters actually move (re-pack on version bump)."""
    torch.manual_seed(0)
    flow = fd.build_flow(st, fd.cfg2_desc(4, 64, 64), 64).to(DEV)
    opt = torch.optim.SGD(flow.parameters(), lr=1e-2)
    data = torch.randn(4096, 64, device=DEV) * 0.5 + 0.3
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = -flow.log_prob(data).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 0.5, losses
    with torch.no_grad():
        assert torch.isfinite(flow.log_prob(data)).all()


def test_unsupported_flows_evaluate_without_graph():
    torch.manual_seed(0)
    # an unknown keyword argument cannot be routed through the layer-wise training path: no graph, and a warning says so
    flow = st.NormalizingFlow(st.UnitNormal(4), [st.AffineLU(4), st.MatrixExponential(4)]).to(DEV)
    x = torch.randn(10, 4, device=DEV, requires_grad=True)
    with pytest.warns(RuntimeWarning):
        lp = flow.log_prob(x, t=torch.rand(10, 1, device=DEV), foo=1)
    assert not lp.requires_grad


def test_matrix_exponential_with_per_row_time_is_differentiable():
    """Round 2: log_prob(x, t=...) with a per-row time (affine.py:236-270, 287-288) builds a graph; gradients (incl. dL/dt)
    against fp64 autograd of the oracle."""
    torch.manual_seed(31)
    dim = 6
    desc = [{'kind': 'affine_lu', 'dim': dim}, {'kind': 'matrix_exp', 'dim': dim, 'bias': True, 'log_time': True},
            {'kind': 'coupling_affine', 'dim': dim, 'hidden': [16], 'mask': 'ordered_left_half', 'latent_dim': 0},
            {'kind': 'matrix_exp', 'dim': dim, 'bias': False, 'log_time': False}]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x, t = torch.randn(200, dim), torch.rand(200, 1) * 2 - 0.5
    leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items()}
    spec = fd.flow_spec(desc, leaves)
    x64, t64 = x.double().requires_grad_(True), t.double().requires_grad_(True)
    want = -orc.flow_log_prob(spec, x64, t=t64).mean()
    want.backward()
    xg, tg = x.to(DEV).requires_grad_(True), t.to(DEV).requires_grad_(True)
    lp = flow.log_prob(xg, t=tg)
    assert lp.requires_grad and lp.shape == (200, 1)
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item()) + 1e-5
    for got, ref in [(xg.grad, x64.grad), (tg.grad, t64.grad)] + [(p.grad, leaves[n].grad) for n, p in flow.named_parameters()]:
        ref = ref.float()
        assert ((got.cpu() - ref).abs().max() / (ref.abs().max() + 1e-12)).item() <= 3e-4
    with torch.no_grad():
        close(flow.log_prob(x.to(DEV), t=t.to(DEV)), lp, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('n,dim,hidden,K,layers,masks', [
    (300, 8, 16, 5, 2, ('ordered_right_half', 'ordered_left_half')),
    (257, 64, 64, 16, 2, ('ordered_right_half', 'ordered_left_half')),       # cfg-3 widths
    (100, 10, 12, 3, 3, ('parity_even', 'parity_odd', 'ordered_left_half')),
    (65, 5, 12, 1, 2, ('ordered_right_half', 'parity_odd')),
    (300, 64, 128, 16, 2, ('ordered_right_half', 'ordered_left_half')),      # four hidden tiles in the slab backward (round 5)
    (200, 24, 96, 8, 2, ('parity_even', 'parity_odd')),
    (260, 64, 160, 16, 2, ('ordered_right_half', 'ordered_left_half')),      # five hidden tiles: two slab-backward launches per layer
    (150, 20, 256, 7, 2, ('parity_even', 'ordered_left_half')),
])
def test_spline_flow_log_prob_backward_matches_autograd_of_oracle(n, dim, hidden, K, layers, masks):
    """Training of rational-quadratic spline coupling flows (second half): the spline and its
    hand-written backward (sx_rqs_inverse_bwd) against fp64 autograd through the oracle, inputs reaching into both
    linear tails."""
    torch.manual_seed(11)
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -3, 'upper': 3,
             'mask': masks[i % len(masks)], 'latent_dim': 0} for i in range(layers)]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(n, dim) * 1.6
    want_loss, want_g, want_gx = oracle_grads(desc, state, x)
    xg = x.to(DEV).requires_grad_(True)
    lp = flow.log_prob(xg)
    assert lp.requires_grad and lp.shape == (n, 1)
    with torch.no_grad():
        close(lp, flow.log_prob(x.to(DEV)), rtol=1e-5, atol=1e-4)      # same values as the fused no-graph path
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss) + 1e-5
    sx = want_gx.abs().max().item()
    assert (xg.grad.cpu() - want_gx.float()).abs().max().item() <= 2e-4 * sx + 1e-7
    for name, p in flow.named_parameters():
        assert p.grad is not None, name
        ref = want_g[name].float()
        scale = ref.abs().max().item() + 1e-12
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err <= 3e-4 * scale + 1e-7, (name, err, scale)


@pytest.mark.parametrize('stype', ['quadratic', 'cubic'])
@pytest.mark.parametrize('n,dim,hidden,K,latent_dim,act', [
    (1000, 64, [64], 16, 0, 'Tanh'),            # cfg-3 layer
    (333, 7, [10], 4, 3, 'Tanh'),               # odd live count, hidden rows not 16-byte aligned, conditional flow
    (500, 12, [24, 40], 9, 0, 'ELU'),           # deeper conditioner, another activation (no one-layer fused forward)
    (31, 6, [33], 16, 0, 'Tanh'),               # fewer rows than one chunk; hidden spills into a second tile
    (700, 8, [32], 8, 0, 'Tanh'),               # one full hidden tile (vector loads), run-time bin count
    (4100, 10, [32], 16, 2, 'Tanh'),            # one hidden tile, K = 16 straight-line form, odd number of slabs, several ranges
    (2100, 64, [128], 16, 0, 'Tanh'),           # four hidden tiles (round 5: 65 .. 128 hidden units, one wave per SIMD)
    (900, 20, [96], 16, 3, 'Tanh'),             # three hidden tiles, vector loads, conditional flow
    (450, 9, [70], 6, 0, 'ReLU'),               # three tiles, the last one ragged; run-time bin count
    (1200, 16, [48, 100], 11, 0, 'Tanh'),       # deeper conditioner, four tiles, the last one ragged
    (1500, 64, [160], 16, 0, 'Tanh'),           # five hidden tiles: two launches over tiles [0, 3) and [3, 5) (round 5)
    (700, 18, [200], 9, 2, 'Tanh'),             # seven tiles (4 + 3), the last one ragged, conditional flow
    (400, 10, [256], 16, 0, 'ELU'),             # eight tiles (4 + 4)
    (300, 12, [64, 190], 5, 0, 'Tanh'),         # six tiles (3 + 3) behind a deeper conditioner
])
def test_spline_slab_backward_matches_per_row_parameter_path(monkeypatch, n, dim, hidden, K, latent_dim, act, stype):
    """sx_rqs_slab_bwd (spline backward fused with the last conditioner layer: no [N, n_live*(3K-1)] tensor) against the
    layer-wise path it replaces (torch Linear -> sx_rqs_coupling / sx_rqs_inverse_bwd -> library GEMMs), same weights."""
    torch.manual_seed(21)
    P = 3 * K - 1 if stype == 'quadratic' else 2 * K + 2
    tr = [st.Coupling(st.Spline(dim, K, latent_net=st.net.MLP(dim + latent_dim, hidden, dim * P, activation=act), lower=-2.5,
                                upper=2.5, spline_type=stype),
                      mask='ordered_right_half' if i % 2 == 0 else 'parity_even') for i in range(2)]
    flow = st.NormalizingFlow(st.UnitNormal(dim), tr).to(DEV)
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(torch.randn_like(p) * 0.05)       # the last layer's bias starts at zero (mlp.py:53)
    x = torch.randn(n, dim, device=DEV) * 1.5
    lat = torch.randn(n, latent_dim, device=DEV) if latent_dim else None
    wgt = torch.rand(n, 1, device=DEV) + 0.5

    def grads():
        for p in flow.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        lp = flow.log_prob(xg, latent=lat)
        (-(lp * wgt).mean()).backward()
        return lp.detach(), [xg.grad.clone()] + [p.grad.clone() for p in flow.parameters()]

    lp_f, fused = grads()
    monkeypatch.setenv('STRIBOR_SPLINE_UNFUSED', '1')
    lp_u, unfused = grads()
    # (cubic: the slab op's forward returns the reference's inverse log-det -- minus the forward log-det re-evaluated at the
    #  inverted point, flow.py:42-47 -- and CubicInverse the inverse pass's own: they differ for elements on a bound or knot)
    if stype == 'quadratic':
        close(lp_f, lp_u, rtol=1e-5, atol=1e-4)
    else:
        d = (lp_f - lp_u).abs().flatten()
        assert (d > 1e-4 + 1e-5 * lp_u.abs().flatten()).float().mean().item() < 0.02 and d.max().item() < 5e-2
    names = ['x'] + [k for k, _ in flow.named_parameters()]
    for name, a, b in zip(names, fused, unfused):
        scale = b.abs().max().item() + 1e-12
        # the two paths round the parameters differently (fp16 x 3 MFMA vs the library's fp32 GEMM) and the spline's
        # gradient amplifies that: same bounds as the fp64-oracle tests (cubic: 1 / w^2 of bins as narrow as 1e-2)
        tol = 3e-4 if stype == 'quadratic' else 1e-3
        d = (a - b).abs()
        # (an input within rounding of a knot may fall into neighbouring bins on the two paths -- the spline's derivative jumps there:
        #  with 2,100 x 32 elements per layer one such element shows up; the fp64-oracle test of the same widths holds 2e-4 on every one)
        over = (d > tol * scale + 1e-8).float().mean().item()
        # (beyond 128 hidden units the two paths also sum 160 .. 256 products in different orders and arithmetics: up to 1e-3 of the
        #  elements sit between 1 x and 2 x the bound; against fp64 autograd of the oracle the same widths hold 3e-4 everywhere)
        assert over <= (1e-3 if max(hidden) > 128 else 1e-4) and d.max().item() <= 10 * tol * scale + 1e-8, (name, d.max().item(), scale, over)


def test_spline_slab_backward_at_scale_is_additive_over_row_partitions():
    """65,537 rows of the cfg-3 layer shape: 32 row ranges x 8 slab pairs (XCD-aware workgroup ids, 16-17 passes per wave, a
    ragged last chunk) -- the partitioning the small cases do not reach.  With loss = sum over rows the parameter gradients of
    the whole batch equal the sum of those of two unequal parts, which the kernel partitions differently (per-row arithmetic
    is identical in both runs, so bin choices are too: the bound is fp32 summation order, not the spline's conditioning --
    against fp64 or the per-row parameter path single inputs within rounding of a knot flip bins and move gradients by 1e-2 at
    this size, `tools/diag_slab.py 0.0 65537`).  Input gradients must agree row by row."""
    torch.manual_seed(23)
    flow = fd.build_flow(st, fd.cfg3_desc(2, 64, 64, 16), 64).to(DEV)
    n, n1 = (1 << 16) + 1, 20011
    x = torch.randn(n, 64, device=DEV) * 1.3

    def grads(rows):
        for p in flow.parameters():
            p.grad = None
        xg = rows.clone().requires_grad_(True)
        (-flow.log_prob(xg).sum() * 1e-4).backward()
        return xg.grad.clone(), [p.grad.clone() for p in flow.parameters()]

    gx, whole = grads(x)
    gx1, part1 = grads(x[:n1])
    gx2, part2 = grads(x[n1:])
    assert torch.isfinite(gx).all()
    gxp = torch.cat([gx1, gx2])
    # (not bit-identical: the power-of-two adjoint scale follows each call's largest gradient, which moves where the fp16 x 3
    #  operands' low parts go subnormal)
    assert (gx - gxp).abs().max().item() <= 2e-4 * gx.abs().max().item()      # (1.0e-4 measured: the scale is taken from sampled rows)
    for (name, _), a, b, c in zip(flow.named_parameters(), whole, part1, part2):
        scale = a.abs().max().item() + 1e-12
        assert (a - (b + c)).abs().max().item() <= 1e-4 * scale, (name, (a - (b + c)).abs().max().item(), scale)


@pytest.mark.parametrize('hidden', [64, 160])
def test_spline_slab_backward_in_row_blocks_under_a_scratch_budget(monkeypatch, hidden):
    """the slab backward's dh-partial scratch grows with the rows of one call (10.7 GB at 2^20 rows and 160 hidden
    units).  Under a budget the op cuts the batch into row blocks (`flows/spline.py: _slab_row_blocks`): same gradients as the
    one-call form -- hidden 64 goes through RQSCouplingSlabL1 (first layer inside the op), hidden 160 through RQSCouplingSlab --,
    and the library scratch stays under the budget."""
    import stribor_amd.flows.spline as spl
    from stribor_amd import _hip
    torch.manual_seed(5)
    flow = fd.build_flow(st, fd.cfg3_desc(2, 64, hidden, 16), 64).to(DEV)
    n = (1 << 16) + 77                  # (blocks are never cut below 8192 rows)
    x = torch.randn(n, 64, device=DEV) * 1.2

    def grads():
        for p in flow.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        (-flow.log_prob(xg).sum() * 1e-4).backward()
        return xg.grad.clone(), [p.grad.clone() for p in flow.parameters()]

    gx, whole = grads()
    lib = _hip.lib()
    one_call = lib.sx_rqs_slab_scratch_floats(n, 32, hidden) * 4
    monkeypatch.setattr(spl, '_SLAB_SCRATCH_BUDGET', one_call // 3)
    blocks = spl._slab_row_blocks(lib, n, 32, hidden)
    assert len(blocks) >= 3 and blocks[0][0] == 0 and blocks[-1][1] == n and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    assert all(lib.sx_rqs_slab_scratch_floats(r1 - r0, 32, hidden) * 4 <= one_call // 3 for r0, r1 in blocks)
    _hip._scratch.clear()
    gxb, parts = grads()
    # (the same per-stream block also serves the first-layer kernel and the weight-gradient reductions: + 16 MB)
    assert max(t.numel() for t in _hip._scratch.values()) * 4 <= max(one_call // 3, 4 << 20) + (16 << 20)
    # per-row arithmetic is the same in both runs (the adjoint scale is the whole batch's in both): input gradients agree row by
    # row; parameter gradients up to fp32 summation order
    assert (gx - gxb).abs().max().item() <= 1e-6 * gx.abs().max().item()
    for (name, _), a, b in zip(flow.named_parameters(), whole, parts):
        scale = a.abs().max().item() + 1e-12
        assert (a - b).abs().max().item() <= 1e-4 * scale, (name, (a - b).abs().max().item(), scale)


def test_spline_slab_backward_reports_fp16_range_and_exact_mode_bypasses_it():
    """The slab backward's GEMM operands are fp16 x 3: a hidden activation beyond 65504 must not come back as a plausible
    gradient (NaN rows + GemmRangeError at the next check), and set_gemm_precision('exact') takes the per-row path."""
    torch.manual_seed(5)
    dim, K = 8, 6
    P = 3 * K - 1

    def make():
        net = st.net.MLP(dim, [16], dim * P, activation='ReLU')
        return st.NormalizingFlow(st.UnitNormal(dim), [st.Coupling(st.Spline(dim, K, latent_net=net, lower=-3, upper=3,
                                                                           spline_type='quadratic'), mask='ordered_right_half')]).to(DEV)
    flow = make()
    with torch.no_grad():
        lin0 = flow.transforms[0].transform.latent_net.net[0]
        lin0.weight.mul_(1e6)                              # ReLU hidden activations ~1e6
        flow.transforms[0].transform.latent_net.net[2].weight.mul_(1e-7)
    x = torch.randn(200, dim, device=DEV).requires_grad_(True)
    st.set_gemm_precision('exact')
    try:
        lp = flow.log_prob(x)
        (-lp.mean()).backward()
        torch.cuda.synchronize()
        st.check_errors()
        g_exact = [p.grad.clone() for p in flow.parameters()]
        assert all(torch.isfinite(g).all() for g in g_exact)
    finally:
        st.set_gemm_precision('fast')
    for p in flow.parameters():
        p.grad = None
    x2 = x.detach().clone().requires_grad_(True)
    with pytest.raises(st.GemmRangeError):
        lp = flow.log_prob(x2)
        (-lp.mean()).backward()
        torch.cuda.synchronize()
        st.check_errors()


def test_spline_flow_training_step_reduces_loss():
    torch.manual_seed(0)
    flow = st.NormalizingFlow(st.UnitNormal(8), [
        st.Coupling(st.Spline(8, 8, latent_net=st.net.MLP(8, [32], 8 * 23), lower=-4, upper=4, spline_type='quadratic'),
                    mask='ordered_right_half' if i % 2 == 0 else 'ordered_left_half') for i in range(4)] +
        [st.Flip([-1])]).to(DEV)
    opt = torch.optim.Adam(flow.parameters(), lr=3e-3)
    data = torch.randn(4096, 8, device=DEV) * 0.6 + 0.5
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = -flow.log_prob(data).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 0.3, (losses[0], losses[-1])
    with torch.no_grad():
        assert torch.isfinite(flow.log_prob(data)).all()


def test_backward_blocks_over_the_batch(monkeypatch):
    """The scratch for the per-row gradient factors is bounded: walking the batch in blocks gives the same grads."""
    from stribor_amd.flow import _FusedLogProb
    torch.manual_seed(9)
    flow = fd.build_flow(st, fd.cfg2_desc(3, 64, 64), 64).to(DEV)
    x = torch.randn(3000, 64, device=DEV)

    def grads():
        for p in flow.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        (-flow.log_prob(xg).mean()).backward()
        return [p.grad.clone() for p in flow.parameters()] + [xg.grad.clone()]

    whole = grads()
    monkeypatch.setattr(_FusedLogProb, 'SIDE_BYTES', 3 * 224 * 4 * 700)          # 700-row blocks, ragged tail
    blocked = grads()
    for a, b in zip(whole, blocked):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_backward_through_flips_realnvp_style():
    """RealNVP pattern: the same mask everywhere, Flip between couplings (slot relabelling in the backward plan)."""
    torch.manual_seed(13)
    c = lambda: {'kind': 'coupling_affine', 'dim': 64, 'hidden': [48], 'mask': 'ordered_right_half', 'latent_dim': 0}
    desc = [c(), {'kind': 'flip'}, c(), {'kind': 'flip'}, c(), {'kind': 'flip'}]
    flow = fd.build_flow(st, desc, 64)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(500, 64)
    want_loss, want_g, want_gx = oracle_grads(desc, state, x)
    xg = x.to(DEV).requires_grad_(True)
    lp = flow.log_prob(xg)
    assert lp.requires_grad
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss)
    close(xg.grad, want_gx.float(), rtol=1e-4, atol=1e-6)
    for name, p in flow.named_parameters():
        ref = want_g[name].float()
        assert (p.grad.cpu() - ref).abs().max().item() <= 2e-4 * (ref.abs().max().item() + 1e-12) + 1e-7, name


@pytest.mark.parametrize('dim,mask,hidden', [(64, 'parity_even', 40), (10, 'ordered_left_half', 13), (2, 'ordered_right_half', 64),
                                              (5, 'parity_odd', 13), (40, 'ordered_right_half', 32)])
def test_backward_dense_masks_and_small_dims(dim, mask, hidden):
    """Masks that do not align with the 32-column tiles (parity_*) and widths below 64 (the reference suite's own
    shapes, stribor/test/test_coupling.py:7 with check_gradients_not_nan) take the dense backward variant."""
    torch.manual_seed(17)
    other = {'ordered_left_half': 'ordered_right_half', 'ordered_right_half': 'ordered_left_half',
             'parity_even': 'parity_odd', 'parity_odd': 'parity_even'}[mask]
    desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': m, 'latent_dim': 0}
            for m in (mask, other, mask)]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(333, dim)
    want_loss, want_g, want_gx = oracle_grads(desc, state, x)
    xg = x.to(DEV).requires_grad_(True)
    lp = flow.log_prob(xg)
    assert lp.requires_grad
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss)
    close(xg.grad, want_gx.float(), rtol=1e-4, atol=1e-6)
    for name, p in flow.named_parameters():
        ref = want_g[name].float()
        assert not torch.isnan(p.grad).any()                                     # base.py:76-81
        assert (p.grad.cpu() - ref).abs().max().item() <= 2e-4 * (ref.abs().max().item() + 1e-12) + 1e-7, name


def test_layerwise_backward_conditional_and_mixed_flows():
    """Layer-wise training path (AffineCouplingOp / RQSInverse + torch conditioners): a conditional affine flow
    (latent input, deep conditioner), and a flow mixing affine couplings, spline couplings, Permute and Flip --
    input, latent and parameter gradients against fp64 autograd of the oracle."""
    torch.manual_seed(21)
    cases = [
        ('conditional affine', [{'kind': 'coupling_affine', 'dim': 6, 'hidden': [16, 16], 'latent_dim': 3,
                                 'mask': 'ordered_right_half' if i % 2 == 0 else 'parity_odd'} for i in range(3)], 6, 3),
        ('mixed', [{'kind': 'coupling_affine', 'dim': 10, 'hidden': [24], 'latent_dim': 0, 'mask': 'ordered_left_half'},
                   {'kind': 'permute', 'dim': 10},
                   {'kind': 'coupling_rqs', 'dim': 10, 'hidden': [24], 'n_bins': 6, 'lower': -3, 'upper': 3,
                    'mask': 'parity_even', 'latent_dim': 0},
                   {'kind': 'flip'},
                   {'kind': 'coupling_affine', 'dim': 10, 'hidden': [24], 'latent_dim': 0, 'mask': 'parity_odd'}], 10, 0),
    ]
    for name, desc, dim, ld in cases:
        flow = fd.build_flow(st, desc, dim)
        state = {k: v.clone() for k, v in flow.state_dict().items()}
        for i, (d, f) in enumerate(zip(desc, flow.transforms)):
            if d['kind'] == 'permute':
                state[f'transforms.{i}.permutation'] = f.permutation.clone()
        flow = flow.to(DEV)
        n = 200
        x = torch.randn(n, dim) * 1.3
        lat = torch.randn(n, ld) if ld else None
        leaves = {k: (v.detach().double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in state.items()}
        spec = fd.flow_spec(desc, leaves)
        xin = x.double().clone().requires_grad_(True)
        lin = None if lat is None else lat.double().clone().requires_grad_(True)
        want_loss = -orc.flow_log_prob(spec, xin, latent=lin).mean()
        want_loss.backward()
        xg = x.to(DEV).requires_grad_(True)
        lg = None if lat is None else lat.to(DEV).requires_grad_(True)
        lp = flow.log_prob(xg) if lg is None else flow.log_prob(xg, latent=lg)
        assert lp.requires_grad, name
        loss = -lp.mean()
        loss.backward()
        assert abs(loss.item() - want_loss.item()) <= 1e-5 * abs(want_loss.item()) + 1e-5, name
        tol = lambda ref: 3e-4 * ref.abs().max().item() + 1e-7
        assert (xg.grad.cpu() - xin.grad.float()).abs().max().item() <= tol(xin.grad), name
        if lg is not None:
            assert (lg.grad.cpu() - lin.grad.float()).abs().max().item() <= tol(lin.grad), name
        for pname, p in flow.named_parameters():
            ref = leaves[pname].grad.float()
            assert (p.grad.cpu() - ref).abs().max().item() <= tol(ref), (name, pname)


def test_layerwise_backward_elementwise_affine_and_spline_layers():
    """Element-wise st.Affine / st.Spline layers (own parameters, or a latent_net fed by `latent`) in a trainable flow."""
    torch.manual_seed(31)
    dim, ld, n = 6, 4, 150
    desc = [{'kind': 'affine', 'dim': dim},
            {'kind': 'coupling_affine', 'dim': dim, 'hidden': [16], 'latent_dim': ld, 'mask': 'ordered_left_half'},
            {'kind': 'rqs', 'dim': dim, 'n_bins': 4, 'lower': -3, 'upper': 3, 'hidden': [12], 'latent_dim': 0},
            {'kind': 'affine_latent', 'dim': dim, 'hidden': [10], 'latent_dim': ld},
            {'kind': 'rqs', 'dim': dim, 'n_bins': 5, 'lower': -4, 'upper': 4, 'hidden': [12], 'latent_dim': ld},
            {'kind': 'flip'}]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x, lat = torch.randn(n, dim) * 1.2, torch.randn(n, ld)
    leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items()}
    spec = fd.flow_spec(desc, leaves)
    xin, lin = x.double().clone().requires_grad_(True), lat.double().clone().requires_grad_(True)
    want = -orc.flow_log_prob(spec, xin, latent=lin).mean()
    want.backward()
    xg, lg = x.to(DEV).requires_grad_(True), lat.to(DEV).requires_grad_(True)
    lp = flow.log_prob(xg, latent=lg)
    assert lp.requires_grad
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item()) + 1e-5
    tol = lambda ref: 3e-4 * ref.abs().max().item() + 1e-7
    assert (xg.grad.cpu() - xin.grad.float()).abs().max().item() <= tol(xin.grad)
    assert (lg.grad.cpu() - lin.grad.float()).abs().max().item() <= tol(lin.grad)
    for pname, p in flow.named_parameters():
        ref = leaves[pname].grad.float()
        assert p.grad is not None and (p.grad.cpu() - ref).abs().max().item() <= tol(ref), pname


def test_layerwise_backward_dense_linear_layers():
    """cfg-4 family (AffineLU + MatrixExponential with / without bias and log-time + affine couplings) trains through the
    layer-wise path; gradients vs fp64 autograd of the oracle."""
    torch.manual_seed(41)
    dim, n = 8, 120
    desc = [{'kind': 'affine_lu', 'dim': dim},
            {'kind': 'coupling_affine', 'dim': dim, 'hidden': [16], 'latent_dim': 0, 'mask': 'ordered_right_half'},
            {'kind': 'matrix_exp', 'dim': dim, 'bias': True, 'log_time': False},
            {'kind': 'coupling_affine', 'dim': dim, 'hidden': [16], 'latent_dim': 0, 'mask': 'ordered_left_half'},
            {'kind': 'matrix_exp', 'dim': dim, 'bias': False, 'log_time': True}]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(n, dim)
    want_loss, want_g, want_gx = oracle_grads(desc, state, x)
    xg = x.to(DEV).requires_grad_(True)
    lp = flow.log_prob(xg)
    assert lp.requires_grad
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss) + 1e-5
    tol = lambda ref: 3e-4 * ref.abs().max().item() + 1e-7
    assert (xg.grad.cpu() - want_gx.float()).abs().max().item() <= tol(want_gx)
    for pname, p in flow.named_parameters():
        ref = want_g[pname].float()
        assert p.grad is not None and (p.grad.cpu() - ref).abs().max().item() <= tol(ref), pname


@pytest.mark.parametrize('n,dim,hidden,K,layers,masks', [
    (300, 8, 16, 5, 2, ('ordered_right_half', 'ordered_left_half')),
    (129, 64, 64, 16, 2, ('ordered_right_half', 'ordered_left_half')),
    (100, 10, 12, 3, 3, ('parity_even', 'parity_odd', 'ordered_left_half')),
    (65, 5, 12, 1, 2, ('ordered_right_half', 'parity_odd')),
])
def test_cubic_spline_flow_log_prob_backward_matches_autograd_of_oracle(n, dim, hidden, K, layers, masks):
    """Training of cubic-spline (the reference's default spline_type) coupling flows: sx_cubic_inverse_bwd differentiates
    the cubic solve implicitly; truth = fp64 autograd THROUGH the oracle's Cardano / trigonometric root formulas."""
    torch.manual_seed(13)
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -3, 'upper': 3,
             'mask': masks[i % len(masks)], 'latent_dim': 0, 'spline_type': 'cubic'} for i in range(layers)]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(n, dim) * 1.6
    want_loss, want_g, want_gx = oracle_grads(desc, state, x)
    xg = x.to(DEV).requires_grad_(True)
    lp = flow.log_prob(xg)
    assert lp.requires_grad and lp.shape == (n, 1)
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss) + 1e-5
    sx = want_gx.abs().max().item()
    # (round-2 note: this bound was 1e-3 "because the cubic's coefficients carry 1 / w^2" -- it was the quadratic-fallback
    #  branch's gradient, test_cubic_coupling_gradients_in_the_quadratic_fallback_branch; now the spline tests' 3e-4)
    assert (xg.grad.cpu() - want_gx.float()).abs().max().item() <= 3e-4 * sx + 1e-7
    for name, p in flow.named_parameters():
        ref = want_g[name].float()
        scale = ref.abs().max().item() + 1e-12
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err <= 3e-4 * scale + 1e-7, (name, err, scale)


def test_layerwise_backward_reference_stack_and_pointwise_flows():
    """The on-path part of test_normalizing_flow.py's stack (affine coupling with a two-hidden-layer conditioner -> Flip ->
    Sigmoid -> cubic-spline coupling -> Logit) trains end to end, and so do ELU / LeakyReLU / Cumsum / Diff / Identity
    layers: gradients vs fp64 autograd of the oracle."""
    torch.manual_seed(51)
    dim = 2
    stack = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [32, 64], 'mask': 'ordered_1', 'latent_dim': 0},
             {'kind': 'flip'}, {'kind': 'sigmoid'},
             {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [32, 64], 'mask': 'ordered_0', 'latent_dim': 0, 'n_bins': 5,
              'lower': 0, 'upper': 1, 'spline_type': 'cubic'},
             {'kind': 'logit'}]
    d6 = 6
    zoo = [{'kind': 'coupling_affine', 'dim': d6, 'hidden': [16], 'mask': 'ordered_right_half', 'latent_dim': 0},
           {'kind': 'leaky_relu', 'negative_slope': 0.3}, {'kind': 'cumsum'},
           {'kind': 'coupling_affine', 'dim': d6, 'hidden': [16], 'mask': 'parity_odd', 'latent_dim': 0},
           {'kind': 'diff'}, {'kind': 'identity'},
           {'kind': 'coupling_rqs', 'dim': d6, 'hidden': [16], 'mask': 'ordered_left_half', 'latent_dim': 0, 'n_bins': 4,
            'lower': -3, 'upper': 3},
           {'kind': 'elu'}]            # log_prob inverts ELU first: its argument must stay above -1
    for name, desc, dd, x in [('stack', stack, dim, torch.randn(90, dim)), ('zoo', zoo, d6, torch.rand(120, d6) * 2.5 - 0.6)]:
        flow = fd.build_flow(st, desc, dd)
        state = {k: v.clone() for k, v in flow.state_dict().items()}
        flow = flow.to(DEV)
        want_loss, want_g, want_gx = oracle_grads(desc, state, x)
        xg = x.to(DEV).requires_grad_(True)
        lp = flow.log_prob(xg)
        assert lp.requires_grad, name
        loss = -lp.mean()
        loss.backward()
        assert abs(loss.item() - want_loss) <= 2e-5 * abs(want_loss) + 1e-5, name
        tol = lambda ref: 1e-3 * ref.abs().max().item() + 1e-7
        assert (xg.grad.cpu() - want_gx.float()).abs().max().item() <= tol(want_gx), name
        for pname, p in flow.named_parameters():
            ref = want_g[pname].float()
            assert p.grad is not None and (p.grad.cpu() - ref).abs().max().item() <= tol(ref), (name, pname)


def test_neural_flow_forward_is_differentiable():
    """ContinuousAffineCoupling inside NeuralFlow with a graph (conditioner + time net through torch, affine map through
    AffineCouplingOp): gradients of an MSE objective on f(x, t) w.r.t. x, t-net scales and conditioner weights vs fp64
    autograd of the oracle; values equal the kernel path's."""
    torch.manual_seed(61)
    dim, n = 4, 80
    desc = [{'kind': 'continuous_affine_coupling', 'dim': dim, 'hidden': [16], 'mask': m, 'latent_dim': 0, 'time_kind': tk}
            for m, tk in (('ordered_left_half', 'linear'), ('ordered_right_half', 'tanh'), ('parity_odd', 'log'))]
    nf = st.NeuralFlow([fd.build_transform(st, d) for d in desc])
    state = {k: v.clone() for k, v in nf.state_dict().items()}
    nf = nf.to(DEV)
    x, t, target = torch.randn(n, dim), torch.rand(n, 1) * 2, torch.randn(n, dim)
    leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items()}
    spec = [fd.transform_spec(d, leaves, f'transforms.{i}.') for i, d in enumerate(desc)]
    xin = x.double().clone().requires_grad_(True)
    want = ((orc.neural_flow_forward(spec, xin, t.double()) - target.double()) ** 2).mean()
    want.backward()
    xg = x.to(DEV).requires_grad_(True)
    out = nf(xg, t=t.to(DEV))
    with torch.no_grad():
        close(out, nf(x.to(DEV), t=t.to(DEV)), rtol=1e-5, atol=1e-5)
    loss = ((out - target.to(DEV)) ** 2).mean()
    loss.backward()
    assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item()) + 1e-6
    tol = lambda ref: 3e-4 * ref.abs().max().item() + 1e-7
    assert (xg.grad.cpu() - xin.grad.float()).abs().max().item() <= tol(xin.grad)
    for pname, p in nf.named_parameters():
        ref = leaves[pname].grad.float()
        assert p.grad is not None and (p.grad.cpu() - ref).abs().max().item() <= tol(ref), pname


@pytest.mark.parametrize('n,M,Nc,width,a0,b0', [(1, 64, 64, 224, 160, 32), (1000, 50, 33, 224, 96, 0),
                                                (4097, 128, 128, 256, 0, 128), (77, 96, 40, 224, 32, 128),
                                                (100003, 64, 32, 224, 96, 0),
                                                # several groups per wave: the steady state of the LDS-DMA prefetch and its counted
                                                # waits (fp16 x 3 form), 128 x 128 / 128 x 64 / 96 x 128 outputs
                                                (70001, 128, 128, 256, 0, 128), (70001, 128, 64, 320, 192, 64), (66000, 96, 128, 320, 0, 160)])
@pytest.mark.parametrize('f16', [False, True])
def test_wgrad_contracts_row_groups_with_maps(n, M, Nc, width, a0, b0, f16):
    """sx_wgrad on its own: feature-major 32-row groups (garbage in the ragged tail), feature counts that are not
    multiples of 32, row / column maps with dropped entries, accumulation into non-zero dW / db; fp64 truth.  f16: the same
    contraction on the matrix pipe (SX_WGRAD_ROW_GROUPS_F16X3, operands split hi + lo in fp16)."""
    from stribor_amd import _hip
    layout = _hip.WGRAD_ROW_GROUPS_F16X3 if f16 else _hip.WGRAD_ROW_GROUPS
    g = torch.Generator(device='cpu').manual_seed(n + M)
    ng = (n + 31) // 32
    side = torch.randn(ng, width, 32, generator=g)
    rows = side.permute(0, 2, 1).reshape(ng * 32, width)[:n]                  # [n, width] view of the valid rows
    dirty = side.clone()                                                      # rows past n hold garbage
    if n % 32:
        dirty[-1, :, n % 32:] = float('nan')
    row_map = torch.randperm(M, generator=g).to(torch.int32)
    row_map[::7] = -1
    col_map = torch.randperm(Nc, generator=g).to(torch.int32)
    col_map[1::5] = -1
    dW0, db0 = torch.randn(M, Nc, generator=g), torch.randn(M, generator=g)
    A, B = rows[:, a0:a0 + M].double(), rows[:, b0:b0 + Nc].double()
    full, fb = A.T @ B, A.sum(0)
    wantW, wantb = dW0.double().clone(), db0.double().clone()
    for i in range(M):
        if row_map[i] < 0:
            continue
        wantb[row_map[i]] += fb[i]
        for j in range(Nc):
            if col_map[j] >= 0:
                wantW[row_map[i], col_map[j]] += full[i, j]
    sd = dirty.to(DEV)
    dW, db = dW0.to(DEV), db0.to(DEV)
    rm, cm = row_map.to(DEV), col_map.to(DEV)
    rc = _hip.lib().sx_wgrad(sd[0, a0].data_ptr(), width * 32, M, sd[0, b0].data_ptr(), width * 32, Nc, n,
                             layout, dW.data_ptr(), Nc, db.data_ptr(), rm.data_ptr(), cm.data_ptr(),
                             _hip.scratch(DEV, _hip.lib().sx_wgrad_scratch_floats(M, Nc, layout)).data_ptr(), _hip.stream())
    _hip.check(rc, 'sx_wgrad')
    scale = max(1.0, float(n) ** 0.5)
    assert (dW.cpu().double() - wantW).abs().max().item() <= 2e-5 * scale
    assert (db.cpu().double() - wantb).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize('prec', ['f16x3', 'f32'])
def test_pack_table_writes_what_the_single_packs_write(prec):
    """sx_pack_linear_batch (one launch for every Linear of a program after an optimizer step) against sx_pack_linear /
    sx_pack_linear_bound one by one: plain, transposed, scaled rows + folded bias, with a bound slot, ragged maps with dropped
    slots, different tile shapes in one table.  Bit-identical blobs, flag word and bounds; the table is re-used unchanged on the
    second run and re-uploaded when a record changes."""
    from stribor_amd import _hip
    code = _hip.GEMM_F16X3 if prec == 'f16x3' else _hip.GEMM_F32
    g = torch.Generator(device='cpu').manual_seed(5)
    specs = [  # out, in, m_tiles, k_tiles, transpose, scaled, bound
        (64, 32, 2, 1, 0, False, False), (94, 64, 3, 2, 0, True, True), (64, 47, 2, 2, 1, False, False),
        (128, 128, 4, 4, 0, True, False), (20, 10, 1, 1, 0, False, True), (1504, 64, 4, 2, 0, True, True)]
    total = sum(_hip.packed_linear_floats(m, k) + 4 for (_, _, m, k, _, _, _) in specs)
    keep, records, singles, off = [], [], [], 1
    blob_a = torch.zeros(total + 1, dtype=torch.float32, device=DEV)
    blob_b = torch.zeros(total + 1, dtype=torch.float32, device=DEV)
    for (out, inn, m, k, tr, scaled, bound) in specs:
        W = torch.randn(out, inn, generator=g).to(DEV)
        W[0, 0] = 7.0e4 if (out, inn) == (64, 32) else W[0, 0]                  # one weight beyond fp16: the flag word
        b = None if tr else torch.randn(out, generator=g).to(DEV)
        n_rows, n_cols = (inn, out) if tr else (out, inn)                       # transposed: row slots index W's columns
        ri = torch.full((32 * m,), -1, dtype=torch.int32)
        sel = torch.randperm(n_rows, generator=g)[:min(n_rows, 32 * m)].to(torch.int32)
        ri[torch.randperm(32 * m, generator=g)[:sel.numel()]] = sel
        ci = torch.full((32 * k,), -1, dtype=torch.int32)
        selc = torch.randperm(n_cols, generator=g)[:min(n_cols, 32 * k)].to(torch.int32)
        ci[torch.randperm(32 * k, generator=g)[:selc.numel()]] = selc
        ri, ci = ri.to(DEV), ci.to(DEV)
        rs = (torch.rand(32 * m, generator=g) + 0.5).to(DEV) if scaled else None
        bs = (torch.rand(32 * m, generator=g) + 0.5).to(DEV) if scaled else None
        fold = 1.0 if scaled else 0.0
        nlin = _hip.packed_linear_floats(m, k)
        keep += [W, b, ri, ci, rs, bs]
        rec = lambda blob: (W.data_ptr(), _hip.ptr(b) or 0, out, inn, ri.data_ptr(), ci.data_ptr(), m, k, _hip.ptr(rs) or 0,
                            _hip.ptr(bs) or 0, fold, tr, blob.data_ptr() + 4 * off, (blob.data_ptr() + 4 * (off + nlin)) if bound else 0)
        records.append(rec(blob_a))
        singles.append(rec(blob_b))
        off += nlin + 4
    table = _hip.PackTable()
    table.run(blob_a, records, code, blob_a.data_ptr())
    first_table = table._table
    for r in singles:
        name = 'sx_pack_linear_bound' if r[13] else 'sx_pack_linear'
        args = [r[0], r[1] or None, r[2], r[3], r[4], r[5], r[6], r[7], r[8] or None, r[9] or None, r[10], r[11], code, blob_b.data_ptr(),
                r[12]] + ([r[13]] if r[13] else [])
        _hip.call(name, blob_b, *args)
    torch.cuda.synchronize()
    assert torch.equal(blob_a.view(torch.int32), blob_b.view(torch.int32))
    assert (int(blob_a[:1].view(torch.int32).item()) & _hip.FLAG_F16_RANGE != 0) == (prec == 'f16x3')
    table.run(blob_a, records, code, blob_a.data_ptr())
    assert table._table is first_table                                           # unchanged records: no upload
    blob_c = torch.zeros_like(blob_a)
    moved = [r[:12] + (r[12] - blob_a.data_ptr() + blob_c.data_ptr(), (r[13] - blob_a.data_ptr() + blob_c.data_ptr()) if r[13] else 0)
             for r in records]
    table.run(blob_c, moved, code, blob_c.data_ptr())
    torch.cuda.synchronize()
    assert table._table is not first_table and torch.equal(blob_c.view(torch.int32), blob_b.view(torch.int32))


def test_wgrad_reduce_table_adds_what_the_single_reductions_add():
    """sx_wgrad_reduce_batch (the 2 L reductions that close a layer-major backward pass, one launch) against sx_wgrad_reduce per
    job: different tile shapes, maps with dropped entries, a job without bias, accumulation into non-zero gradients."""
    import ctypes as C
    from stribor_amd import _hip
    lib = _hip.lib()
    g = torch.Generator(device='cpu').manual_seed(11)
    n_part = 37
    shapes = [(64, 64, 64, 50, True, False, True), (64, 32, 50, 32, False, True, True), (32, 32, 20, 31, True, True, False),
              (128, 32, 128, 32, False, False, True)]        # M32, N32, m_valid, n_valid, row map?, col map?, bias?
    part = torch.randn(sum(n_part * (M * N + M) for (M, N, *_rest) in shapes), generator=g).to(DEV)
    out_a = torch.randn(sum(128 * 64 + 128 for _ in shapes), generator=g).to(DEV)
    out_b = out_a.clone()
    jobs = (_hip.sx_reduce_job * len(shapes))()
    keep, po, oo = [], 0, 0
    for j, (M, N, mv, nv, rmap, cmap, bias) in zip(jobs, shapes):
        rm = torch.randperm(128, generator=g)[:M].to(torch.int32)
        rm[::5] = -1
        cm = torch.randperm(64, generator=g)[:N].to(torch.int32)
        cm[1::4] = -1
        rm, cm = rm.to(DEV), cm.to(DEV)
        keep += [rm, cm]
        (j.part_off, j.dW_off, j.db_off, j.ldw, j.row_map, j.col_map, j.M32, j.N32, j.m_valid, j.n_valid) = (
            po, oo, (oo + 128 * 64) if bias else -1, 64, rm.data_ptr() if rmap else None, cm.data_ptr() if cmap else None, M, N, mv, nv)
        _hip.check(lib.sx_wgrad_reduce(part.data_ptr() + 4 * po, n_part, M, N, out_b.data_ptr() + 4 * oo, 64,
                                       (out_b.data_ptr() + 4 * (oo + 128 * 64)) if bias else None, mv, nv,
                                       rm.data_ptr() if rmap else None, cm.data_ptr() if cmap else None, _hip.stream()), 'sx_wgrad_reduce')
        po += n_part * (M * N + M)
        oo += 128 * 64 + 128
    table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(DEV)
    _hip.check(lib.sx_wgrad_reduce_batch(part.data_ptr(), out_a.data_ptr(), table.data_ptr(), len(shapes), n_part,
                                         max(M * N + M for (M, N, *_r) in shapes), _hip.stream()), 'sx_wgrad_reduce_batch')
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)


@pytest.mark.parametrize('n,in_dim,out_dim', [(5000, 64, 64), (4097, 33, 50), (8192, 64, 1504), (6001, 128, 300)])
def test_batch_linear_weight_gradients_match_autograd(n, in_dim, out_dim):
    """BatchLinear (layer-wise training path): dL/dW, dL/db from sx_wgrad on row-major operands -- M up to 2048 in
    128-row slabs, ragged n, strided dL/dy -- against torch's own Linear backward in fp64."""
    from stribor_amd.net.mlp import BatchLinear
    torch.manual_seed(n)
    x = torch.randn(n, in_dim, device=DEV, requires_grad=True)
    W = torch.randn(out_dim, in_dim, device=DEV, requires_grad=True)
    b = torch.randn(out_dim, device=DEV, requires_grad=True)
    weight = torch.randn(n, out_dim, device=DEV)
    (BatchLinear.apply(x, W, b).tanh() * weight).sum().backward()
    xd, Wd, bd = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    (torch.nn.functional.linear(xd, Wd, bd).tanh() * weight.double()).sum().backward()
    scale = float(n) ** 0.5
    assert (W.grad.double() - Wd.grad).abs().max().item() <= 3e-5 * scale
    assert (b.grad.double() - bd.grad).abs().max().item() <= 3e-5 * scale
    assert (x.grad.double() - xd.grad).abs().max().item() <= 1e-3


@pytest.mark.parametrize('n,in_dim,out_dim', [(4096, 64, 64), (5000, 128, 256), (4099, 33, 7), (8192, 100, 130), (4096, 1, 1),
                                              # a spline conditioner's last layer (dL/dx contracts 1504 columns: 12 accumulating launches),
                                              # inputs beyond 128 columns, ragged everything
                                              (4100, 64, 1504), (4097, 200, 64), (4096, 300, 333)])
def test_batch_linear_forward_and_input_gradient_run_as_programs(n, in_dim, out_dim, monkeypatch):
    """Round 6: `BatchLinear`'s forward and dL/dx are fused-kernel launches (SX_STEP_MLP_INPUT + OUT_TILE steps,
    v_mfma_f32_32x32x2_f32; one launch per 128 contracted columns, the later ones adding into the output) instead of library GEMMs
    (net/mlp.py:48-58 layer by layer; flows/affine.py:157-163): no torch `linear` / `matmul` is called, values against fp64, rows
    of 1e6 included (the exact arithmetic has no operand range).  A second call after an in-place weight update re-packs the kept
    programs."""
    from stribor_amd.net.mlp import BatchLinear
    monkeypatch.setattr(BatchLinear, 'PROGRAM_MAX_IN', 2048)       # (the default keeps contractions beyond 128 columns with the library: slower here)
    torch.manual_seed(n + in_dim)
    x = torch.randn(n, in_dim, device=DEV)
    x[3] *= 1.0e6
    x.requires_grad_(True)
    W = torch.randn(out_dim, in_dim, device=DEV, requires_grad=True)
    b = torch.randn(out_dim, device=DEV, requires_grad=True)
    weight = torch.randn(n, out_dim, device=DEV)
    calls = []
    real_linear = torch.nn.functional.linear
    monkeypatch.setattr(torch.nn.functional, 'linear', lambda *a, **k: (calls.append('linear'), real_linear(*a, **k))[1])
    real_matmul = torch.Tensor.__matmul__
    monkeypatch.setattr(torch.Tensor, '__matmul__', lambda *a: (calls.append('matmul'), real_matmul(*a))[1])
    y = BatchLinear.apply(x, W, b)
    (y * weight).sum().backward()
    kept = len(BatchLinear._programs)
    with torch.no_grad():
        W.mul_(0.5)                                                                  # same storage, next version: the kept programs re-pack
    y2 = BatchLinear.apply(x.detach(), W, b)
    monkeypatch.undo()
    assert calls == (['matmul'] if out_dim > BatchLinear.MAX_OUT or in_dim > 128 else []), calls     # (a wide layer's dL/dW stays the library's: mlp.py)
    assert len(BatchLinear._programs) == kept
    y2d = torch.nn.functional.linear(x.detach().double(), W.detach().double(), b.detach().double())
    W.data.mul_(2.0)
    xd, Wd, bd = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    yd = torch.nn.functional.linear(xd, Wd, bd)
    (yd * weight.double()).sum().backward()
    row_scale = xd.detach().abs().amax(1, keepdim=True).clamp_min(1.0) * float(in_dim) ** 0.5
    assert ((y.double() - yd).abs() / row_scale).max().item() <= 2e-6
    assert ((y2.double() - y2d).abs() / row_scale).max().item() <= 2e-6
    assert (x.grad.double() - xd.grad).abs().max().item() <= 2e-5 * float(out_dim) ** 0.5
    st.check_errors()


@pytest.mark.parametrize('n,M,lda,Nc,ldb', [(4099, 50, 52, 33, 36), (1000, 300, 300, 128, 128), (31, 64, 64, 64, 64),
                                            (2050, 7, 7, 5, 5)])
def test_wgrad_row_major_layouts(n, M, lda, Nc, ldb):
    """sx_wgrad on row-major operands: the 16-byte path (aligned rows, feature counts that end inside a quad, 128-row
    slabs of a wide A) and the per-element path (odd strides), ragged n; fp64 truth."""
    from stribor_amd import _hip
    g = torch.Generator(device='cpu').manual_seed(n + M)
    Af, Bf = torch.randn(n, lda, generator=g), torch.randn(n, ldb, generator=g)
    want = Af[:, :M].double().T @ Bf[:, :Nc].double()
    wantb = Af[:, :M].double().sum(0)
    A, B = Af.to(DEV), Bf.to(DEV)
    dW, db = torch.zeros(M, Nc, device=DEV), torch.zeros(M, device=DEV)
    rc = _hip.lib().sx_wgrad(A.data_ptr(), lda, M, B.data_ptr(), ldb, Nc, n, _hip.WGRAD_ROW_MAJOR, dW.data_ptr(), Nc,
                             db.data_ptr(), None, None,
                             _hip.scratch(DEV, _hip.lib().sx_wgrad_scratch_floats(M, Nc, _hip.WGRAD_ROW_MAJOR)).data_ptr(), _hip.stream())
    _hip.check(rc, 'sx_wgrad')
    scale = max(1.0, float(n) ** 0.5)
    assert (dW.cpu().double() - want).abs().max().item() <= 2e-5 * scale
    assert (db.cpu().double() - wantb).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize('make,dim,n', [('cfg2', 64, 2048), ('cfg3', 64, 512), ('cfg4', 128, 512)])
def test_training_step_replays_from_a_hip_graph(make, dim, n):
    """Small-batch training is launch-bound: the whole step (forward, hand-written / layer-wise backward, parameter
    gradients) captured once into a HIP graph and replayed on new data gives the gradients of an eager step bit for bit
    (no allocation outside torch's graph pool, no synchronisation, scratch and counters created by the warm-up)."""
    torch.manual_seed(3)
    desc = {'cfg2': lambda: fd.cfg2_desc(4, dim, 64), 'cfg3': lambda: fd.cfg3_desc(2, dim, 64, 16),
            'cfg4': lambda: fd.cfg4_desc(1, dim, 64)}[make]()
    flow = fd.build_flow(st, desc, dim).to(DEV)
    params = list(flow.parameters())
    static_x = torch.randn(n, dim, device=DEV)

    def step():
        loss = -flow.log_prob(static_x).mean()
        return (loss.detach(),) + tuple(torch.autograd.grad(loss, params))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        static_out = step()
    for seed in range(2):
        fresh = torch.randn(n, dim, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
        static_x.copy_(fresh)
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in static_out]
        want = step()
        for a, b in zip(got, want):
            assert torch.equal(a, b), (make, seed)


@pytest.mark.parametrize('n,M,lda', [(1, 7, 7), (513, 1504, 1504), (100_003, 50, 52), (4096, 300, 300), (77, 1025, 1028)])
def test_colsum_matches_fp64(n, M, lda):
    from stribor_amd import _hip
    g = torch.Generator(device='cpu').manual_seed(n + M)
    A = torch.randn(n, lda, generator=g)
    base = torch.randn(M, generator=g)
    want = base.double() + A[:, :M].double().sum(0)
    Ad, out = A.to(DEV), base.to(DEV)
    _hip.check(_hip.lib().sx_colsum(Ad.data_ptr(), lda, n, M, out.data_ptr(), _hip.scratch(DEV, 256 * M).data_ptr(), _hip.stream()), 'sx_colsum')
    assert (out.cpu().double() - want).abs().max().item() <= 2e-6 * max(1.0, float(n) ** 0.5) * 4


@pytest.mark.parametrize('k,D', [(1, 1), (3, 7), (4, 128), (2, 65)])
def test_tri_inverse_f64_and_its_backward(k, D):
    """sx_tri_inverse_f64 (unit-lower and upper) against torch.linalg on the CPU in fp64, forward and gradient."""
    from stribor_amd.flows.linear import TriInverse
    g = torch.Generator(device='cpu').manual_seed(k * 1000 + D)
    W = torch.randn(k, D, D, generator=g, dtype=torch.float64) * 0.3
    eye = torch.eye(D, dtype=torch.float64)
    for lower, unit in ((True, True), (False, False)):
        def build(w):
            e = eye.to(w.device)
            return torch.tril(w, -1) + e if lower else torch.triu(w, 1) + e * (1.5 + torch.diagonal(w, dim1=-2, dim2=-1).tanh()).unsqueeze(-1)
        wc = W.clone().requires_grad_(True)
        Xc = torch.linalg.inv(build(wc))
        G = torch.randn(k, D, D, generator=g, dtype=torch.float64)
        (Xc * G).sum().backward()
        wd = W.clone().to(DEV).requires_grad_(True)
        Xd = TriInverse.apply(build(wd), lower, unit)
        (Xd * G.to(DEV)).sum().backward()
        scale = Xc.abs().max().item()
        assert (Xd.detach().cpu() - Xc.detach()).abs().max().item() <= 1e-10 * scale
        assert (wd.grad.cpu() - wc.grad).abs().max().item() <= 1e-9 * max(1.0, wc.grad.abs().max().item())


def test_layer_major_backward_matches_factor_path_and_oracle():
    """Round 2: the layer-major backward (sx_flow_bwd_run: one launch per coupling, weight gradients contracted in the
    kernel through MFMA-turned factor tiles, state streamed in fragment order) against the round-1 path (per-row factors
    in HBM + sx_wgrad_layer) on cfg-2 shaped flows -- ragged batch sizes (padded tail rows must contribute nothing),
    hidden 64 and 40, 1 / 3 / 8 layers -- and against fp64 autograd of the oracle."""
    import os
    from stribor_amd.flow import _layer_major_ok
    for layers, hidden, n in [(8, 64, 4096), (3, 40, 1000), (1, 64, 77), (2, 32, 128 * 300 + 5), (4, 64, 1 << 17)]:
        torch.manual_seed(layers)
        desc = fd.cfg2_desc(layers, 64, hidden)
        flow = fd.build_flow(st, desc, 64)
        spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
        flow = flow.to(DEV)
        x = torch.randn(n, 64)
        w = torch.rand(n, 1) + 0.5                                   # non-uniform dL/dlog_prob per row
        params = list(flow.parameters())

        def grads():
            xg = x.to(DEV).requires_grad_(True)
            loss = -(flow.log_prob(xg) * w.to(DEV)).sum() / n
            return torch.autograd.grad(loss, [xg] + params)

        bprog, lay = flow._backward_program(64, torch.device(DEV))
        assert _layer_major_ok(bprog, lay, 32 * bprog.prog.h_tiles)
        new = [t.detach().clone() for t in grads()]
        os.environ['STRIBOR_BWD_FACTORS'] = '1'
        try:
            old = [t.detach().clone() for t in grads()]
        finally:
            del os.environ['STRIBOR_BWD_FACTORS']
        for a, b in zip(new, old):
            scale = b.abs().max().clamp_min(1e-30)
            assert ((a - b).abs().max() / scale).item() <= 5e-5, (layers, hidden, n)
        # fp64 autograd of the oracle
        spec64 = orc.spec_to(spec, torch.float64)
        x64 = x.double().requires_grad_(True)
        loss64 = -(orc.flow_log_prob(spec64, x64) * w.double()).sum() / n
        gx64 = torch.autograd.grad(loss64, [x64])[0]
        # 2^17 rows of a mean loss: dL/dlog_prob ~ 1e-5 per row (the pass runs on a power-of-two multiple of it)
        assert ((new[0].cpu().double() - gx64).abs().max() / gx64.abs().max()).item() <= 2e-4, (layers, hidden, n)


def test_backward_is_accurate_for_tiny_loss_scales():
    """dL/dlog_prob of 1e-9 per row (a mean over a huge batch, or a down-weighted loss term): the fp16 x 3 backward GEMMs
    would see operands far below fp16's normal range; the pass rescales by a power of two, so the gradients keep their
    relative accuracy against fp64 autograd of the oracle."""
    torch.manual_seed(12)
    desc = fd.cfg2_desc(4, 64, 64)
    flow = fd.build_flow(st, desc, 64)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(2000, 64)
    for scale in (1e-9, 1.0, 3e4):
        for p in flow.parameters():
            p.grad = None
        xg = x.to(DEV).requires_grad_(True)
        (-flow.log_prob(xg).sum() * scale).backward()
        _, want_g, want_gx = oracle_grads(desc, state, x)
        k = scale * x.shape[0]                                 # oracle_grads uses the mean
        ref = want_gx.float() * k
        assert ((xg.grad.cpu() - ref).abs().max() / ref.abs().max()).item() <= 2e-4, scale
        for name, p in flow.named_parameters():
            ref = want_g[name].float() * k
            assert ((p.grad.cpu() - ref).abs().max() / (ref.abs().max() + 1e-30)).item() <= 2e-4, (name, scale)


def test_forward_and_inverse_directions_are_differentiable():
    """Round 2: forward / forward_and_log_det_jacobian / rsample / inverse(_and_log_det_jacobian) build autograd
    graphs like the reference's methods do (VI-style losses from reparametrised samples + log-dets): gradients against fp64
    autograd of the oracle.  Forward direction: affine couplings, Flip / Permute, element-wise Affine, point-wise flows,
    AffineLU / MatrixExponential; inverse direction: spline couplings as well."""
    torch.manual_seed(21)
    dim = 8
    desc = [{'kind': 'affine_lu', 'dim': dim},
            {'kind': 'coupling_affine', 'dim': dim, 'hidden': [16], 'mask': 'ordered_right_half', 'latent_dim': 0},
            {'kind': 'flip'},
            {'kind': 'matrix_exp', 'dim': dim, 'bias': True, 'log_time': False},
            {'kind': 'affine', 'dim': dim},
            {'kind': 'coupling_affine', 'dim': dim, 'hidden': [16], 'mask': 'parity_even', 'latent_dim': 0},
            {'kind': 'leaky_relu', 'negative_slope': 0.1}]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(300, dim)

    def oracle(direction):
        leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items()}
        spec = fd.flow_spec(desc, leaves)
        xin = x.double().clone().requires_grad_(True)
        y, ldj = (orc.flow_forward_and_ldj if direction == 'fwd' else orc.flow_inverse_and_ldj)(spec, xin)
        loss = (y ** 2).sum() * 0.1 + (ldj * torch.linspace(0.5, 1.5, x.shape[0]).double().unsqueeze(-1)).sum()
        loss.backward()
        return loss.item(), {k: v.grad for k, v in leaves.items()}, xin.grad

    w = torch.linspace(0.5, 1.5, x.shape[0]).unsqueeze(-1).to(DEV)
    for direction in ('fwd', 'inv'):
        for p in flow.parameters():
            p.grad = None
        xg = x.to(DEV).requires_grad_(True)
        y, ldj = flow.forward_and_log_det_jacobian(xg) if direction == 'fwd' else flow.inverse_and_log_det_jacobian(xg)
        assert y.requires_grad and ldj.requires_grad and ldj.shape == (300, 1)
        loss = (y ** 2).sum() * 0.1 + (ldj * w).sum()
        loss.backward()
        want_loss, want_g, want_gx = oracle(direction)
        assert abs(loss.item() - want_loss) <= 1e-4 * abs(want_loss) + 1e-3
        ref = want_gx.float()
        assert ((xg.grad.cpu() - ref).abs().max() / ref.abs().max()).item() <= 3e-4, direction
        for name, p in flow.named_parameters():
            assert p.grad is not None, (name, direction)
            ref = want_g[name].float()
            assert ((p.grad.cpu() - ref).abs().max() / (ref.abs().max() + 1e-12)).item() <= 3e-4, (name, direction)
    # values equal the fused (no-graph) kernels'
    with torch.no_grad():
        yk, lk = flow.forward_and_log_det_jacobian(x.to(DEV))
    yg, lg = flow.forward_and_log_det_jacobian(x.to(DEV))
    close(yg, yk, rtol=1e-5, atol=1e-5)
    close(lg, lk, rtol=1e-5, atol=1e-4)
    s = flow.rsample((5,))
    assert s.shape == (5, dim) and s.requires_grad
    assert flow.forward(x.to(DEV)).requires_grad
    # spline flows (quadratic and cubic) are differentiable in both directions: next test


@pytest.mark.parametrize('K,dim,hidden,stype,tol', [(5, 8, 16, 'quadratic', 3e-4), (16, 12, 24, 'quadratic', 3e-4),
                                                    (1, 5, 12, 'quadratic', 3e-4), (6, 8, 16, 'cubic', 1e-3),
                                                    (16, 10, 24, 'cubic', 1e-3)])
def test_spline_forward_direction_backward_matches_autograd_of_oracle(K, dim, hidden, stype, tol):
    """forward_and_log_det_jacobian / rsample of spline flows build a graph too (sx_rqs_forward_bwd / sx_cubic_forward_bwd:
    reverse mode through the FORWARD spline, bin searched on the widths): spline couplings + an element-wise Spline, against
    fp64 autograd of the oracle, inputs reaching into both linear tails (cubic: the inverse test's 1e-3 bound)."""
    torch.manual_seed(31)
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -2, 'upper': 2,
             'mask': 'ordered_right_half', 'latent_dim': 0, 'spline_type': stype},
            {'kind': 'rqs', 'dim': dim, 'n_bins': K, 'lower': -2.5, 'upper': 2.5, 'hidden': [], 'latent_dim': 0, 'spline_type': stype},
            {'kind': 'flip'},
            {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -2, 'upper': 2, 'mask': 'parity_odd',
             'latent_dim': 0, 'spline_type': stype}]
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(257, dim) * 1.4
    w = torch.linspace(0.5, 1.5, x.shape[0]).unsqueeze(-1)
    leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items()}
    xin = x.double().clone().requires_grad_(True)
    y64, l64 = orc.flow_forward_and_ldj(fd.flow_spec(desc, leaves), xin)
    want = (y64 ** 2).sum() * 0.1 + (l64 * w.double()).sum()
    want.backward()
    xg = x.to(DEV).requires_grad_(True)
    y, ldj = flow.forward_and_log_det_jacobian(xg)
    assert y.requires_grad and ldj.requires_grad and ldj.shape == (x.shape[0], 1)
    loss = (y ** 2).sum() * 0.1 + (ldj * w.to(DEV)).sum()
    loss.backward()
    assert abs(loss.item() - want.item()) <= 1e-4 * abs(want.item()) + 1e-3
    ref = xin.grad.float()
    assert ((xg.grad.cpu() - ref).abs().max() / ref.abs().max()).item() <= tol
    for name, p in flow.named_parameters():
        if p.numel() == 0:                       # K = 1: an element-wise Spline has no derivative parameters
            continue
        assert p.grad is not None, name
        ref = leaves[name].grad.float()
        assert ((p.grad.cpu() - ref).abs().max() / (ref.abs().max() + 1e-12)).item() <= tol, name
    with torch.no_grad():
        yk, lk = flow.forward_and_log_det_jacobian(x.to(DEV))
    close(y, yk, rtol=1e-5, atol=1e-5)
    close(ldj, lk, rtol=1e-5, atol=1e-4)
    assert flow.rsample((7,)).requires_grad


def test_flow_of_column_shuffles_only_carries_the_input_gradient():
    """Found by tools/fuzz_train.py --mix: a flow of nothing but Permute / Flip has no parameter, but the reference's
    log_prob is still differentiable in its input (flow.py:69-84: inverse, then the base density).  It used to be evaluated
    without a graph (warning + `loss.backward()` raising)."""
    torch.manual_seed(2)
    dim = 47
    desc = [{'kind': 'flip'}, {'kind': 'permute', 'dim': dim}, {'kind': 'flip'}]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(181, dim)
    xin = x.double().requires_grad_(True)
    want = -orc.flow_log_prob(fd.flow_spec(desc, state), xin, None).mean()
    want.backward()
    xg = x.to(DEV).requires_grad_(True)
    loss = -flow.log_prob(xg).mean()
    loss.backward()
    close(loss.detach().cpu(), want.detach().float(), rtol=1e-6, atol=1e-6)
    close(xg.grad.cpu(), xin.grad.float(), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize('unfused', [False, True])
@pytest.mark.parametrize('K,hidden,last_scale', [(8, [24], 1e-5), (16, [64], 1e-2), (5, [20, 12], 3e-2)])
def test_cubic_coupling_gradients_in_the_quadratic_fallback_branch(K, hidden, last_scale, unfused, monkeypatch):
    """Found by tools/fuzz_train.py: where |a| < 1e-3 the reference's inverse solves the bin's QUADRATIC (cubic_spline.py:216-222),
    so its root -- and autograd through it -- does not depend on `a`; the backward used to differentiate the full cubic there,
    whose a-path is NOT small (da/dtheta carries 1/w^2): 1e-2 of a weight gradient's scale on single fuzz cases.  A near-identity
    spline (small last-layer weights: every freshly initialised or zero-initialised conditioner) has a ~ 0 in every bin, so here
    most elements take that branch (last_scale 1e-5: all interior bins; the other two: a mix.  Scales that put many elements AT the
    threshold -- a ~ 20 x last_scale here -- are avoided: there fp32 and fp64 take different branches for a percent of the
    elements, in the reference too, and the branches' gradients differ).  Both backward paths (slab kernel, per-row kernel)
    against fp64 autograd of the oracle.  The same flows pin the inverse's VALUES: the reference's quadratic formula
    (-c + sqrt(c^2 - 4 b q)) / (2 b) cancels catastrophically as b -> 0 (1e-2 of x in fp32 for a near-identity spline); the
    kernel evaluates the same root as -2 q / (c + sqrt(.)) and must match the fp64 oracle per row."""
    if unfused:
        monkeypatch.setenv('STRIBOR_SPLINE_UNFUSED', '1')
    torch.manual_seed(21)
    dim, n = 12, 700
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': hidden, 'n_bins': K, 'lower': -3, 'upper': 3, 'mask': m,
             'latent_dim': 0, 'spline_type': 'cubic'} for m in ('ordered_right_half', 'ordered_left_half')]
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for t in flow.transforms:
            last = [m for m in t.transform.latent_net.modules() if isinstance(m, torch.nn.Linear)][-1]
            last.weight.mul_(last_scale)
            last.bias.mul_(last_scale)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(n, dim) * 1.3
    want_loss, want_g, want_gx = oracle_grads(desc, state, x)
    xg = x.to(DEV).requires_grad_(True)
    lp = flow.log_prob(xg)
    want_lp = orc.flow_log_prob(fd.flow_spec(desc, {k: v.double() for k, v in state.items()}), x.double())
    close(lp.detach(), want_lp.float(), rtol=1e-5, atol=5e-5)
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss) + 1e-5
    sx = want_gx.abs().max().item()
    assert (xg.grad.cpu() - want_gx.float()).abs().max().item() <= 2e-4 * sx + 1e-7
    if last_scale < 1e-4:
        # bb ~ 1e-7 here, and fp64 AUTOGRAD through the reference's (-c + sqrt(c^2 - 4 b q)) / (2 b) is off by ~10 % per element
        # (two terms of size t / b cancel to t^2); the oracle's VALUES are fine, so the truth is a central difference of its loss
        spec_of = lambda st_: fd.flow_spec(desc, st_)
        s64 = {k: v.double() for k, v in state.items()}
        for tname in ('transforms.0.transform.latent_net.net.2.weight', 'transforms.1.transform.latent_net.net.2.bias'):
            ours = dict(flow.named_parameters())[tname].grad.cpu().double()
            for idx in ours.abs().flatten().topk(6).indices.tolist():
                vals = []
                for sgn in (1.0, -1.0):
                    s2 = {k: v.clone() for k, v in s64.items()}
                    s2[tname].view(-1)[idx] += sgn * 1e-6
                    vals.append(-orc.flow_log_prob(spec_of(s2), x.double()).mean().item())
                fdg = (vals[0] - vals[1]) / 2e-6
                assert abs(ours.flatten()[idx].item() - fdg) <= 1e-3 * abs(fdg) + 1e-7, (tname, idx, ours.flatten()[idx].item(), fdg)
        return
    for name, p in flow.named_parameters():
        ref = want_g[name].float()
        scale = ref.abs().max().item() + 1e-12
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err <= 2e-4 * scale + 1e-9, (name, err, scale)


@pytest.mark.parametrize('stype', ['quadratic', 'cubic'])
def test_spline_training_beyond_the_program_tiles(stype):
    """Found by tools/fuzz_train.py --wide: 121 columns + 3 latent inputs need five tiles of 32; inference falls back to the
    generic tier, but the training path's no-grad evaluation called the MLP-program tier directly and raised
    NotImplementedError.  Gradients against fp64 autograd of the oracle.

    d log|f'(x)| / dx = f'' / f' JUMPS at a spline's knots (the splines are C1), and the knots come out of the conditioner: an element
    within rounding of a knot takes either side, and one such row moves the parameter gradients by percents (round 5: the nearest-rounding
    operand split moved row 20 of this batch across one; the reference's own fp32 autograd is off by 50 % of the x-gradient scale on
    another row: tools/experiments/dbg_cubic121.py).  Such rows are first shown to BE knot rows -- the fp64 oracle's gradient at
    x -+ 1e-5 brackets the product's -- and then left out of the batch the gradients are compared on."""
    torch.manual_seed(17)
    dim, latent, n = 121, 3, 150
    desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [40], 'n_bins': 6, 'lower': -3.0, 'upper': 3.0, 'mask': 'ordered_left_half',
             'latent_dim': latent, 'spline_type': stype}]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x_all, lat_all = torch.randn(n, dim) * 1.3, torch.randn(n, latent)

    def oracle(x, lat, scale=1.0):
        leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items()}
        xin = x.double().clone().requires_grad_(True)
        want = -orc.flow_log_prob(fd.flow_spec(desc, leaves), xin, lat.double()).sum() * scale
        want.backward()
        return want, xin.grad, {k: v.grad for k, v in leaves.items()}

    def product(x, lat):
        for p in flow.parameters():
            p.grad = None
        xg = x.to(DEV).requires_grad_(True)
        loss = -flow.log_prob(xg, latent=lat.to(DEV)).mean()
        loss.backward()
        return loss, xg.grad.cpu().double()

    want, gx, _ = oracle(x_all, lat_all, 1.0 / n)
    loss, ours = product(x_all, lat_all)
    sx = gx.abs().max().item()
    bad = torch.nonzero((ours - gx).abs() > 3e-4 * sx + 1e-8)
    knot_rows = sorted({int(r) for r, _ in bad.tolist()})
    assert len(knot_rows) <= 2, bad.tolist()
    for r in knot_rows:         # the row's worst element sits at the knot; its other elements (the conditioner's inputs) follow from it
        c = int((ours[r] - gx[r]).abs().argmax())
        sides = []
        for d in (1e-5, -1e-5):
            xr = x_all[r:r + 1].clone()
            xr[0, c] += d
            sides.append(oracle(xr, lat_all[r:r + 1], 1.0 / n)[1][0, c].item())
        lo_, hi_ = min(sides + [gx[r, c].item()]), max(sides + [gx[r, c].item()])
        assert lo_ - 3e-4 * sx <= ours[r, c].item() <= hi_ + 3e-4 * sx and hi_ - lo_ > 3e-4 * sx, \
            ('not a knot row', r, c, ours[r, c].item(), gx[r, c].item(), sides)
    keep = torch.tensor([i for i in range(n) if i not in knot_rows])
    x, lat = x_all[keep], lat_all[keep]
    want, gx, gp = oracle(x, lat, 1.0 / len(keep))
    loss, ours = product(x, lat)
    assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item()) + 1e-5
    sx = gx.abs().max().item()
    assert (ours - gx).abs().max().item() <= 3e-4 * sx + 1e-8
    for name, p in flow.named_parameters():
        ref = gp[name]
        assert (p.grad.cpu().double() - ref).abs().max().item() <= 3e-4 * ref.abs().max().item() + 1e-8, name


@pytest.mark.parametrize('kind,stype', [('coupling_affine', None), ('coupling_rqs', 'quadratic'), ('coupling_rqs', 'cubic')])
@pytest.mark.parametrize('direction', ['log_prob', 'forward'])
def test_set_data_couplings_are_differentiable(kind, stype, direction):
    """Coupling(set_data=True) (coupling.py:48-53: the mask runs over the SET axis; a transformed element sees only cat[0, latent])
    trains like every other layer (round 2: it used to be evaluated without a graph): gradients of a loss on log_prob, or on
    forward_and_log_det_jacobian, against fp64 autograd of the oracle; inputs (7, 4, 5) as in the reference's test_coupling.py."""
    torch.manual_seed(12)
    dim, latent = 5, 2
    desc = []
    for m in ('ordered_right_half', 'parity_odd'):
        d = {'kind': kind, 'dim': dim, 'hidden': [24], 'mask': m, 'latent_dim': latent, 'set_data': True}
        if stype is not None:
            d.update(n_bins=5, lower=-3.0, upper=3.0, spline_type=stype)
        desc.append(d)
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x, lat = torch.randn(7, 4, dim) * 1.2, torch.randn(7, 4, latent)
    leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items()}
    xin = x.double().clone().requires_grad_(True)
    spec = fd.flow_spec(desc, leaves)
    xg = x.to(DEV).requires_grad_(True)
    if direction == 'log_prob':
        want = -orc.flow_log_prob(spec, xin, lat.double()).mean()
        loss = -flow.log_prob(xg, latent=lat.to(DEV)).mean()
    else:
        y64, l64 = orc.flow_forward_and_ldj(spec, xin, lat.double())
        want = ((y64 ** 2).sum() * 0.1 + l64.sum()) / 28
        yg, lg = flow.forward_and_log_det_jacobian(xg, latent=lat.to(DEV))
        loss = ((yg ** 2).sum() * 0.1 + lg.sum()) / 28
    want.backward()
    loss.backward()
    assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item()) + 1e-6
    assert (xg.grad.cpu().double() - xin.grad).abs().max().item() <= 3e-4 * xin.grad.abs().max().item() + 1e-8
    for name, p in flow.named_parameters():
        ref = leaves[name].grad
        assert (p.grad.cpu().double() - ref).abs().max().item() <= 3e-4 * ref.abs().max().item() + 1e-8, name


@pytest.mark.parametrize('n_live,H', [(33, 48), (32, 64), (5, 20), (61, 64)])
def test_slab_packs_stay_inside_their_buffer(n_live, H, monkeypatch):
    """Found by tools/fuzz_train.py as a SEQUENCE-dependent 1e-1 gradient error: the transposed pack of the slab backward carries a
    bias block of 32 * ceil(H / 32) floats, 32 had been reserved -- for H > 32 the pack kernel wrote 128 B past the buffer, and where
    the allocator's block size left no padding (n_live = 33, H = 48) that zeroed the start of the next tensor in memory (the incoming
    log-det adjoints).  The buffers are handed out here from inside a canary region that must come back untouched."""
    from stribor_amd.flows import spline as sp_mod
    torch.manual_seed(0)
    K = 12
    P = 3 * K - 1
    W2, b2 = torch.randn(n_live * P, H, device=DEV), torch.randn(n_live * P, device=DEV)
    x2 = torch.randn(64, 2 * n_live, device=DEV)
    slot_rows = torch.from_numpy(sp_mod.slab_slot_rows(n_live, K, False)).to(DEV)
    hid = torch.full((((H + 31) // 32) * 32,), -1, dtype=torch.int32, device=DEV)
    hid[:H] = torch.arange(H, dtype=torch.int32, device=DEV)
    real_empty, canaries = torch.empty, []

    def guarded_empty(*size, **kw):
        n = int(size[0]) if len(size) == 1 and not isinstance(size[0], (tuple, list)) else None
        if n is None or kw.get('dtype', torch.float32) != torch.float32:
            return real_empty(*size, **kw)
        buf = torch.full((n + 1024,), float('nan'), dtype=torch.float32, device=kw.get('device'))
        canaries.append(buf[n:])
        return buf[:n]
    monkeypatch.setattr(torch, 'empty', guarded_empty)
    packs, n_fwd = sp_mod._slab_packs(x2, W2, b2, slot_rows, hid, n_live, H)
    monkeypatch.setattr(torch, 'empty', real_empty)
    torch.cuda.synchronize()
    assert canaries and all(torch.isnan(c).all().item() for c in canaries)
    assert torch.isfinite(packs).all()


def _cfg4_like(dim, hidden, blocks, bias=False, log_time=False):
    out = []
    for _ in range(blocks):
        out += [{'kind': 'affine_lu', 'dim': dim},
                {'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': 'ordered_right_half', 'latent_dim': 0},
                {'kind': 'matrix_exp', 'dim': dim, 'bias': bias, 'log_time': log_time},
                {'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': 'ordered_left_half', 'latent_dim': 0}]
    return out


@pytest.mark.parametrize('n,dim,hidden,blocks,bias,log_time', [
    (300, 128, 64, 1, False, False),        # cfg-4 widths (BASELINE configs[3]), one block
    (257, 128, 64, 2, True, False),
    (200, 100, 40, 1, True, True),          # ragged width: padded slots, hidden 40
    (130, 12, 16, 2, True, True),           # a narrow flow runs the same 4 + 4 tile program (zero-padded columns)
    (64, 70, 32, 1, False, False),
])
def test_fused_backward_of_dense_linear_flows_matches_autograd_of_oracle(monkeypatch, n, dim, hidden, blocks, bias, log_time):
    """cfg-4 family training without library GEMMs: AffineLU / MatrixExponential (affine.py:156-171,243-288) join
    the single-launch backward program (SX_STEP_LINEAR_BWD: v = A u + b on the x tiles, dL/dv = W^T dL/du on the adjoint tiles;
    dL/dW by sx_wgrad on the stored factors; the D x D parameter algebra through autograd of the batched fp64 derivation).
    Gradients -- every parameter and the input -- against fp64 autograd of the oracle at the suite's 3e-4 bound."""
    torch.manual_seed(51)
    desc = _cfg4_like(dim, hidden, blocks, bias, log_time)
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for p in flow.parameters():            # off the near-identity init
            p.add_(0.02 * torch.randn_like(p))
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(n, dim)
    want_loss, want_g, want_gx = oracle_grads(desc, state, x)
    # the fused path or nothing: the layer-wise path (library GEMMs) must not be what answers
    monkeypatch.setattr(type(flow), '_log_prob_layerwise_autograd', lambda *a, **k: (_ for _ in ()).throw(AssertionError('layer-wise path')))
    xg = x.to(DEV).requires_grad_(True)
    assert flow._can_backward(xg)
    lp = flow.log_prob(xg)
    assert lp.requires_grad and lp.shape == (n, 1)
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss) + 1e-5
    tol = lambda ref: 3e-4 * ref.abs().max().item() + 1e-7
    assert (xg.grad.cpu() - want_gx.float()).abs().max().item() <= tol(want_gx)
    for pname, p in flow.named_parameters():
        ref = want_g[pname].float()
        assert p.grad is not None, pname
        assert (p.grad.cpu() - ref).abs().max().item() <= tol(ref), (pname, (p.grad.cpu() - ref).abs().max().item(), ref.abs().max().item())


def test_fused_backward_of_dense_linear_flows_blocks_and_trains(monkeypatch):
    """Blocked batches give the same gradients (the factor scratch is bounded), a parameter update re-derives the matrices
    (one batched fp64 derivation per version), and a few SGD steps lower the loss."""
    from stribor_amd.flow import _FusedLogProb
    torch.manual_seed(52)
    flow = fd.build_flow(st, _cfg4_like(128, 64, 1), 128).to(DEV)
    x = torch.randn(1500, 128, device=DEV) * 0.7 + 0.2

    def grads():
        for p in flow.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        (-flow.log_prob(xg).mean()).backward()
        return [p.grad.clone() for p in flow.parameters()] + [xg.grad.clone()]

    whole = grads()
    monkeypatch.setattr(_FusedLogProb, 'SIDE_BYTES', 4 * 320 * 4 * 400)           # 400-row blocks, ragged tail
    blocked = grads()
    for a, b in zip(whole, blocked):
        assert torch.allclose(a, b, rtol=2e-4, atol=1e-6 + 2e-4 * a.abs().max().item())
    monkeypatch.undo()
    opt = torch.optim.SGD(flow.parameters(), lr=2e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = -flow.log_prob(x).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses
    with torch.no_grad():
        assert torch.isfinite(flow.log_prob(x)).all()


def test_fused_backward_of_dense_linear_flows_in_the_exact_arithmetic():
    """The same backward program through the sx_f32x kernels (set_gemm_precision('exact'): v_mfma_f32_32x32x2_f32, fp32 sx_wgrad)."""
    old = st.set_gemm_precision('exact')
    try:
        torch.manual_seed(53)
        dim, n = 128, 150
        desc = _cfg4_like(dim, 64, 1, True, False)
        flow = fd.build_flow(st, desc, dim)
        with torch.no_grad():
            for p in flow.parameters():
                p.add_(0.02 * torch.randn_like(p))
        state = {k: v.clone() for k, v in flow.state_dict().items()}
        flow = flow.to(DEV)
        x = torch.randn(n, dim)
        want_loss, want_g, want_gx = oracle_grads(desc, state, x)
        xg = x.to(DEV).requires_grad_(True)
        assert flow._can_backward(xg)
        loss = -flow.log_prob(xg).mean()
        loss.backward()
        assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss) + 1e-5
        tol = lambda ref: 3e-4 * ref.abs().max().item() + 1e-7
        assert (xg.grad.cpu() - want_gx.float()).abs().max().item() <= tol(want_gx)
        for pname, p in flow.named_parameters():
            ref = want_g[pname].float()
            assert p.grad is not None and (p.grad.cpu() - ref).abs().max().item() <= tol(ref), pname
        st.check_errors()
    finally:
        st.set_gemm_precision(old)


@pytest.mark.parametrize('stype,layers,dim,hidden,K,n', [('quadratic', 8, 64, 64, 16, 4099), ('quadratic', 3, 64, 40, 7, 1000),
                                                         ('cubic', 2, 64, 32, 16, 777), ('quadratic', 1, 32, 16, 4, 33),
                                                         ('quadratic', 3, 37, 40, 7, 500)])
def test_spline_flow_training_runs_its_forward_as_one_launch(monkeypatch, stype, layers, dim, hidden, K, n):
    """Round 3: the graph path of a flow of spline couplings (Linear - Tanh - Linear conditioners) runs the whole-flow fused program
    ONCE, with the side outputs of its SX_STEP_RQS_HIDDEN steps (tanh h per layer, the state each layer but the first received) as
    the saved tensors of the per-layer backward ops -- log_prob and every gradient equal the per-layer-forward path's."""
    torch.manual_seed(layers * 100 + dim)
    desc = [dict(d, spline_type=stype) for d in fd.cfg3_desc(layers, dim, hidden, K)]
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(n, dim) * 1.3
    # (halves that are not whole 32-column tiles are relabelled inside the fused program: such flows keep the per-layer forward)
    once = flow._spline_forward_once(x.to(DEV))
    assert once is not None or dim % 32 != 0
    if once is not None:
        assert len(once) == layers and all(h.shape == (n, hidden) for _, _, h in once)

    def grads(per_layer):
        if per_layer:
            monkeypatch.setenv('STRIBOR_SPLINE_FORWARD_PER_LAYER', '1')
        else:
            monkeypatch.delenv('STRIBOR_SPLINE_FORWARD_PER_LAYER', raising=False)
        for p in flow.parameters():
            p.grad = None
        xg = x.to(DEV).requires_grad_(True)
        lp = flow.log_prob(xg)
        (-lp.mean()).backward()
        return lp.detach().cpu(), xg.grad.cpu(), {k: p.grad.cpu().clone() for k, p in flow.named_parameters()}
    lp1, gx1, gp1 = grads(False)
    lp2, gx2, gp2 = grads(True)
    assert torch.equal(lp1, lp2) or (lp1 - lp2).abs().max().item() <= 1e-5 * (1 + lp2.abs().max().item())
    assert (gx1 - gx2).abs().max().item() <= 1e-5 * (gx2.abs().max().item() + 1e-12)
    for k in gp1:
        assert (gp1[k] - gp2[k]).abs().max().item() <= 1e-5 * (gp2[k].abs().max().item() + 1e-12), k
    # (against the oracle: log_prob here, the gradients through the per-layer path's own tests -- rows within rounding of a knot
    #  have one-sided gradients, DESIGN.md 2.1, and a mean loss over a few thousand rows shows each of them at 1e-3 of a parameter's)
    want = orc.flow_log_prob(fd.flow_spec(desc, {k: v.double() for k, v in state.items()}), x.double())
    assert ((lp1.double().reshape(-1) - want.reshape(-1)).abs() / (1.0 + want.reshape(-1).abs())).max().item() <= 2e-5


def test_fused_backward_of_dense_linear_flows_with_narrow_hidden_layers_at_scale():
    """Hidden <= 32 (one hidden tile) on the 4 + 4 tile backward program, at a size where the weight DMA of the next step is still
    in flight when a half-step ends: a coupling's step B then issues only 32 factor stores behind that DMA, fewer than the counted
    wait (`vmcnt(63)`) needs to cover it, so the wait at the loop head must be a full one.  A race shows as
    run-to-run differences and as a gap between the two GEMM arithmetics; both are checked over repeats."""
    torch.manual_seed(54)
    dim, n = 128, 1 << 18
    flow = fd.build_flow(st, _cfg4_like(dim, 32, 2), dim)
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(0.02 * torch.randn_like(p))
    flow = flow.to(DEV)
    x = torch.randn(n, dim, device=DEV)

    def grads():
        for p in flow.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        assert flow._can_backward(xg)
        (-flow.log_prob(xg).mean()).backward()
        return [xg.grad.clone()] + [p.grad.clone() for p in flow.parameters()]

    fast = [grads() for _ in range(4)]
    for rep in fast[1:]:
        for a, b in zip(fast[0], rep):
            # the weight gradients are reduced with float atomics in a run-dependent order: equal to rounding, not bit for bit
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 + 1e-5 * a.abs().max().item())
    assert torch.equal(fast[0][0], fast[1][0]), 'dL/dx has no cross-row reduction: repeats must agree bit for bit'
    old = st.set_gemm_precision('exact')
    try:
        exact = grads()
    finally:
        st.set_gemm_precision(old)
    for a, b in zip(fast[0], exact):
        assert (a - b).abs().max().item() <= 3e-4 * b.abs().max().item() + 1e-7
    st.check_errors()


@pytest.mark.gpu
@pytest.mark.parametrize('desc_kind', ['affine_140', 'spline_150_69', 'affine_128_190'])
def test_training_and_inference_with_hidden_layers_beyond_128(desc_kind):
    """Found by tools/fuzz_train.py --fat in round 4: a ProgramBuilder made for a hidden width beyond four tiles carries the CHUNK
    width of the forward's hidden-chunk steps in h_tiles; the backward-program builder and the MLP-program builder indexed their
    32 * h_tiles tables with the full width (ValueError) instead of leaving the layer to the next tier.  Values and gradients
    against fp64 autograd of the oracle."""
    torch.manual_seed(23)
    dim, n = 24, 180
    if desc_kind == 'affine_140':
        desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [140], 'mask': 'ordered_right_half', 'latent_dim': 0},
                {'kind': 'coupling_affine', 'dim': dim, 'hidden': [140], 'mask': 'ordered_left_half', 'latent_dim': 0}]
    elif desc_kind == 'affine_128_190':
        desc = [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [128, 190], 'mask': 'ordered_right_half', 'latent_dim': 0}]
    else:
        desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [150, 69], 'n_bins': 9, 'lower': -3.0, 'upper': 3.0,
                 'mask': 'parity_even', 'latent_dim': 0, 'spline_type': 'quadratic'}]
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(DEV)
    x = torch.randn(n, dim) * 1.2
    want_loss, want_g, want_gx = oracle_grads(desc, state, x)
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV))
    spec = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
    close(lp, orc.flow_log_prob(spec, x.double()).float(), rtol=1e-5, atol=2e-4)
    xg = x.to(DEV).requires_grad_(True)
    loss = -flow.log_prob(xg).mean()
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-5 * abs(want_loss) + 1e-5
    close(xg.grad, want_gx.float(), rtol=3e-4, atol=1e-6)
    for name, p in flow.named_parameters():
        ref = want_g[name].float()
        assert (p.grad.cpu() - ref).abs().max().item() <= 3e-4 * (ref.abs().max().item() + 1e-12) + 1e-7, name
    st.check_errors()


def test_second_backward_through_a_dense_flow_with_retain_graph():
    """the fused log_prob op of a dense-linear flow keeps the graph of its D x D fp64 derivation for the
    backward; autograd.grad on it without retain_graph made `loss.backward(retain_graph=True)` followed by another backward
    raise 'backward through the graph a second time'.  The second pass adds the same gradients again."""
    torch.manual_seed(52)
    dim, n = 24, 150
    desc = _cfg4_like(dim, 16, 1, True, False)
    flow = fd.build_flow(st, desc, dim).to(DEV)
    xg = (torch.randn(n, dim) * 0.8).to(DEV).requires_grad_(True)
    loss = -flow.log_prob(xg).mean()
    loss.backward(retain_graph=True)
    once = {k: p.grad.clone() for k, p in flow.named_parameters()}
    gx = xg.grad.clone()
    loss.backward()
    close(xg.grad, 2 * gx, rtol=1e-6, atol=1e-9)
    for k, p in flow.named_parameters():
        close(p.grad, 2 * once[k], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('stype', ['quadratic', 'cubic'])
def test_one_coupling_module_twice_in_a_spline_flow_trains(stype):
    """the one-launch training forward handed every layer its precomputed tensors through an attribute of the
    (shared) module; they travel as an argument of the call now.  The same Coupling twice in one flow: gradients of the shared
    parameters are the sum over both positions, against fp64 autograd of the oracle on the equivalent three-layer description."""
    torch.manual_seed(53)
    dim, n, K = 16, 220, 8
    d = {'kind': 'coupling_rqs', 'dim': dim, 'hidden': [32], 'n_bins': K, 'lower': -3.0, 'upper': 3.0, 'mask': 'ordered_right_half',
         'latent_dim': 0, 'spline_type': stype}
    desc = [d, {'kind': 'flip'}, d]
    built = fd.build_flow(st, desc, dim)
    shared = built.transforms[0]
    flow = st.NormalizingFlow(st.UnitNormal(dim), [shared, built.transforms[1], shared])
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    assert any(k.startswith('transforms.2.') for k in state)
    flow = flow.to(DEV)
    x = torch.randn(n, dim) * 1.2
    leaves = {k: v.detach().double().clone().requires_grad_(True) for k, v in state.items() if k.startswith('transforms.0.')}
    both = dict(leaves)
    both.update({k.replace('transforms.0.', 'transforms.2.'): v for k, v in leaves.items()})
    for k, v in state.items():
        if k not in both:
            both[k] = v.double()
    xin = x.double().clone().requires_grad_(True)
    want = -orc.flow_log_prob(fd.flow_spec(desc, both), xin).mean()
    want.backward()
    xg = x.to(DEV).requires_grad_(True)
    loss = -flow.log_prob(xg).mean()
    loss.backward()
    assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item()) + 1e-5
    tol = 3e-4 if stype == 'quadratic' else 2e-3
    assert (xg.grad.cpu().double() - xin.grad).abs().max().item() <= tol * xin.grad.abs().max().item() + 1e-8
    for name, p in shared.named_parameters():
        ref = leaves['transforms.0.' + name].grad
        assert (p.grad.cpu().double() - ref).abs().max().item() <= tol * ref.abs().max().item() + 1e-8, name
