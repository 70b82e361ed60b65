"""Spline flows (reference: stribor/flows/spline.py:11-143): ``spline_type='quadratic'`` ->
stribor/util/rational_quadratic_spline.py, ``spline_type='cubic'`` (the reference default) ->
stribor/util/cubic_spline.py.

Same constructor as the reference.  The arithmetic runs in ``sx_rqs_coupling`` / ``sx_cubic_coupling``: one lane
per element, the element's 3K-1 (2K+2) parameters staged through the wave's own LDS slice, tails predicated.
Quadratic-spline couplings also join fused flow programs; cubic ones run layer by layer (MLP program + this kernel).

Error behaviour: the reference raises ``ValueError('Minimal bin width too large ...')`` when
``1e-3 * n_bins > 1`` (rational_quadratic_spline.py:96-99) — same here, from Python, before launch.
Its ``assert (discriminant >= 0).all()`` (:223) becomes a device flag that ``check_errors()`` reads.
"""
from typing import Optional

import os

import torch
import torch.nn as nn

from .. import _hip
from ..flow import ElementwiseTransform, flatten_rows, graph_rows, graph_wanted

__all__ = ['Spline', 'run_rqs_kernel', 'run_cubic_kernel', 'RQSInverse', 'RQSForward', 'CubicInverse', 'CubicForward', 'RQSCouplingSlab', 'RQSCouplingSlabL1', 'slab_slot_rows']

def check_errors(device=None) -> None:
    """Raise what the reference would have raised for data-dependent failures (synchronises): the spline's
    ``assert (discriminant >= 0).all()`` (rational_quadratic_spline.py:223) -> AssertionError."""
    _hip.check_errors(device)


def run_rqs_kernel(x2, params, params_stride, live_idx, live_start, n_live, n_bins, left, right, bottom, top,
                   reverse, want_ldj, want_ldiag, ldj_scale=1.0):
    """Launch sx_rqs_coupling on [N, D] rows -> (y, ldj | None, ldiag | None)."""
    if 1e-3 * n_bins > 1.0:
        raise ValueError('Minimal bin width too large for the number of bins')      # :96-97
    n, d = x2.shape
    y = torch.empty_like(x2)
    ldj = torch.empty(n, dtype=torch.float32, device=x2.device) if want_ldj else None
    ldiag = torch.empty(n, d, dtype=torch.float32, device=x2.device) if want_ldiag else None
    _hip.call('sx_rqs_coupling', x2, x2.data_ptr(), y.data_ptr(), _hip.ptr(ldj), _hip.ptr(ldiag), params.data_ptr(),
                                    params_stride, _hip.ptr(live_idx), live_start, n_live, n_bins, float(left),
                                    float(right), float(bottom), float(top), n, d, _hip.dtype_code(x2),
                                    int(reverse), 0, float(ldj_scale), _hip.err_flag(x2.device))
    return y, ldj, ldiag


class _SplineElementOp(torch.autograd.Function):
    """(out, row log-det[, per-element log-derivative]) of the spline of the live columns as a differentiable op -- one class for
    the four (spline type, direction) pairs: forward = sx_rqs_coupling / sx_cubic_coupling, backward = sx_rqs_inverse_bwd /
    sx_rqs_forward_bwd / sx_cubic_inverse_bwd / sx_cubic_forward_bwd (hand-written reverse mode through
    rational_quadratic_spline.py:101-107,180-248 resp. cubic_spline.py:103-247; the cubic solve is differentiated implicitly, so
    that op also keeps its output).  Gradients flow to the input and to the per-row parameter tensor [N, n_live * P]; whatever
    produced the parameters (a conditioner evaluated with torch's own Linear layers, or nn.Parameters) gets its gradient from
    autograd.  `want_ldiag`: a third output [N, D] (Spline.log_diag_jacobian with a graph; its adjoint rides on the same
    backward kernel as `gldiag`, ABI v3)."""

    @staticmethod
    def forward(ctx, x2, params, cubic, reverse, live_idx, live_start, n_live, n_bins, lower, upper, ldj_scale, want_ldiag):
        x2 = x2.contiguous()
        params = params.contiguous()
        if cubic:
            y, ldj, ldiag = run_cubic_kernel(x2, params, params.stride(0), live_idx, live_start, n_live, n_bins, lower, upper,
                                             bool(reverse), True, bool(want_ldiag), ldj_scale)
        else:
            y, ldj, ldiag = run_rqs_kernel(x2, params, params.stride(0), live_idx, live_start, n_live, n_bins, lower, upper,
                                           lower, upper, bool(reverse), True, bool(want_ldiag), ldj_scale)
        ctx.save_for_backward(x2, params, y if (cubic and reverse) else None)
        ctx.meta = (bool(cubic), bool(reverse), live_idx, live_start, n_live, n_bins, float(lower), float(upper), float(ldj_scale))
        if want_ldiag:
            return y, ldj, ldiag
        return y, ldj

    @staticmethod
    def backward(ctx, gy, gldj, gldiag=None):
        x2, params, yout = ctx.saved_tensors
        cubic, reverse, live_idx, live_start, n_live, n_bins, lower, upper, ldj_scale = ctx.meta
        n, d = x2.shape
        gy = (torch.zeros_like(x2) if gy is None else gy).to(torch.float32).contiguous()
        gldj = (torch.zeros(n, device=x2.device) if gldj is None else gldj).to(torch.float32).contiguous()
        gldiag = None if gldiag is None else gldiag.to(torch.float32).contiguous()
        gx = gy.clone()                     # pass-through columns: y = x; the kernel overwrites the live columns
        gparams = torch.empty_like(params)
        if cubic and reverse:
            _hip.call('sx_cubic_inverse_bwd', x2, x2.data_ptr(), yout.data_ptr(), gy.data_ptr(), gldj.data_ptr(), _hip.ptr(gldiag),
                      params.data_ptr(), params.stride(0), gx.data_ptr(), gparams.data_ptr(), _hip.ptr(live_idx), live_start,
                      n_live, n_bins, lower, upper, n, d, ldj_scale)
        elif cubic:
            _hip.call('sx_cubic_forward_bwd', x2, x2.data_ptr(), gy.data_ptr(), gldj.data_ptr(), _hip.ptr(gldiag), params.data_ptr(),
                      params.stride(0), gx.data_ptr(), gparams.data_ptr(), _hip.ptr(live_idx), live_start, n_live, n_bins, lower,
                      upper, n, d, ldj_scale)
        else:
            _hip.call('sx_rqs_inverse_bwd' if reverse else 'sx_rqs_forward_bwd', x2, x2.data_ptr(), gy.data_ptr(), gldj.data_ptr(),
                      _hip.ptr(gldiag), params.data_ptr(), params.stride(0), gx.data_ptr(), gparams.data_ptr(), _hip.ptr(live_idx),
                      live_start, n_live, n_bins, lower, upper, lower, upper, n, d, ldj_scale)
        return (gx, gparams) + (None,) * 10


class _SplineOpAlias:
    """`X.apply(x2, params, live_idx, live_start, n_live, n_bins, lower, upper, ldj_scale[, want_ldiag])` -> (out, row log-det[,
    ldiag]): the historical per-(type, direction) op names over _SplineElementOp."""
    cubic = reverse = False

    @classmethod
    def apply(cls, x2, params, live_idx, live_start, n_live, n_bins, lower, upper, ldj_scale, want_ldiag=False):
        return _SplineElementOp.apply(x2, params, cls.cubic, cls.reverse, live_idx, live_start, n_live, n_bins, lower, upper,
                                      ldj_scale, want_ldiag)


class RQSInverse(_SplineOpAlias):
    """inverse rational-quadratic spline -- the direction log_prob evaluates (sx_rqs_coupling(reverse=1) / sx_rqs_inverse_bwd)."""
    cubic, reverse = False, True


class RQSForward(_SplineOpAlias):
    """FORWARD rational-quadratic spline (forward / rsample of spline flows; sx_rqs_forward_bwd)."""
    cubic, reverse = False, False


def _adjoint_scale(gy, gldj):
    """max |adjoint| on the device, for the slab kernels' power-of-two normalisation of their fp16 x 3 operands: ONE launch of
    sx_absmax2 over EVERY element (adjoints are row-local: a loss with strongly non-uniform row weights can hide its largest
    rows from any strided sample -- ADVICE r2 -- and the sampled torch form took four launches)."""
    out = torch.zeros(1, dtype=torch.float32, device=gy.device)
    _hip.call('sx_absmax2', gy, gy.data_ptr(), gy.numel(), gldj.data_ptr(), gldj.numel(), out.data_ptr())
    return out


def slab_slot_rows(n_live: int, n_bins: int, cubic: bool = False):
    """sx_rqs_slab_bwd's slot -> parameter-row map (include/stribor_hip.h): slab s holds transformed columns 2s, 2s+1 in
    three 32-slot tiles (widths | heights | derivatives); slot 32 t + R carries parameter (R&3) + 4 (R>>3) of column
    2 s + ((R>>2)&1).  Rows index the [n_live * P] selected rows of the conditioner's last layer (spline.py:82-86): per column
    K widths, K heights, then K-1 knot derivatives (quadratic, P = 3K-1) or the 2 boundary derivatives (cubic, P = 2K+2)."""
    import numpy as np
    K = n_bins
    P = 2 * K + 2 if cubic else 3 * K - 1
    n_third = 2 if cubic else K - 1
    n_slabs = (n_live + 1) // 2
    rows = np.full(n_slabs * 96, -1, dtype=np.int32)
    for s in range(n_slabs):
        for t in range(3):
            for R in range(32):
                k, ci = (R & 3) + 4 * (R >> 3), 2 * s + ((R >> 2) & 1)
                if ci < n_live and k < (n_third if t == 2 else K):
                    rows[s * 96 + 32 * t + R] = ci * P + t * K + k
    return rows


# The slab backward's scratch grows with the rows of ONE call (dh partials: n_groups x n_chunks x HT KiB-tiles -- 10.7 GB for 2^20
# rows at 160 hidden units, more than the parameter tensor the tier exists to avoid) and `_hip.scratch` keeps the
# block for the life of the (device, stream).  Calls are therefore cut into row blocks whose scratch stays under a budget
# (STRIBOR_SLAB_SCRATCH_MB, default 2048): dW / db of the blocks add up, everything else is per row.  The BASELINE shapes (cfg 3,
# 2^18 .. 2^20 rows at 64 hidden units: 0.5 .. 2 GB) stay one call.
_SLAB_SCRATCH_BUDGET = int(os.environ.get('STRIBOR_SLAB_SCRATCH_MB', '2048')) << 20


def _slab_row_blocks(lib, n: int, n_live: int, H: int):
    rows = n
    while rows > 8192 and lib.sx_rqs_slab_scratch_floats(rows, n_live, H) * 4 > _SLAB_SCRATCH_BUDGET:
        rows = ((rows + 1) // 2 + 127) // 128 * 128
    return [(r0, min(n, r0 + rows)) for r0 in range(0, n, rows)] if n > 0 else [(0, 0)]


class RQSCouplingSlab(torch.autograd.Function):
    """(x_out, row log-det) of an inverse quadratic-spline COUPLING as a differentiable op of (x, h, W2, b2): h [N, H] is the
    conditioner's last hidden activation (torch graph upstream; or, with pre_tanh, the pre-activation of a final Tanh, which the
    op then applies and differentiates itself -- the tanh backward rides on the kernel that reduces dL/dh), (W2, b2) the rows of its last Linear that parameterise the
    transformed columns.  The forward is the coupling's own no-graph evaluation (`evaluate(x2)` -> (y, ldj)); the backward is
    sx_rqs_slab_bwd: spline reverse mode + dW2 / db2 / dL/dh in one kernel that keeps each slab of W2 in LDS, so the
    [N, n_live * (3K-1)] parameter tensor (spline.py:82-86) and its gradient never reach HBM.  Needs H <= 256, K <= 16 and the
    fp16 x 3 GEMM arithmetic.  cubic=True: monotone cubic splines (the op then also keeps the pass's output)."""

    @staticmethod
    def eligible(hidden: int, n_bins: int) -> bool:
        return hidden <= 256 and n_bins <= 16 and _hip.get_gemm_precision() != 'exact'

    @staticmethod
    def forward(ctx, x2, h, W2, b2, evaluate, plan, live_idx, live_start, n_live, n_bins, lower, upper, pre_tanh=False,
                cubic=False):
        x2 = x2.contiguous()
        with torch.no_grad():
            y, ldj = evaluate(x2)
            if pre_tanh:                    # `h` is the pre-activation of a Tanh: applied here, differentiated in backward
                h = torch.tanh(h)
        # cubic splines: the backward differentiates the solve at the pass's output
        ctx.save_for_backward(x2, h, W2, b2, y if cubic else None)
        ctx.meta = (plan, live_idx, live_start, n_live, n_bins, float(lower), float(upper), bool(pre_tanh))
        return y, ldj

    @staticmethod
    def backward(ctx, gy, gldj):
        x2, h, W2, b2, yout = ctx.saved_tensors
        (slot_rows, hid_idx), live_idx, live_start, n_live, n_bins, lower, upper, pre_tanh = ctx.meta
        n, d = x2.shape
        dev = x2.device
        H = h.shape[1]
        gy = (torch.zeros_like(x2) if gy is None else gy).to(torch.float32).contiguous()
        gldj = (torch.zeros(n, device=dev) if gldj is None else gldj).to(torch.float32).contiguous()
        h = h if h.stride(1) == 1 else h.contiguous()
        W2, b2 = W2.detach().contiguous(), b2.detach().contiguous()
        lib = _hip.lib()
        flag = _hip.err_flag(dev)
        packs, n_fwd = _slab_packs(x2, W2, b2, slot_rows, hid_idx, n_live, H)
        gx = gy.clone()                     # pass-through columns: y = x; the kernel overwrites the transformed columns
        gh = torch.empty(n, H, dtype=torch.float32, device=dev)
        gW, gb = torch.empty_like(W2), torch.empty_like(b2)      # every selected row sits in exactly one slot: all written
        # the kernels normalise the adjoints by a power of two derived from their largest magnitude (exact): the parameter
        # gradients are fp16 x 3 GEMM operands, and dL/dlog_prob = 1/N of a mean loss would put them under fp16's normal
        # range.  The maximum stays on the device (no host sync).
        scale = _adjoint_scale(gy, gldj)
        blocks = _slab_row_blocks(lib, n, n_live, H)
        for i, (r0, r1) in enumerate(blocks):
            gWb, gbb = (gW, gb) if i == 0 else (torch.empty_like(W2), torch.empty_like(b2))
            with _hip.device_of(x2):
                sc = _hip.scratch(dev, lib.sx_rqs_slab_scratch_floats(r1 - r0, n_live, H))
            _hip.call('sx_rqs_slab_bwd', x2, x2[r0:r1].data_ptr(), gy[r0:r1].data_ptr(), gldj[r0:r1].data_ptr(),
                      None if yout is None else yout[r0:r1].data_ptr(), h[r0:r1].data_ptr(), h.stride(0), H,
                      packs.data_ptr(), packs.data_ptr() + 4 * n_fwd, slot_rows.data_ptr(), gx[r0:r1].data_ptr(), gh[r0:r1].data_ptr(),
                      gh.stride(0), gWb.data_ptr(), gWb.stride(0), gbb.data_ptr(), _hip.ptr(live_idx), live_start, n_live, n_bins,
                      lower, upper, lower, upper, r1 - r0, d, 1.0, int(pre_tanh), scale.data_ptr(), sc.data_ptr(), flag)
            if i:
                gW += gWb
                gb += gbb
        return gx, gh, gW, gb, None, None, None, None, None, None, None, None, None, None


def _slab_packs(x2, W2, b2, slot_rows, hid_idx, n_live, H):
    """The two fragment packs of the selected last-layer rows sx_rqs_slab_bwd consumes -> (buffer, float offset of the
    transposed pack)."""
    lib = _hip.lib()
    slots, ht = lib.sx_rqs_slab_slots(n_live), (H + 31) // 32
    mt = slots // 32
    # (the transposed pack carries its own -- unused, zero -- bias block of 32 ht floats behind the tiles: reserving 32 for it
    #  let hidden widths above 32 write 128 B past the buffer; where the block size left no padding that zeroed the start of
    #  the next tensor in memory, e.g. the incoming log-det adjoints: tools/fuzz_train.py found it as a sequence-dependent
    #  1e-1 gradient error)
    n_fwd, n_bwd = _hip.packed_linear_floats(mt, ht), _hip.packed_linear_floats(ht, mt)
    packs = torch.empty(n_fwd + n_bwd, dtype=torch.float32, device=x2.device)
    flag = _hip.err_flag(x2.device)
    _hip.call('sx_pack_linear', x2, W2.data_ptr(), b2.data_ptr(), W2.shape[0], H, slot_rows.data_ptr(), hid_idx.data_ptr(),
              mt, ht, None, None, 0.0, 0, _hip.GEMM_F16X3, flag, packs.data_ptr())
    _hip.call('sx_pack_linear', x2, W2.data_ptr(), None, W2.shape[0], H, hid_idx.data_ptr(), slot_rows.data_ptr(),
              ht, mt, None, None, 0.0, 1, _hip.GEMM_F16X3, flag, packs.data_ptr() + 4 * n_fwd)
    return packs, n_fwd


class RQSCouplingSlabL1(torch.autograd.Function):
    """RQSCouplingSlab with the conditioner's FIRST layer inside the op as well, for Linear - Tanh - Linear conditioners without
    a latent input (cfg 3): a differentiable op of (x, W1, b1, W2, b2).  Forward: h = tanh(x (W1 * mask)^T + b1) (an MFMA program launch
    of net/mlp.py, kept for the backward) and the coupling's no-graph evaluation.  Backward: sx_rqs_slab_bwd leaves its dh partials in
    scratch and sx_rqs_slab_l1_bwd finishes the layer in one pass over the rows -- partial sum, tanh', dL/dx of the
    conditioning columns (= dL/dout + W1m^T da), dW1, db1 -- so no library GEMM, clone or add is left in the backward."""

    @staticmethod
    def eligible(dim: int, hidden: int, n_bins: int) -> bool:
        return dim <= 64 and hidden <= 128 and RQSCouplingSlab.eligible(hidden, n_bins)

    @staticmethod
    def forward(ctx, x2, W1, b1, W2, b2, mask_t, evaluate, plan, live_idx, live_start, n_live, n_bins, lower, upper, cubic=False):
        x2 = x2.contiguous()
        pre = getattr(evaluate, 'precomputed', None)
        if pre is not None:
            # the whole flow ran as ONE fused program that left every layer's output, log-det share and tanh h behind
            # (NormalizingFlow._layerwise_autograd): nothing to launch here, the backward below is the same
            y, ldj, h = pre
        else:
          with torch.no_grad():
            # h = tanh(x (W1 * mask)^T + b1) comes out of the forward program itself (side output of its hidden step) when the
            # coupling runs as one fused program; otherwise one Linear program (net/mlp.py BatchLinear) + tanh
            h = torch.empty(x2.shape[0], W1.shape[0], dtype=torch.float32, device=x2.device)
            got = evaluate(x2, h)
            if len(got) == 3 and got[2]:
                y, ldj = got[0], got[1]
            else:
                y, ldj = got[0], got[1]
                from ..net.mlp import BatchLinear
                h = BatchLinear._program_linear(x2, (W1 * mask_t).contiguous(), b1, False)       # (an MFMA program: net/mlp.py)
                h = torch.tanh(h if h is not None else torch.addmm(b1, x2, (W1 * mask_t).t()))     # (fewer than 4096 rows: the library)
        ctx.save_for_backward(x2, h, W1, W2, b2, mask_t, y if cubic else None)
        ctx.meta = (plan, live_idx, live_start, n_live, n_bins, float(lower), float(upper))
        return y, ldj

    @staticmethod
    def backward(ctx, gy, gldj):
        import ctypes as C
        x2, h, W1, W2, b2, mask_t, yout = ctx.saved_tensors
        plan, live_idx, live_start, n_live, n_bins, lower, upper = ctx.meta
        slot_rows, hid_idx, col_slots, col_map, cond_words = plan[:5]
        full_w2 = len(plan) > 5 and plan[5]          # W2 / b2 are the conditioner's WHOLE last layer, slot_rows indexes its rows
        n, d = x2.shape
        dev = x2.device
        H = h.shape[1]
        gy = (torch.zeros_like(x2) if gy is None else gy).to(torch.float32).contiguous()
        gldj = (torch.zeros(n, device=dev) if gldj is None else gldj).to(torch.float32).contiguous()
        W2, b2 = W2.detach().contiguous(), b2.detach().contiguous()
        lib = _hip.lib()
        flag = _hip.err_flag(dev)
        ht, xt = (H + 31) // 32, (d + 31) // 32
        cache = plan[6] if len(plan) > 6 else None
        if cache is None:
            packs, n_fwd = _slab_packs(x2, W2, b2, slot_rows, hid_idx, n_live, H)
            W1m = (W1.detach() * mask_t).contiguous()
            w1t = torch.empty(_hip.packed_linear_floats(xt, ht), dtype=torch.float32, device=dev)       # tiles + the pack's bias block
            _hip.call('sx_pack_linear', x2, W1m.data_ptr(), None, H, d, col_slots.data_ptr(), hid_idx.data_ptr(), xt, ht, None, None,
                      0.0, 1, _hip.GEMM_F16X3, flag, w1t.data_ptr())
        else:
            # the layer's three fragment packs (W2 slabs, their transpose, (W1 * mask)^T) live with the layer's plan and are re-made --
            # ONE sx_pack_linear_batch launch into the same buffers -- when a parameter changed (3 launches + a product per call before)
            stamp = tuple((p_.data_ptr(), p_._version) for p_ in (W1, W2, b2, mask_t))
            mt = lib.sx_rqs_slab_slots(n_live) // 32
            n_fwd, n_bwd = _hip.packed_linear_floats(mt, ht), _hip.packed_linear_floats(ht, mt)
            if cache.get('stamp') != stamp or cache['packs'].device != dev:
                if 'packs' not in cache or cache['packs'].device != dev:
                    cache['packs'] = torch.empty(n_fwd + n_bwd, dtype=torch.float32, device=dev)
                    cache['w1t'] = torch.empty(_hip.packed_linear_floats(xt, ht), dtype=torch.float32, device=dev)
                    cache['W1m'] = torch.empty(H, d, dtype=torch.float32, device=dev)
                    cache['table'] = _hip.PackTable()
                torch.mul(W1.detach(), mask_t, out=cache['W1m'])
                pk, sr, hi = cache['packs'].data_ptr(), slot_rows.data_ptr(), hid_idx.data_ptr()
                cache['table'].run(x2, [
                    (W2.data_ptr(), b2.data_ptr(), W2.shape[0], H, sr, hi, mt, ht, 0, 0, 0.0, 0, pk, 0),
                    (W2.data_ptr(), 0, W2.shape[0], H, hi, sr, ht, mt, 0, 0, 0.0, 1, pk + 4 * n_fwd, 0),
                    (cache['W1m'].data_ptr(), 0, H, d, col_slots.data_ptr(), hi, xt, ht, 0, 0, 0.0, 1, cache['w1t'].data_ptr(), 0)],
                    _hip.GEMM_F16X3, flag)
                cache['stamp'] = stamp
            packs, w1t = cache['packs'], cache['w1t']
        gx = torch.empty_like(gy)           # every column is written: transformed ones by the slab kernel, the rest by the l1 kernel
        # every selected row sits in exactly one slot: all written; the whole layer's other rows (parameters of the columns the
        # coupling leaves alone) have a zero gradient
        gW2, gb2 = (torch.zeros_like(W2), torch.zeros_like(b2)) if full_w2 else (torch.empty_like(W2), torch.empty_like(b2))
        gW1 = torch.zeros(H, d, dtype=torch.float32, device=dev)
        gb1 = torch.zeros(H, dtype=torch.float32, device=dev)
        n_l1 = lib.sx_rqs_slab_l1_scratch_floats(d, H)
        scale = _adjoint_scale(gy, gldj)
        words = (C.c_uint32 * 2)(*cond_words)
        blocks = _slab_row_blocks(lib, n, n_live, H)
        for i, (r0, r1) in enumerate(blocks):
            # (dW2 / db2 are written per call, dW1 / db1 accumulated by the first-layer kernel: the blocks' shares add up)
            gW2b, gb2b = (gW2, gb2) if i == 0 else (torch.zeros_like(W2), torch.zeros_like(b2)) if full_w2 else (torch.empty_like(W2), torch.empty_like(b2))
            n_slab = lib.sx_rqs_slab_scratch_floats(r1 - r0, n_live, H)
            with _hip.device_of(x2):
                sc = _hip.scratch(dev, n_slab + n_l1 + 64)
            _hip.call('sx_rqs_slab_bwd', x2, x2[r0:r1].data_ptr(), gy[r0:r1].data_ptr(), gldj[r0:r1].data_ptr(),
                      None if yout is None else yout[r0:r1].data_ptr(), h[r0:r1].data_ptr(), h.stride(0), H,
                      packs.data_ptr(), packs.data_ptr() + 4 * n_fwd, slot_rows.data_ptr(), gx[r0:r1].data_ptr(), None, 0,
                      gW2b.data_ptr(), gW2b.stride(0), gb2b.data_ptr(), _hip.ptr(live_idx), live_start, n_live, n_bins,
                      lower, upper, lower, upper, r1 - r0, d, 1.0, 1, scale.data_ptr(), sc.data_ptr(), flag)
            l1_scratch = sc.data_ptr() + 4 * ((n_slab + 3) // 4 * 4)
            _hip.call('sx_rqs_slab_l1_bwd', x2, sc.data_ptr(), h[r0:r1].data_ptr(), h.stride(0), H, x2[r0:r1].data_ptr(), gy[r0:r1].data_ptr(),
                      w1t.data_ptr(), C.cast(words, C.c_void_p), gx[r0:r1].data_ptr(), gW1.data_ptr(), gW1.stride(0), gb1.data_ptr(),
                      col_map.data_ptr(), n_live, r1 - r0, d, scale.data_ptr(), l1_scratch, flag)
            if i:
                gW2 += gW2b
                gb2 += gb2b
        return gx, gW1, gb1, gW2, gb2, None, None, None, None, None, None, None, None, None, None


def run_cubic_kernel(x2, params, params_stride, live_idx, live_start, n_live, n_bins, lower, upper, reverse, want_ldj,
                     want_ldiag, ldj_scale=1.0):
    """Launch sx_cubic_coupling on [N, D] rows -> (y, ldj | None, ldiag | None)."""
    if 1e-2 * n_bins > 1.0:
        raise ValueError('Minimal bin width too large for the number of bins')      # cubic_spline.py:93-94
    n, d = x2.shape
    y = torch.empty_like(x2)
    ldj = torch.empty(n, dtype=torch.float32, device=x2.device) if want_ldj else None
    ldiag = torch.empty(n, d, dtype=torch.float32, device=x2.device) if want_ldiag else None
    _hip.call('sx_cubic_coupling', x2, x2.data_ptr(), y.data_ptr(), _hip.ptr(ldj), _hip.ptr(ldiag), params.data_ptr(),
                                      params_stride, _hip.ptr(live_idx), live_start, n_live, n_bins, float(lower),
                                      float(upper), n, d, _hip.dtype_code(x2), int(reverse), 0, float(ldj_scale))
    return y, ldj, ldiag


class CubicInverse(_SplineOpAlias):
    """inverse monotone cubic spline (sx_cubic_coupling(reverse=1) / sx_cubic_inverse_bwd: the solve differentiated implicitly,
    then reverse mode through cubic_spline.py:103-137).  params: [N, n_live*(2K+2)]."""
    cubic, reverse = True, True


class CubicForward(_SplineOpAlias):
    """FORWARD monotone cubic spline (sx_cubic_forward_bwd: the polynomial's own derivatives, then the chain of CubicInverse
    through the Steffen knot derivatives, cumsums and softmax)."""
    cubic, reverse = True, False


class Spline(ElementwiseTransform):
    def __init__(self, dim: int, n_bins: int, latent_net: Optional[nn.Module] = None, lower: Optional[float] = 0,
                 upper: Optional[float] = 1, spline_type: Optional[str] = 'cubic', **kwargs):
        super().__init__()
        self.lower, self.upper = lower, upper
        self.dim, self.n_bins = dim, n_bins
        self.latent_net = latent_net
        self.spline_type = spline_type
        if spline_type == 'quadratic':
            self.derivative_dim = n_bins - 1                                         # spline.py:56-57
        elif spline_type == 'cubic':
            self.derivative_dim = 2                                                  # spline.py:59-61
        else:
            raise ValueError('spline_type must be either `quadratic` or `cubic`')    # spline.py:63
        if latent_net is None:
            self.width = nn.Parameter(torch.empty(dim, n_bins))                      # spline.py:65-69
            self.height = nn.Parameter(torch.empty(dim, n_bins))
            self.derivative = nn.Parameter(torch.empty(dim, self.derivative_dim))
            self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.width)                                          # spline.py:71-74
        nn.init.xavier_uniform_(self.height)
        if self.derivative_dim > 0:
            nn.init.xavier_uniform_(self.derivative)

    # ---- parameters: [rows, D*(3K-1)] in the conditioner's own layout (spline.py:82-86) ----------------
    def _params(self, x2, latent):
        if self.latent_net is None:
            p = torch.cat([self.width, self.height, self.derivative], dim=-1).detach()
            return p.reshape(-1).to(device=x2.device, dtype=torch.float32).contiguous(), 0
        if latent is None:
            raise ValueError('Spline with a latent_net needs `latent`')
        p = self.latent_net(latent.reshape(-1, latent.shape[-1]))
        return p, p.stride(0)

    @property
    def params_per_element(self) -> int:
        return 2 * self.n_bins + self.derivative_dim                                 # spline.py:83

    def _bounds(self):
        return self.lower, self.upper, self.lower, self.upper                        # left, right, bottom, top

    def _launch(self, x, latent, reverse, want_ldj, want_ldiag, ldj_scale=1.0):
        _hip.require_device(x, 'x')
        x2, lead = flatten_rows(x)
        d = x2.shape[1]
        params, stride = self._params(x2, latent)
        l, r, b, t = self._bounds()
        if self.spline_type == 'cubic':
            y, ldj, ldiag = run_cubic_kernel(x2, params, stride, None, 0, d, self.n_bins, self.lower, self.upper, reverse,
                                             want_ldj, want_ldiag, ldj_scale)
        else:
            y, ldj, ldiag = run_rqs_kernel(x2, params, stride, None, 0, d, self.n_bins, l, r, b, t, reverse, want_ldj,
                                           want_ldiag, ldj_scale)
        return (y.reshape(*lead, d), None if ldj is None else ldj.reshape(*lead, 1),
                None if ldiag is None else ldiag.reshape(*lead, d))

    # ---- training (layer-wise autograd path) ---------------------------------------------------------------------
    def _autograd_supported(self) -> bool:
        return True          # own parameters, a net.MLP (forward_autograd) or any nn.Module latent_net (torch's graph)

    def _autograd_from_params(self, x2: torch.Tensor, params: torch.Tensor, reverse: bool, want_ldiag: bool = False):
        """(out, log-det [N][, log-diag [N, D]]) of every column from a per-row parameter tensor [N, D * P] (spline.py:82-86)
        with a graph; reverse: the inverse spline and its own (negated) log-derivative (rational_quadratic_spline.py:234)."""
        if self.spline_type == 'cubic':
            op = CubicInverse if reverse else CubicForward
        else:
            op = RQSInverse if reverse else RQSForward
        return op.apply(x2, params.to(torch.float32), None, 0, x2.shape[1], self.n_bins, self.lower, self.upper, 1.0, want_ldiag)

    def _autograd_inverse(self, x2: torch.Tensor, lat2=None, reverse: bool = True, want_ldiag: bool = False):
        """inverse_and_log_diag_jacobian summed over the columns, on fp32 rows, with a graph (RQSInverse)."""
        n, d = x2.shape
        if self.latent_net is None:
            p = torch.cat([self.width, self.height, self.derivative], dim=-1).reshape(1, -1)     # spline.py:78-79
            params = p.to(device=x2.device, dtype=torch.float32).expand(n, p.shape[1])
        else:
            if lat2 is None:
                raise ValueError('Spline with a latent_net needs `latent`')
            net = self.latent_net
            params = net.forward_autograd(lat2) if hasattr(net, 'forward_autograd') else net(lat2)          # spline.py:82-86
        return self._autograd_from_params(x2, params, reverse, want_ldiag)

    def _autograd_forward(self, x2: torch.Tensor, lat2=None):
        """forward_and_log_diag_jacobian summed over the columns, with a graph (quadratic splines: RQSForward)."""
        return self._autograd_inverse(x2, lat2, reverse=False)

    # ---- reference method set (spline.py:89-143): differentiable like the reference's (see flow.graph_wanted) -----------
    def _graph(self, x, latent, reverse: bool, want_ldiag: bool = False):
        """(out [..., D], log-det [..., 1], log-diag [..., D] | None) with a graph."""
        _hip.require_device(x, 'x')
        x2, lat2, lead = graph_rows(x, latent)
        d = x2.shape[1]
        out = self._autograd_inverse(x2, lat2, reverse=reverse, want_ldiag=want_ldiag)
        return out[0].reshape(*lead, d), out[1].reshape(*lead, 1), (out[2].reshape(*lead, d) if want_ldiag else None)

    def forward(self, x, latent=None, **kwargs):
        if graph_wanted(self, x, latent):
            return self._graph(x, latent, False)[0]
        return self._launch(x, latent, False, False, False)[0]

    def inverse(self, y, latent=None, **kwargs):
        if graph_wanted(self, y, latent):
            return self._graph(y, latent, True)[0]
        return self._launch(y, latent, True, False, False)[0]

    def forward_and_log_diag_jacobian(self, x, latent=None, *, reverse=False, **kwargs):
        if graph_wanted(self, x, latent):
            y, _, ld = self._graph(x, latent, reverse, True)
            return y, ld
        y, _, ld = self._launch(x, latent, reverse, False, True)
        return y, ld

    def inverse_and_log_diag_jacobian(self, y, latent=None, **kwargs):
        # the inverse spline already returns the negated value (rational_quadratic_spline.py:234): no extra sign
        return self.forward_and_log_diag_jacobian(y, latent, reverse=True)

    def forward_and_log_det_jacobian(self, x, latent=None, **kwargs):
        if graph_wanted(self, x, latent):
            return self._graph(x, latent, False)[:2]
        y, ldj, _ = self._launch(x, latent, False, True, False)
        return y, ldj

    def inverse_and_log_det_jacobian(self, y, latent=None, **kwargs):
        if graph_wanted(self, y, latent):
            return self._graph(y, latent, True)[:2]
        x, ldj, _ = self._launch(y, latent, True, True, False)
        return x, ldj

    def log_det_jacobian(self, x, y=None, latent=None, **kwargs):
        if graph_wanted(self, x, latent):
            return self._graph(x, latent, False)[1]
        return self._launch(x, latent, False, True, False)[1]

    def log_diag_jacobian(self, x, y=None, latent=None, **kwargs):
        if graph_wanted(self, x, latent):
            return self._graph(x, latent, False, True)[2]
        return self._launch(x, latent, False, False, True)[2]
