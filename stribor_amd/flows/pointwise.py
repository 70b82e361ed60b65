"""Parameter-free element-wise flows (reference: stribor/flows/sigmoid.py:9-56, flows/activations.py:11-101,
flows/cumsum.py:40-92, flows/identity.py:6-25): ``Sigmoid``, ``Logit``, ``ELU``, ``LeakyReLU``, ``Cumsum``, ``Diff``,
``Identity``.  Same constructors and method sets; every call is one ``sx_pointwise`` launch (values, per-element
log-derivative and its row sum in the same pass).  They do not join fused programs: a flow that contains one runs
layer by layer.
"""
import math

import torch

from .. import _hip
from ..flow import ElementwiseTransform, flatten_rows, graph_rows, graph_wanted

__all__ = ['Sigmoid', 'Logit', 'ELU', 'LeakyReLU', 'Cumsum', 'Diff', 'Identity']

PW_SIGMOID, PW_LOGIT, PW_ELU, PW_ELU_INV, PW_LEAKY, PW_LEAKY_INV, PW_CUMSUM, PW_DIFF = range(1, 9)


def run_pointwise(x, kind, param=0.0, want_y=True, want_ldj=False, want_ldiag=False):
    """Launch sx_pointwise on x[..., D] -> (y | None, ldj[..., 1] | None, ldiag[..., D] | None)."""
    _hip.require_device(x, 'x')
    x2, lead = flatten_rows(x)
    n, d = x2.shape
    y = torch.empty_like(x2) if want_y else None
    ldj = torch.empty(n, dtype=torch.float32, device=x2.device) if want_ldj else None
    ldiag = torch.empty(n, d, dtype=torch.float32, device=x2.device) if want_ldiag else None
    _hip.call('sx_pointwise', x2, x2.data_ptr(), _hip.ptr(y), _hip.ptr(ldj), _hip.ptr(ldiag), n, d, _hip.dtype_code(x2),
                                 kind, float(param), 0)
    return (None if y is None else y.reshape(*lead, d), None if ldj is None else ldj.reshape(*lead, 1),
            None if ldiag is None else ldiag.reshape(*lead, d))


class PointwiseOp(torch.autograd.Function):
    """(out, row log-det[, per-element log-derivative]) of one sx_pointwise kind as a differentiable op; backward =
    sx_pointwise_bwd (the adjoint of the per-element output rides on it as `gldiag`, ABI v3)."""

    @staticmethod
    def forward(ctx, x2, kind, param, want_ldiag=False):
        x2 = x2.contiguous()
        y, ldj, ldiag = run_pointwise(x2, kind, param, want_ldj=True, want_ldiag=bool(want_ldiag))
        ctx.save_for_backward(x2)
        ctx.meta = (kind, float(param))
        if want_ldiag:
            return y, ldj.reshape(-1), ldiag
        return y, ldj.reshape(-1)

    @staticmethod
    def backward(ctx, gy, gldj, gldiag=None):
        (x2,) = ctx.saved_tensors
        kind, param = ctx.meta
        n, d = x2.shape
        gy = (torch.zeros_like(x2) if gy is None else gy).to(torch.float32).contiguous()
        gl = None if gldj is None else gldj.to(torch.float32).contiguous()
        gd = None if gldiag is None else gldiag.to(torch.float32).contiguous()
        gx = torch.empty_like(x2)
        _hip.call('sx_pointwise_bwd', x2, x2.data_ptr(), gy.data_ptr(), _hip.ptr(gl), _hip.ptr(gd), gx.data_ptr(), n, d, kind, param)
        return gx, None, None, None


class _Pointwise(ElementwiseTransform):
    """fwd / inv kernel kinds + parameters; the inverse kinds return MINUS the forward log-derivative at the value they
    produce, which is what Transform.inverse_and_log_det_jacobian returns (flow.py:42-47).  Every method is differentiable like
    the reference's (see flow.graph_wanted): with a graph wanted the call is PointwiseOp."""
    _fwd = _inv = None

    def _p(self, reverse):
        return 0.0

    # ---- training (layer-wise autograd path) ------------------------------------------------------------------------
    def _autograd_supported(self) -> bool:
        return True

    def _autograd_inverse(self, x2, lat2=None):
        return PointwiseOp.apply(x2, self._inv, self._p(True))

    def _autograd_forward(self, x2, lat2=None):
        return PointwiseOp.apply(x2, self._fwd, self._p(False))

    # ---- fused-program hook: one SX_STEP_POINTWISE step (Cumsum / Diff mix columns: they stay out) -----------------------
    def _plan(self, builder, reverse, ldj_scale):
        kind = self._inv if reverse else self._fwd
        if kind in (PW_CUMSUM, PW_DIFF):
            return False
        param = self._p(reverse)
        log_slope = 0.0
        if kind == PW_LEAKY:
            log_slope = math.log(param)
        elif kind == PW_LEAKY_INV:
            log_slope = -math.log(param)                 # param = 1 / slope (sx_pointwise: log of the forward slope)
        # every kind's own log-derivative is that of the function it applies.  ldj_scale is the coefficient of the layer's FORWARD
        # log-det (flow.py:47: -1 for a flow's inverse pass); on the inverse pass the kind applied is the layer's inverse, whose
        # log-derivative is minus the layer's forward one at the value produced
        builder.add_pointwise(kind, param, log_slope, -ldj_scale if reverse else ldj_scale)
        return True

    def _graph(self, x, reverse: bool, want_ldiag: bool = False):
        """(out [..., D], log-det [..., 1], log-diag [..., D] | None) with a graph."""
        _hip.require_device(x, 'x')
        x2, _, lead = graph_rows(x)
        d = x2.shape[1]
        out = PointwiseOp.apply(x2, self._inv if reverse else self._fwd, self._p(reverse), want_ldiag)
        return out[0].reshape(*lead, d), out[1].reshape(*lead, 1), (out[2].reshape(*lead, d) if want_ldiag else None)

    def forward(self, x, **kwargs):
        if graph_wanted(None, x):
            return self._graph(x, False)[0]
        return run_pointwise(x, self._fwd, self._p(False))[0]

    def inverse(self, y, **kwargs):
        if graph_wanted(None, y):
            return self._graph(y, True)[0]
        return run_pointwise(y, self._inv, self._p(True))[0]

    def log_diag_jacobian(self, x, y=None, **kwargs):
        if graph_wanted(None, x):
            return self._graph(x, False, True)[2]
        return run_pointwise(x, self._fwd, self._p(False), want_y=False, want_ldiag=True)[2]

    def log_det_jacobian(self, x, y=None, **kwargs):
        if graph_wanted(None, x):
            return self._graph(x, False)[1]
        return run_pointwise(x, self._fwd, self._p(False), want_y=False, want_ldj=True)[1]

    def forward_and_log_det_jacobian(self, x, **kwargs):
        if graph_wanted(None, x):
            return self._graph(x, False)[:2]
        y, ldj, _ = run_pointwise(x, self._fwd, self._p(False), want_ldj=True)
        return y, ldj

    def inverse_and_log_det_jacobian(self, y, **kwargs):
        if graph_wanted(None, y):
            return self._graph(y, True)[:2]
        x, ldj, _ = run_pointwise(y, self._inv, self._p(True), want_ldj=True)
        return x, ldj

    def forward_and_log_diag_jacobian(self, x, **kwargs):
        if graph_wanted(None, x):
            y, _, ld = self._graph(x, False, True)
            return y, ld
        y, _, ld = run_pointwise(x, self._fwd, self._p(False), want_ldiag=True)
        return y, ld

    def inverse_and_log_diag_jacobian(self, y, **kwargs):
        if graph_wanted(None, y):
            x, _, ld = self._graph(y, True, True)
            return x, ld
        x, _, ld = run_pointwise(y, self._inv, self._p(True), want_ldiag=True)
        return x, ld


class Sigmoid(_Pointwise):
    """sigmoid.py:9-44 (values clamped to [tiny, 1 - eps] like the reference)."""
    _fwd, _inv = PW_SIGMOID, PW_LOGIT

    def __init__(self, **kwargs):
        super().__init__()


class Logit(_Pointwise):
    """sigmoid.py:46-56: the inverse of Sigmoid."""
    _fwd, _inv = PW_LOGIT, PW_SIGMOID

    def __init__(self, **kwargs):
        super().__init__()


class ELU(_Pointwise):
    """activations.py:11-63."""
    _fwd, _inv = PW_ELU, PW_ELU_INV

    def __init__(self, *args, **kwargs):             # the reference's ELU has no __init__ of its own: anything goes
        super().__init__()


class LeakyReLU(_Pointwise):
    """activations.py:66-101."""
    _fwd, _inv = PW_LEAKY, PW_LEAKY_INV

    def __init__(self, negative_slope: float = 0.01, **kwargs):
        super().__init__()
        assert negative_slope > 0, '`negative_slope` must be positive'              # activations.py:78
        self.negative_slope = negative_slope

    def _p(self, reverse):
        return 1 / self.negative_slope if reverse else self.negative_slope           # activations.py:91


class Cumsum(_Pointwise):
    """cumsum.py:40-74: cumulative sum over the last axis (sequential, torch.cumsum's order); log-det 0."""
    _fwd, _inv = PW_CUMSUM, PW_DIFF

    def __init__(self, dim: int):
        super().__init__()
        assert dim == -1, '`dim` must be equal to -1'                                # cumsum.py:55
        self.dim = dim


class Diff(Cumsum):
    """cumsum.py:76-85."""
    _fwd, _inv = PW_DIFF, PW_CUMSUM


class Identity(ElementwiseTransform):
    """identity.py:6-25."""
    def __init__(self, **kwargs):
        super().__init__()

    def forward(self, x, **kwargs):
        _hip.require_device(x, 'x')
        return x

    def inverse(self, y, **kwargs):
        _hip.require_device(y, 'y')
        return y

    def log_det_jacobian(self, x, y=None, **kwargs):
        return torch.zeros_like(x[..., :1], dtype=torch.float32)

    def log_diag_jacobian(self, x, y=None, **kwargs):
        return torch.zeros_like(x, dtype=torch.float32)

    def forward_and_log_det_jacobian(self, x, **kwargs):
        return self.forward(x), self.log_det_jacobian(x)

    def inverse_and_log_det_jacobian(self, y, **kwargs):
        return self.inverse(y), self.log_det_jacobian(y)

    def _autograd_supported(self) -> bool:
        return True

    def _autograd_inverse(self, x2, lat2=None):
        return x2, torch.zeros(x2.shape[0], dtype=torch.float32, device=x2.device)

    def _autograd_forward(self, x2, lat2=None):
        return x2, torch.zeros(x2.shape[0], dtype=torch.float32, device=x2.device)
