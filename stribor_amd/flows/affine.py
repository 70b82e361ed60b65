"""Elementwise affine flow (reference: stribor/flows/affine.py:13-123).

``Affine(dim, *, latent_net=None, scale=None, shift=None)`` — keyword-only like the reference.
Arithmetic runs in ``sx_affine_coupling`` (one HBM pass: read x, read params, write y, wave-shuffle
row sums for the log-det); inside a fused flow a parameter-only Affine is a constant step.
"""
from numbers import Number
from typing import Optional

import torch
import torch.nn as nn

from .. import _hip
from ..flow import ElementwiseTransform, flatten_rows, graph_rows, graph_wanted

__all__ = ['Affine']


def run_affine_kernel(x2, params, params_stride, live_idx, live_start, n_live, reverse, want_y, want_ldj,
                      ldj_scale=1.0):
    """Thin launcher of sx_affine_coupling on [N, D] rows."""
    n, d = x2.shape
    y = torch.empty_like(x2) if want_y else torch.empty_like(x2)     # the kernel always writes y
    ldj = torch.empty(n, dtype=torch.float32, device=x2.device) if want_ldj else None
    _hip.call('sx_affine_coupling', x2, x2.data_ptr(), y.data_ptr(), _hip.ptr(ldj), params.data_ptr(), params_stride,
                                       _hip.ptr(live_idx), live_start, n_live, n, d, _hip.dtype_code(x2),
                                       int(reverse), 0, float(ldj_scale))
    return y, ldj


class AffineCouplingOp(torch.autograd.Function):
    """(y, row log-det) of the element-wise affine map on the live columns as a differentiable op (layer-wise training
    path): forward = sx_affine_coupling, backward = sx_affine_coupling_bwd.  params = [N, 2*n_live] (log_scale | shift)."""

    @staticmethod
    def forward(ctx, x2, params, live_idx, live_start, n_live, reverse, ldj_scale):
        x2, params = x2.contiguous(), params.contiguous()
        y, ldj = run_affine_kernel(x2, params, params.stride(0), live_idx, live_start, n_live, reverse, True, True, ldj_scale)
        ctx.save_for_backward(x2, params)
        ctx.meta = (live_idx, live_start, n_live, bool(reverse), float(ldj_scale))
        return y, ldj

    @staticmethod
    def backward(ctx, gy, gldj):
        x2, params = ctx.saved_tensors
        live_idx, live_start, n_live, reverse, ldj_scale = ctx.meta
        n, d = x2.shape
        gy = (torch.zeros_like(x2) if gy is None else gy).to(torch.float32).contiguous()
        gldj = (torch.zeros(n, device=x2.device) if gldj is None else gldj).to(torch.float32).contiguous()
        gx = gy.clone()
        gparams = torch.empty_like(params)
        _hip.call('sx_affine_coupling_bwd', x2, x2.data_ptr(), gy.data_ptr(), gldj.data_ptr(), params.data_ptr(),
                                               params.stride(0), gx.data_ptr(), gparams.data_ptr(), _hip.ptr(live_idx),
                                               live_start, n_live, n, d, int(reverse), ldj_scale)
        return gx, gparams, None, None, None, None, None


class Affine(ElementwiseTransform):
    def __init__(self, dim: int, *, latent_net: Optional[nn.Module] = None, scale=None, shift=None, **kwargs):
        super().__init__()
        self.dim = dim
        self.latent_net = latent_net
        if latent_net is None:
            if scale is None:
                self.log_scale = nn.Parameter(torch.empty(1, dim))            # affine.py:47-50
                self.shift = nn.Parameter(torch.empty(1, dim))
                nn.init.xavier_uniform_(self.log_scale)
                nn.init.xavier_uniform_(self.shift)
            else:
                if isinstance(scale, Number):
                    scale, shift = torch.tensor([float(scale)]), torch.tensor([float(shift)])
                assert torch.all(scale > 0), '`scale` mush have positive values'  # affine.py:55
                # buffers (not plain attributes, quirk Q6) so .to(device) moves them; kept out of state_dict
                self.register_buffer('log_scale', scale.float().log(), persistent=False)
                self.register_buffer('shift', shift.float().clone(), persistent=False)

    # ---- parameters ---------------------------------------------------------------------------------------
    def _const_params(self, device) -> torch.Tensor:
        ls = self.log_scale.detach().reshape(-1)
        sh = self.shift.detach().reshape(-1)
        if ls.numel() == 1:
            ls, sh = ls.expand(self.dim), sh.expand(self.dim)
        return torch.cat([ls, sh]).to(device=device, dtype=torch.float32).contiguous()

    def _params(self, x2, latent):
        """-> (params [rows, 2D] fp32, row stride)"""
        if self.latent_net is None:
            return self._const_params(x2.device), 0
        if latent is None:
            raise ValueError('Affine with a latent_net needs `latent`')
        p = self.latent_net(latent.reshape(-1, latent.shape[-1]))            # affine.py:66
        return p, p.stride(0)

    def _launch(self, x, latent, reverse, want_y, want_ldj, ldj_scale=1.0):
        _hip.require_device(x, 'x')
        x2, lead = flatten_rows(x)
        d = x2.shape[1]
        params, stride = self._params(x2, latent)
        y, ldj = run_affine_kernel(x2, params, stride, None, 0, d, reverse, want_y, want_ldj, ldj_scale)
        return (y.reshape(*lead, d) if want_y else None), (None if ldj is None else ldj.reshape(*lead, 1))

    # ---- reference method set (affine.py:69-123): differentiable like the reference's (see flow.graph_wanted) -----------
    def _graph(self, x, latent, reverse: bool):
        _hip.require_device(x, 'x')
        x2, lat2, lead = graph_rows(x, latent)
        y, ldj = self._autograd_inverse(x2, lat2, reverse=reverse)
        return y.reshape(*lead, x2.shape[1]), ldj.reshape(*lead, 1)

    def forward(self, x, latent=None, **kwargs):
        if graph_wanted(self, x, latent):
            return self._graph(x, latent, False)[0]
        return self._launch(x, latent, False, True, False)[0]

    def inverse(self, y, latent=None, **kwargs):
        if graph_wanted(self, y, latent):
            return self._graph(y, latent, True)[0]
        return self._launch(y, latent, True, True, False)[0]

    def log_det_jacobian(self, x, y=None, latent=None, **kwargs):
        if graph_wanted(self, x, latent):
            return self.log_diag_jacobian(x, y, latent=latent).sum(-1, keepdim=True)      # affine.py:119-120
        return self._launch(x, latent, False, False, True)[1]

    def forward_and_log_det_jacobian(self, x, latent=None, *, reverse: bool = False, **kwargs):
        if graph_wanted(self, x, latent):
            y, ldj = self._graph(x, latent, reverse)
            return y, (-ldj if reverse else ldj)                               # affine.py:97-109: +sum(log_scale) either way
        return self._launch(x, latent, reverse, True, True)                    # affine.py:97-109

    def inverse_and_log_det_jacobian(self, y, latent=None, **kwargs):
        if graph_wanted(self, y, latent):
            return self._graph(y, latent, True)
        return self._launch(y, latent, True, True, True, ldj_scale=-1.0)      # affine.py:111-113

    def log_diag_jacobian(self, x, y=None, latent=None, **kwargs):
        x2, lead = flatten_rows(x)
        if graph_wanted(self, x, latent):                                     # the parameters keep their graph
            d = x2.shape[1]
            if self.latent_net is None:
                ls = self.log_scale.reshape(-1)
                ls = (ls.expand(d) if ls.numel() == 1 else ls).to(x.device, torch.float32).expand(x2.shape[0], d)
            else:
                if latent is None:
                    raise ValueError('Affine with a latent_net needs `latent`')
                ls = self.latent_net(latent.reshape(-1, latent.shape[-1]))[..., :d]       # affine.py:66,122-123
            return ls.reshape(*lead, d)
        params, stride = self._params(x2, latent)
        ls = params[..., :x2.shape[1]]                                        # affine.py:122-123
        if stride == 0:
            ls = ls.reshape(1, -1).expand(x2.shape[0], -1)
        return ls.reshape(*lead, x2.shape[1])

    # ---- training (layer-wise autograd path) ---------------------------------------------------------------------
    def _autograd_supported(self) -> bool:
        return True

    def _autograd_forward(self, x2: torch.Tensor, lat2=None):
        return self._autograd_inverse(x2, lat2, reverse=False)

    def _autograd_from_params(self, x2: torch.Tensor, params: torch.Tensor, reverse: bool):
        """(out, log-det [N]) of every column from a per-row parameter tensor [N, 2D] (log_scale | shift, affine.py:66) with a
        graph: inverse_and_log_det_jacobian (reverse) or forward_and_log_det_jacobian (affine.py:97-113)."""
        return AffineCouplingOp.apply(x2, params.to(torch.float32).contiguous(), None, 0, x2.shape[1], bool(reverse),
                                      -1.0 if reverse else 1.0)

    def _autograd_inverse(self, x2: torch.Tensor, lat2=None, reverse: bool = True):
        """inverse_and_log_det_jacobian (or, reverse=False, forward_and_log_det_jacobian) on fp32 rows with a graph
        (AffineCouplingOp over all columns); the parameters are the module's own (broadcast over the rows) or the
        latent_net's output through torch's Linear layers."""
        n, d = x2.shape
        if self.latent_net is None:
            ls, sh = self.log_scale.reshape(-1), self.shift.reshape(-1)
            if ls.numel() == 1:
                ls, sh = ls.expand(d), sh.expand(d)
            params = torch.cat([ls, sh]).to(device=x2.device, dtype=torch.float32).unsqueeze(0).expand(n, 2 * d)
        else:
            if lat2 is None:
                raise ValueError('Affine with a latent_net needs `latent`')
            net = self.latent_net
            params = net.forward_autograd(lat2) if hasattr(net, 'forward_autograd') else net(lat2)   # affine.py:66
        return self._autograd_from_params(x2, params, reverse)

    # ---- fused-program hooks --------------------------------------------------------------------------------
    def _plan_hidden_width(self):
        return 0

    def _plan(self, builder, reverse, ldj_scale):
        if self.latent_net is not None:
            return False
        builder.add_affine_const(self.log_scale, self.shift, reverse, ldj_scale)
        return True
