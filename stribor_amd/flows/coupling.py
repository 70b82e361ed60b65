"""Coupling layer (reference: stribor/flows/coupling.py:10-95).

``Coupling(transform, mask, set_data=False)`` wraps an elementwise transform whose ``latent_net`` maps the
masked input (optionally concatenated with ``latent``) to the transform's parameters.

* ``Coupling(Affine(latent_net=MLP))`` is ONE launch of the fused MFMA kernel per call: masked GEMM-1,
  tanh, pruned GEMM-2, affine, blend and per-sample log-det all stay in registers; the conditioner runs
  once even for ``*_and_log_det_jacobian`` (the reference runs it twice, quirk Q2).
* ``Coupling(Spline(quadratic, latent_net=MLP))`` runs the conditioner with the MFMA kernel (pruned to the
  transformed columns) and the spline in ``sx_rqs_coupling`` (parameters staged through LDS per wavefront).
"""
from typing import Optional

import numpy as np
import torch

from .. import _hip
from ..flow import Transform, flatten_rows
from ..fused import ProgramBuilder
from ..net.mlp import MLP, _chunk_mlp_program
from ..util.mask import get_mask
from .affine import Affine

__all__ = ['Coupling']


class Coupling(Transform):
    def __init__(self, transform, mask: str, set_data: bool = False, **kwargs):
        super().__init__()
        if set_data:
            raise NotImplementedError('stribor_amd.Coupling: set_data=True (masking over a set axis) is outside '
                                      'the coupling-flow hot path (SURVEY 8(f))')
        self.transform = transform
        self.mask_name = mask
        self.mask_func = get_mask(mask)                       # raises NotImplementedError like mask.py:20
        self.set_data = False
        self._masks = {}
        self._programs = {}

    # ---- mask: built once per width (the reference rebuilds it from numpy every call, quirk Q4) --------
    def mask_vector(self, dim: int) -> np.ndarray:
        if dim not in self._masks:
            m = self.mask_func(dim).numpy().astype(np.float64).reshape(-1)
            self._masks[dim] = np.full(dim, m[0]) if m.size == 1 else m
        return self._masks[dim]

    def _get_mask(self, x: torch.Tensor) -> torch.Tensor:
        return torch.from_numpy(self.mask_vector(x.shape[-1])).to(x).expand_as(x)     # coupling.py:48-53

    def _net(self) -> MLP:
        net = getattr(self.transform, 'latent_net', None)
        if not isinstance(net, MLP):
            raise NotImplementedError('stribor_amd.Coupling needs a transform with a stribor_amd.net.MLP latent_net')
        return net

    # ---- affine: one fused single-step program per (direction, width, latent width, device) -----------
    def _affine_program(self, reverse: bool, ldj_scale: float, dim: int, latent_dim: int, device):
        key = ('affine', reverse, ldj_scale, dim, latent_dim, str(device))
        if key not in self._programs:
            b = ProgramBuilder(dim, latent_dim, self._net().hidden_width)
            if not self._plan(b, reverse, ldj_scale):
                raise NotImplementedError('this coupling cannot run on the fused kernel')
            self._programs[key] = b.build(device)
        return self._programs[key]

    def _run(self, x, latent, reverse, want_y, want_ldj, ldj_scale=1.0):
        _hip.require_device(x, 'x')
        x2, lead = flatten_rows(x)
        d = x2.shape[1]
        lat2 = None if latent is None else latent.reshape(-1, latent.shape[-1])
        ld = 0 if lat2 is None else lat2.shape[1]
        if isinstance(self.transform, Affine):
            try:
                prog = self._affine_program(reverse, ldj_scale, d, ld, x.device)
            except NotImplementedError:
                prog = None            # e.g. a conditioner with several hidden layers: MLP kernel + element-wise kernel
            if prog is not None:
                y, ldj, _ = prog.run(x2, lat2, want_y, want_ldj, False)
            else:
                y, ldj = self._run_affine_unfused(x2, lat2, reverse, want_ldj, ldj_scale)
        else:
            from .spline import Spline
            if not isinstance(self.transform, Spline):
                raise NotImplementedError(f'Coupling({type(self.transform).__name__}) is not on the hot path')
            y, ldj = self._run_spline(x2, lat2, reverse, want_ldj, ldj_scale)
        return (None if y is None else y.reshape(*lead, d)), (None if ldj is None else ldj.reshape(*lead, 1))

    # ---- affine, unfused: pruned conditioner (MFMA program) + HBM-bound element-wise kernel -----------------
    def _affine_unfused_program(self, dim: int, latent_dim: int, device):
        key = ('affine-unfused', dim, latent_dim, str(device))
        if key not in self._programs:
            net = self._net()
            m = self.mask_vector(dim)
            live = np.nonzero(m <= 0.5)[0]
            cond = m > 0.5
            if dim == 1:
                cond = np.zeros(1, dtype=bool)
            out_rows = np.concatenate([live, dim + live])                           # (log_scale | shift) of live columns
            b = ProgramBuilder(dim, latent_dim, net.hidden_width)
            b.add_mlp(net.linears(), net.act_code, cond, out_rows)
            contiguous = len(live) > 0 and np.array_equal(live, np.arange(live[0], live[0] + len(live)))
            live_idx = None if contiguous else torch.from_numpy(live.astype(np.int32)).to(device)
            self._programs[key] = (_chunk_mlp_program(b, device), live_idx, int(live[0]) if len(live) else 0,
                                   len(live))
        return self._programs[key]

    def _run_affine_unfused(self, x2, lat2, reverse, want_ldj, ldj_scale):
        from .affine import run_affine_kernel
        n, d = x2.shape
        if not (self.mask_vector(d) <= 0.5).any():
            return x2.clone(), (torch.zeros(n, dtype=torch.float32, device=x2.device) if want_ldj else None)
        progs, live_idx, live_start, n_live = self._affine_unfused_program(d, 0 if lat2 is None else lat2.shape[1],
                                                                           x2.device)
        params = torch.empty(n, 2 * n_live, dtype=torch.float32, device=x2.device)
        for p in progs:
            p.run(x2, lat2, mlp_out=params)
        return run_affine_kernel(x2, params, params.stride(0), live_idx, live_start, n_live, reverse, True, want_ldj,
                                 ldj_scale)

    # ---- spline: pruned conditioner (MFMA) + LDS-staged spline kernel --------------------------------------
    def _spline_program(self, dim: int, latent_dim: int, device):
        key = ('spline', dim, latent_dim, str(device))
        if key not in self._programs:
            net, sp = self._net(), self.transform
            m = self.mask_vector(dim)
            live = np.nonzero(m <= 0.5)[0]
            cond = m > 0.5
            if dim == 1:
                cond = np.zeros(1, dtype=bool)                                       # coupling.py:62-63
            P = sp.params_per_element                                                # 3K-1 quadratic, 2K+2 cubic
            out_rows = (live[:, None] * P + np.arange(P)[None, :]).reshape(-1)       # spline.py:82-86, pruned
            b = ProgramBuilder(dim, latent_dim, net.hidden_width)
            b.add_mlp(net.linears(), net.act_code, cond, out_rows)
            contiguous = len(live) > 0 and np.array_equal(live, np.arange(live[0], live[0] + len(live)))
            live_idx = None if contiguous else torch.from_numpy(live.astype(np.int32)).to(device)
            self._programs[key] = (_chunk_mlp_program(b, device), live_idx, int(live[0]) if len(live) else 0,
                                   len(live), len(out_rows))
        return self._programs[key]

    def _run_spline(self, x2, lat2, reverse, want_ldj, ldj_scale):
        from .spline import run_rqs_kernel
        sp = self.transform
        n, d = x2.shape
        if not (self.mask_vector(d) <= 0.5).any():      # dim == 1: mask = [1], nothing is transformed (mask.py:37-38)
            return x2.clone(), (torch.zeros(n, dtype=torch.float32, device=x2.device) if want_ldj else None)
        progs, live_idx, live_start, n_live, width = self._spline_program(d, 0 if lat2 is None else lat2.shape[1],
                                                                          x2.device)
        params = torch.empty(n, width, dtype=torch.float32, device=x2.device)
        for p in progs:
            p.run(x2, lat2, mlp_out=params)
        if sp.spline_type == 'cubic':
            from .spline import run_cubic_kernel
            # one pass: the inverse kernel returns its own (already negated) log-derivative, like the quadratic path.
            # (The reference evaluates MINUS the FORWARD log-det at the inverted point, flow.py:42-47; the two differ
            # only for elements within an ulp of the domain boundary, where the log-derivative jumps to the tails' 0.)
            y, ldj, _ = run_cubic_kernel(x2, params, params.stride(0), live_idx, live_start, n_live, sp.n_bins,
                                         sp.lower, sp.upper, reverse, want_ldj, False, ldj_scale)
        else:
            y, ldj, _ = run_rqs_kernel(x2, params, params.stride(0), live_idx, live_start, n_live, sp.n_bins,
                                       sp.lower, sp.upper, sp.lower, sp.upper, reverse, want_ldj, False, ldj_scale)
        return y, ldj

    # ---- training (autograd): spline couplings, inverse direction --------------------------------------------------
    def _autograd_supported(self) -> bool:
        from .spline import Spline
        if not isinstance(getattr(self.transform, 'latent_net', None), MLP):
            return False
        return isinstance(self.transform, Affine) or \
            (isinstance(self.transform, Spline) and self.transform.spline_type == 'quadratic')

    def _autograd_inverse(self, x2: torch.Tensor, lat2=None):
        """inverse_and_log_det_jacobian on fp32 rows [N, D] with a graph: the conditioner runs through torch's own
        Linear layers (rocBLAS; only the rows of the last layer that parameterise transformed columns), the transform
        and its backward are the HIP kernels behind ``RQSInverse`` / ``AffineCouplingOp``.
        Returns (x_out [N, D], ldj [N])."""
        from .spline import RQSInverse, Spline
        from .affine import AffineCouplingOp
        sp, net = self.transform, self._net()
        is_spline = isinstance(sp, Spline)
        n, d = x2.shape
        m = self.mask_vector(d)
        live = np.nonzero(m <= 0.5)[0]
        if len(live) == 0:
            return x2, torch.zeros(n, dtype=torch.float32, device=x2.device)
        key = ('autograd', d, str(x2.device))
        if key not in self._programs:
            if is_spline:
                P = sp.params_per_element
                rows = (live[:, None] * P + np.arange(P)[None, :]).reshape(-1)      # spline.py:82-86
            else:
                rows = np.concatenate([live, d + live])                               # affine.py:66 (log_scale | shift)
            contiguous = np.array_equal(live, np.arange(live[0], live[0] + len(live)))
            self._programs[key] = (torch.from_numpy(m.astype(np.float32)).to(x2.device),
                                   torch.from_numpy(rows.astype(np.int64)).to(x2.device),
                                   None if contiguous else torch.from_numpy(live.astype(np.int32)).to(x2.device))
        mask_t, rows_t, live_idx = self._programs[key]
        z = x2 * mask_t                                                              # coupling.py:61
        if d == 1:
            z = z * 0                                                                # coupling.py:62-63
        if lat2 is not None:
            z = torch.cat([z, lat2], -1)                                             # coupling.py:64-65
        layers = list(net.net)
        h = z
        for layer in layers[:-1]:
            h = layer(h)
        last = layers[-1]
        params = torch.nn.functional.linear(h, last.weight.index_select(0, rows_t), last.bias.index_select(0, rows_t))
        if is_spline:
            return RQSInverse.apply(x2, params, live_idx, int(live[0]), len(live), sp.n_bins, sp.lower, sp.upper, 1.0)
        # Transform.inverse_and_log_det_jacobian: minus the forward log-det (flow.py:47)
        return AffineCouplingOp.apply(x2, params, live_idx, int(live[0]), len(live), True, -1.0)

    # ---- reference method set (coupling.py:69-95) -----------------------------------------------------------
    def forward(self, x, latent=None, reverse: bool = False, **kwargs):
        return self._run(x, latent, reverse, True, False)[0]

    def inverse(self, y, latent=None, **kwargs):
        return self._run(y, latent, True, True, False)[0]                            # coupling.py:81-82 (Q3)

    def log_det_jacobian(self, x, y=None, latent=None, **kwargs):
        return self._run(x, latent, False, False, True)[1]

    def forward_and_log_det_jacobian(self, x, latent=None, **kwargs):
        return self._run(x, latent, False, True, True)

    def inverse_and_log_det_jacobian(self, y, latent=None, **kwargs):
        if isinstance(self.transform, Affine):
            # the log-scales that invert y are the forward log-det at x (same conditioner input): one launch
            return self._run(y, latent, True, True, True, ldj_scale=-1.0)
        # spline: inverse kernel returns the already-negated log-diag-Jacobian (rational_quadratic_spline.py:234)
        return self._run(y, latent, True, True, True, ldj_scale=1.0)

    # ---- fused-program hooks --------------------------------------------------------------------------------
    def _plan_hidden_width(self):
        return self._net().hidden_width if isinstance(getattr(self.transform, 'latent_net', None), MLP) else 0

    def _plan_spline(self, builder: ProgramBuilder, reverse: bool, ldj_scale: float) -> bool:
        sp = self.transform
        net = getattr(sp, 'latent_net', None)
        if not isinstance(net, MLP) or net.activation_name != 'Tanh' or sp.n_bins > 16 or sp.spline_type != 'quadratic':
            return False             # cubic-spline couplings run layer by layer (MLP program + sx_cubic_coupling)
        lin = net.linears()
        if len(lin) != 2:
            return False
        (W1, b1), (W2, b2) = lin
        if W1.shape[1] != builder.dim + builder.latent_dim or W2.shape[0] != builder.dim * (3 * sp.n_bins - 1):
            raise ValueError(f'latent_net maps {W1.shape[1]} -> {W2.shape[0]}, expected '
                             f'{builder.dim + builder.latent_dim} -> {builder.dim * (3 * sp.n_bins - 1)}')
        builder.add_coupling_rqs(W1, b1, W2, b2, self.mask_vector(builder.dim), reverse, ldj_scale, W1.shape[0],
                                 sp.n_bins, sp.lower, sp.upper, sp.lower, sp.upper)
        return True

    def _plan_first_mask(self, dim):
        return self.mask_vector(dim)

    def _plan(self, builder: ProgramBuilder, reverse: bool, ldj_scale: float) -> bool:
        from .spline import Spline
        if isinstance(self.transform, Spline):
            return self._plan_spline(builder, reverse, ldj_scale)
        if not isinstance(self.transform, Affine) or not isinstance(getattr(self.transform, 'latent_net', None), MLP):
            return False
        net = self._net()
        lin = net.linears()
        if len(lin) != 2:
            return False                  # deeper conditioners take the per-layer path
        (W1, b1), (W2, b2) = lin
        if W1.shape[1] != builder.dim + builder.latent_dim or W2.shape[0] != 2 * builder.dim:
            raise ValueError(f'latent_net maps {W1.shape[1]} -> {W2.shape[0]}, expected '
                             f'{builder.dim + builder.latent_dim} -> {2 * builder.dim}')
        builder.add_coupling_affine(W1, b1, W2, b2, self.mask_vector(builder.dim), net.act_code, reverse, ldj_scale,
                                    W1.shape[0])
        return True
