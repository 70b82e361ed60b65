"""Flip / Permute (reference: stribor/flows/permute.py:11-82): bit-exact column moves, log-det 0.

Stand-alone they run ``sx_permute`` (a byte gather); inside a fused flow they are free — the planner only
relabels which column each state slot holds.  The permutation is a registered buffer (the reference
keeps it as a plain attribute that ``state_dict`` and ``.to()`` miss, quirks Q6/Q7).
"""
from typing import List

import numpy as np
import torch

from .. import _hip
from ..flow import ElementwiseTransform, flatten_rows, graph_wanted

__all__ = ['Flip', 'Permute']


def _gather_columns(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _hip.require_device(x, 'x')
    if graph_wanted(None, x):            # differentiable like the reference's x[..., perm] (permute.py:71,75)
        return x.index_select(-1, idx.to(device=x.device, dtype=torch.long))
    if x.element_size() not in (2, 4):
        raise TypeError(f'sx_permute moves 2- or 4-byte elements (got {x.dtype})')
    x2, lead = flatten_rows(x)
    y = torch.empty_like(x2)
    idx = idx.to(device=x.device, dtype=torch.int32)
    _hip.call('sx_permute', x2, x2.data_ptr(), y.data_ptr(), idx.data_ptr(), x2.shape[0], x2.shape[1],
                               x2.element_size())
    return y.reshape(*lead, x2.shape[1])


class _ColumnShuffle(ElementwiseTransform):
    def _perm(self, dim: int) -> torch.Tensor:
        raise NotImplementedError

    def _feature_only(self) -> bool:
        """The layer only moves columns (what `_perm` describes); Flip over other axes overrides this."""
        return True

    def _inv(self, dim: int) -> torch.Tensor:
        p = self._perm(dim)
        inv = torch.empty_like(p)
        inv[p] = torch.arange(p.numel(), device=p.device, dtype=p.dtype)
        return inv

    def forward(self, x, **kwargs):
        return _gather_columns(x, self._perm(x.shape[-1]))

    def inverse(self, y, **kwargs):
        return _gather_columns(y, self._inv(y.shape[-1]))

    def log_det_jacobian(self, x, y=None, **kwargs):
        return torch.zeros_like(x[..., :1])                                  # permute.py:41,78

    def log_diag_jacobian(self, x, y=None, **kwargs):
        # permute.py:44,82: log of the diagonal of a permutation matrix (0 on fixed points, -inf elsewhere)
        p = self._perm(x.shape[-1]).to(x.device)
        fixed = p == torch.arange(p.numel(), device=x.device)
        d = torch.where(fixed, torch.zeros((), device=x.device), torch.full((), float('-inf'), device=x.device))
        return d.to(x.dtype).expand_as(x)

    def _plan(self, builder, reverse, ldj_scale):
        builder.add_permutation(self._perm(builder.dim).cpu().numpy(), reverse)
        return True


class Flip(_ColumnShuffle):
    """permute.py:11-44.  dims=[-1] (the default, what flows use between couplings) is the `sx_permute` byte gather and
    a free relabelling inside fused programs; any other axis list moves whole rows around -- `torch.flip`, a plain
    copy -- and such a flow runs layer by layer."""

    def __init__(self, dims: List[int] = [-1]):
        super().__init__()
        self.dims = list(dims)

    def _feature_only(self) -> bool:
        return self.dims == [-1]

    def _perm(self, dim):
        return torch.arange(dim - 1, -1, -1)

    def forward(self, x, **kwargs):
        if self._feature_only() or (x.dim() == 1 and self.dims in ([0], [-1])):
            return super().forward(x, **kwargs)
        _hip.require_device(x, 'x')
        return torch.flip(x, self.dims)                                       # permute.py:35

    def inverse(self, y, **kwargs):
        if self._feature_only() or (y.dim() == 1 and self.dims in ([0], [-1])):
            return super().inverse(y, **kwargs)
        _hip.require_device(y, 'y')
        return torch.flip(y, self.dims)                                       # permute.py:38

    def log_diag_jacobian(self, x, y=None, **kwargs):
        if self._feature_only():
            return super().log_diag_jacobian(x, y, **kwargs)
        return torch.eye(x.shape[-1], device=x.device).flip(self.dims).diag().log().to(x.dtype).expand_as(x)   # permute.py:44

    def _plan(self, builder, reverse, ldj_scale):
        if not self._feature_only():
            return False
        return super()._plan(builder, reverse, ldj_scale)


class Permute(_ColumnShuffle):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        self.register_buffer('permutation', torch.randperm(dim))              # permute.py:65
        inv = torch.empty(dim, dtype=torch.long)
        inv[self.permutation] = torch.arange(dim)                             # permute.py:67-68
        self.register_buffer('inverse_permutation', inv, persistent=False)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        if prefix + 'permutation' in missing_keys:       # a reference checkpoint never holds it (quirk Q7)
            missing_keys.remove(prefix + 'permutation')
        inv = torch.empty_like(self.permutation)
        inv[self.permutation] = torch.arange(self.dim, device=self.permutation.device)
        self.inverse_permutation = inv

    def _perm(self, dim):
        assert dim == self.dim
        return self.permutation

    def _plan_guards(self):
        return [self.permutation]          # baked into the fused programs' slot relabelling at plan time
