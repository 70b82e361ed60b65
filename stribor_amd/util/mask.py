"""Coupling masks by name (reference: stribor/util/mask.py:6-57).

``get_mask(name)`` returns a callable ``dim -> float tensor [dim]`` of 0/1 like the reference:
1 = the column passes through and feeds the conditioner, 0 = the column is transformed.
Unlike the reference (quirk Q4) callers build the vector once at construction; ``random_half`` is
drawn once per call of the generator, so a Coupling stores its draw (quirk Q5).
"""
from typing import Callable

import torch

__all__ = ['get_mask']

_ORDERED = {'ordered_right_half': False, 'ordered_0': False, 'ordered_left_half': True, 'ordered_1': True}
_PARITY = {'parity_even': False, 'parity_odd': True}


def _zero_count(dim: int, ratio: float = 0.5) -> int:
    return min(max(int(dim * ratio), 1), dim - 1)


def _make(kind: str, flip: bool) -> Callable[[int], torch.Tensor]:
    def gen(dim: int) -> torch.Tensor:
        if kind == 'none':
            return torch.zeros(1)
        if dim == 1:
            return torch.ones(1)
        m = torch.ones(dim)
        if kind == 'ordered':
            m[:_zero_count(dim)] = 0.0
        elif kind == 'parity':
            m[0::2] = 0.0
        elif kind == 'random':
            m = torch.zeros(dim)
            m[torch.randperm(dim)[:_zero_count(dim)]] = 1.0
            return m
        return 1.0 - m if flip else m
    return gen


def get_mask(mask: str) -> Callable[[int], torch.Tensor]:
    if mask == 'none':
        return _make('none', False)
    if mask in _ORDERED:
        return _make('ordered', _ORDERED[mask])
    if mask in _PARITY:
        return _make('parity', _PARITY[mask])
    if mask == 'random_half':
        return _make('random', False)
    raise NotImplementedError(mask)
