"""AffineLU / MatrixExponential — filled in by the linear-layer milestone."""
from ..flow import Transform

__all__ = ['AffineLU', 'MatrixExponential']


class AffineLU(Transform):
    def __init__(self, *a, **k):
        raise NotImplementedError

    def forward(self, x, **kw): ...
    def inverse(self, y, **kw): ...
    def log_det_jacobian(self, x, y, **kw): ...


class MatrixExponential(Transform):
    def __init__(self, *a, **k):
        raise NotImplementedError

    def forward(self, x, **kw): ...
    def inverse(self, y, **kw): ...
    def log_det_jacobian(self, x, y, **kw): ...
