from . import mask
from .mask import get_mask


def quadratic_spline_latent_dim(dim: int, n_bins: int) -> int:
    """Conditioner output width of a rational-quadratic spline (util/rational_quadratic_spline.py:7-8)."""
    return dim * (3 * n_bins - 1)
