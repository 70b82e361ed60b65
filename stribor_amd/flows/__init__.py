from .affine import Affine
from .coupling import ContinuousAffineCoupling, Coupling
from .permute import Flip, Permute
from .spline import Spline
from .linear import AffineLU, MatrixExponential
from .pointwise import ELU, Cumsum, Diff, Identity, LeakyReLU, Logit, Sigmoid
