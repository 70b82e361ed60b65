"""stribor_amd — MI355X-native drop-in for stribor's coupling-flow hot path.

Same names and constructor signatures as ``stribor`` for the classes on the path
(``NormalizingFlow``, ``Coupling``, ``Affine``, ``Spline``, ``AffineLU``, ``MatrixExponential``,
``Permute``/``Flip``, ``Sigmoid``/``Logit``, ``ELU``, ``LeakyReLU``, ``Cumsum``/``Diff``, ``Identity``,
``UnitNormal``, ``net.MLP``, ``util.get_mask``); the arithmetic is hand-written
HIP for gfx950 behind the C ABI in ``include/stribor_hip.h``.  There is no CPU fallback.
"""
from . import net, util
from .dist import *          # noqa: F401,F403
from .dist.normal import UnitNormal
from .flow import ElementwiseTransform, NeuralFlow, NormalizingFlow, Transform
from .flows import (ELU, Affine, AffineLU, ContinuousAffineCoupling, Coupling, Cumsum, Diff, Flip, Identity, LeakyReLU, Logit, MatrixExponential,
                    Permute, Sigmoid, Spline)

from ._hip import GemmRangeError, check_errors, get_gemm_precision, set_gemm_precision, set_sync_errors

__version__ = '0.2.0'
