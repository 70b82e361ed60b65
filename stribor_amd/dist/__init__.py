from .normal import UnitNormal
