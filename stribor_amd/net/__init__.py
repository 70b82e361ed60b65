from .mlp import MLP
from .time_net import TimeFourier, TimeFourierBounded, TimeIdentity, TimeLinear, TimeLog, TimeTanh
