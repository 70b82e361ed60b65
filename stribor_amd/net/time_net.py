"""Time embeddings (reference: stribor/net/time_net.py:6-47): ``TimeIdentity``, ``TimeLinear``, ``TimeTanh``, ``TimeLog``.

Inside ``ContinuousAffineCoupling`` they are evaluated by the HIP kernel (``sx_time_affine_coupling`` takes the kind and
the scale vector); ``forward`` / ``derivative`` here serve stand-alone calls with plain tensor ops on the caller's
device.  ``TimeFourier`` is not on the path and raises.
"""
import torch
import torch.nn as nn

__all__ = ['TimeIdentity', 'TimeLinear', 'TimeTanh', 'TimeLog', 'TimeFourier', 'TimeFourierBounded']


class TimeIdentity(nn.Module):
    kind = 0                                                     # SX_TIME_IDENTITY

    def __init__(self, out_dim: int, **kwargs):
        super().__init__()
        self.out_dim = out_dim

    def forward(self, t):
        return t.repeat_interleave(self.out_dim, dim=-1)        # time_net.py:11

    def derivative(self, t):
        return torch.ones_like(t).repeat_interleave(self.out_dim, dim=-1)


class TimeLinear(nn.Module):
    kind = 1                                                     # SX_TIME_LINEAR

    def __init__(self, out_dim: int, **kwargs):
        super().__init__()
        self.out_dim = out_dim
        self.scale = nn.Parameter(torch.randn(1, out_dim))      # time_net.py:19-20
        nn.init.xavier_uniform_(self.scale)

    def forward(self, t):
        return self.scale * t

    def derivative(self, t):
        return self.scale * torch.ones_like(t)


class TimeTanh(TimeLinear):
    kind = 2                                                     # SX_TIME_TANH

    def forward(self, t):
        return torch.tanh(self.scale * t)                       # time_net.py:30

    def derivative(self, t):
        return self.scale * (1 - self.forward(t) ** 2)


class TimeLog(TimeLinear):
    kind = 3                                                     # SX_TIME_LOG

    def forward(self, t):
        return torch.log(self.scale.exp() * t + 1)              # time_net.py:38

    def derivative(self, t):
        return self.scale.exp() / (self.scale.exp() * t + 1)


class TimeFourier(nn.Module):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError('stribor_amd.net.TimeFourier is outside the coupling-flow path')


TimeFourierBounded = TimeFourier
