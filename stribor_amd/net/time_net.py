"""Time embeddings (reference: stribor/net/time_net.py:6-47): ``TimeIdentity``, ``TimeLinear``, ``TimeTanh``, ``TimeLog``.

Inside ``ContinuousAffineCoupling`` they are evaluated by the HIP kernel (``sx_time_affine_coupling`` takes the kind and
the scale vector, or -- in the fused SX_STEP_COUPLING_TIME step -- the per-column constants in the step's blob); ``forward`` /
``derivative`` here serve stand-alone calls with plain tensor ops on the caller's device.  ``TimeFourier`` /
``TimeFourierBounded`` are an in-kernel kind too (round 3); a user-written time net is called as a module.
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
    """Fourier features sum_k x_k sin(s_k t) (time_net.py:49-82).  Inside a fused ContinuousAffineCoupling / NeuralFlow program
    the sum is evaluated in the kernel (time kind 4: per-column weights and frequencies ride in the step's blob, hidden_dim <= 64);
    `forward` / `derivative` here serve stand-alone calls and the autograd path with tensor ops."""
    kind = 4                                                     # SX_TIME_FOURIER

    def __init__(self, out_dim: int, hidden_dim: int, lmbd: float = 0.5, bounded: bool = False, **kwargs):
        super().__init__()
        self.bounded = bounded
        self.hidden_dim = hidden_dim
        self.shift = nn.Parameter(-torch.log(1 - torch.rand(out_dim, hidden_dim)) / lmbd)      # time_net.py:63
        self.weight = nn.Parameter(torch.empty(out_dim, hidden_dim))
        nn.init.xavier_normal_(self.weight)

    def get_scale(self):
        if self.bounded:
            return torch.softmax(self.weight, -1) / 2                                          # time_net.py:69-70
        return self.weight / self.hidden_dim

    def forward(self, t):
        t = t.unsqueeze(-1)
        return (self.get_scale() * torch.sin(self.shift * t)).sum(-1)                          # time_net.py:74-79

    def derivative(self, t):
        t = t.unsqueeze(-1)
        return (self.shift * self.get_scale() * torch.cos(self.shift * t)).sum(-1)             # time_net.py:81-86


class TimeFourierBounded(TimeFourier):
    """Same as TimeFourier but between 0 and 1 (time_net.py:88-91)."""

    def __init__(self, out_dim: int, hidden_dim: int, lmbd: float = 0.5, **kwargs):
        super().__init__(out_dim, hidden_dim, lmbd, True)
