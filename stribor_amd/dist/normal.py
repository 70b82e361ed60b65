"""Base density (reference: stribor/dist/normal.py:40-54 -> torch Independent(Normal(0, 1), 1)).

Only ``UnitNormal`` is on the coupling-flow path.  It is an nn.Module here so that ``flow.to(device)``
moves it (the reference pins it to the CPU, quirk Q6).  ``log_prob`` is one HIP kernel
(row-wise sum of squares by wave shuffles); sampling uses torch's device RNG.
"""
import torch
import torch.nn as nn

from .. import _hip


class UnitNormal(nn.Module):
    def __init__(self, dim: int, **kwargs):
        super().__init__()
        self.dim = dim
        self.register_buffer('loc', torch.zeros(dim), persistent=False)
        self.register_buffer('scale', torch.ones(dim), persistent=False)

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        """[..., dim] -> [...]  (sum_d -x^2/2 - dim*log(sqrt(2*pi)), dist/normal.py:37)."""
        _hip.require_device(x, 'x')
        assert x.shape[-1] == self.dim
        if torch.is_grad_enabled() and x.requires_grad:       # differentiable like torch's Independent(Normal) (normal.py:37)
            xf = x.to(torch.float32)
            return -0.5 * (xf * xf).sum(-1) - self.dim * 0.9189385332046727
        x2 = x.reshape(-1, self.dim).contiguous()
        out = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
        _hip.call('sx_unit_normal_logprob', x2, x2.data_ptr(), None, out.data_ptr(), x2.shape[0], self.dim,
                                               _hip.dtype_code(x2))
        return out.reshape(x.shape[:-1])

    def sample(self, sample_shape=()) -> torch.Tensor:
        if isinstance(sample_shape, int):
            sample_shape = (sample_shape,)
        with torch.no_grad():
            return torch.randn(*sample_shape, self.dim, device=self.loc.device)

    def rsample(self, sample_shape=()) -> torch.Tensor:
        return self.sample(sample_shape)

    def forward(self, x):
        return self.log_prob(x)
