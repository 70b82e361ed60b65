"""Conditioner MLP (reference: stribor/net/mlp.py:6-65).

Same constructor, same ``state_dict`` keys (``net.{0,2,...}.{weight,bias}``), last bias zero-filled
(mlp.py:53).  ``forward`` runs the whole network in ONE launch of the fused MFMA kernel
(hidden activations never reach HBM); inside a Coupling the network is not called at all — its
weights are consumed directly by the coupling kernel.
"""
from typing import Callable, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from ..fused import ProgramBuilder


class MLP(nn.Module):
    def __init__(self, in_dim: int, hidden_dims: List[int], out_dim: int, activation: Union[str, Callable] = 'Tanh',
                 final_activation: Optional[str] = None, nn_linear_wrapper_func: Optional[Callable] = None, **kwargs):
        super().__init__()
        if nn_linear_wrapper_func is not None:
            raise NotImplementedError('stribor_amd.net.MLP: nn_linear_wrapper_func (spectral norm) is used by '
                                      'IResNet only, which is outside the coupling-flow path')
        if final_activation is not None:
            raise NotImplementedError('stribor_amd.net.MLP: final_activation is not used on the coupling-flow path')
        act_name = activation if isinstance(activation, str) else type(activation).__name__
        if act_name not in _hip.ACT_CODES:
            raise NotImplementedError(f'activation {act_name!r}; supported: {sorted(_hip.ACT_CODES)}')
        self.activation_name = act_name
        self.in_dim, self.out_dim = in_dim, out_dim
        widths = [in_dim] + list(hidden_dims) + [out_dim]
        act = getattr(nn, act_name)()
        layers: List[nn.Module] = []
        for i in range(len(widths) - 1):
            if i:
                layers.append(act)                        # keeps the reference's indices 0, 2, 4, ...
            layers.append(nn.Linear(widths[i], widths[i + 1]))
        with torch.no_grad():
            layers[-1].bias.zero_()                       # mlp.py:53
        self.net = nn.Sequential(*layers)
        self._programs = {}

    # -- pieces the coupling / spline planners consume -------------------------------------------------
    def linears(self):
        return [(m.weight, m.bias) for m in self.net if isinstance(m, nn.Linear)]

    @property
    def act_code(self) -> int:
        return _hip.ACT_CODES[self.activation_name]

    @property
    def hidden_width(self) -> int:
        ls = self.linears()
        return max([w.shape[0] for (w, _) in ls[:-1]] + [1])

    # -- standalone evaluation ---------------------------------------------------------------------------
    def _program(self, device):
        key = str(device)
        if key not in self._programs:
            b = ProgramBuilder(self.in_dim, 0, self.hidden_width)
            b.add_mlp(self.linears(), self.act_code, None, np.arange(self.out_dim))
            self._programs[key] = _chunk_mlp_program(b, device)
        return self._programs[key]

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        _hip.require_device(x, 'MLP input')
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).to(torch.float32).contiguous()
        out = torch.empty(x2.shape[0], self.out_dim, dtype=torch.float32, device=x.device)
        for prog in self._program(x.device):
            prog.run(x2, mlp_out=out)
        return out.reshape(*lead, self.out_dim)


def _chunk_mlp_program(builder: ProgramBuilder, device):
    """A program holds at most SX_MAX_STEPS steps; very wide outputs are split over several launches
    (each recomputes the hidden layers, which is cheap next to the output GEMM)."""
    n_hidden = sum(1 for s in builder.steps if s['kind'] != _hip.STEP_MLP_OUT_TILE)
    outs = [s for s in builder.steps if s['kind'] == _hip.STEP_MLP_OUT_TILE]
    head = builder.steps[:n_hidden]
    room = _hip.SX_MAX_STEPS - n_hidden
    progs = []
    all_steps = builder.steps
    for i in range(0, max(len(outs), 1), room):
        builder.steps = head + outs[i:i + room]
        progs.append(builder.build(device))
    builder.steps = all_steps
    # the chunks share one set of pack jobs / one blob buffer
    for p in progs[1:]:
        p.blobs, p.jobs = progs[0].blobs, []
    return progs
