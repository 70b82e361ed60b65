"""Conditioner MLP (reference: stribor/net/mlp.py:6-65).

Same constructor, same ``state_dict`` keys (``net.{0,2,...}.{weight,bias}``), last bias zero-filled
(mlp.py:53).  ``forward`` runs the whole network in ONE launch of the fused MFMA kernel
(hidden activations never reach HBM); inside a Coupling the network is not called at all — its
weights are consumed directly by the coupling kernel.
"""
from collections import OrderedDict
from typing import Callable, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from ..fused import ProgramBuilder, ProgramCache, StructureTracked


class BatchLinear(torch.autograd.Function):
    """y = x W^T + b for the layer-wise training path.  Forward and dL/dx: MFMA programs (round 6; up to 128 contracted and 4064
    output features, library GEMMs beyond).  dL/dW = (dL/dy)^T x and
    dL/db = sum_n dL/dy contract over the batch: for narrow layers (output <= 256 features) that is a tall-skinny product
    library GEMMs run on a handful of workgroups (0.63 ms for a 64 x 64 gradient over 2^18 rows) -- sx_wgrad computes
    both; for wide layers (a spline conditioner's 1504 rows) the library GEMM fills the chip and keeps dL/dW, and the
    bias gradient is sx_colsum (deterministic, and unlike torch's multi-block column sum on this build it replays
    correctly from a HIP graph)."""

    MIN_ROWS = 4096        # below this the library's weight gradient is as fast
    MAX_OUT = 256          # wider outputs: the library GEMM wins (measured on the 1504-wide layer)

    @staticmethod
    def eligible(x: torch.Tensor, W: torch.Tensor) -> bool:
        return x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and W.dtype == torch.float32

    # Round 6: forward and dL/dx as hand-written MFMA programs too (SX_STEP_MLP_INPUT: the output tiles contract the
    # program's input; v_mfma_f32_32x32x2_f32).  One launch holds 128 contracted features and up to 127 output tiles (a spline
    # conditioner's 64 -> 1504 layer: 0.61 ms, the library's 0.61); every pack of a program is ONE sx_pack_linear_batch launch, and the
    # programs of a weight are kept (LRU) and re-packed when its version moves.  A wider contraction runs as one launch per 128 columns,
    # the later ones adding into y (tested: PROGRAM_MAX_IN = 2048) -- but measures 2.4 - 4.3 x the library GEMM at 2^18 rows (200 -> 64:
    # 0.30 vs 0.09 ms; dL/dx of 64 -> 1504: 1.94 vs 0.45 ms; tools/experiments/linear_program_vs_library.py), so by default contractions
    # beyond 128 columns stay with the library.
    PROGRAM_MAX_IN, PROGRAM_MAX_OUT = 128, 127 * 32
    _programs = OrderedDict()          # (weight storage, shape, orientation, bias storage, device) -> programs, least recently used first
    _PROGRAMS_KEPT = 64

    @staticmethod
    def _linear_programs(W, b, transpose: bool, device):
        key = (W.data_ptr(), tuple(W.shape), bool(transpose), None if b is None else b.data_ptr(), str(device))
        progs = BatchLinear._programs.get(key)
        if progs is not None:
            BatchLinear._programs.move_to_end(key)
            return progs
        out_dim, in_dim = (W.shape[1], W.shape[0]) if transpose else W.shape
        progs = []
        for k0 in range(0, in_dim, 128):
            kw = min(128, in_dim - k0)
            bld = ProgramBuilder(kw, 0, kw)
            if in_dim > 128:
                bld.x_cols, bld.x_stride = np.arange(k0, k0 + kw), in_dim
            bld.add_single_linear(W, b if k0 == 0 else None, np.arange(out_dim), transpose=transpose, k0=k0, accumulate=k0 > 0)
            progs.append(bld.build(device))
        # (the pack jobs keep `W` / `b` alive, so their storage cannot be handed to another tensor while the entry exists)
        BatchLinear._programs[key] = progs
        while len(BatchLinear._programs) > BatchLinear._PROGRAMS_KEPT:
            BatchLinear._programs.popitem(last=False)
        return progs

    @staticmethod
    def _program_linear(x, W, b, transpose: bool):
        """x [N, K] @ (W^T | W) + b as fused-kernel launches (one per 128 input features), or None where the shape is not the
        programs'."""
        out_dim, in_dim = (W.shape[1], W.shape[0]) if transpose else W.shape
        if (x.shape[0] < BatchLinear.MIN_ROWS or in_dim > BatchLinear.PROGRAM_MAX_IN or out_dim > BatchLinear.PROGRAM_MAX_OUT
                or not x.is_contiguous() or not W.is_contiguous() or x.shape[1] != in_dim):
            return None
        try:
            progs = BatchLinear._linear_programs(W.detach(), None if b is None else b.detach(), transpose, x.device)
        except NotImplementedError:
            return None
        y = torch.empty(x.shape[0], out_dim, dtype=torch.float32, device=x.device)
        # (the exact-fp32 arithmetic: a single layer is HBM-bound either way -- 400 MB against 17 Gflop for [2^18, 128] x [128, 256]
        #  --, fp32 MFMA has no operand range and is torch's own arithmetic for this op)
        for prog in progs:
            prog.run(x, mlp_out=y, exact=True)
        return y

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        y = BatchLinear._program_linear(x, W, b, False)
        return y if y is not None else torch.nn.functional.linear(x, W, b)

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        gx = None
        if ctx.needs_input_grad[0]:
            gyc = gy if gy.is_contiguous() else gy.contiguous()
            gx = BatchLinear._program_linear(gyc, W, None, True)
            if gx is None:
                gx = gy @ W
        gW = gb = None
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if not (want_w or want_b):
            return gx, None, None
        lib = _hip.lib()
        gy2 = gy if gy.stride(1) == 1 else gy.contiguous()
        out_dim, in_dim = W.shape
        n = x.shape[0]
        if n >= BatchLinear.MIN_ROWS and in_dim <= 128 and out_dim <= BatchLinear.MAX_OUT:
            x2 = x if x.stride(1) == 1 else x.contiguous()
            gW = torch.zeros(out_dim, in_dim, dtype=torch.float32, device=x.device)
            gb = torch.zeros(out_dim, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            with _hip.device_of(x):
                sc = _hip.scratch(x.device, lib.sx_wgrad_scratch_floats(out_dim, in_dim, _hip.WGRAD_ROW_MAJOR))
                _hip.call('sx_wgrad', x, gy2.data_ptr(), gy2.stride(0), out_dim, x2.data_ptr(), x2.stride(0), in_dim, n,
                          _hip.WGRAD_ROW_MAJOR, gW.data_ptr(), in_dim, _hip.ptr(gb), None, None, sc.data_ptr())
            return gx, gW, gb
        if want_w:
            gW = gy.t() @ x
        if want_b:
            gb = torch.zeros(out_dim, dtype=torch.float32, device=x.device)
            with _hip.device_of(x):
                sc = _hip.scratch(x.device, 256 * out_dim)
                _hip.call('sx_colsum', x, gy2.data_ptr(), gy2.stride(0), n, out_dim, gb.data_ptr(), sc.data_ptr())
        return gx, gW, gb


class SelectRows(torch.autograd.Function):
    """p.index_select(0, rows) for DISTINCT rows, with a backward that writes (index_copy_) instead of accumulating
    (index_add_) into the zero-filled gradient: the same values, and -- unlike autograd's own index_select backward on
    this torch / ROCm build -- correct when the step is replayed from a HIP graph (from the second replay on the
    accumulating form returned stale sums for the bias; reproduced with plain torch ops)."""

    @staticmethod
    def forward(ctx, p, rows):
        ctx.save_for_backward(rows)
        ctx.shape = p.shape
        return p.index_select(0, rows)

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        out.index_copy_(0, rows, g)
        return out, None


def batch_linear(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.functional.linear with the batch-axis weight gradient on sx_wgrad where that applies."""
    if torch.is_grad_enabled() and (W.requires_grad or (b is not None and b.requires_grad)) and BatchLinear.eligible(x, W):
        return BatchLinear.apply(x, W, b)
    return torch.nn.functional.linear(x, W, b)


class _Linear(StructureTracked, nn.Linear):
    """nn.Linear whose re-assigned weight / bias invalidates the programs that packed the old tensors."""


class _Sequential(StructureTracked, nn.Sequential):
    """nn.Sequential whose replaced layers invalidate cached programs (same state_dict keys as the reference's)."""

    def __setitem__(self, idx, module):
        from ..fused import bump_structure_epoch
        bump_structure_epoch()
        super().__setitem__(idx, module)

    def __delitem__(self, idx):
        from ..fused import bump_structure_epoch
        bump_structure_epoch()
        super().__delitem__(idx)


class MLP(StructureTracked, nn.Module):
    def __init__(self, in_dim: int, hidden_dims: List[int], out_dim: int, activation: Union[str, Callable] = 'Tanh',
                 final_activation: Optional[str] = None, nn_linear_wrapper_func: Optional[Callable] = None, **kwargs):
        super().__init__()
        act_name = activation if isinstance(activation, str) else type(activation).__name__
        # activations / wrappers the fused kernel does not know (mlp.py:38-39 takes ANY torch.nn name, mlp.py:41-42 any
        # nn.Linear wrapper such as spectral norm): the network then runs through its torch layers (library GEMMs)
        self.activation_name = act_name
        self._wrapped = nn_linear_wrapper_func is not None
        wrap = nn_linear_wrapper_func or (lambda m: m)
        self.in_dim, self.out_dim = in_dim, out_dim
        widths = [in_dim] + list(hidden_dims) + [out_dim]
        act = getattr(nn, act_name)() if isinstance(activation, str) else activation
        layers: List[nn.Module] = []
        for i in range(len(widths) - 1):
            if i:
                layers.append(act)                        # keeps the reference's indices 0, 2, 4, ...
            lin = _Linear(widths[i], widths[i + 1])
            layers.append(wrap(lin) if i else lin)        # mlp.py:46,50: the first layer is not wrapped
        with torch.no_grad():
            layers[-1].bias.zero_()                       # mlp.py:53
        self.final_activation_name = final_activation
        if final_activation is not None:
            layers.append(getattr(nn, final_activation)())   # mlp.py:55-56
        self.net = _Sequential(*layers)
        self._programs = ProgramCache()

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

    # -- differentiable evaluation (layer-wise training path) -------------------------------------------
    def hidden_autograd(self, x2: torch.Tensor, col_mask: Optional[torch.Tensor] = None, pre_tanh: bool = False,
                        want_flag: bool = False):
        """Everything before the last Linear, with a graph -> (last hidden activation [N, H], the last Linear, the layers
        after it (a final activation or none)).  `col_mask` [in_dim] (0 / 1): the first Linear sees x2 * col_mask -- applied
        to its [H, in] weight instead of the [N, in] rows (x_j (w_ij m_j) = (x_j m_j) w_ij exactly for a 0 / 1 mask), which
        saves a pass over the batch in the forward and another in the backward.  `pre_tanh`: when the last hidden activation is
        Tanh, stop BEFORE it (-> the pre-activation; with want_flag the 4th result says whether that happened)."""
        layers = list(self.net)
        tail = []
        if self.final_activation_name is not None:                            # mlp.py:55-56
            layers, tail = layers[:-1], layers[-1:]
        body = layers[:-1]
        if pre_tanh and len(body) >= 2 and isinstance(body[-1], nn.Tanh):
            body = body[:-1]                      # the caller applies (and differentiates) the last tanh itself
        else:
            pre_tanh = False
        h = x2
        for i, layer in enumerate(body):
            if isinstance(layer, nn.Linear):
                W = layer.weight * col_mask if (i == 0 and col_mask is not None) else layer.weight
                h = batch_linear(h, W, layer.bias)
            else:
                if i == 0 and col_mask is not None:
                    h = h * col_mask
                h = layer(h)
        return (h, layers[-1], tail, pre_tanh) if want_flag else (h, layers[-1], tail)

    def forward_autograd(self, x2: torch.Tensor, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The network on [N, in] rows with a graph: torch activations, `batch_linear` layers.  `rows` selects the
        output rows of the last layer (a coupling only needs the parameters of its transformed columns)."""
        h, last, tail = self.hidden_autograd(x2)
        W, b = last.weight, last.bias
        if rows is not None:
            W, b = SelectRows.apply(W, rows), SelectRows.apply(b, rows)       # rows are distinct
        h = batch_linear(h, W, b)
        for layer in tail:
            h = layer(h)
        return h

    # -- standalone evaluation ---------------------------------------------------------------------------
    def _program(self, device):
        def build():
            H = self.hidden_width
            if H <= 128:
                b = ProgramBuilder(self.in_dim, 0, H)
                b.add_mlp(self.linears(), self.act_code, None, np.arange(self.out_dim))
                return _chunk_mlp_program(b, device)
            # one hidden layer wider than the kernel's four hidden tiles: one program per chunk of 128 hidden units, later chunks
            # accumulating into the output (W2 act(W1 x + b1) + b2 is a sum over hidden-unit chunks)
            progs = []
            for h0 in range(0, H, 128):
                hsel = np.arange(h0, min(h0 + 128, H))
                b = ProgramBuilder(self.in_dim, 0, len(hsel))
                b.add_mlp(self.linears(), self.act_code, None, np.arange(self.out_dim), hidden_rows=hsel, accumulate=h0 > 0)
                progs += _chunk_mlp_program(b, device)
            return progs
        return self._programs.get(str(device), build)

    def fusable(self) -> bool:
        """The planner can consume this network weight by weight: plain Linear layers, an activation the kernel knows,
        no final activation."""
        return (not self._wrapped and self.final_activation_name is None and self.activation_name in _hip.ACT_CODES
                and len(self.linears()) >= 2)

    def _fits_program(self) -> bool:
        """One launch of the fused MFMA kernel holds inputs and hidden layers of up to 128 columns; a single hidden layer of any
        width runs as one launch per 128 hidden units."""
        return self.fusable() and self.in_dim <= 128 and (self.hidden_width <= 128 or len(self.linears()) == 2)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        _hip.require_device(x, 'MLP input')
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).to(torch.float32).contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self.forward_autograd(x2).reshape(*lead, self.out_dim)      # differentiable like the reference's
        progs = None
        if self._fits_program():
            try:
                progs = self._program(x.device)
            except NotImplementedError:
                progs = None
        if progs is None:
            # wider than the fused kernel's tiles (or a final activation): the Linear layers are plain library GEMMs
            # (rocBLAS / hipBLASLt through torch), as the kernel playbook prescribes for plain GEMMs
            return self.net(x2).reshape(*lead, self.out_dim)
        out = torch.empty(x2.shape[0], self.out_dim, dtype=torch.float32, device=x.device)
        for prog in progs:
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
    total_out = builder.mlp_out_dim
    for i in range(0, max(len(outs), 1), room):
        chunk = outs[i:i + room]
        # each launch writes its own window of the output: tile indices restart at 0 (the device step keeps them in 8 bits --
        # a 348-tile output, e.g. 121 columns x 92 spline parameters, used to wrap at tile 256 and leave the columns beyond
        # 8192 unwritten: tools/fuzz_train.py --fat --wide)
        first = chunk[0]['t0'] if chunk else 0
        builder.steps = head + [dict(s, t0=s['t0'] - first) for s in chunk]
        builder.mlp_out_dim = min(32 * len(chunk), total_out - 32 * first) if chunk else total_out
        prog = builder.build(device)
        prog.mlp_col0 = 32 * first
        progs.append(prog)
    builder.steps = all_steps
    builder.mlp_out_dim = total_out
    # the chunks share one set of pack jobs / one blob buffer
    for p in progs[1:]:
        p.share_weights_of(progs[0])
    return progs
