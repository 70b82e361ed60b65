"""Dense invertible linear layers (reference: stribor/flows/affine.py:126-299).

``AffineLU(dim)``: ``y = x (L U) + b`` with ``L = tril(W,-1)+I``, ``U = triu(W,1)+diag(exp(log_diag))``
(affine.py:148-157); the reference inverts by two right-side triangular solves (:159-163).
``MatrixExponential(dim, bias, log_time)``: ``y = L U e^{diag t} U^-1 L^-1 x (+ b)`` with
``U = triu(W)+I`` (affine.py:222-270); the reference solves two batched [N, D, 1] triangular systems
(97 % of cfg-4 time).

Here every direction is one (or, for a per-row ``t``, two) dense D x D products on the fp32 matrix cores
inside the fused flow kernel.  The matrices — ``(LU)^T``, ``((LU)^-1)^T``, ``L U e^{diag t} (L U)^-1`` — are
derived from the parameters in fp64 on the device once per parameter version and rounded to fp32, which is
closer to the exact inverse than the reference's own fp32 substitution (fp32-vs-fp64 self error of the
reference on cfg 4: 1e-6..4e-6 rel).  ``log_det_jacobian`` is parameter-only.
"""
import math
from numbers import Number
from typing import Optional

import torch
import torch.nn as nn

from .. import _hip
from ..flow import Transform, flatten_rows, graph_rows, graph_wanted
from ..fused import ProgramBuilder, ProgramCache
from ..net.mlp import batch_linear

__all__ = ['AffineLU', 'MatrixExponential', 'DenseDeriver']


class _DenseLinear(Transform):
    """Shared launcher: single-transform fused programs keyed by (direction, ldj sign, t kind, device)."""

    def _program(self, key, build):
        cache = self.__dict__.get('_programs')
        if cache is None:
            cache = self._programs = ProgramCache()
        return cache.get(key, build)

    def _run(self, x, reverse, want_y, want_ldj, ldj_scale, t=None):
        _hip.require_device(x, 'x')
        x2, lead = flatten_rows(x)
        d = x2.shape[1]
        t_kind = 'tensor' if torch.is_tensor(t) else (None if t is None else float(t))
        key = (reverse, ldj_scale, t_kind, d, str(x.device))

        def build():
            b = ProgramBuilder(d, 0, 32)
            b.t = t_kind
            assert self._plan(b, reverse, ldj_scale)
            return b.build(x.device)

        prog = self._program(key, build)
        row_t = t.reshape(-1) if torch.is_tensor(t) else None
        y, ldj, _ = prog.run(x2, None, want_y, want_ldj, False, row_t=row_t)
        return (None if y is None else y.reshape(*lead, d)), (None if ldj is None else ldj.reshape(*lead, 1))


_TE_CACHE = {}


class TriInverse(torch.autograd.Function):
    """T^-1 for a batch [k, D, D] of fp64 triangular matrices on the device (sx_tri_inverse_f64: one workgroup per matrix
    instead of one 57 us library triangular solve against the identity per matrix and direction); backward is the
    inverse's own  dT = -X^T dX X^T  as two batched matmuls."""

    @staticmethod
    def forward(ctx, T, lower: bool, unit: bool):
        T = T.contiguous()
        X = torch.empty_like(T)
        k, D = T.shape[0], T.shape[-1]
        _hip.call('sx_tri_inverse_f64', T, T.data_ptr(), X.data_ptr(), k, D, int(lower), int(unit))
        ctx.save_for_backward(X)
        return X

    @staticmethod
    def backward(ctx, gX):
        (X,) = ctx.saved_tensors
        Xt = X.transpose(-1, -2)
        return -(Xt @ gX @ Xt), None, None


def _lu_inverse_batched(L, U):
    """(L U)^-1 = U^-1 L^-1 for batches of unit-lower L and upper U (fp64, on the device)."""
    return _lu_inverse_multi([(L, U)])[0]


def _lu_inverse_multi(pairs):
    """[(L U)^-1 for (L, U) in pairs], each a batch [k_i, D, D] of unit-lower L (ones on the diagonal, explicit) and upper U.
    The triangular inverses of ALL pairs of one width go through ONE sx_tri_inverse_f64 launch (one workgroup per matrix, the
    launch is latency-bound: 0.3 ms whether it inverts 4 or 16 matrices): U^-1 = ((U^T)^-1)^T makes every operand lower
    triangular with an explicit diagonal."""
    out = [None] * len(pairs)
    by_dim = {}
    for i, (L, U) in enumerate(pairs):
        if L.is_cuda and L.dtype == torch.float64 and L.shape[-1] <= 128:
            by_dim.setdefault(L.shape[-1], []).append(i)
        else:
            eye = torch.eye(L.shape[-1], dtype=L.dtype, device=L.device).expand_as(L)
            out[i] = torch.linalg.solve_triangular(U, torch.linalg.solve_triangular(L, eye, upper=False), upper=True)
    for idx in by_dim.values():
        stack = torch.cat([t for i in idx for t in (pairs[i][1].transpose(-1, -2), pairs[i][0])])
        X = TriInverse.apply(stack, True, False)
        parts = X.split([pairs[i][0].shape[0] for i in idx for _ in (0, 1)])        # (one cat in the backward, not a zero-fill per slice)
        for n, i in enumerate(idx):
            out[i] = parts[2 * n].transpose(-1, -2) @ parts[2 * n + 1]
    return out


def derive_dense_batched(layers, dev, reverse: bool = True):
    """The matrices of all AffineLU / MatrixExponential layers of a flow in a few BATCHED fp64 torch ops (one launch per op
    for every group of same-shaped layers instead of one per layer: a training step of a cfg-4-like flow is dominated by
    these tiny launches at small batch sizes), for the inverse direction (what log_prob evaluates) or the forward one.
    -> {id(layer): (W [out, in] fp32, b fp32 | None, log-det scalar tensor)} for the layers' _autograd_inverse /
    _autograd_forward; differentiable."""
    out, groups = {}, {}
    for f in layers:
        if isinstance(f, AffineLU):
            groups.setdefault(('lu', f.dim), []).append(f)
        elif isinstance(f, MatrixExponential):
            groups.setdefault(('mx', f.dim, f.bias is not None), []).append(f)
    # pass 1: the triangular factors of every group; pass 2 (after ONE batched inverse of all of them): the matrices
    prep = {}
    for key, fs in groups.items():
        D = key[1]
        eye = torch.eye(D, dtype=torch.float64, device=dev)
        if key[0] == 'lu':
            W = torch.stack([f.weight for f in fs]).to(dev, torch.float64)
            ld = torch.stack([f.log_diag.reshape(-1) for f in fs]).to(dev, torch.float64)
            b = torch.stack([f.bias.reshape(-1) for f in fs]).to(dev, torch.float64)
            L, U = torch.tril(W, -1) + eye, torch.triu(W, 1) + torch.diag_embed(ld.exp())      # affine.py:148-154
            prep[key] = (L, U, ld, b)
        else:
            W = torch.stack([f._weight for f in fs]).to(dev, torch.float64)
            dg = torch.stack([f.diag for f in fs]).to(dev, torch.float64)
            L, U = torch.tril(W, diagonal=-1) + eye, torch.triu(W) + eye                       # affine.py:222-226
            prep[key] = (L, U, dg, None)
    need = [key for key in groups if key[0] == 'mx' or reverse]
    inv = dict(zip(need, _lu_inverse_multi([(prep[k][0], prep[k][1]) for k in need])))
    for key, fs in groups.items():
        L, U, third, b = prep[key]
        if key[0] == 'lu':
            ld = third
            if reverse:
                Ainv = inv[key]
                Wm = Ainv.transpose(-1, -2).to(torch.float32).contiguous()                     # x = (y - b) A^-1 (:159-163)
                bm = (-(b.unsqueeze(1) @ Ainv).squeeze(1)).to(torch.float32)
                ldj = (-ld.sum(-1)).to(torch.float32)                                          # :171, negated
            else:
                Wm = (L @ U).transpose(-1, -2).to(torch.float32).contiguous()                  # y = x (L U) + b (:157)
                bm = b.to(torch.float32)
                ldj = ld.sum(-1).to(torch.float32)
            # (unbind, not [i]: the backward of k selects is k zero-filled [k, D, D] tensors and k copies -- 40 launches per cfg-4
            #  step; UnbindBackward stacks the pieces once)
            for f, Wi, bi, li in zip(fs, Wm.unbind(0), bm.unbind(0), ldj.unbind(0)):
                out[id(f)] = (Wi, bi, li)
        else:
            dg = third
            tkey = (tuple(f._t_eff(1.0) for f in fs), str(dev))
            te = _TE_CACHE.get(tkey)                      # constants: uploaded once (a host copy would break graph capture)
            if te is None:
                te = _TE_CACHE[tkey] = torch.tensor(tkey[0], dtype=torch.float64, device=dev).unsqueeze(-1)
            Ainv = inv[key]
            sg = -te if reverse else te
            M = ((L @ U) * (dg * sg).exp().unsqueeze(-2)) @ Ainv                               # :254-266 (t -> -t inverse)
            Wm = M.to(torch.float32).contiguous()
            ldj = ((dg.sum(-1) * sg.squeeze(-1))).to(torch.float32)                            # :287-288 (negated inverse)
            if key[2]:
                bias = torch.stack([f.bias for f in fs]).to(dev, torch.float64)
                bm = (-(M @ bias.unsqueeze(-1)).squeeze(-1)).to(torch.float32) if reverse else bias.to(torch.float32)
            bms = bm.unbind(0) if key[2] else [None] * len(fs)
            for f, Wi, bi, li in zip(fs, Wm.unbind(0), bms, ldj.unbind(0)):
                out[id(f)] = (Wi, bi, li)
    return out


class DenseDeriver:
    """The matrices of ALL AffineLU / MatrixExponential (default t) layers of one flow from a few batched fp64 ops
    (`derive_dense_batched`), cached per parameter version and direction.  Fused programs built inside a flow take their packed
    matrices from here (one batched derivation per parameter update instead of a chain of tiny fp64 launches -- and an LU
    factorisation -- per layer), and the fused training backward asks for the same tensors WITH a graph so that dL/d(matrix)
    reaches the parameters through autograd of the D x D algebra."""

    def __init__(self, layers):
        self.layers = [f for f in layers if isinstance(f, (AffineLU, MatrixExponential))]
        self._cache = {}

    def _versions(self):
        return tuple((p.data_ptr(), p._version) for f in self.layers for p in f.parameters())

    def get(self, dev, reverse: bool, graph: bool = False):
        """-> {id(layer): (W [out, in] fp32, b fp32 | None, log-det 0-dim)} for the direction (reverse = what log_prob applies)."""
        key = (str(dev), bool(reverse))
        v = self._versions()
        ent = self._cache.get(key)
        if ent is not None and ent[0] == v and not graph:
            return ent[1]
        with (torch.enable_grad() if graph else torch.no_grad()):
            out = derive_dense_batched(self.layers, dev, reverse)
        # the cache keeps detached tensors (a cached graph would outlive the step that built it)
        self._cache[key] = (v, {k: tuple(None if t is None else t.detach() for t in ts) for k, ts in out.items()} if graph else out)
        return out


class AffineLU(_DenseLinear):
    def __init__(self, dim: int, **kwargs):
        super().__init__()
        self.dim = dim
        self.weight = nn.Parameter(torch.empty(dim, dim))
        self.log_diag = nn.Parameter(torch.empty(1, dim))
        self.bias = nn.Parameter(torch.empty(1, dim))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)                    # affine.py:143-146
        nn.init.xavier_uniform_(self.log_diag)
        nn.init.xavier_uniform_(self.bias)

    # parameter-level accessors of the reference (D x D tensor ops on the parameters' device, differentiable)
    @property
    def L(self):
        return torch.tril(self.weight, -1) + torch.eye(self.dim).to(self.weight)                     # affine.py:148-150

    @property
    def U(self):
        return torch.triu(self.weight, 1) + torch.eye(self.dim).to(self.weight) * self.log_diag.exp()   # affine.py:152-154

    def _lu64(self, dev):
        W = self.weight.detach().to(dev, torch.float64)
        eye = torch.eye(self.dim, dtype=torch.float64, device=dev)
        L = torch.tril(W, -1) + eye                             # affine.py:148-150
        U = torch.triu(W, 1) + eye * self.log_diag.detach().to(dev, torch.float64).exp()    # :152-154
        return L @ U

    def _matrices(self, reverse):
        def fn(dev):
            A = self._lu64(dev)
            b = self.bias.detach().to(dev, torch.float64).reshape(-1)
            if not reverse:
                return A.T, b                                   # y = x A + b            (:157)
            Ainv = torch.linalg.inv(A)
            return Ainv.T, -(b @ Ainv)                          # x = (y - b) A^-1       (:159-163)
        return fn

    # ---- training (layer-wise autograd path): the D x D matrix is derived with differentiable torch ops in fp64, the
    #      [N, D] x [D, D] product is a plain library GEMM --------------------------------------------------------------
    def _autograd_supported(self) -> bool:
        return True

    def _autograd_inverse(self, x2: torch.Tensor, lat2=None, derived=None):
        dev = x2.device
        if derived is None:
            derived = derive_dense_batched([self], dev)[id(self)]
        Wm, bm, ldj = derived
        return batch_linear(x2, Wm, bm), ldj.expand(x2.shape[0])                                           # :159-163, :171

    def _autograd_forward(self, x2: torch.Tensor, lat2=None, derived=None):
        if derived is None:
            derived = derive_dense_batched([self], x2.device, reverse=False)[id(self)]
        Wm, bm, ldj = derived
        return batch_linear(x2, Wm, bm), ldj.expand(x2.shape[0])                                           # :157, :171

    def _plan(self, builder, reverse, ldj_scale):
        ldj = lambda dev: ldj_scale * self.log_diag.detach().to(dev, torch.float64).sum()       # affine.py:171
        fn = self._matrices(reverse)
        deriver = getattr(builder, 'dense_deriver', None)
        if deriver is not None and any(f is self for f in deriver.layers):
            fn = lambda dev: tuple(t.detach() if t is not None else None for t in deriver.get(dev, reverse)[id(self)][:2])
        builder.add_linear([self.weight, self.log_diag, self.bias], fn, ldj)
        return True

    def _bwd_matrices(self, deriver):
        """(fn_fwd, fn_adj) of the fused backward program's two SX_STEP_LINEAR_BWD steps: v = A^T-form u + b (affine.py:157) on the
        x tiles, dL/dv = W^T dL/du on the adjoint tiles, W = the matrix log_prob applied ((LU)^-1 in Linear layout)."""
        fwd = lambda dev: tuple(t.detach() if t is not None else None for t in deriver.get(dev, False)[id(self)][:2])
        adj = lambda dev: (deriver.get(dev, True)[id(self)][0].detach().t().contiguous(), None)
        return fwd, adj

    # ---- reference method set (affine.py:156-179): differentiable like the reference's (see flow.graph_wanted) -------
    def _graph(self, x, reverse: bool):
        _hip.require_device(x, 'x')
        x2, _, lead = graph_rows(x)
        y, ldj = (self._autograd_inverse if reverse else self._autograd_forward)(x2)
        return y.reshape(*lead, x2.shape[1]), ldj.reshape(*lead, 1)

    def forward(self, x, **kwargs):
        if graph_wanted(self, x):
            return self._graph(x, False)[0]
        return self._run(x, False, True, False, 1.0)[0]

    def inverse(self, y, **kwargs):
        if graph_wanted(self, y):
            return self._graph(y, True)[0]
        return self._run(y, True, True, False, 1.0)[0]

    def log_det_jacobian(self, x, y=None, **kwargs):
        if graph_wanted(self, x):
            return self.log_diag.to(x.device, torch.float32).sum().expand(*x.shape[:-1], 1)     # affine.py:171, with its graph
        ld = self.log_diag.detach().to(x.device, torch.float32).sum()
        return ld.expand(*x.shape[:-1], 1).clone()                          # affine.py:171

    def forward_and_log_det_jacobian(self, x, **kwargs):
        if graph_wanted(self, x):
            return self._graph(x, False)
        return self._run(x, False, True, True, 1.0)

    def inverse_and_log_det_jacobian(self, y, **kwargs):
        if graph_wanted(self, y):
            return self._graph(y, True)
        return self._run(y, True, True, True, -1.0)

    def jacobian(self, x, y=None, **kwargs):
        return ((self.L @ self.U).T).to(x.device).expand(*x.shape[:-1], -1, -1)   # affine.py:173-179


class MatrixExponential(_DenseLinear):
    def __init__(self, dim: int, bias: bool = False, log_time: bool = False, **kwargs):
        super().__init__()
        self.dim, self.log_time = dim, log_time
        self._weight = nn.Parameter(torch.empty(dim, dim))
        self.diag = nn.Parameter(torch.empty(dim))
        if bias:
            self.bias = nn.Parameter(torch.empty(dim))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self._weight, a=math.sqrt(5))               # affine.py:213-220
        fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self._weight)
        bound = 1 / math.sqrt(fan_in)
        nn.init.uniform_(self.diag, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def _sources(self):
        return [p for p in (self._weight, self.diag, self.bias) if p is not None]

    # parameter-level accessors of the reference (D x D tensor ops on the parameters' device, differentiable)
    def lu(self):
        eye = torch.eye(self.dim).to(self._weight)
        return torch.tril(self._weight, diagonal=-1) + eye, torch.triu(self._weight) + eye      # affine.py:222-226

    @property
    def weight(self):
        L, U = self.lu()
        W = L @ U
        return (W * self.diag) @ torch.linalg.inv(W)                                          # affine.py:228-234

    def get_time(self, t, shape):
        if isinstance(t, Number):
            t = torch.ones(*shape[:-1], 1, device=self._weight.device) * t                     # affine.py:236-241
        if self.log_time:
            t = torch.log1p(t.abs())
        return t

    def jacobian(self, x, y=None, t=1.0, **kwargs):
        t = self.get_time(t, x.shape).to(x.device)
        W = torch.matrix_exp(self.weight.to(x.device) * t.unsqueeze(-1))                      # affine.py:290-299
        return W.expand(*x.shape[:-1], -1, -1)

    def _lu64(self, dev):
        W = self._weight.detach().to(dev, torch.float64)
        eye = torch.eye(self.dim, dtype=torch.float64, device=dev)
        return (torch.tril(W, diagonal=-1) + eye) @ (torch.triu(W) + eye)    # affine.py:222-226

    def _t_eff(self, t: float) -> float:
        return math.log1p(abs(t)) if self.log_time else t                    # affine.py:239-240

    # ---- training (layer-wise autograd path, default t = 1): as AffineLU ----------------------------------------------
    def _autograd_supported(self) -> bool:
        return True

    def _autograd_inverse(self, x2: torch.Tensor, lat2=None, derived=None):
        if derived is None:
            derived = derive_dense_batched([self], x2.device)[id(self)]
        Wm, bm, ldj = derived
        return batch_linear(x2, Wm, bm), ldj.expand(x2.shape[0])                                           # :254-266, :287-288

    def _autograd_forward(self, x2: torch.Tensor, lat2=None, derived=None):
        if derived is None:
            derived = derive_dense_batched([self], x2.device, reverse=False)[id(self)]
        Wm, bm, ldj = derived
        return batch_linear(x2, Wm, bm), ldj.expand(x2.shape[0])                                           # :243-270, :287-288

    def _autograd_time(self, x2: torch.Tensor, t, reverse: bool):
        """forward / inverse with an explicit time (a number or a per-row tensor [N, 1]) and a graph: the three stages of
        affine.py:254-269 -- v = (LU)^-1 (x [- b]); v *= exp(+-diag t_n); y = (LU) v [+ b] -- with the matrices derived
        in fp64 by differentiable ops and the two [N, D] x [D, D] products through BatchLinear."""
        dev = x2.device
        W = self._weight.to(dev, torch.float64)
        eye = torch.eye(self.dim, dtype=torch.float64, device=dev)
        L, U = torch.tril(W, diagonal=-1) + eye, torch.triu(W) + eye                          # affine.py:222-226
        A = L @ U
        Ainv = _lu_inverse_batched(L.unsqueeze(0), U.unsqueeze(0))[0]
        tt = t.reshape(-1, 1).to(dev, torch.float32) if torch.is_tensor(t) else torch.full((x2.shape[0], 1), float(t), device=dev)
        if self.log_time:
            tt = torch.log1p(tt.abs())                                                         # affine.py:239-240
        sg = -tt if reverse else tt                                                            # :254
        xin = x2 - self.bias.to(dev, torch.float32) if (reverse and self.bias is not None) else x2          # :255-256
        v = batch_linear(xin, Ainv.to(torch.float32), None)                                    # :260-261
        v = v * torch.exp(self.diag.to(dev, torch.float32) * sg)                               # :263
        y = batch_linear(v, A.to(torch.float32), None)                                         # :265-266
        if not reverse and self.bias is not None:
            y = y + self.bias.to(dev, torch.float32)                                           # :268-269
        return y, (self.diag.to(dev, torch.float32).sum() * sg).reshape(-1)                    # :287-288 (negated: flow.py:47)

    def _plan(self, builder, reverse, ldj_scale):
        t = getattr(builder, 't', None)
        t = 1.0 if t is None else t                                          # default t = 1.0 (affine.py:246)
        has_bias = self.bias is not None
        if t == 'tensor':
            # v = (LU)^-1 (x [- b]);  v *= exp(+-diag t_n);  y = (LU) v [+ b]       (affine.py:254-269)
            def solve(dev):
                Ainv = torch.linalg.inv(self._lu64(dev))
                b = -(Ainv @ self.bias.detach().to(dev, torch.float64)) if (reverse and has_bias) else None
                return Ainv, b

            def mult(dev):
                b = self.bias.detach().to(dev, torch.float64) if (not reverse and has_bias) else None
                return self._lu64(dev), b
            builder.add_linear(self._sources(), solve)
            builder.add_row_scale_exp(self.diag, reverse, ldj_scale, self.log_time, 0.0)
            builder.add_linear(self._sources(), mult)
            return True
        te = self._t_eff(float(t))
        sg = -te if reverse else te

        def collapsed(dev):
            A = self._lu64(dev)
            M = (A * (self.diag.detach().to(dev, torch.float64) * sg).exp()) @ torch.linalg.inv(A)
            if not has_bias:
                return M, None
            b = self.bias.detach().to(dev, torch.float64)
            return (M, b) if not reverse else (M, -(M @ b))
        ldj = lambda dev: (ldj_scale * te) * self.diag.detach().to(dev, torch.float64).sum()    # affine.py:287-288
        deriver = getattr(builder, 'dense_deriver', None)
        if deriver is not None and float(t) == 1.0 and any(f is self for f in deriver.layers):
            collapsed = lambda dev: tuple(x.detach() if x is not None else None for x in deriver.get(dev, reverse)[id(self)][:2])
        builder.add_linear(self._sources(), collapsed, ldj)
        return True

    def _bwd_matrices(self, deriver):
        """As AffineLU._bwd_matrices, for the collapsed matrix at the default t = 1 (affine.py:243-270)."""
        fwd = lambda dev: tuple(t.detach() if t is not None else None for t in deriver.get(dev, False)[id(self)][:2])
        adj = lambda dev: (deriver.get(dev, True)[id(self)][0].detach().t().contiguous(), None)
        return fwd, adj

    # ---- reference method set (affine.py:243-288): differentiable like the reference's (see flow.graph_wanted) -------
    def _graph(self, x, t, reverse: bool):
        _hip.require_device(x, 'x')
        x2, _, lead = graph_rows(x)
        if torch.is_tensor(t) or float(t) != 1.0:
            y, ldj = self._autograd_time(x2, t, reverse)
        else:
            y, ldj = (self._autograd_inverse if reverse else self._autograd_forward)(x2)
        return y.reshape(*lead, x2.shape[1]), ldj.reshape(*lead, 1)

    def forward(self, x, t=1.0, *, reverse: bool = False, **kwargs):
        if graph_wanted(self, x, t):
            return self._graph(x, t, reverse)[0]
        return self._run(x, reverse, True, False, 1.0, t)[0]

    def inverse(self, y, t=1.0, **kwargs):
        if graph_wanted(self, y, t):
            return self._graph(y, t, True)[0]
        return self._run(y, True, True, False, 1.0, t)[0]                    # affine.py:272-278

    def log_det_jacobian(self, x, y=None, t=1.0, **kwargs):
        graph = graph_wanted(self, x, t)
        s = (self.diag if graph else self.diag.detach()).to(x.device, torch.float32).sum()
        if torch.is_tensor(t):
            tt = t.to(x.device, torch.float32)
            tt = torch.log1p(tt.abs()) if self.log_time else tt
            return s * tt                                                    # [..., 1]
        out = (s * self._t_eff(float(t))).expand(*x.shape[:-1], 1)
        return out if graph else out.clone()

    def forward_and_log_det_jacobian(self, x, t=1.0, **kwargs):
        if graph_wanted(self, x, t):
            return self._graph(x, t, False)
        return self._run(x, False, True, True, 1.0, t)

    def inverse_and_log_det_jacobian(self, y, t=1.0, **kwargs):
        if graph_wanted(self, y, t):
            return self._graph(y, t, True)
        return self._run(y, True, True, True, -1.0, t)
