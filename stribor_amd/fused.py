"""Host-side planner for the fused flow kernel (``sx_flow_run``).

Turns a sequence of transforms (in execution order, each with a direction) into

  * an ``sx_program`` (step table: which 32-column tiles condition, which are transformed, where each
    step's weight blob lives),
  * one device buffer of weight blobs in MFMA fragment order (written by ``sx_pack_linear`` on the
    device; re-packed only when a parameter's version changes),
  * the slot <-> column maps used to load / store the state.

The flow state is kept in *slots* (tile t = slots 32t..32t+31).  Permute / Flip layers cost nothing:
they only relabel which logical column a slot holds (stribor/flows/permute.py:71,75), and the
relabelling is folded into the row / column index lists of the neighbouring layers' weights.
Masks (stribor/util/mask.py) become tile sets: when a coupling's conditioning columns fill exactly
the low (or high) half of the tiles, the dead half of both GEMMs is pruned (SURVEY finding 3: exact,
zeros contribute exactly 0); otherwise the layer runs dense over all tiles with zeroed weights.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from abc import ABCMeta

from . import _hip

ALLOWED_TILES = (1, 2, 4)


# ---- validity of cached programs -------------------------------------------------------------------------------
# A built program bakes plan-time facts into host tables: which transforms a flow holds and in which order, every
# Permute's index vector (slot relabelling), masks, which Parameter objects feed which pack job.  Parameter VALUES are
# tracked per job ((data_ptr, _version) -> re-pack); everything else is covered by
#   * a process-wide structure epoch, bumped whenever a module / parameter / buffer of one of this package's modules is
#     (re)assigned or deleted, or an attribute the module's constructor DECLARED is re-assigned (`StructureTracked.__setattr__`:
#     mask names, bin counts, bounds, flags -- whatever __init__ set is what a planner may have read), and
#   * guard tensors (buffers read at plan time, e.g. Permute.permutation): (data_ptr, _version) snapshots, so an
#     in-place `load_state_dict` or a `.to()` re-plans, and
#   * an owner-supplied fingerprint (NormalizingFlow: the ids of its transforms, which ModuleList can change without
#     passing through any __setattr__ of ours).
# Structure events are rare (model surgery), so one process-wide counter keeps the per-call validity check at one integer
# compare; what is NOT an event: attributes a user hangs on a module after construction (`flow.step = i` in a training loop
# used to re-plan every flow of the process on every call) and tensors a forward pre-hook recomputes.
_STRUCT_EPOCH = [0]


def bump_structure_epoch() -> None:
    _STRUCT_EPOCH[0] += 1


class _TrackedMeta(ABCMeta):
    """Records, once __init__ has returned, which plain attributes the constructor declared (`_sx_declared`)."""

    def __call__(cls, *args, **kwargs):
        obj = super().__call__(*args, **kwargs)
        object.__setattr__(obj, '_sx_declared', frozenset(k for k in obj.__dict__ if not k.startswith('_')))
        return obj


class StructureTracked(metaclass=_TrackedMeta):
    """Mixin for nn.Module subclasses whose attributes are read at plan time."""

    _UNTRACKED = ('training',)

    def __setattr__(self, name, value):
        if not name.startswith('_') and name not in StructureTracked._UNTRACKED:
            d = self.__dict__
            registered = name in d.get('_parameters', ()) or name in d.get('_buffers', ()) or name in d.get('_modules', ())
            if torch.is_tensor(value) or isinstance(value, torch.nn.Module):
                # a plain tensor (not a Parameter) assigned to a name that is neither a registered parameter nor a buffer is a value
                # a forward pre-hook recomputes on every call (torch's spectral_norm: setattr(module, 'weight', w / sigma)); such
                # wrapped layers never join fused programs
                hook_value = torch.is_tensor(value) and not isinstance(value, torch.nn.Parameter) and not registered
                if not hook_value:
                    _STRUCT_EPOCH[0] += 1
            elif registered or name in d.get('_sx_declared', ()) or '_sx_declared' not in d and name in d:
                # a registered slot set to None / replaced by a non-tensor, or a declared attribute re-assigned
                # (during __init__ itself nothing can have been planned from this object yet, but a re-assignment inside a
                #  constructor of an attribute the BASE class set is kept an event: cheap, and obviously safe)
                _STRUCT_EPOCH[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        d = self.__dict__
        if (name in d.get('_parameters', ()) or name in d.get('_buffers', ()) or name in d.get('_modules', ())
                or name in d.get('_sx_declared', ()) or '_sx_declared' not in d):
            _STRUCT_EPOCH[0] += 1
        super().__delattr__(name)


def _snap(guards):
    return [(g.data_ptr(), g._version) for g in guards]


class ProgramCache:
    """key -> built value, rebuilt when the structure epoch, the owner's fingerprint or a guard tensor changed.  Bounded: keys
    may carry call-dependent sizes (e.g. the row count of a set_data batch), so the least recently BUILT entries beyond
    `MAX_ENTRIES` are dropped (dict order = insertion order; a hit does not reorder -- the bound only has to stop growth)."""

    MAX_ENTRIES = 256

    def __init__(self):
        self._d = {}

    def get(self, key, build, guards=(), fingerprint=None):
        ent = self._d.get(key)
        if ent is not None:
            epoch, fp, gs, snap, value = ent
            if epoch == _STRUCT_EPOCH[0] and fp == fingerprint and (not gs or _snap(gs) == snap):
                return value
        gs = list(guards() if callable(guards) else guards)
        value = build()
        self._d.pop(key, None)
        self._d[key] = (_STRUCT_EPOCH[0], fingerprint, gs, _snap(gs), value)
        while len(self._d) > ProgramCache.MAX_ENTRIES:
            self._d.pop(next(iter(self._d)))
        return value

    def clear(self):
        self._d.clear()


def _ceil_div(a: int, b: int) -> int:
    return -(-a // b)


def _round_tiles(n: int, what: str) -> int:
    for t in ALLOWED_TILES:
        if n <= t:
            return t
    raise NotImplementedError(f'stribor_amd fused kernel: {what} needs {n} tiles of 32 (max 4, i.e. 128 columns)')


def _kmap(r: int, h: int) -> int:
    return (r & 3) + 8 * (r >> 2) + 4 * h


class _PackJob:
    """One sx_pack_linear call: (W, b) -> blob[dst_off : dst_off + n]."""

    def __init__(self, W, b, row_idx: np.ndarray, col_idx: np.ndarray, m_tiles: int, k_tiles: int, dst_off: int,
                 row_scale: Optional[np.ndarray] = None, bias_scale: Optional[np.ndarray] = None,
                 fold_ones: float = 0.0, transpose: bool = False, bound_off: Optional[int] = None):
        self.transpose = int(transpose)
        # blob slot that keeps the running bound on |output| of the packed rows (sx_pack_linear_bound), or None
        self.bound_off = bound_off
        self.W, self.b = W, b
        self.row_idx_host, self.col_idx_host = row_idx.astype(np.int32), col_idx.astype(np.int32)
        self.m_tiles, self.k_tiles, self.dst_off = m_tiles, k_tiles, dst_off
        self.row_idx = self.col_idx = None
        self.row_scale_host = None if row_scale is None else row_scale.astype(np.float32)
        self.bias_scale_host = None if bias_scale is None else bias_scale.astype(np.float32)
        self.row_scale = self.bias_scale = None
        self.fold_ones = float(fold_ones)

    def _operands(self, dev):
        if self.row_idx is None:
            self.row_idx = torch.from_numpy(self.row_idx_host).to(dev)
            self.col_idx = torch.from_numpy(self.col_idx_host).to(dev)
            if self.row_scale_host is not None:
                self.row_scale = torch.from_numpy(self.row_scale_host).to(dev)
            if self.bias_scale_host is not None:
                self.bias_scale = torch.from_numpy(self.bias_scale_host).to(dev)
        W = self.W.detach()
        b = None if self.b is None else self.b.detach()
        if W.dtype != torch.float32 or not W.is_contiguous():
            raise TypeError('stribor_amd: conditioner weights must be contiguous float32')
        return W, b

    def record(self, blobs: torch.Tensor):
        """This pack as one record of a _hip.PackTable (sx_pack_linear_batch)."""
        W, b = self._operands(blobs.device)
        out_dim, in_dim = W.shape
        return (W.data_ptr(), _hip.ptr(b) or 0, out_dim, in_dim, self.row_idx.data_ptr(), self.col_idx.data_ptr(), self.m_tiles,
                self.k_tiles, _hip.ptr(self.row_scale) or 0, _hip.ptr(self.bias_scale) or 0, self.fold_ones, self.transpose,
                blobs.data_ptr() + 4 * self.dst_off, 0 if self.bound_off is None else blobs.data_ptr() + 4 * self.bound_off)

    def run(self, blobs: torch.Tensor, prec: int = _hip.GEMM_F16X3) -> None:
        W, b = self._operands(blobs.device)
        out_dim, in_dim = W.shape
        if self.bound_off is not None:
            _hip.call('sx_pack_linear_bound', blobs, W.data_ptr(), _hip.ptr(b), out_dim, in_dim, self.row_idx.data_ptr(),
                      self.col_idx.data_ptr(), self.m_tiles, self.k_tiles, _hip.ptr(self.row_scale), _hip.ptr(self.bias_scale),
                      self.fold_ones, self.transpose, prec, blobs.data_ptr(), blobs.data_ptr() + 4 * self.dst_off,
                      blobs.data_ptr() + 4 * self.bound_off)
            return
        _hip.call('sx_pack_linear', blobs, W.data_ptr(), _hip.ptr(b), out_dim, in_dim, self.row_idx.data_ptr(),
                  self.col_idx.data_ptr(), self.m_tiles, self.k_tiles, _hip.ptr(self.row_scale), _hip.ptr(self.bias_scale),
                  self.fold_ones, self.transpose, prec, blobs.data_ptr(), blobs.data_ptr() + 4 * self.dst_off)

    def params(self):
        return [p for p in (self.W, self.b) if p is not None]


class _DerivedLinearJob:
    """A dense layer whose matrix is DERIVED from parameters (AffineLU: (L U)^T or its inverse;
    MatrixExponential: L U e^{diag t} U^-1 L^-1): `fn()` -> (W [out, in] fp32, b [out] fp32 | None) on the
    device, recomputed (in fp64 inside fn) only when a source parameter changes, then packed slab by slab.
    `ldj_fn(device)` -> the layer's (signed, scaled) log-det as a 0-dim device tensor, written behind the bias of the
    blob by a device-side copy (no host read-back), so it can never go stale against the matrix."""

    def __init__(self, sources, fn, targets, ldj_fn=None, ldj_off: int = 0):
        self.sources, self.fn, self.targets = list(sources), fn, targets   # targets: [(row_idx, col_idx, k_tiles, off, m_tiles)]
        self.ldj_fn, self.ldj_off = ldj_fn, ldj_off
        self._dev_idx = None
        self._keep = None

    def _derive(self, dev):
        """fn() into this job's own fp32 operands (the same storage at every re-pack: the records of a PackTable stay valid)."""
        W, b = self.fn(dev)
        if self._keep is None or self._keep[0].shape != W.shape or self._keep[0].device != W.device or (self._keep[1] is None) != (b is None):
            self._keep = (torch.empty(W.shape, dtype=torch.float32, device=W.device),
                          None if b is None else torch.empty(b.shape, dtype=torch.float32, device=W.device))
        self._keep[0].copy_(W)
        if b is not None:
            self._keep[1].copy_(b)
        if self._dev_idx is None:
            self._dev_idx = [(torch.from_numpy(r.astype(np.int32)).to(dev), torch.from_numpy(c.astype(np.int32)).to(dev))
                             for (r, c, _, _, _) in self.targets]
        return self._keep

    def records(self, blobs: torch.Tensor):
        """Derive the matrix, then this layer's packs as records of a _hip.PackTable (sx_pack_linear_batch)."""
        W, b = self._derive(blobs.device)
        return [(W.data_ptr(), _hip.ptr(b) or 0, W.shape[0], W.shape[1], ri.data_ptr(), ci.data_ptr(), m_tiles, k_tiles, 0, 0, 0.0, 0,
                 blobs.data_ptr() + 4 * off, 0) for (ri, ci), (_, _, k_tiles, off, m_tiles) in zip(self._dev_idx, self.targets)]

    def write_ldj(self, blobs: torch.Tensor) -> None:
        if self.ldj_fn is not None:
            blobs[self.ldj_off:self.ldj_off + 1] = self.ldj_fn(blobs.device).reshape(1).to(torch.float32)

    def run(self, blobs: torch.Tensor, prec: int = _hip.GEMM_F16X3) -> None:
        W, b = self._derive(blobs.device)
        for (ri, ci), (_, _, k_tiles, off, m_tiles) in zip(self._dev_idx, self.targets):
            _hip.call('sx_pack_linear', blobs, W.data_ptr(), _hip.ptr(b), W.shape[0], W.shape[1], ri.data_ptr(),
                      ci.data_ptr(), m_tiles, k_tiles, None, None, 0.0, 0, prec, blobs.data_ptr(),
                      blobs.data_ptr() + 4 * off)
        self.write_ldj(blobs)

    def params(self):
        return self.sources


class _ConstJob:
    """Per-slot constants (st.Affine without latent_net): blob[dst_off:] = cat(ls, sh)[gather]."""

    def __init__(self, log_scale, shift, dim: int, gather: np.ndarray, dst_off: int):
        self.log_scale, self.shift, self.dim = log_scale, shift, dim
        self.gather_host, self.dst_off = gather.astype(np.int64), dst_off
        self.gather = None

    def run(self, blobs: torch.Tensor, prec: int = 0) -> None:
        dev = blobs.device
        if self.gather is None:
            self.gather = torch.from_numpy(self.gather_host).to(dev)
        ls = self.log_scale.detach().to(dev, torch.float32).reshape(-1).expand(self.dim) \
            if self.log_scale.numel() == 1 else self.log_scale.detach().to(dev, torch.float32).reshape(-1)
        sh = self.shift.detach().to(dev, torch.float32).reshape(-1).expand(self.dim) \
            if self.shift.numel() == 1 else self.shift.detach().to(dev, torch.float32).reshape(-1)
        src = torch.cat([ls, sh, torch.zeros(1, device=dev)])
        n = self.gather.numel()
        blobs[self.dst_off:self.dst_off + n] = src[self.gather]

    def params(self):
        return [self.log_scale, self.shift]


class _ScalarsJob:
    """blob[dst_off : dst_off + len(values)] = values (host constants, e.g. spline bounds)."""

    def __init__(self, values, dst_off: int):
        self.values, self.dst_off = [float(v) for v in values], dst_off

    def run(self, blobs: torch.Tensor, prec: int = 0) -> None:
        blobs[self.dst_off:self.dst_off + len(self.values)] = torch.tensor(self.values, dtype=torch.float32,
                                                                          device=blobs.device)

    def params(self):
        return []


class _PadJob:
    """blob[offsets] = values, re-applied whenever `W` is re-packed (the pack job of its rows writes 0 into the bias slots of rows
    it does not map: the parked values of a K < 16 spline tile sit in exactly those slots).  One scatter per layer."""

    def __init__(self, W, offsets: np.ndarray, values: np.ndarray):
        self.W, self.offsets_host, self.values_host = W, offsets, values
        self.offsets = self.values = None

    def run(self, blobs: torch.Tensor, prec: int = 0) -> None:
        if self.offsets is None or self.offsets.device != blobs.device:
            self.offsets = torch.from_numpy(self.offsets_host).to(blobs.device)
            self.values = torch.from_numpy(self.values_host).to(blobs.device)
        blobs.index_copy_(0, self.offsets, self.values)

    def params(self):
        return [self.W]


class _VectorJob:
    """blob[dst_off:] = cat(vec, 0)[gather] (per-slot constants in C-fragment order)."""

    def __init__(self, vec, dim: int, gather: np.ndarray, dst_off: int):
        self.vec, self.dim, self.gather_host, self.dst_off = vec, dim, gather.astype(np.int64), dst_off
        self.gather = None

    def run(self, blobs: torch.Tensor, prec: int = 0) -> None:
        dev = blobs.device
        if self.gather is None:
            self.gather = torch.from_numpy(self.gather_host).to(dev)
        src = torch.cat([self.vec.detach().to(dev, torch.float32).reshape(-1), torch.zeros(1, device=dev)])
        blobs[self.dst_off:self.dst_off + self.gather.numel()] = src[self.gather]

    def params(self):
        return [self.vec]


class _TimeConstJob:
    """Per-column constants of a time net (net/time_net.py:6-91) in the SX_STEP_COUPLING_TIME layout: per data tile [c_ls (32) |
    c_sh (32)] in C-fragment order for TimeLinear / TimeTanh (the scale) / TimeLog (exp(scale)); for TimeFourier(Bounded) per tile
    and feature k [w_ls | s_ls | w_sh | s_sh].  The embedding's first half scales log_scale, the second the shift (coupling.py:195),
    each `dim` wide or 1 wide (broadcast)."""

    def __init__(self, time_net, kind: int, K: int, dim: int, x_tiles: int, ls_col: np.ndarray, dst_off: int):
        self.time_net, self.kind, self.K, self.dim, self.x_tiles, self.dst_off = time_net, kind, K, dim, x_tiles, dst_off
        # slot (in C-fragment order per tile) -> column, -1 = not transformed
        order = np.full(32 * x_tiles, -1, dtype=np.int64)
        for t in range(x_tiles):
            for h in range(2):
                for r in range(16):
                    order[t * 32 + h * 16 + r] = ls_col[32 * t + _kmap(r, h)]
        self.order_host = order
        self._idx = None

    def params(self):
        return [p for p in self.time_net.parameters()]

    def run(self, blobs: torch.Tensor, prec: int = 0) -> None:
        dev = blobs.device
        tn = self.time_net
        with torch.no_grad():
            if self.kind == 4:
                S, P = tn.get_scale().to(dev, torch.float32), tn.shift.detach().to(dev, torch.float32)      # [out, K]
            else:
                sc = tn.scale.detach().reshape(-1, 1).to(dev, torch.float32)                               # [out, 1]
                S, P = (sc.exp() if self.kind == 3 else sc), None
            out = S.shape[0]
            half = out // 2
            if half not in (1, self.dim):
                raise ValueError(f'time_net width {out} does not broadcast against 2 x {self.dim}')
            if self._idx is None or self._idx[0].device != dev:
                o = self.order_host
                ls = np.where(o >= 0, o if half == self.dim else 0, out)                # `out` = the appended zero row
                sh = np.where(o >= 0, half + (o if half == self.dim else 0), out)
                self._idx = (torch.from_numpy(ls).to(dev), torch.from_numpy(sh).to(dev))
            ls_i, sh_i = self._idx
            zrow = torch.zeros(1, S.shape[1], dtype=torch.float32, device=dev)
            Sz = torch.cat([S, zrow])
            if self.kind == 4:
                Pz = torch.cat([P, zrow])
                X, K = self.x_tiles, self.K
                parts = [Sz[ls_i], Pz[ls_i], Sz[sh_i], Pz[sh_i]]                            # each [X * 32, K]
                val = torch.stack([q.reshape(X, 32, K).permute(0, 2, 1) for q in parts], dim=2)   # [X, K, 4, 32]
            else:
                val = torch.stack([Sz[ls_i].reshape(self.x_tiles, 32), Sz[sh_i].reshape(self.x_tiles, 32)], dim=1)   # [X, 2, 32]
            val = val.reshape(-1).contiguous()
            blobs[self.dst_off:self.dst_off + val.numel()] = val


class CompiledProgram:
    def __init__(self, prog: _hip.sx_program, blob_floats: int, jobs: List, in_col: Optional[np.ndarray],
                 out_col: Optional[np.ndarray], device: torch.device, mlp_out_dim: int = 0):
        self.prog = prog
        self.device = device
        self.jobs = jobs
        self.blob_floats = max(blob_floats, 256)
        # weight blobs per GEMM arithmetic (the fragment layouts differ): packed on first use / parameter change
        self._blobs = {}
        self._versions = {}
        self._pack_tables = {}
        self.in_col = None if in_col is None else torch.from_numpy(in_col.astype(np.int32)).to(device)
        self.out_col = None if out_col is None else torch.from_numpy(out_col.astype(np.int32)).to(device)
        self.mlp_out_dim = mlp_out_dim
        self.mlp_col0 = 0                       # first output column of this launch's window (chunked MLP programs)
        self.accumulates = False                # the launch adds into mlp_out instead of writing it
        self._tracked = None
        self._owner = None

    @property
    def blobs(self) -> torch.Tensor:
        """The packed weights of the default arithmetic (introspection / tools)."""
        return self.blobs_for(_hip.GEMM_F16X3)

    # -- parameter tracking ----------------------------------------------------------------------
    def _current_versions(self):
        ps = self._tracked
        if ps is None:                      # the jobs are fixed once the program is built
            ps = self._tracked = [p for j in self.jobs for p in j.params()]
        return [(p.data_ptr(), p._version) for p in ps]

    def blobs_for(self, prec: int) -> torch.Tensor:
        """Packed weights for arithmetic `prec`, re-packed when a tracked parameter changed."""
        if self._owner is not None:                 # a later chunk of a wide MLP program: the first chunk owns the pack jobs
            return self._owner.blobs_for(prec)
        blobs = self._blobs.get(prec)
        if blobs is None:
            blobs = self._blobs[prec] = torch.zeros(self.blob_floats, dtype=torch.float32, device=self.device)
        if self.jobs:
            v = self._current_versions()
            if v != self._versions.get(prec):
                first = prec not in self._versions
                if not first:
                    blobs[:1].zero_()              # header word 0: flags of the previous packing
                # every Linear of the program in ONE launch (sx_pack_linear_batch; 13 packs per spline coupling, re-run after each
                # optimizer step), then the other jobs in their order (a _PadJob follows the pack whose bias slots it overwrites)
                batched = sum(1 if type(j) is _PackJob else len(j.targets) if type(j) is _DerivedLinearJob else 0 for j in self.jobs) > 1
                if batched:
                    records = []
                    for j in self.jobs:
                        if type(j) is _PackJob:
                            records.append(j.record(blobs))
                        elif type(j) is _DerivedLinearJob:      # (the layer's matrix is derived here, into the job's own operands)
                            records += j.records(blobs)
                    table = self._pack_tables.get(prec)
                    if table is None:
                        table = self._pack_tables[prec] = _hip.PackTable()
                    table.run(blobs, records, prec, blobs.data_ptr())
                for j in self.jobs:
                    # constants (spline bounds, live-slot masks: no parameter behind them) are written once per blob buffer; a
                    # training step re-packs the weights only (each constant was a host-to-device copy per step before)
                    if batched and type(j) is _DerivedLinearJob:
                        j.write_ldj(blobs)
                    elif (first or j.params()) and not (batched and type(j) is _PackJob):
                        j.run(blobs, prec)
                self._versions[prec] = v
                if prec == _hip.GEMM_F16X3:
                    self._f16_overflow = None       # unknown for this weight version (weights_beyond_fp16)
        return blobs

    def weights_beyond_fp16(self) -> bool:
        """Do the weights of the current version exceed what fp16 x 3 fragments can hold (|w'| > 65504 after the folded constants)?
        sx_pack_linear raises SX_FLAG_F16_RANGE in header word 0 of the blob it packs; this reads that word ONCE per weight version
        (a 4-byte device-to-host copy: one synchronisation when weights change, none on the calls that follow) so that a 'fast' call
        can take the exact-fp32 objects for such a program instead of returning NaN rows + GemmRangeError -- the reference takes any
        finite weight (net/mlp.py:65, flows/affine.py:156-163).  Not asked while a graph is being built or captured (training steps
        re-pack every step: there the flag is reported as before)."""
        if self._owner is not None:
            return self._owner.weights_beyond_fp16()
        blobs = self.blobs_for(_hip.GEMM_F16X3)
        if getattr(self, '_f16_overflow', None) is None:
            word = int(blobs[:1].view(torch.int32).item())
            self._f16_overflow = bool(word & _hip.FLAG_F16_RANGE)
        return self._f16_overflow

    def jobs_or_owner(self) -> bool:
        return bool(self.jobs) or self._owner is not None

    def refresh(self) -> None:
        self.blobs_for(_hip.GEMM_F16X3 if _hip.get_gemm_precision() != 'exact' else _hip.GEMM_F32)

    def share_weights_of(self, other: 'CompiledProgram') -> None:
        """Chunks of one wide MLP program: same blobs; every request for them goes through the first chunk, which owns the
        pack jobs (so a precision first asked for by a later chunk -- 'auto' re-running one launch exactly -- is packed too)."""
        self._owner, self.jobs = other, []

    # -- launch -----------------------------------------------------------------------------------
    def run(self, x: torch.Tensor, latent: Optional[torch.Tensor] = None, want_y: bool = False,
            want_ldj: bool = False, want_logp: bool = False, sum_out: Optional[torch.Tensor] = None,
            mlp_out: Optional[torch.Tensor] = None, row_t: Optional[torch.Tensor] = None,
            side: Optional[torch.Tensor] = None, exact: bool = False):
        """x: [N, dim] contiguous on the program's device.  Returns (y | None, ldj | None, logp | None).
        exact: run on the exact-fp32 objects whatever the global arithmetic (HBM-bound single layers: net/mlp.py BatchLinear)."""
        _hip.require_device(x, 'x')
        assert x.dim() == 2 and x.shape[1] == (self.prog.pad_ or self.prog.dim), (x.shape, self.prog.dim, self.prog.pad_)
        if not x.is_contiguous():
            x = x.contiguous()
        _hip.poll_errors(device=x.device)   # a data-dependent condition of an EARLIER call on x's device (its current stream) surfaces here
        n = x.shape[0]
        if n == 0:        # empty batch: nothing to launch (the reference returns empty tensors too)
            e = lambda *shape, dt=torch.float32: torch.empty(*shape, dtype=dt, device=x.device)
            return (e(0, x.shape[1], dt=x.dtype) if want_y else None, e(0) if want_ldj else None,
                    e(0) if want_logp else None)
        y = torch.empty_like(x) if want_y else None
        ldj = torch.empty(n, dtype=torch.float32, device=x.device) if want_ldj else None
        logp = torch.empty(n, dtype=torch.float32, device=x.device) if want_logp else None
        if latent is not None:
            _hip.require_device(latent, 'latent')
            latent = latent.to(torch.float32).contiguous()
            assert latent.shape == (n, self.prog.latent_dim), (latent.shape, self.prog.latent_dim)
        elif self.prog.latent_dim:
            raise ValueError('this transform was built for a latent input but none was given')
        stride = mlp_out.stride(0) if mlp_out is not None else 0
        # (spline-coupling programs: an optional side output, the conditioner's last hidden activation [N, H] -- see SX_STEP_RQS_HIDDEN)
        mlp_dim = self.mlp_out_dim if (self.mlp_out_dim or mlp_out is None) else mlp_out.shape[1]
        if row_t is not None:
            _hip.require_device(row_t, 't')
            row_t = row_t.reshape(-1).to(torch.float32).contiguous()
            assert row_t.numel() == n, (row_t.shape, n)
        mode = 'exact' if exact else _hip.get_gemm_precision()
        with _hip.device_of(x):                     # the library launches on the CURRENT device's stream
            work = _hip.work_counters(x.device)
            flag = _hip.err_flag(x.device)

            def launch(prec, redo=False):
                # redo: the fp16 x 3 launch names the samples whose operands left fp16's range and the exact-fp32 kernel evaluates
                # them in a second launch (sx_flow_run2) -- any finite fp32 row at fp32-grade error, no flag, no synchronisation
                redo = redo and prec == _hip.GEMM_F16X3 and side is None and _hip.redo_allowed()
                rc = _hip.lib().sx_flow_run2(C.byref(self.prog), self.blobs_for(prec).data_ptr(),
                                             self.blobs_for(_hip.GEMM_F32).data_ptr() if redo else None,
                                             _hip.redo_list(x.device, n).data_ptr() if redo else None, x.data_ptr(),
                                             _hip.ptr(latent), _hip.ptr(self.in_col), _hip.ptr(self.out_col), _hip.ptr(y),
                                             _hip.ptr(ldj), _hip.ptr(logp), _hip.ptr(sum_out),
                                             None if mlp_out is None else mlp_out.data_ptr() + 4 * self.mlp_col0, stride,
                                             mlp_dim, _hip.ptr(row_t), _hip.ptr(side), n, _hip.dtype_code(x),
                                             prec, work.data_ptr(), flag, _hip.stream())
                if rc != 0:
                    work.zero_()                    # a failed launch may leave the ticket pair armed
                    if redo:
                        _hip.redo_list(x.device, n).zero_()      # ... and, where the second launch did not happen, a list that is not empty
                _hip.check(rc, 'sx_flow_run')
                if mode != 'auto':
                    _hip.after_launch()             # STRIBOR_SYNC_ERRORS: the data-dependent error leaves THIS call

            if mode == 'exact':
                launch(_hip.GEMM_F32)
            elif mode == 'fast':
                # (weights beyond fp16's range: the same program on the exact-fp32 objects, silently -- a parameter's magnitude is not
                #  an error in the reference.  Decided from the pack's own flag, once per weight version, on calls without a graph.)
                # (`redo_allowed`: not inside a graph-building flow call -- autograd runs a Function's forward with grad mode off, so
                #  grad mode alone would let every training step pay this read-back)
                quiet = _hip.redo_allowed() and not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()
                launch(_hip.GEMM_F32 if (quiet and self.jobs_or_owner() and self.weights_beyond_fp16()) else _hip.GEMM_F16X3, redo=True)
            else:                                   # 'auto': never hand back a NaN-poisoned result
                keep = None if sum_out is None else sum_out.clone()
                # an accumulating launch (mlp_out += chunk) is not idempotent: the exact re-run must start from what the output
                # held BEFORE the fp16 x 3 attempt, or the unflagged rows get the chunk's contribution twice
                keep_out = mlp_out.clone() if (self.accumulates and mlp_out is not None) else None
                torch.cuda.current_stream().synchronize()
                _hip.poll_errors(device=x.device)   # a flag an EARLIER call left behind is that call's: raise it, do not swallow it
                launch(_hip.GEMM_F16X3)
                torch.cuda.current_stream().synchronize()
                if _hip.take_flag(x.device, _hip.FLAG_F16_RANGE):
                    if keep is not None:
                        sum_out.copy_(keep)
                    if keep_out is not None:
                        mlp_out.copy_(keep_out)
                    launch(_hip.GEMM_F32)
        return y, ldj, logp

    def launch_info(self, n_rows: int) -> Tuple[int, int, int]:
        g, b, l = C.c_int32(), C.c_int32(), C.c_int32()
        _hip.check(_hip.lib().sx_flow_launch_info(C.byref(self.prog), n_rows, C.byref(g), C.byref(b), C.byref(l)),
                   'sx_flow_launch_info')
        return g.value, b.value, l.value


class ProgramTooLong(NotImplementedError):
    """The flow plans, but its steps do not fit one program (SX_MAX_STEPS): the one failure a run of SEGMENTS can cure."""


class ProgramBuilder:
    """Accumulates steps; tracks which logical column each state slot holds."""

    def __init__(self, dim: int, latent_dim: int = 0, hidden_width: int = 32, min_x_tiles: int = 1, time_slots: int = 0):
        self.dim, self.latent_dim = dim, latent_dim
        # time-conditioned programs (ContinuousAffineCoupling / NeuralFlow): t and t0 take the `time_slots` (0..2) slots behind the
        # latent columns in the latent tiles; the kernel fills them from row_t / the second time vector
        self.time_slots = time_slots
        self._spline_layers = 0          # spline couplings added so far (ordinal of the next one's side outputs)
        lat_tiles = _ceil_div(latent_dim + time_slots, 32)
        need = max(_ceil_div(dim, 32), min_x_tiles) + lat_tiles
        # 129 .. 256 columns (round 4): eight state tiles at one wave per SIMD.  Only affine couplings whose masks split the tiles
        # fuse at that width, as hidden-chunk steps of 32 units (the only step whose weights fit the LDS ring there: 48 KB), beside
        # element-wise affines and column shuffles; everything else keeps its layer-by-layer tier.
        self.wide_state = 4 < need <= 8 and lat_tiles == 0
        self.tiles = 8 if self.wide_state else _round_tiles(need, f'dim {dim} + latent {latent_dim}')
        self.x_tiles = self.tiles - lat_tiles
        # hidden layers wider than four tiles: single-hidden-layer Tanh couplings run as CHUNK steps (add_coupling_affine: the
        # conditioner's output is a sum over hidden-unit chunks, accumulated in registers across the steps); the chunk is as wide
        # as one step's weights allow in the LDS ring (128 hidden units up to 64 columns, 64 beyond)
        ht = _ceil_div(hidden_width, 32)
        self.h_tiles = _round_tiles(ht, f'hidden width {hidden_width}') if ht <= 4 else (2 if self.tiles >= 4 else 4)

        self.n_slots = 32 * self.x_tiles
        # slot -> logical column (-1 = padding); identity until choose_layout / permutations change it
        self.col_of_slot = np.full(self.n_slots, -1, dtype=np.int64)
        self.col_of_slot[:dim] = np.arange(dim)
        self.in_col: Optional[np.ndarray] = None     # set at the first step
        self.steps: List[dict] = []
        self.jobs: List = []
        self.blob_floats = 256             # 1 KiB header: word 0 = flags raised while packing (sx_flow_run reports them)
        self.mlp_out_dim = 0
        # conditioner programs of couplings wider than the state tiles read a COLUMN SUBSET of the wide rows: `x_cols` (column of
        # x per state column of this program) and the rows' stride
        self.x_cols: Optional[np.ndarray] = None
        self.x_stride = 0

    # -- planning a flow in segments: a layer that would not fit the program is taken back ---------------------------
    def snapshot(self):
        """Every field of the builder (shallow copy of __dict__; lists, arrays and dicts copied one level deep), so that a field a
        _plan starts to mutate in a later round is covered without touching this method."""
        snap = {}
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                snap[k] = v.copy()
            elif isinstance(v, (list, dict, set)):
                snap[k] = type(v)(v)
            else:
                snap[k] = v
        return snap

    def restore(self, snap) -> None:
        self.__dict__.clear()
        for k, v in snap.items():
            self.__dict__[k] = v.copy() if isinstance(v, np.ndarray) else type(v)(v) if isinstance(v, (list, dict, set)) else v

    # -- layout ------------------------------------------------------------------------------------
    def choose_layout(self, first_mask: Optional[np.ndarray]) -> None:
        """Put the first coupling's conditioning columns in the low tiles and its transformed columns
        in the high tiles when both fit, so masks that are not contiguous (parity_*) still prune."""
        assert not self.steps
        if first_mask is None or self.tiles < 2 or self.latent_dim or self.x_tiles != self.tiles:
            return
        half = 16 * self.tiles
        cond = np.nonzero(first_mask > 0.5)[0]
        live = np.nonzero(first_mask <= 0.5)[0]
        if len(cond) == 0 or len(live) == 0 or len(cond) > half or len(live) > half:
            return
        # identity already splits along the tile halves (ordered_* masks at full tiles): keep vector loads
        if (cond.max() < half <= live.min()) or (live.max() < half <= cond.min()):
            return
        lay = np.full(self.n_slots, -1, dtype=np.int64)
        lay[:len(cond)] = cond
        lay[half:half + len(live)] = live
        self.col_of_slot = lay

    def _freeze_input(self) -> None:
        if self.in_col is None:
            self.in_col = self.col_of_slot.copy()

    def _narrow_only(self, what: str) -> None:
        if self.wide_state:
            raise NotImplementedError(f'{what} fuse up to 128 columns (129 .. 256: affine couplings with split masks only)')

    def slot_of_col(self) -> np.ndarray:
        s = np.full(self.dim, -1, dtype=np.int64)
        ok = self.col_of_slot >= 0
        s[self.col_of_slot[ok]] = np.nonzero(ok)[0]
        return s

    def _alloc(self, n_floats: int) -> Tuple[int, int]:
        n = _ceil_div(max(n_floats, 1), 256) * 256
        # one step's weights must fit the double-buffered LDS ring (2 x n floats + the ticket slots in 160 KiB); a layer
        # that does not (e.g. a dense-mask coupling with a 128-wide conditioner) makes the flow fall back to its
        # layer-by-layer form, where the conditioner runs as its own program of smaller steps
        if n * 8 + 16 > 160 * 1024:
            raise NotImplementedError(f'a fused step would need {n * 4} B of weights per LDS buffer (> 80 KiB)')
        off = self.blob_floats
        self.blob_floats += n
        return off, n

    # -- steps -------------------------------------------------------------------------------------
    def add_permutation(self, perm: np.ndarray, reverse: bool) -> None:
        """forward: y[j] = x[perm[j]]; inverse: x[j] = y[inv[j]]  (permute.py:71,75) — a relabelling."""
        self._freeze_input()
        perm = np.asarray(perm, dtype=np.int64)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(perm))
        relabel = perm if reverse else inv        # new label of the slot that held old label j
        ok = self.col_of_slot >= 0
        self.col_of_slot[ok] = relabel[self.col_of_slot[ok]]

    def add_coupling_affine(self, W1, b1, W2, b2, mask: np.ndarray, act: int, reverse: bool, ldj_scale: float,
                            hidden: int) -> None:
        self._freeze_input()
        D, T, HT = self.dim, self.tiles, self.h_tiles
        mask = np.asarray(mask, dtype=np.float64).reshape(-1)
        if mask.size == 1:
            mask = np.full(D, mask[0])
        cond_col = mask > 0.5
        if D == 1:
            cond_col = np.zeros(1, dtype=bool)          # z = z * 0 (coupling.py:62-63)
        live_col = mask <= 0.5
        col = self.col_of_slot
        slot_cond = np.array([c >= 0 and cond_col[c] for c in col])
        slot_live = np.array([c >= 0 and live_col[c] for c in col])
        half = 16 * T
        variant = 'dense'
        if T >= 2 and self.latent_dim == 0 and self.x_tiles == T:
            if not slot_cond[half:].any() and not slot_live[:half].any():
                variant = 'low'
            elif not slot_cond[:half].any() and not slot_live[half:].any():
                variant = 'high'
        if variant == 'low':
            c0, ct, t0, tt = 0, T // 2, T // 2, T // 2
        elif variant == 'high':
            c0, ct, t0, tt = T // 2, T // 2, 0, T // 2
        else:
            c0, ct, t0, tt = 0, T, 0, T
        # GEMM-1 operand: hidden rows x conditioning slots
        k_slots = np.arange(32 * c0, 32 * (c0 + ct))
        col_idx = np.full(len(k_slots), -1, dtype=np.int64)
        for i, p in enumerate(k_slots):
            if p < self.n_slots:
                if slot_cond[p]:
                    col_idx[i] = col[p]
            else:
                li = p - self.n_slots
                if li < self.latent_dim:
                    col_idx[i] = D + li                    # cat([z, latent]) (coupling.py:64-65)
        n1 = _hip.packed_linear_floats(HT, ct)
        n2 = _hip.packed_linear_floats(2 * tt, HT)
        wide = hidden > 32 * HT
        if (wide or self.wide_state) and (act != _hip.ACT_CODES['Tanh'] or variant == 'dense'):
            raise NotImplementedError('hidden layers wider than the program\'s hidden tiles: Tanh conditioners, masks that split the tiles')
        # Tanh hot path: fold the constants of tanh(z) = 1 - 2/(exp2(2 log2e z) + 1) and exp(x) = exp2(log2e x)
        # into the packed weights (exact re-parametrisation; fp32 MFMA shares the VALU, so every saved VALU
        # instruction is saved wall time):  hidden r = 1/(exp2(c z) + 1);  W2 tanh + b2 = (-2 W2) r + (b2 + W2 1).
        folded = act == _hip.ACT_CODES['Tanh']
        LOG2E = 1.4426950408889634
        rs1 = bs1 = rs2 = bs2 = None
        fold = 0.0
        if folded:
            kk = (-LOG2E) if reverse else LOG2E              # exp(-ls) / exp(+ls) -> exp2(kk ls)
            rs1 = np.full(32 * HT, 2.0 * LOG2E)
            bs1 = rs1
            rs2 = np.empty(32 * 2 * tt)
            bs2 = np.empty(32 * 2 * tt)
            for t in range(tt):
                rs2[32 * (2 * t):32 * (2 * t + 1)] = -2.0 * kk
                bs2[32 * (2 * t):32 * (2 * t + 1)] = kk
                rs2[32 * (2 * t + 1):32 * (2 * t + 2)] = -2.0
                bs2[32 * (2 * t + 1):32 * (2 * t + 2)] = 1.0
            fold = 1.0
            act = _hip.ACT_TANH_FOLDED
            ldj_scale = ldj_scale / kk
        # GEMM-2 operand: per transformed tile, 32 log_scale rows then 32 shift rows (affine.py:66)
        row2 = np.full(32 * 2 * tt, -1, dtype=np.int64)
        for t in range(tt):
            for i in range(32):
                p = 32 * (t0 + t) + i
                if p < self.n_slots and slot_live[p]:
                    row2[32 * (2 * t) + i] = col[p]
                    row2[32 * (2 * t + 1) + i] = D + col[p]
        if self.wide_state:
            # eight state tiles: hidden step + one step per transformed tile (sx_flow_kernel.h wide_hidden / wide_affine_tile)
            if wide:
                raise NotImplementedError('129 .. 256 columns: hidden layers of up to 128 units')
            row_idx = np.full(32 * HT, -1, dtype=np.int64)
            row_idx[:hidden] = np.arange(hidden)
            col2 = np.full(32 * HT, -1, dtype=np.int64)
            col2[:hidden] = np.arange(hidden)
            off, n = self._alloc(n1)
            self.jobs.append(_PackJob(W1, b1, row_idx, col_idx, HT, ct, off, rs1, bs1, 0.0))
            self.steps.append(dict(kind=_hip.STEP_WIDE_HIDDEN, c0=c0, ct=ct, t0=t0, tt=tt, reverse=int(reverse), act=act, blob_off=off,
                                   blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
            nt = _hip.packed_linear_floats(2, HT)
            for t in range(tt):
                if not (row2[64 * t:64 * t + 64] >= 0).any():
                    continue                              # a tile of padding only
                off, n = self._alloc(nt)
                self.jobs.append(_PackJob(W2, b2, row2[64 * t:64 * t + 64], col2, 2, HT, off, rs2[64 * t:64 * t + 64], bs2[64 * t:64 * t + 64], fold))
                self.steps.append(dict(kind=_hip.STEP_WIDE_AFFINE_TILE, c0=0, ct=0, t0=t0 + t, tt=1, reverse=int(reverse), act=act,
                                       blob_off=off, blob_floats=n, ldj_scale=ldj_scale, ldj_const=0.0))
            return
        # one step per chunk of 32 HT hidden units (one chunk = the whole layer up to that width).  W2 tanh(W1 z + b1) + b2 is a SUM
        # over hidden-unit chunks: a chunk step evaluates its slice of the hidden layer and adds its share of (log_scale, shift) to
        # accumulator tiles that live in registers across the steps (kernel MODE 9); the first chunk carries b2, every chunk its own
        # share of the folded constant W2 1 (the packer folds over the chunk's columns), the last one applies the affine map
        chunk = 32 * HT
        n_chunks = _ceil_div(hidden, chunk)
        for c in range(n_chunks):
            hsel = np.arange(c * chunk, min(hidden, (c + 1) * chunk))
            row_idx = np.full(32 * HT, -1, dtype=np.int64)
            row_idx[:len(hsel)] = hsel
            col2 = np.full(32 * HT, -1, dtype=np.int64)
            col2[:len(hsel)] = hsel
            off, n = self._alloc(n1 + n2)
            self.jobs.append(_PackJob(W1, b1, row_idx, col_idx, HT, ct, off, rs1, bs1, 0.0))
            self.jobs.append(_PackJob(W2, b2 if c == 0 else None, row2, col2, 2 * tt, HT, off + n1, rs2, bs2, fold))
            step = dict(kind=_hip.STEP_COUPLING_AFFINE_HC if wide else _hip.STEP_COUPLING_AFFINE, c0=c0, ct=ct, t0=t0, tt=tt,
                        reverse=int(reverse), act=act, blob_off=off, blob_floats=n, ldj_scale=ldj_scale, ldj_const=0.0)
            if wide:
                step['pad_'] = int(c == 0) | (int(c == n_chunks - 1) << 1)
            self.steps.append(step)

    def add_coupling_affine_deep(self, linears: Sequence[Tuple], mask: np.ndarray, act: int, reverse: bool,
                                 ldj_scale: float) -> None:
        """Affine coupling whose conditioner has two or more hidden layers (linears = [(W1, b1), ..., (W_out, b_out)]):
        CPL_HIDDEN (state -> hidden), CPL_HIDDEN2 for every middle layer, COUPLING_AFFINE_DEEP (last hidden layer,
        output layer, affine map); the hidden activations stay in registers between the steps."""
        self._narrow_only('deep conditioners')
        self._freeze_input()
        assert len(linears) >= 3
        D, T, HT = self.dim, self.tiles, self.h_tiles
        if max(W.shape[0] for (W, _) in linears[:-1]) > 32 * HT:
            raise NotImplementedError('conditioners with several hidden layers fuse up to 128 units per layer')
        mask = np.asarray(mask, dtype=np.float64).reshape(-1)
        if mask.size == 1:
            mask = np.full(D, mask[0])
        cond_col = mask > 0.5
        if D == 1:
            cond_col = np.zeros(1, dtype=bool)
        live_col = mask <= 0.5
        col = self.col_of_slot
        slot_cond = np.array([c >= 0 and cond_col[c] for c in col])
        slot_live = np.array([c >= 0 and live_col[c] for c in col])
        c0, ct, t0, tt = self._cond_variant(slot_cond, slot_live)
        k_slots = np.arange(32 * c0, 32 * (c0 + ct))
        col_idx = np.full(len(k_slots), -1, dtype=np.int64)
        for i, p in enumerate(k_slots):
            if p < self.n_slots:
                if slot_cond[p]:
                    col_idx[i] = col[p]
            else:
                li = p - self.n_slots
                if li < self.latent_dim:
                    col_idx[i] = D + li

        def hidden_idx(width):
            idx = np.full(32 * HT, -1, dtype=np.int64)
            idx[:width] = np.arange(width)
            return idx

        (W1, b1) = linears[0]
        off, n = self._alloc(_hip.packed_linear_floats(HT, ct))
        self.jobs.append(_PackJob(W1, b1, hidden_idx(W1.shape[0]), col_idx, HT, ct, off))
        self.steps.append(dict(kind=_hip.STEP_CPL_HIDDEN, c0=c0, ct=ct, t0=0, tt=0, reverse=0, act=act, blob_off=off,
                               blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
        prev = W1.shape[0]
        for (Wk, bk) in linears[1:-2]:
            off, n = self._alloc(_hip.packed_linear_floats(HT, HT))
            self.jobs.append(_PackJob(Wk, bk, hidden_idx(Wk.shape[0]), hidden_idx(prev), HT, HT, off))
            self.steps.append(dict(kind=_hip.STEP_CPL_HIDDEN2, c0=0, ct=0, t0=0, tt=0, reverse=0, act=act, blob_off=off,
                                   blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
            prev = Wk.shape[0]
        (WL, bL), (Wo, bo) = linears[-2], linears[-1]
        n1 = _hip.packed_linear_floats(HT, HT)
        n2 = _hip.packed_linear_floats(2 * tt, HT)
        off, n = self._alloc(n1 + n2)
        self.jobs.append(_PackJob(WL, bL, hidden_idx(WL.shape[0]), hidden_idx(prev), HT, HT, off))
        row2 = np.full(32 * 2 * tt, -1, dtype=np.int64)
        for t in range(tt):
            for i in range(32):
                p = 32 * (t0 + t) + i
                if p < self.n_slots and slot_live[p]:
                    row2[32 * (2 * t) + i] = col[p]                 # log_scale rows, then shift rows (affine.py:66)
                    row2[32 * (2 * t + 1) + i] = D + col[p]
        self.jobs.append(_PackJob(Wo, bo, row2, hidden_idx(WL.shape[0]), 2 * tt, HT, off + n1))
        self.steps.append(dict(kind=_hip.STEP_COUPLING_AFFINE_DEEP, c0=0, ct=0, t0=t0, tt=tt, reverse=int(reverse),
                               act=act, blob_off=off, blob_floats=n, ldj_scale=ldj_scale, ldj_const=0.0))

    def _cond_variant(self, slot_cond, slot_live):
        """(c0, ct, t0, tt) tile ranges: pruned low / high halves when the masks align with the tiles, else dense."""
        T = self.tiles
        half = 16 * T
        if T >= 2 and self.latent_dim == 0 and self.x_tiles == T:
            if not slot_cond[half:].any() and not slot_live[:half].any():
                return 0, T // 2, T // 2, T // 2
            if not slot_cond[:half].any() and not slot_live[half:].any():
                return T // 2, T // 2, 0, T // 2
        return 0, T, 0, T

    def add_coupling_rqs(self, W1, b1, W2, b2, mask: np.ndarray, reverse: bool, ldj_scale: float, hidden: int,
                         n_bins: int, left: float, right: float, bottom: float, top: float, middle=(), cubic: bool = False) -> None:
        """Rational-quadratic spline coupling, fused: hidden step(s) + 12 phase steps per transformed tile.
        Tanh conditioners; with two or more hidden layers (`middle` = [(W, b), ...] between W1 and the output layer W2)
        the earlier layers run as CPL_HIDDEN / CPL_HIDDEN2 steps and the last hidden layer (folded tanh) as the
        RQS_HIDDEN step; n_bins <= 16.  cubic=True: the monotone cubic spline (2K+2 parameters per element: widths, heights,
        two boundary derivatives; phase steps with act = 1, kernel MODE 12 / 13)."""
        self._narrow_only('spline couplings')
        self._freeze_input()
        D, T, HT, K = self.dim, self.tiles, self.h_tiles, n_bins
        if max([hidden] + [Wk.shape[0] for (Wk, _) in middle]) > 32 * HT:
            raise NotImplementedError('spline couplings fuse with hidden layers of up to 128 units')
        if K > (16 if cubic else 32):
            raise NotImplementedError('fused spline coupling supports n_bins <= 16 (cubic) / 32 (rational-quadratic)')
        P = 2 * K + 2 if cubic else 3 * K - 1
        mask = np.asarray(mask, dtype=np.float64).reshape(-1)
        if mask.size == 1:
            mask = np.full(D, mask[0])
        cond_col = mask > 0.5
        if D == 1:
            cond_col = np.zeros(1, dtype=bool)
        live_col = mask <= 0.5
        col = self.col_of_slot
        slot_cond = np.array([c >= 0 and cond_col[c] for c in col])
        slot_live = np.array([c >= 0 and live_col[c] for c in col])
        c0, ct, t0, tt = self._cond_variant(slot_cond, slot_live)
        LOG2E = 1.4426950408889634
        # hidden layer (folded tanh: r = 1/(exp2(2 log2e z) + 1))
        k_slots = np.arange(32 * c0, 32 * (c0 + ct))
        col_idx = np.full(len(k_slots), -1, dtype=np.int64)
        for i, p in enumerate(k_slots):
            if p < self.n_slots:
                if slot_cond[p]:
                    col_idx[i] = col[p]
            else:
                li = p - self.n_slots
                if li < self.latent_dim:
                    col_idx[i] = D + li
        def hidden_idx(width):
            idx = np.full(32 * HT, -1, dtype=np.int64)
            idx[:width] = np.arange(width)
            return idx

        sc = np.full(32 * HT, 2.0 * LOG2E)
        if not middle:
            off, n = self._alloc(_hip.packed_linear_floats(HT, ct))
            self.jobs.append(_PackJob(W1, b1, hidden_idx(hidden), col_idx, HT, ct, off, sc, sc, 0.0))
            step = dict(kind=_hip.STEP_RQS_HIDDEN, c0=c0, ct=ct, t0=t0, tt=tt, reverse=int(reverse),
                        act=_hip.ACT_TANH_FOLDED, blob_off=off, blob_floats=n, ldj_scale=0.0, ldj_const=0.0)
            step['pad_'] = self._spline_layers << 8          # the layer's ordinal (side outputs of a whole-flow program)
            self.steps.append(step)
        else:
            tanh = _hip.ACT_CODES['Tanh']
            off, n = self._alloc(_hip.packed_linear_floats(HT, ct))
            self.jobs.append(_PackJob(W1, b1, hidden_idx(W1.shape[0]), col_idx, HT, ct, off))
            self.steps.append(dict(kind=_hip.STEP_CPL_HIDDEN, c0=c0, ct=ct, t0=0, tt=0, reverse=0, act=tanh, blob_off=off,
                                   blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
            prev = W1.shape[0]
            for (Wk, bk) in middle[:-1]:
                off, n = self._alloc(_hip.packed_linear_floats(HT, HT))
                self.jobs.append(_PackJob(Wk, bk, hidden_idx(Wk.shape[0]), hidden_idx(prev), HT, HT, off))
                self.steps.append(dict(kind=_hip.STEP_CPL_HIDDEN2, c0=0, ct=0, t0=0, tt=0, reverse=0, act=tanh,
                                       blob_off=off, blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
                prev = Wk.shape[0]
            WL, bL = middle[-1]
            hidden = WL.shape[0]
            off, n = self._alloc(_hip.packed_linear_floats(HT, HT))
            self.jobs.append(_PackJob(WL, bL, hidden_idx(hidden), hidden_idx(prev), HT, HT, off, sc, sc, 0.0))
            step = dict(kind=_hip.STEP_RQS_HIDDEN, c0=0, ct=0, t0=t0, tt=tt, reverse=int(reverse),
                        act=_hip.ACT_TANH_FOLDED, blob_off=off, blob_floats=n, ldj_scale=0.0, ldj_const=0.0)
            step['pad_'] = 1 | (self._spline_layers << 8)    # 1: source = the hidden state kept by the CPL_HIDDEN steps
            self.steps.append(step)
        col2 = np.full(32 * HT, -1, dtype=np.int64)
        col2[:hidden] = np.arange(hidden)
        # forward searches the widths on [left, right], inverse the heights on [bottom, top]
        blocks = [(K, K, bottom, top), (0, K, left, right)] if reverse else [(0, K, left, right), (K, K, bottom, top)]
        blocks.append((2 * K, 2, left, right) if cubic else (2 * K, K - 1, 0.0, 0.0))      # cubic: the evaluation needs the domain
        pad_off, pad_val = [], []           # bias slots of unused tile registers (K < 16), written behind the layer's pack jobs
        for t in range(t0, t0 + tt):
            if t >= self.x_tiles:
                continue
            live_mask = 0
            for i in range(32):
                if slot_live[32 * t + i]:
                    live_mask |= 1 << i
            if live_mask == 0:
                continue
            for g in range(4):
                if not any((live_mask >> (q + 8 * g + 4 * h)) & 1 for q in range(4) for h in range(2)):
                    continue
                # up to 16 bins: one triple of steps per group, an element's parameters = ONE output tile (four elements per step);
                # 17 .. 32 bins (rational-quadratic only): TWO tiles per element, so a group is two triples of two elements each
                # (step.act bit 1 = which pair)
                wide = K > 16
                for half in ((0, 1) if wide else (0,)):
                    elems = (2 * half, 2 * half + 1) if wide else (0, 1, 2, 3)
                    if wide and not any((live_mask >> (q + 8 * g + 4 * h)) & 1 for q in elems for h in range(2)):
                        continue
                    bound_slot = None
                    for phase, (start, count, lo, hi) in enumerate(blocks):
                        # output tile of the step -> register k of lane half h = row kmap(k, h) of the tile (sx_flow_spline.h RQS_P)
                        rows = np.full(128, -1, dtype=np.int64)
                        for qi, q in enumerate(elems):
                            for h in range(2):
                                slot = 32 * t + q + 8 * g + 4 * h
                                if not slot_live[slot]:
                                    continue
                                for k in range(min(count, 32 if wide else 16)):
                                    tile = (2 * qi + (k >> 4)) if wide else q
                                    rows[32 * tile + _kmap(k & 15, h)] = col[slot] * P + start + k
                        nlin = _hip.packed_linear_floats(4, HT)
                        off, n = self._alloc(nlin + 4)
                        # the two softmax blocks are packed in base 2 (rows and bias times log2 e): the kernel's softmax is then
                        # v_exp_f32(p - max) with no multiply (32 instructions per element)
                        sc2 = LOG2E if phase < 2 else 1.0
                        # (the two softmax blocks of a group leave the bound on their logits in the slot behind (lo, hi) of the FIRST
                        #  block's blob: the K = 16 phases run without a running maximum below it -- sx_flow_spline.h rqs16_sums --
                        #  and the kernel decides once per group; the blob buffer starts zeroed and the slot only grows)
                        if phase == 0:
                            bound_slot = off + nlin + 2
                        self.jobs.append(_PackJob(W2, b2, rows, col2, 4, HT, off, np.full(128, -2.0 * sc2), np.full(128, sc2), 1.0,
                                                  bound_off=bound_slot if phase < 2 else None))
                        self.jobs.append(_ScalarsJob([lo, hi], off + nlin))
                        if (K < (32 if wide else 16) and phase < 2) or (phase == 2 and not cubic):
                            # The unused registers of an element's tile (16 slots; 32 for 17 .. 32 bins) are parked where the
                            # straight-line code of the kernel needs no predicate for them (sx_flow_spline.h, rqs16_c / rqs32_c): logits
                            # of bins >= K at -1e30 (exp2 = 0: every sum is that of the K bins), derivative rows >= K - 1 -- there is
                            # always at least one -- at the boundary-derivative constant log(exp(1 - 1e-3) - 1) of
                            # rational_quadratic_spline.py:81 (D[K - 1] then needs no select).
                            # Bias block of the pack: [tile][lane half][register] behind the 4 x HT fragment tiles.
                            first, pad = (K, -1e30) if phase < 2 else (K - 1, 0.5397424172369522)
                            for qi, q in enumerate(elems):
                                for h in range(2):
                                    if slot_live[32 * t + q + 8 * g + 4 * h]:
                                        if wide:        # 17 .. 32 bins: parameter k of the step's element qi = register k & 15 of tile 2 qi + (k >> 4)
                                            for k in range(first, 32):
                                                pad_off.append(off + 4 * HT * 1024 + 32 * (2 * qi + (k >> 4)) + 16 * h + (k & 15))
                                                pad_val.append(pad)
                                        else:
                                            base = off + 4 * HT * 1024 + 32 * q + 16 * h
                                            pad_off += list(range(base + first, base + 16))
                                            pad_val += [pad] * (16 - first)
                        s_scale = (-ldj_scale if reverse else ldj_scale) if phase == 2 else 0.0
                        step = dict(kind=_hip.STEP_RQS_PHASE, c0=g, ct=phase, t0=t, tt=K, reverse=int(reverse),
                                    act=int(cubic) | (half << 1), blob_off=off, blob_floats=n, ldj_scale=s_scale, ldj_const=0.0)
                        step['pad_'] = live_mask - (1 << 32) if live_mask >= (1 << 31) else live_mask
                        self.steps.append(step)

        if pad_off:
            self.jobs.append(_PadJob(W2, np.asarray(pad_off, dtype=np.int64), np.asarray(pad_val, dtype=np.float32)))
        self._spline_layers += 1

    def enable_adjoint_tiles(self) -> None:
        """Backward programs carry dL/dx beside x: tiles [0, x_tiles) = x, [x_tiles, 2 x_tiles) = adjoint."""
        self._narrow_only('backward programs')
        if self.x_tiles > 4 or self.latent_dim:
            raise NotImplementedError('training backward is built for up to 128 columns without latent inputs')
        if self.x_tiles == 4 and self.h_tiles > 2:
            raise NotImplementedError('the backward program of 128-column flows is built for hidden widths up to 64')
        self.tiles = 2 * self.x_tiles

    def add_coupling_affine_bwd(self, W1, b1, W2, b2, mask: np.ndarray, hidden: int, layer_slot: int) -> dict:
        """Backward of one affine coupling of a log_prob pass; returns the slot maps the caller needs to turn the
        kernel's per-row factors into weight gradients."""
        self._narrow_only('backward programs')
        self._freeze_input()
        XT = self.x_tiles
        assert self.tiles == 2 * XT
        D, HT = self.dim, self.h_tiles
        if hidden > 32 * HT:         # (h_tiles of a builder for > 128 hidden units is the forward's chunk width: no backward chunk steps)
            raise NotImplementedError(f'backward programs hold hidden layers of up to {32 * HT} units')
        mask = np.asarray(mask, dtype=np.float64).reshape(-1)
        if mask.size == 1:
            mask = np.full(D, mask[0])
        cond_col, live_col = mask > 0.5, mask <= 0.5
        if D == 1:
            cond_col = np.zeros(1, dtype=bool)
        col = self.col_of_slot
        slot_cond = np.array([c >= 0 and cond_col[c] for c in col])
        slot_live = np.array([c >= 0 and live_col[c] for c in col])
        half = 16 * XT
        if XT >= 2 and not slot_cond[half:].any() and not slot_live[:half].any():
            c0, ct, t0, tt = 0, XT // 2, XT // 2, XT // 2
        elif XT >= 2 and not slot_cond[:half].any() and not slot_live[half:].any():
            c0, ct, t0, tt = XT // 2, XT // 2, 0, XT // 2
        else:
            c0, ct, t0, tt = 0, XT, 0, XT                     # dense: any mask, zero weights outside it
        LOG2E = 1.4426950408889634
        kk = -LOG2E                                              # log_prob direction: scale = exp(-log_scale)
        col_idx = np.array([col[p] if slot_cond[p] else -1 for p in range(32 * c0, 32 * (c0 + ct))], dtype=np.int64)
        row_h = np.full(32 * HT, -1, dtype=np.int64)
        row_h[:hidden] = np.arange(hidden)
        row2 = np.full(64 * tt, -1, dtype=np.int64)
        for t in range(tt):
            for i in range(32):
                p = 32 * (t0 + t) + i
                if slot_live[p]:
                    row2[64 * t + i] = col[p]
                    row2[64 * t + 32 + i] = D + col[p]
        n1, n2 = _hip.packed_linear_floats(HT, ct), _hip.packed_linear_floats(2 * tt, HT)
        n3, n4 = _hip.packed_linear_floats(HT, 2 * tt), _hip.packed_linear_floats(ct, HT)
        rs1 = np.full(32 * HT, 2.0 * LOG2E)
        rs2 = np.concatenate([np.concatenate([np.full(32, -2.0 * kk), np.full(32, -2.0)]) for _ in range(tt)])
        bs2 = np.concatenate([np.concatenate([np.full(32, kk), np.full(32, 1.0)]) for _ in range(tt)])
        if XT == 4:
            # 128-column flows: the four operands (100 KB) do not fit the double-buffered LDS ring together -> two steps, the
            # forward operands (BWD_A) and the transposed ones (BWD_B); halves only (a dense mask would need 8 x 8 tiles)
            if (ct, tt) != (2, 2):
                raise NotImplementedError('backward of 128-column couplings: conditioner / transformed columns must be the tile halves')
            offa, na = self._alloc(n1 + n2)
            self.jobs.append(_PackJob(W1, b1, row_h, col_idx, HT, ct, offa, rs1, rs1, 0.0))
            self.jobs.append(_PackJob(W2, b2, row2, row_h, 2 * tt, HT, offa + n1, rs2, bs2, 1.0))
            offb, nb = self._alloc(n3 + n4)
            self.jobs.append(_PackJob(W2, None, row_h, row2, HT, 2 * tt, offb, transpose=True))            # W2^T
            self.jobs.append(_PackJob(W1, None, col_idx, row_h, ct, HT, offb + n3, transpose=True))        # W1^T
            for kind, off, n in ((_hip.STEP_COUPLING_AFFINE_BWD_A, offa, na), (_hip.STEP_COUPLING_AFFINE_BWD_B, offb, nb)):
                self.steps.append(dict(kind=kind, c0=c0, ct=ct, t0=t0, tt=layer_slot, reverse=1, act=_hip.ACT_TANH_FOLDED,
                                       blob_off=off, blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
            return dict(kind='coupling', cond_cols=col_idx, out_rows=row2, hidden=hidden, ct=ct, tt=tt,
                        side_width=32 * ct + 64 * HT + 64 * tt)
        off, n = self._alloc(n1 + n2 + n3 + n4)
        self.jobs.append(_PackJob(W1, b1, row_h, col_idx, HT, ct, off, rs1, rs1, 0.0))
        self.jobs.append(_PackJob(W2, b2, row2, row_h, 2 * tt, HT, off + n1, rs2, bs2, 1.0))
        self.jobs.append(_PackJob(W2, None, row_h, row2, HT, 2 * tt, off + n1 + n2, transpose=True))      # W2^T
        self.jobs.append(_PackJob(W1, None, col_idx, row_h, ct, HT, off + n1 + n2 + n3, transpose=True))  # W1^T
        self.steps.append(dict(kind=_hip.STEP_COUPLING_AFFINE_BWD, c0=c0, ct=ct, t0=t0, tt=layer_slot, reverse=1,
                               act=_hip.ACT_TANH_FOLDED, blob_off=off, blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
        return dict(kind='coupling', cond_cols=col_idx, out_rows=row2, hidden=hidden, ct=ct, tt=tt,
                    side_width=32 * ct + 64 * HT + 64 * tt)

    def add_linear_bwd(self, sources, fn_fwd, fn_adj, layer_slot: int) -> dict:
        """Backward of one dense linear layer (AffineLU / MatrixExponential) of a log_prob pass on 4 + 4 tiles: the x tiles
        recover the layer's input, v = M_fwd u + b_fwd (fn_fwd(device) -> (W [out, in], b)), the adjoint tiles become
        dL/dv = W^T dL/du (fn_adj(device) -> (W^T as a Linear weight, None)); the factors dL/du (side features [0, 128)) and v
        ([128, 256)) are what sx_wgrad contracts into dL/dW of the matrix log_prob applied, in slot order (see the maps)."""
        self._narrow_only('backward programs')
        self._freeze_input()
        XT = self.x_tiles
        if XT != 4 or self.tiles != 8:
            raise NotImplementedError('dense layers in a backward program need the 4 + 4 tile form')
        col = self.col_of_slot[:32 * XT].copy()
        n_lin = _hip.packed_linear_floats(XT, XT)
        for c0, fn, t0, rev in ((0, fn_fwd, 4, 0), (4, fn_adj, 0, 1)):
            off, n = self._alloc(n_lin)
            self.jobs.append(_DerivedLinearJob(sources, fn, [(col, col, XT, off, XT)]))
            self.steps.append(dict(kind=_hip.STEP_LINEAR_BWD, c0=c0, ct=XT, t0=t0, tt=layer_slot, reverse=rev, act=0,
                                   blob_off=off, blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
        return dict(kind='dense', slot_cols=col, side_width=256)

    def add_affine_const(self, log_scale, shift, reverse: bool, ldj_scale: float) -> None:
        self._freeze_input()
        D, T = self.dim, self.tiles
        off, n = self._alloc(2 * 32 * T)
        gather = np.full(2 * 32 * T, 2 * D, dtype=np.int64)    # index 2D = the appended zero
        for t in range(self.x_tiles):
            for h in range(2):
                for r in range(16):
                    c = self.col_of_slot[32 * t + _kmap(r, h)]
                    if c >= 0:
                        gather[t * 32 + h * 16 + r] = c
                        gather[(T + t) * 32 + h * 16 + r] = D + c
        self.jobs.append(_ConstJob(log_scale, shift, D, gather, off))
        self.steps.append(dict(kind=_hip.STEP_AFFINE_CONST, c0=0, ct=0, t0=0, tt=T, reverse=int(reverse), act=0,
                               blob_off=off, blob_floats=n, ldj_scale=ldj_scale, ldj_const=0.0))

    def add_coupling_time(self, W1, b1, W2, b2, mask: np.ndarray, act: int, reverse: bool, ldj_scale: float, hidden: int,
                          time_col: Optional[int], time_sel: int, time_net) -> None:
        """ContinuousAffineCoupling (coupling.py:184-213) as one SX_STEP_COUPLING_TIME step: the conditioner reads ALL tiles
        (x * mask | latent | the time slot `time_sel`, through column `time_col` of W1 when time is concatenated), the affine map
        with the time embedding acts on the data tiles.  time_net: a net.Time* module with an in-kernel `kind`."""
        self._narrow_only('time-conditioned couplings')
        self._freeze_input()
        if hidden > 32 * self.h_tiles:
            raise NotImplementedError('time-conditioned couplings fuse with hidden layers of up to 128 units')
        D, T, HT, L = self.dim, self.tiles, self.h_tiles, self.latent_dim
        # transformed tiles [0, XT): the tiles that hold data columns, rounded up to what the kernel dispatches on (all, half or
        # a quarter of the tiles; a latent / padding tile inside the range has zero weights: exp(0) x + 0)
        XT = next(c for c in (max(T // 4, 1), max(T // 2, 1), T) if c >= _ceil_div(D, 32))
        if time_sel >= self.time_slots and time_col is not None:
            raise NotImplementedError('time coupling: no slot for this time')
        mask = np.asarray(mask, dtype=np.float64).reshape(-1)
        if mask.size == 1:
            mask = np.full(D, mask[0])
        cond_col, live_col = mask > 0.5, mask <= 0.5
        if D == 1:
            cond_col = np.zeros(1, dtype=bool)                # coupling.py:151-152
        col = self.col_of_slot
        col_idx = np.full(32 * T, -1, dtype=np.int64)
        for p in range(32 * T):
            if p < self.n_slots:
                if col[p] >= 0 and cond_col[col[p]]:
                    col_idx[p] = col[p]
            else:
                li = p - self.n_slots
                if li < L:
                    col_idx[p] = D + li                       # cat([z, latent]) (coupling.py:153-154)
                elif li == L + time_sel and time_col is not None:
                    col_idx[p] = time_col                     # cat([z, t]) (coupling.py:155-156)
        row_idx = np.full(32 * HT, -1, dtype=np.int64)
        row_idx[:hidden] = np.arange(hidden)
        kind = int(time_net.kind)
        K = int(time_net.hidden_dim) if kind == 4 else 0
        if kind == 4 and not (1 <= K <= 64):
            raise NotImplementedError('in-kernel Fourier time nets hold up to 64 features')
        n1 = _hip.packed_linear_floats(HT, T)
        n2 = _hip.packed_linear_floats(2 * XT, HT)
        n3 = 0 if kind == 0 else (XT * K * 128 if kind == 4 else XT * 64)
        off, n = self._alloc(n1 + n2 + n3)
        folded = act == _hip.ACT_CODES['Tanh']
        LOG2E = 1.4426950408889634
        rs1 = bs1 = rs2 = bs2 = None
        fold = 0.0
        if folded:                                            # as add_coupling_affine: tanh's and exp's constants live in the weights
            kk = (-LOG2E) if reverse else LOG2E
            rs1 = np.full(32 * HT, 2.0 * LOG2E)
            bs1 = rs1
            rs2, bs2 = np.empty(64 * XT), np.empty(64 * XT)
            for t in range(XT):
                rs2[64 * t:64 * t + 32], bs2[64 * t:64 * t + 32] = -2.0 * kk, kk
                rs2[64 * t + 32:64 * t + 64], bs2[64 * t + 32:64 * t + 64] = -2.0, 1.0
            fold = 1.0
            act = _hip.ACT_TANH_FOLDED
            ldj_scale = ldj_scale / kk
        self.jobs.append(_PackJob(W1, b1, row_idx, col_idx, HT, T, off, rs1, bs1, 0.0))
        row2 = np.full(64 * XT, -1, dtype=np.int64)
        ls_col = np.full(32 * XT, -1, dtype=np.int64)         # per data slot: its column when transformed (else -1)
        for t in range(XT):
            for i in range(32):
                p = 32 * t + i
                if p < self.n_slots and col[p] >= 0 and live_col[col[p]]:
                    row2[64 * t + i] = col[p]                 # log_scale rows, then shift rows (coupling.py:194 chunk(2))
                    row2[64 * t + 32 + i] = D + col[p]
                    ls_col[p] = col[p]
        col2 = np.full(32 * HT, -1, dtype=np.int64)
        col2[:hidden] = np.arange(hidden)
        self.jobs.append(_PackJob(W2, b2, row2, col2, 2 * XT, HT, off + n1, rs2, bs2, fold))
        if kind != 0:
            self.jobs.append(_TimeConstJob(time_net, kind, K, D, XT, ls_col, off + n1 + n2))
        step = dict(kind=_hip.STEP_COUPLING_TIME, c0=0, ct=T, t0=0, tt=XT, reverse=int(reverse), act=act, blob_off=off,
                    blob_floats=n, ldj_scale=ldj_scale, ldj_const=0.0)
        step['pad_'] = kind | (int(time_sel) << 8) | (K << 16)
        self.steps.append(step)

    def add_pointwise(self, kind: int, param: float, log_slope: float, ldj_coeff: float) -> None:
        """One point-wise flow on the data columns (sigmoid.py:9-56, activations.py:11-101): `kind` = the sx_pointwise kind for the
        direction taken; the step adds ldj_coeff * (that kind's own log-derivative sum) to the accumulator."""
        self._narrow_only('point-wise flows')
        self._freeze_input()
        T = self.tiles
        off, n = self._alloc(32 * T + 1)
        vals = np.zeros(32 * T + 1, dtype=np.float64)
        for t in range(self.x_tiles):
            for h in range(2):
                for r in range(16):
                    if self.col_of_slot[32 * t + _kmap(r, h)] >= 0:
                        vals[t * 32 + h * 16 + r] = 1.0
        vals[32 * T] = log_slope
        self.jobs.append(_ScalarsJob(vals, off))
        self.steps.append(dict(kind=_hip.STEP_POINTWISE, c0=0, ct=0, t0=0, tt=T, reverse=0, act=int(kind), blob_off=off,
                               blob_floats=n, ldj_scale=float(ldj_coeff), ldj_const=float(param)))

    def add_linear(self, sources, fn, ldj_fn=None) -> None:
        """y = W . x + b on the data columns (W, b = fn(device), torch Linear layout [out, in]): one LINEAR_TILE step
        for the whole layer; ldj_fn(device) -> its log-det term (0-dim tensor, signed and scaled) or None for 0."""
        self._narrow_only('dense linear layers')
        self._freeze_input()
        D, T = self.dim, self.tiles
        col = self.col_of_slot
        col_idx = np.full(32 * T, -1, dtype=np.int64)
        col_idx[:self.n_slots] = col                       # input slot -> W column (= logical column)
        XT = self.x_tiles
        # the whole layer in ONE step (all output slabs, one barrier / one weight refill; at most 4 x 4 tiles = 64 KiB,
        # which fits the LDS ring twice): act = number of slabs
        row_idx = col[:32 * XT].copy()                     # output slot keeps its logical column
        n_lin = _hip.packed_linear_floats(XT, T)
        off, n = self._alloc(n_lin + 1)                    # + the log-det term behind the bias
        targets = [(row_idx, col_idx, T, off, XT)]
        self.steps.append(dict(kind=_hip.STEP_LINEAR_TILE, c0=0, ct=T, t0=0, tt=1, reverse=0, act=XT, blob_off=off,
                               blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
        if ldj_fn is None:
            ldj_fn = lambda dev: torch.zeros((), dtype=torch.float32, device=dev)
        self.jobs.append(_DerivedLinearJob(sources, fn, targets, ldj_fn, off + n_lin))

    def add_row_scale_exp(self, diag, reverse: bool, ldj_scale: float, log_time: bool, t_const: float) -> None:
        """state *= exp(+-diag * t_row) (MatrixExponential with a per-row time, affine.py:263)."""
        self._narrow_only('dense linear layers')
        self._freeze_input()
        T = self.tiles
        off, n = self._alloc(32 * T)
        gather = np.full(32 * T, self.dim, dtype=np.int64)      # index dim = appended zero
        for t in range(self.x_tiles):
            for h in range(2):
                for r in range(16):
                    c = self.col_of_slot[32 * t + _kmap(r, h)]
                    if c >= 0:
                        gather[t * 32 + h * 16 + r] = c
        self.jobs.append(_VectorJob(diag, self.dim, gather, off))
        self.steps.append(dict(kind=_hip.STEP_ROW_SCALE_EXP, c0=0, ct=0, t0=0, tt=T, reverse=int(reverse),
                               act=int(log_time), blob_off=off, blob_floats=n, ldj_scale=ldj_scale,
                               ldj_const=float(t_const)))

    def add_mlp(self, linears: Sequence[Tuple], act: int, in_cols_live: Optional[np.ndarray],
                out_rows: np.ndarray, hidden_rows: Optional[np.ndarray] = None, accumulate: bool = False,
                w1_cols: Optional[np.ndarray] = None, w1_latent_base: Optional[int] = None) -> None:
        """Conditioner as its own program: hidden layers then one OUT_TILE step per 32 output columns.

        linears: [(W, b), ...] torch layout; in_cols_live: bool[D] (False -> that x column is zeroed,
        the z = x*mask of coupling.py:61) or None; out_rows[i] = row of the last W written to column i.
        Conditioners wider than the tiles (one hidden layer only): `hidden_rows` = the hidden units this program covers (the
        network's output is a SUM over hidden-unit chunks: W2 tanh(W1 z + b1) = sum_c W2[:, c] tanh(W1[c] z + b1[c])); `accumulate`:
        this chunk adds to what an earlier one wrote (and leaves the output bias to it); `w1_cols[c]` = column of W1 that state
        column c of this (column-subset) program feeds, `w1_latent_base` = W1's first latent column."""
        self._narrow_only('MLP programs')
        self._freeze_input()
        assert len(linears) >= 2, 'conditioner needs at least one hidden layer'
        assert hidden_rows is None or len(linears) == 2
        D, T, HT = self.dim, self.tiles, self.h_tiles
        col = self.col_of_slot
        W0, b0 = linears[0]
        lat0 = D if w1_latent_base is None else w1_latent_base
        col_idx = np.full(32 * T, -1, dtype=np.int64)
        for p in range(32 * T):
            if p < self.n_slots:
                c = col[p]
                if c >= 0 and (in_cols_live is None or in_cols_live[c]):
                    col_idx[p] = c if w1_cols is None else w1_cols[c]
            else:
                li = p - self.n_slots
                if li < self.latent_dim:
                    col_idx[p] = lat0 + li
        hsel = np.arange(W0.shape[0]) if hidden_rows is None else np.asarray(hidden_rows, dtype=np.int64)
        h_prev = len(hsel)
        # (a builder made for a hidden width beyond four tiles carries the CHUNK width of add_coupling_affine in h_tiles: an MLP
        #  program has no chunk steps -- its caller splits single-hidden-layer conditioners itself, deeper ones take the next tier)
        if max([h_prev] + [W.shape[0] for (W, _) in linears[1:-1]]) > 32 * HT:
            raise NotImplementedError(f'MLP programs hold hidden layers of up to {32 * HT} units here')
        row_idx = np.full(32 * HT, -1, dtype=np.int64)
        row_idx[:h_prev] = hsel
        off, n = self._alloc(_hip.packed_linear_floats(HT, T))
        self.jobs.append(_PackJob(W0, b0, row_idx, col_idx, HT, T, off))
        self.steps.append(dict(kind=_hip.STEP_MLP_HIDDEN, c0=0, ct=T, t0=0, tt=0, reverse=0, act=act, blob_off=off,
                               blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
        for (W, b) in linears[1:-1]:
            h_cur = W.shape[0]
            r = np.full(32 * HT, -1, dtype=np.int64)
            r[:h_cur] = np.arange(h_cur)
            c = np.full(32 * HT, -1, dtype=np.int64)
            c[:h_prev] = np.arange(h_prev)
            off, n = self._alloc(_hip.packed_linear_floats(HT, HT))
            self.jobs.append(_PackJob(W, b, r, c, HT, HT, off))
            self.steps.append(dict(kind=_hip.STEP_MLP_HIDDEN2, c0=0, ct=0, t0=0, tt=0, reverse=0, act=act,
                                   blob_off=off, blob_floats=n, ldj_scale=0.0, ldj_const=0.0))
            h_prev = h_cur
        WL, bL = linears[-1]
        out_rows = np.asarray(out_rows, dtype=np.int64)
        self.mlp_out_dim = len(out_rows)
        cL = np.full(32 * HT, -1, dtype=np.int64)
        cL[:h_prev] = hsel if hidden_rows is not None else np.arange(h_prev)
        for u in range(_ceil_div(len(out_rows), 32)):
            r = np.full(32, -1, dtype=np.int64)
            seg = out_rows[32 * u:32 * u + 32]
            r[:len(seg)] = seg
            off, n = self._alloc(_hip.packed_linear_floats(1, HT))
            self.jobs.append(_PackJob(WL, None if accumulate else bL, r, cL, 1, HT, off))
            self.steps.append(dict(kind=_hip.STEP_MLP_OUT_TILE, c0=0, ct=0, t0=u, tt=1, reverse=int(accumulate), act=0,
                                   blob_off=off, blob_floats=n, ldj_scale=0.0, ldj_const=0.0))

    def add_single_linear(self, W, b, out_rows: np.ndarray, transpose: bool = False, k0: int = 0, accumulate: bool = False) -> None:
        """ONE nn.Linear as a program (round 6, VERDICT r5 #7: the layer-wise training path's `F.linear` / `gy @ W` were library
        GEMMs): an SX_STEP_MLP_INPUT step (hidden = the program's input tiles) and one OUT_TILE step per 32 output columns.
        y[:, i] = sum_c W[out_rows[i], k0 + c] x[:, c] + b[out_rows[i]];  transpose: the operand is W^T (y = x W: the input gradient
        of a Linear; b must be None).  Input width <= 128 per program (the builder was made with hidden_width = dim, so h_tiles >=
        tiles): a wider contraction is one program per 128 input columns -- `k0` = the first, the builder's x_cols / x_stride select
        them from the wide rows -- with `accumulate` on every program but the first (the output tiles ADD into y)."""
        self._narrow_only('MLP programs')
        self._freeze_input()
        T, HT = self.tiles, self.h_tiles
        assert T <= HT and self.latent_dim == 0, (T, HT)
        off, n = self._alloc(256)
        self.steps.append(dict(kind=_hip.STEP_MLP_INPUT, c0=0, ct=T, t0=0, tt=0, reverse=0, act=0, blob_off=off, blob_floats=n,
                               ldj_scale=0.0, ldj_const=0.0))
        out_rows = np.asarray(out_rows, dtype=np.int64)
        self.mlp_out_dim = len(out_rows)
        cL = np.full(32 * HT, -1, dtype=np.int64)
        cL[:self.n_slots] = np.where(self.col_of_slot >= 0, self.col_of_slot + k0, -1)
        for u in range(_ceil_div(len(out_rows), 32)):
            r = np.full(32, -1, dtype=np.int64)
            seg = out_rows[32 * u:32 * u + 32]
            r[:len(seg)] = seg
            off, n = self._alloc(_hip.packed_linear_floats(1, HT))
            self.jobs.append(_PackJob(W, b, r, cL, 1, HT, off, transpose=transpose))
            self.steps.append(dict(kind=_hip.STEP_MLP_OUT_TILE, c0=0, ct=0, t0=u, tt=1, reverse=int(accumulate), act=0,
                                   blob_off=off, blob_floats=n, ldj_scale=0.0, ldj_const=0.0))

    # -- finish ------------------------------------------------------------------------------------
    def build(self, device: torch.device) -> CompiledProgram:
        self._freeze_input()
        if len(self.steps) > _hip.SX_MAX_STEPS:
            raise ProgramTooLong(f'fused program has {len(self.steps)} steps (max {_hip.SX_MAX_STEPS})')
        kinds = {s['kind'] for s in self.steps}
        if self.wide_state and not (_hip.STEP_WIDE_HIDDEN in kinds and
                                    kinds <= {_hip.STEP_WIDE_HIDDEN, _hip.STEP_WIDE_AFFINE_TILE, _hip.STEP_AFFINE_CONST}):
            # sx_flow_run's rule for eight data tiles (sx_flow_fused.hip: `x_tiles == 8 && hc && n_hc8 == n_steps`): such a program
            # exists for the couplings of kinds 22 / 23; a flow (or a segment) of that width made of element-wise affines and
            # column shuffles alone -- or an empty one -- has no kernel to run in and keeps its layer-by-layer tier
            raise NotImplementedError('programs on eight state tiles (129 .. 256 columns) carry at least one affine coupling')
        rqs = kinds & {_hip.STEP_RQS_HIDDEN, _hip.STEP_RQS_PHASE}
        dense = kinds & {_hip.STEP_LINEAR_TILE, _hip.STEP_ROW_SCALE_EXP}
        deep = kinds & {_hip.STEP_CPL_HIDDEN, _hip.STEP_CPL_HIDDEN2, _hip.STEP_COUPLING_AFFINE_DEEP}
        pointwise = _hip.STEP_POINTWISE in kinds
        if _hip.STEP_COUPLING_AFFINE_HC in kinds and (rqs or dense or pointwise or deep or _hip.STEP_COUPLING_TIME in kinds or
                                                      kinds & {_hip.STEP_MLP_HIDDEN, _hip.STEP_MLP_HIDDEN2, _hip.STEP_MLP_OUT_TILE}):
            raise NotImplementedError('couplings with chunked hidden layers fuse with affine couplings / element-wise affines only')
        if rqs and dense:
            raise NotImplementedError('spline couplings cannot share a fused program with dense linear layers')
        # spline couplings beside affine couplings / point-wise steps, or both spline types: the MIXED kernel (MODE 14)
        mixed = bool(rqs) and (bool(kinds & {_hip.STEP_COUPLING_AFFINE, _hip.STEP_COUPLING_AFFINE_DEEP}) or pointwise or
                               len({s['act'] & 1 for s in self.steps if s['kind'] == _hip.STEP_RQS_PHASE}) > 1)
        if mixed and any(s['kind'] == _hip.STEP_RQS_PHASE and s['tt'] > 16 for s in self.steps):
            raise NotImplementedError('spline couplings of 17..32 bins fuse in programs of rational-quadratic couplings only')
        if _hip.STEP_COUPLING_TIME in kinds and len(kinds) > 1:
            raise NotImplementedError('time-conditioned couplings form fused programs of their own')
        if pointwise and (dense or (deep and not mixed) or kinds & {_hip.STEP_MLP_HIDDEN, _hip.STEP_MLP_HIDDEN2, _hip.STEP_MLP_OUT_TILE}):
            raise NotImplementedError('point-wise steps share fused programs with couplings and element-wise affines only')
        if deep and kinds & {_hip.STEP_LINEAR_TILE, _hip.STEP_ROW_SCALE_EXP, _hip.STEP_MLP_HIDDEN, _hip.STEP_MLP_HIDDEN2,
                             _hip.STEP_MLP_OUT_TILE, _hip.STEP_COUPLING_AFFINE_BWD}:
            raise NotImplementedError('deep-conditioner couplings only share a fused program with other couplings')
        prog = _hip.sx_program()
        prog.n_steps = len(self.steps)
        prog.dim, prog.latent_dim = self.dim, self.latent_dim
        prog.x_tiles, prog.tiles, prog.h_tiles = self.x_tiles, self.tiles, self.h_tiles
        ident = np.full(self.n_slots, -1, dtype=np.int64)
        ident[:self.dim] = np.arange(self.dim)
        identity = (self.dim % 4 == 0 and np.array_equal(self.in_col, ident)
                    and np.array_equal(self.col_of_slot, ident)) and self.x_cols is None
        prog.identity_cols = int(identity)
        in_col = self.in_col
        if self.x_cols is not None:                   # state column c of this program = column x_cols[c] of rows x_stride wide
            xc = np.asarray(self.x_cols, dtype=np.int64)
            in_col = np.where(self.in_col >= 0, xc[np.clip(self.in_col, 0, len(xc) - 1)], -1)
            prog.pad_ = int(self.x_stride)
        for i, s in enumerate(self.steps):
            st = prog.steps[i]
            for k, v in s.items():
                setattr(st, k, v)
        cp = CompiledProgram(prog, self.blob_floats, self.jobs, None if identity else in_col,
                             None if identity else self.col_of_slot.copy(), device, self.mlp_out_dim)
        # a later hidden chunk of a wide conditioner ADDS into mlp_out (add_mlp(accumulate=True))
        cp.accumulates = any(s['kind'] == _hip.STEP_MLP_OUT_TILE and s.get('reverse') for s in self.steps)
        return cp
