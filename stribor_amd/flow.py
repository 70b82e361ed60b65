"""Plugin surface and flow container (reference: stribor/flow.py:8-152).

``Transform`` / ``ElementwiseTransform`` keep the reference's abstract methods and defaulted
``*_and_log_det_jacobian`` combinators, so third-party transforms written against stribor plug in
unchanged.  ``NormalizingFlow`` keeps the reference's method set, but instead of looping over the
layers in Python (flow.py:99-125) it asks every layer for its step(s) of a fused program and runs the
whole flow — inverse pass, log-det accumulation and the UnitNormal base density (flow.py:127-130) —
in ONE kernel launch.  Layers that cannot be fused fall back to a per-layer loop of HIP kernels.
"""
import ctypes as C
import functools
import os
import threading
from abc import ABCMeta, abstractmethod
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

HALF_LOG_2PI = 0.9189385332046727          # log(sqrt(2 pi)), dist/normal.py:37

from . import _hip, debug
from .fused import CompiledProgram, ProgramBuilder, ProgramCache, ProgramTooLong, StructureTracked

__all__ = ['Transform', 'ElementwiseTransform', 'NormalizingFlow', 'graph_wanted', 'graph_rows']


class _FusedLogProb(torch.autograd.Function):
    """log_prob with a hand-written backward for flows of affine couplings.

    forward: the fused kernel, which also leaves the latent z in HBM.  backward: flows are invertible, so no activation
    was saved -- each step recomputes its conditioner from the state, un-transforms the state and propagates dL/dx.
    Two forms: the layer-major one (`_backward_layer_major`: one launch per coupling, the weight gradients
    contracted over the batch INSIDE the kernel, `sx_wgrad_reduce` adds the per-workgroup partials) where the shape allows
    it (64-column split couplings, hidden <= 64); otherwise ONE launch of the backward program walks the layers in the
    opposite order and leaves the per-row factors of the weight gradients (z, tanh h, dL/dh_pre, dL/dparams; dense layers:
    dL/du, v) in HBM, which `sx_wgrad_layer` / `sx_wgrad` contract over the batch on the matrix pipe.  No library GEMM."""

    SIDE_BYTES = 8 << 30      # scratch for the per-row gradient factors (8 GiB of the 288 GB: 2^20-row blocks at 8 layers)

    @staticmethod
    def forward(ctx, flow, y2, *params):
        prog = flow._fused_program(True, y2.shape[1], 0, y2.device)
        # dense layers: ONE batched fp64 derivation of their matrices per step, with a graph (kept for backward); the forward
        # program's pack jobs (and later the backward program's) pick up the same tensors, detached
        deriver = flow._dense_deriver()
        ctx.dense_graph = deriver.get(y2.device, True, graph=True) if deriver.layers else None
        z, _, logp = prog.run(y2, None, True, False, True)
        ctx.flow = flow
        ctx.save_for_backward(z)
        ctx.need_input_grad = y2.requires_grad
        return logp

    @staticmethod
    def backward(ctx, grad_logp):
        flow = ctx.flow
        (z,) = ctx.saved_tensors
        n, d = z.shape
        bprog, layers = flow._backward_program(d, z.device)
        g = grad_logp.reshape(-1).to(torch.float32).contiguous()
        # The backward is LINEAR in g = dL/dlog_prob, and its GEMM operands (adjoints, dL/dparams) scale with it.  A mean
        # loss over 2^20 rows makes g ~ 1e-6: below fp16's normal range, where the fp16 x 3 split keeps only a few bits.
        # So the pass runs on g * S, S = the power of two that brings max |g| to [1, 2), and every result is multiplied
        # by 1 / S afterwards -- exact (powers of two), all on the device (no host read-back).
        # (the adjoints start at -g z and grow through the layers -- by the gain of the dense layers' matrices in cfg-4-like flows,
        #  orders of magnitude in their worst directions -- so S leaves headroom: max |g| max(1, max |z|) lands in [1/2, 1) of a
        #  range that reaches 65504 upwards and keeps 22 bits down to 0.125, 14 bits at 1e-3)
        if any(info.get('kind') == 'dense' for _, info in layers):
            # (sx_absmax2: one pass per operand, no |.| temporary -- the torch forms took four launches and wrote 134 MB at 2^18 x 128)
            mx = torch.zeros(2, dtype=torch.float32, device=z.device)
            _hip.call('sx_absmax2', g, g.data_ptr(), g.numel(), g.data_ptr(), 0, mx.data_ptr())
            if z.dtype == torch.float32 and z.is_contiguous():
                _hip.call('sx_absmax2', z, z.data_ptr(), z.numel(), z.data_ptr(), 0, mx.data_ptr() + 4)
            else:
                mx[1] = z.abs().max()
            gmax = mx[0] * (2.0 * mx[1].clamp_min(1.0))
        else:
            gmax = g.abs().max()                                     # (pure coupling flows: max |g| -> [1, 2), as before)
        S = torch.where(gmax > 0, torch.exp2(-torch.floor(torch.log2(gmax.clamp_min(1e-38)))), torch.ones_like(gmax))
        g = g * S
        inv_S = 1.0 / S
        ht = 32 * bprog.prog.h_tiles
        width = max(info['side_width'] for _, info in layers)
        dev = z.device
        # the parameter gradients live in ONE zero-filled buffer; sx_wgrad adds into views of it, mapping the
        # kernel's slot order to the parameters' own rows / columns on the way (no per-layer fills or gathers)
        shapes = []
        dense_layers = [f for f, info in layers if info.get('kind') == 'dense']
        for cpl, info in layers:
            if info.get('kind') == 'dense':       # dL/d(matrix, bias) of the map log_prob applied; the parameters get theirs below
                shapes.append((torch.empty(d, d, device='meta'), torch.empty(d, device='meta')))
                continue
            (W1, b1), (W2, b2) = cpl._net().linears()
            shapes.append((W1, b1, W2, b2))
        dense_graph = None
        if dense_layers:
            # the D x D algebra (LU products, inverses, matrix exponential) WITH a graph: the kernels produce dL/d(matrix), autograd
            # of these batched fp64 ops carries it to the parameters; the backward program packs the same tensors (detached)
            dense_graph = ctx.dense_graph
            flow._dense_deriver().get(z.device, False)
        flat = torch.zeros(sum(p_.numel() for ps in shapes for p_ in ps), dtype=torch.float32, device=dev)
        grads, views, off = {}, [], 0
        for ps in shapes:
            vs = []
            for p_ in ps:
                v = flat[off:off + p_.numel()].view(p_.shape)
                off += p_.numel()
                if not p_.is_meta:
                    grads[id(p_)] = v
                vs.append(v)
            views.append(vs)

        def finish():
            flat.mul_(inv_S)
            if gy is not None:
                gy.mul_(inv_S)
            if dense_graph is not None:
                outs, gouts = [], []
                gsum = grad_logp.reshape(-1).to(torch.float32).sum()
                for (f, info), vs in zip(layers, views):
                    if info.get('kind') != 'dense':
                        continue
                    Wm, bm, ldj = dense_graph[id(f)]
                    outs += [Wm, ldj] + ([bm] if bm is not None and bm.requires_grad else [])
                    gouts += [vs[0], gsum.to(ldj.dtype)] + ([vs[1]] if bm is not None and bm.requires_grad else [])
                ps = [p_ for f in dense_layers for p_ in f.parameters() if p_.requires_grad]
                for p_, gp in zip(ps, torch.autograd.grad(outs, ps, gouts, allow_unused=True, retain_graph=True)):
                    if gp is not None:
                        grads[id(p_)] = gp if id(p_) not in grads else grads[id(p_)] + gp
            out = [grads.get(id(p_)) for p_ in flow._grad_params()]
            return (None, gy if ctx.need_input_grad else None, *out)
        gy = torch.empty_like(z) if ctx.need_input_grad else None
        lib = _hip.lib()
        if _FusedLogProb._layer_major_ok(bprog, layers, ht):
            _FusedLogProb._backward_layer_major(bprog, layers, views, z, g, gy if gy is not None else torch.empty_like(z))
            return finish()
        # the per-row factors are 224 floats per row and layer: bound the scratch by walking the batch in blocks.
        # Layout [layer, 32-row group, feature, 32 rows] (coalesced for the kernel's fragment stores and sx_wgrad's loads)
        block = max(32, min(n, _FusedLogProb.SIDE_BYTES // (len(layers) * width * 4)) // 32 * 32)
        side = torch.empty(len(layers), (min(block, n) + 31) // 32, width, 32, dtype=torch.float32, device=dev)
        # the factors are fp16 x 3 GEMM operands of the backward program already: contract them on the matrix pipe too
        wl = _hip.WGRAD_ROW_GROUPS if _hip.get_gemm_precision() == 'exact' else _hip.WGRAD_ROW_GROUPS_F16X3
        with _hip.device_of(z):
            st = _hip.stream()
            for lo in range(0, n, block):
                m = min(block, n - lo)
                ng = (m + 31) // 32
                sd_all = side if ng == side.shape[1] else torch.empty(len(layers), ng, width, 32, dtype=torch.float32, device=dev)
                gblk, _, _ = bprog.run(z[lo:lo + m], None, True, False, False, row_t=g[lo:lo + m], side=sd_all)
                if gy is not None:
                    gy[lo:lo + m] = gblk
                base, ld = sd_all.data_ptr(), width * 32
                for slot in range(len(layers)):
                    info = layers[slot][1]
                    p0 = base + slot * ng * ld * 4
                    if info.get('kind') == 'dense':
                        # dL/dW[out, in] = sum_n dL/du_n v_n^T, dL/dc = sum_n dL/du_n of u = W v + c: features [0, 128) = dL/du,
                        # [128, 256) = v, both in slot order (slot -> logical column through the maps)
                        gWm, gbm = views[slot]
                        sc = _hip.scratch(dev, lib.sx_wgrad_scratch_floats(128, 128, _hip.WGRAD_ROW_GROUPS))
                        _hip.check(lib.sx_wgrad(p0, ld, 128, p0 + 128 * 128, ld, 128, m, wl, gWm.data_ptr(),
                                                gWm.stride(0), gbm.data_ptr(), info['slot_map'].data_ptr(), info['slot_map'].data_ptr(),
                                                sc.data_ptr(), st), 'sx_wgrad')
                        continue
                    H = info['hidden']
                    zc, pc = 32 * info['ct'], 64 * info['tt']                  # features: z | tanh h | dL/dh_pre | dL/dparams
                    gW1, gb1, gW2, gb2 = views[slot]
                    if info['ct'] == 1 and info['tt'] == 1 and ht <= 64:
                        # pruned half masks: both gradients of the layer in one pass over its 28 KB row groups
                        sc = _hip.scratch(dev, lib.sx_wgrad_layer_scratch_floats(1, ht // 32, 1))
                        _hip.check(lib.sx_wgrad_layer(p0, ld, m, 1, ht // 32, 1, H, gW2.data_ptr(), gW2.stride(0), gb2.data_ptr(),
                                                      info['row_map'].data_ptr(), gW1.data_ptr(), gW1.stride(0), gb1.data_ptr(),
                                                      info['col_map'].data_ptr(), sc.data_ptr(), st), 'sx_wgrad_layer')
                        continue
                    sc = _hip.scratch(dev, max(lib.sx_wgrad_scratch_floats(pc, H, _hip.WGRAD_ROW_GROUPS),
                                               lib.sx_wgrad_scratch_floats(H, zc, _hip.WGRAD_ROW_GROUPS)))
                    _hip.check(lib.sx_wgrad(p0 + 128 * (zc + 2 * ht), ld, pc, p0 + 128 * zc, ld, H, m, wl, gW2.data_ptr(),
                                            gW2.stride(0), gb2.data_ptr(), info['row_map'].data_ptr(), None, sc.data_ptr(), st), 'sx_wgrad')
                    _hip.check(lib.sx_wgrad(p0 + 128 * (zc + ht), ld, H, p0, ld, zc, m, wl, gW1.data_ptr(),
                                            gW1.stride(0), gb1.data_ptr(), None, info['col_map'].data_ptr(), sc.data_ptr(), st), 'sx_wgrad')
        return finish()


def _layer_major_ok(bprog, layers, ht) -> bool:
    """The layer-major backward (weight gradients contracted in-kernel, sx_flow_bwd_run) covers 64-column flows whose
    couplings condition one 32-column half on the other in the flow's own column order, hidden <= 64, fp16 x 3."""
    if debug.on('STRIBOR_BWD_FACTORS') or _hip.get_gemm_precision() == 'exact':
        return False
    p = bprog.prog
    return (p.identity_cols == 1 and p.x_tiles == 2 and p.tiles == 4 and ht <= 64
            and all(info['ct'] == 1 and info['tt'] == 1 for _, info in layers))


def _backward_layer_major(bprog, layers, views, z, g, gy) -> None:
    """One launch per coupling (forward order = the backward pass's order): the state (x | dL/dx) streams through one
    fragment-order buffer in place (512 B per row), each launch leaves one partial of its layer's dW2 / db2 / dW1 / db1
    per workgroup, and sx_wgrad_reduce adds them into the parameter-order gradient views."""
    lib = _hip.lib()
    n, dev = z.shape[0], z.device
    L = len(layers)
    G = max(1, int(lib.sx_flow_bwd_max_steps()))      # layers per launch this build's register budget allows (1)
    subs = getattr(bprog, '_layer_programs', None)
    if subs is None or bprog._layer_group != G:       # programs of G consecutive steps over the same blobs
        subs = []
        for k0 in range(0, L, G):
            sp = _hip.sx_program()
            C.memmove(C.byref(sp), C.byref(bprog.prog), C.sizeof(_hip.sx_program))
            sp.n_steps = min(G, L - k0)
            for j in range(sp.n_steps):
                sp.steps[j] = bprog.prog.steps[k0 + j]
            subs.append((k0, sp))
        bprog._layer_programs, bprog._layer_group = subs, G
    n_part, part_floats = C.c_int32(), C.c_int64()
    _hip.check(lib.sx_flow_bwd_partials(C.byref(subs[0][1]), n, C.byref(n_part), C.byref(part_floats)), 'sx_flow_bwd_partials')
    n_part, part_floats = n_part.value, part_floats.value
    H32 = 32 * bprog.prog.h_tiles
    E2 = 64 * H32 + 64
    with _hip.device_of(z):
        blobs = bprog.blobs_for(_hip.GEMM_F16X3)
        frag = torch.empty((n + 31) // 32 * 4096, dtype=torch.float32, device=dev) if len(subs) > 1 else None
        acc = torch.empty(L, n_part * part_floats, dtype=torch.float32, device=dev)
        work, flag, st = _hip.work_counters(dev), _hip.err_flag(dev), _hip.stream()
        for i, (k0, sp) in enumerate(subs):
            first, last = i == 0, i == len(subs) - 1
            rc = lib.sx_flow_bwd_run(C.byref(sp), blobs.data_ptr(), z.data_ptr() if first else None, g.data_ptr(),
                                     None if first else frag.data_ptr(), None if last else frag.data_ptr(),
                                     gy.data_ptr() if last else None, acc[k0].data_ptr(), n, work.data_ptr(), flag, st)
            if rc != 0:
                work.zero_()
            _hip.check(rc, 'sx_flow_bwd_run')
        # the 2 L reductions in ONE launch: the table holds offsets into `acc` and into the flat gradient buffer the views
        # partition (the same in every step), so it is uploaded once per backward program
        out0 = views[0][0].data_ptr()
        table = getattr(bprog, '_reduce_table', None)
        if table is None or table[1] != (n_part, part_floats) or table[0].device != dev:
            jobs = (_hip.sx_reduce_job * (2 * L))()
            for k in range(L):
                info = layers[k][1]
                gW1, gb1, gW2, gb2 = views[k]
                off = lambda v: (v.data_ptr() - out0) // 4
                j2, j1 = jobs[2 * k], jobs[2 * k + 1]
                (j2.part_off, j2.dW_off, j2.db_off, j2.ldw, j2.row_map, j2.col_map, j2.M32, j2.N32, j2.m_valid, j2.n_valid) = (
                    k * n_part * part_floats, off(gW2), off(gb2), gW2.stride(0), info['row_map'].data_ptr(), None, 64, H32, 64, info['hidden'])
                (j1.part_off, j1.dW_off, j1.db_off, j1.ldw, j1.row_map, j1.col_map, j1.M32, j1.N32, j1.m_valid, j1.n_valid) = (
                    k * n_part * part_floats + n_part * E2, off(gW1), off(gb1), gW1.stride(0), None, info['col_map'].data_ptr(), H32, 32,
                    info['hidden'], 32)
            table = bprog._reduce_table = (torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev), (n_part, part_floats))
        _hip.check(lib.sx_wgrad_reduce_batch(acc.data_ptr(), out0, table[0].data_ptr(), 2 * L, n_part, max(E2, H32 * 32 + H32), st),
                   'sx_wgrad_reduce_batch')


_FusedLogProb._layer_major_ok = staticmethod(_layer_major_ok)
_FusedLogProb._backward_layer_major = staticmethod(_backward_layer_major)


def graph_wanted(module, *tensors) -> bool:
    """Does this call have to build an autograd graph?  In the reference every Transform method is differentiable
    (flow.py:35-47, coupling.py:69-95; its own harness differentiates stand-alone f and f.inverse, test/base.py:24-33), so a
    stand-alone layer call under grad mode whose input (latent, t) or parameters require grad runs through the layer's autograd
    ops (HIP kernels with hand-written backwards) instead of the no-graph kernels."""
    if not torch.is_grad_enabled():
        return False
    for t in tensors:
        if torch.is_tensor(t) and t.requires_grad:
            return True
    return module is not None and any(p.requires_grad for p in module.parameters())


def graph_rows(x, latent=None):
    """[..., D] (any storage dtype) -> fp32 rows for the autograd ops: (x2 [N, D], lat2 [N, Ld] | None, lead shape)."""
    x2, lead = flatten_rows(x.to(torch.float32))
    lat2 = None if latent is None else latent.reshape(-1, latent.shape[-1]).to(torch.float32)
    return x2, lat2, lead


def flatten_rows(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Size]:
    """[..., D] -> contiguous [N, D] plus the leading shape (the kernels see rows = samples)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    return (x2 if x2.is_contiguous() else x2.contiguous()), lead


class Transform(StructureTracked, nn.Module, metaclass=ABCMeta):
    """flow.py:8-47.  Subclasses may additionally implement the planner hooks
    ``_plan_hidden_width()`` and ``_plan(builder, reverse, ldj_scale)`` to join fused programs.
    (``StructureTracked``: re-assigning a sub-module, parameter, buffer or public attribute invalidates every cached
    program, see ``fused.ProgramCache``.)"""

    @abstractmethod
    def forward(self, x, **kwargs):
        ...

    @abstractmethod
    def inverse(self, y, **kwargs):
        ...

    @abstractmethod
    def log_det_jacobian(self, x, y, **kwargs):
        ...

    def jacobian(self, x, y, **kwargs):
        raise NotImplementedError

    def forward_and_log_det_jacobian(self, x, **kwargs):
        y = self.forward(x, **kwargs)
        return y, self.log_det_jacobian(x, y, **kwargs)

    def inverse_and_log_det_jacobian(self, y, **kwargs):
        x = self.inverse(y, **kwargs)
        return x, -self.log_det_jacobian(x, y, **kwargs)

    # ---- fused-program hooks (default: not fusable) --------------------------------------------------
    def _plan_hidden_width(self) -> int:
        return 0

    def _plan_first_mask(self, dim: int):
        return None

    def _plan(self, builder: ProgramBuilder, reverse: bool, ldj_scale: float) -> bool:
        return False

    def _plan_guards(self) -> list:
        """Buffers whose VALUES `_plan` bakes into host tables (a changed (data_ptr, _version) re-plans)."""
        return []


class ElementwiseTransform(Transform):
    """flow.py:50-69."""

    @abstractmethod
    def log_diag_jacobian(self, x, y, **kwargs):
        ...

    def forward_and_log_diag_jacobian(self, x, **kwargs):
        y = self.forward(x, **kwargs)
        return y, self.log_diag_jacobian(x, y, **kwargs)

    def inverse_and_log_diag_jacobian(self, y, **kwargs):
        x = self.inverse(y, **kwargs)
        return x, -self.log_diag_jacobian(x, y, **kwargs)


_call_depth = threading.local()


def _errors_leave_the_call(fn):
    """The reference raises a data-dependent failure inside the failing op: the rational-quadratic spline's inverse asserts its
    discriminant synchronously (rational_quadratic_spline.py:175-178,223).  Here kernels flag it and a later poll raises.  A call
    that builds an autograd graph -- a training step -- of a flow that HOLDS such a spline ends with one stream synchronisation + poll
    (`_hip.end_of_flow_call`; mode 'grad', the default), so the exception leaves THIS call; only the outermost of nested public calls
    (log_prob -> inverse_and_log_det_jacobian) pays it.  Measured (tools/experiments/sync_mode_cost.sh): +2 .. 3 % on the cfg-3
    training step; the same rule on flows without that op would cost cfg 2 +13 .. 24 % and cfg 4 +11 .. 14 % of a step for an
    exception the reference does not raise there, so they -- and every inference call -- stay asynchronous."""
    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        x = args[0] if args else kwargs.get('x', kwargs.get('y'))          # the data argument, however the caller spelled it
        wants = torch.is_tensor(x) and x.is_cuda and self._wants_grad(x)
        if wants:
            # a graph-building call: its launches run without the exact redo pass (`_hip.no_redo`: the pass needs the weights packed
            # a second time, and a training step re-packs them every step)
            with _hip.no_redo():
                return wrapped_inner(self, x, *args, **kwargs)
        return fn(self, *args, **kwargs)

    def wrapped_inner(self, x, *args, **kwargs):
        if _hip._sync_mode != 'grad' or not self._holds_asserting_op():
            return fn(self, *args, **kwargs)
        depth = getattr(_call_depth, 'n', 0)
        _call_depth.n = depth + 1
        try:
            out = fn(self, *args, **kwargs)
        finally:
            _call_depth.n = depth
        if depth == 0:
            _hip.end_of_flow_call(x)
        return out
    return wrapped


def _fp32_between_layers(x):
    """bf16 is a STORAGE format here (bf16 in, fp32 arithmetic): a flow that runs layer by layer keeps fp32 between
    its layers -- as the fused kernel does in registers -- and rounds once, on the way out.  -> (tensor to run on, cast back)."""
    if torch.is_tensor(x) and x.dtype == torch.bfloat16:
        return x.to(torch.float32), (lambda r: r.to(torch.bfloat16))
    return x, (lambda r: r)


class NormalizingFlow(Transform):
    """flow.py:72-152.  ``base_dist`` needs ``log_prob / sample / rsample`` (flow.py:129,139,141)."""

    def __init__(self, base_dist, transforms: List[Transform]):
        super().__init__()
        self.base_dist = base_dist
        self.transforms = nn.ModuleList(transforms)
        self._fused = ProgramCache()
        self._too_long = set()           # keys of _fused_program whose plan exceeded SX_MAX_STEPS (see _fused_segments)

    # ---- fused program cache -----------------------------------------------------------------------------
    # Programs bake the transform list, every Permute's index vector and the masks into host tables; the cache entry
    # is therefore tied to the ids of the transforms (ModuleList edits), the permutation buffers' (data_ptr, _version)
    # (load_state_dict after a first call, permute.py:62-82) and the package-wide structure epoch.
    def _fingerprint(self):
        return tuple(map(id, self.transforms))

    def _plan_guards(self) -> list:
        return [g for f in self.transforms for g in f._plan_guards()]

    def _holds_asserting_op(self) -> bool:
        """Does a layer evaluate the reference op that asserts on its data (the rational-quadratic spline; see _errors_leave_the_call)?"""
        def build():
            from .flows.spline import Spline
            return any(isinstance(m, Spline) and m.spline_type == 'quadratic' for f in self.transforms for m in f.modules())
        return self._cached(('asserting-op',), build)

    def _cached(self, key, build):
        return self._fused.get(key, build, self._plan_guards, self._fingerprint())

    def _fused_program(self, reverse: bool, dim: int, latent_dim: int, device, t_kind=None) -> Optional[CompiledProgram]:
        key = (reverse, dim, latent_dim, str(device), t_kind)
        return self._cached(key, lambda: self._build_fused(reverse, dim, latent_dim, device, t_kind))

    def _build_fused(self, reverse, dim, latent_dim, device, t_kind=None) -> Optional[CompiledProgram]:
        order = list(reversed(self.transforms)) if reverse else list(self.transforms)
        try:
            hw = max([f._plan_hidden_width() for f in order] + [1])
            b = ProgramBuilder(dim, latent_dim, hw)
            b.t = t_kind                      # None | float | 'tensor' (MatrixExponential's time)
            b.dense_deriver = self._dense_deriver()      # AffineLU / MatrixExponential matrices: one batched fp64 derivation
            for f in order:
                m = f._plan_first_mask(dim)
                if m is not None:
                    b.choose_layout(m)
                    break
            # Transform.inverse_and_log_det_jacobian negates the forward log-det (flow.py:47)
            scale = -1.0 if reverse else 1.0
            for f in order:
                if not f._plan(b, reverse, scale):
                    return None
            return b.build(device)
        except ProgramTooLong:
            self._too_long.add((reverse, dim, latent_dim, str(device), t_kind))         # plans, but needs segments
            return None
        except NotImplementedError:
            return None

    def _fused_segments(self, reverse: bool, dim: int, latent_dim: int, device, t_kind=None):
        # only a flow whose one-program plan failed on the STEP LIMIT is planned a second time (any other refusal -- layer kinds,
        # mixing rules -- would refuse every segment too); called after _fused_program returned None for the same key
        if (reverse, dim, latent_dim, str(device), t_kind) not in self._too_long:
            return None
        key = ('segments', reverse, dim, latent_dim, str(device), t_kind)
        return self._cached(key, lambda: self._build_segments(reverse, dim, latent_dim, device, t_kind))

    def _build_segments(self, reverse, dim, latent_dim, device, t_kind=None):
        """The flow as a SEQUENCE of fused programs over contiguous runs of layers, for flows whose steps do not fit one program
        (SX_MAX_STEPS = 128 steps travel in the kernel's argument segment: 16 spline couplings of 16 bins are 208, one of 24 bins is
        25).  The state crosses HBM once per segment -- not once or twice per LAYER plus the conditioner's output, as on the
        layer-by-layer path -- and the log-dets of the segments are added on the device.  None when a layer cannot be planned at all."""
        order = list(reversed(self.transforms)) if reverse else list(self.transforms)
        scale = -1.0 if reverse else 1.0
        try:
            hw = max([f._plan_hidden_width() for f in order] + [1])
            segs, i = [], 0
            while i < len(order):
                b = ProgramBuilder(dim, latent_dim, hw)
                b.t = t_kind
                b.dense_deriver = self._dense_deriver()
                for f in order[i:]:
                    m = f._plan_first_mask(dim)
                    if m is not None:
                        b.choose_layout(m)
                        break
                j = i
                while j < len(order):
                    snap = b.snapshot()
                    if not order[j]._plan(b, reverse, scale):
                        return None
                    if len(b.steps) > _hip.SX_MAX_STEPS:
                        b.restore(snap)
                        break
                    j += 1
                if j == i:
                    return None                     # a single layer beyond the program size
                segs.append(b.build(device))
                i = j
            return segs if len(segs) > 1 else None
        except NotImplementedError:
            return None

    def _dense_deriver(self):
        from .flows.linear import DenseDeriver
        return self._cached(('dense-deriver',), lambda: DenseDeriver(list(self.transforms)))

    # ---- training (autograd) -----------------------------------------------------------------------------
    def _grad_params(self):
        return [p for p in self.parameters()]

    def _backward_program(self, dim: int, device):
        """The backward program of log_prob (layers in forward order) + the slot maps for the weight gradients."""
        got = self._cached(('bwd', dim, str(device)), lambda: self._build_backward_program(dim, device))
        if got == 'unsupported':
            raise NotImplementedError('training backward unsupported for this flow')
        return got

    def _build_backward_program(self, dim: int, device):
        try:
            import numpy as np
            from .flows.coupling import Coupling
            from .flows.affine import Affine
            from .flows.permute import _ColumnShuffle
            from .flows.linear import AffineLU, MatrixExponential
            order = list(self.transforms)
            hw = max([f._plan_hidden_width() for f in order] + [1])
            has_dense = any(isinstance(f, (AffineLU, MatrixExponential)) for f in order)
            # dense linear layers (and flows wider than 64 columns) run the 4 + 4 tile form of the backward program
            b = ProgramBuilder(dim, 0, hw, min_x_tiles=4 if (has_dense or dim > 64) else 1)
            deriver = self._dense_deriver()
            # the backward pass starts in the slot layout the forward (log_prob) program ends in; affine-coupling
            # flows never move columns, so that is the layout chosen from the first mask the forward pass sees
            rev = list(reversed(order))
            for f in rev:
                m = f._plan_first_mask(dim)
                if m is not None:
                    b.choose_layout(m)
                    break
            # Permute / Flip only relabel slots: replay the forward program's relabelling to find the layout it
            # ends in (= the layout z is read back in), then undo it layer by layer on the way back
            for f in rev:
                if isinstance(f, _ColumnShuffle):
                    if not f._feature_only():
                        raise NotImplementedError('Flip over a non-feature axis moves rows: layer-wise path')
                    b.add_permutation(f._perm(dim).cpu().numpy(), True)
            b.in_col = None                      # the backward program's input layout is the forward's final one
            b.steps = []
            b.enable_adjoint_tiles()
            layers = []
            for f in order:
                if isinstance(f, _ColumnShuffle):
                    b.add_permutation(f._perm(dim).cpu().numpy(), False)
                    continue
                if isinstance(f, (AffineLU, MatrixExponential)):
                    fwd, adj = f._bwd_matrices(deriver)
                    info = b.add_linear_bwd(list(f.parameters()), fwd, adj, len(layers))
                    info['slot_map'] = torch.from_numpy(np.ascontiguousarray(info['slot_cols'], dtype=np.int32)).to(device)
                    layers.append((f, info))
                    continue
                if not (isinstance(f, Coupling) and isinstance(f.transform, Affine)) or f.set_data:
                    raise NotImplementedError('training backward is implemented for flows of Coupling(Affine), AffineLU / '
                                              'MatrixExponential layers and Permute / Flip')
                net = f._net()
                lin = net.linears()
                if len(lin) != 2 or net.activation_name != 'Tanh':
                    raise NotImplementedError('training backward needs Linear-Tanh-Linear conditioners')
                (W1, b1), (W2, b2) = lin
                info = b.add_coupling_affine_bwd(W1, b1, W2, b2, f.mask_vector(dim), W1.shape[0], len(layers))
                # slot order -> parameter order for sx_wgrad (negative = padding slot)
                info['row_map'] = torch.from_numpy(np.ascontiguousarray(info['out_rows'], dtype=np.int32)).to(device)
                info['col_map'] = torch.from_numpy(np.ascontiguousarray(info['cond_cols'], dtype=np.int32)).to(device)
                layers.append((f, info))
            if not any(info.get('kind', 'coupling') == 'coupling' for _, info in layers):
                raise NotImplementedError('no coupling in the flow: the layer-wise path carries dL/dx')
            return (b.build(device), layers)
        except NotImplementedError:
            return 'unsupported'

    def _can_backward(self, y) -> bool:
        try:
            d = y.shape[-1]
            if self._fused_program(True, d, 0, y.device) is None:
                return False
            self._backward_program(d, y.device)
            return True
        except NotImplementedError:
            return False

    # ---- training, layer by layer: couplings (affine / quadratic spline), element-wise Affine / Spline, Permute / Flip ----
    def _layerwise_autograd_ok(self) -> bool:
        from .flows.permute import _ColumnShuffle
        ok = [isinstance(f, _ColumnShuffle) or (hasattr(f, '_autograd_supported') and f._autograd_supported())
              for f in self.transforms]
        return all(ok)         # (a flow of nothing but Permute / Flip still carries dL/dx: index_select ops + the base density)

    def _layerwise_autograd(self, x, latent=None, reverse: bool = True, t=None):
        """The flow layer by layer WITH an autograd graph: each layer's transform (and its backward) is a HIP kernel behind
        an autograd op, conditioners and column shuffles are torch ops on the device.  fp32 state.
        -> (rows [N, D], accumulated log-det [N]); reverse = the direction log_prob evaluates."""
        from .flows.permute import _ColumnShuffle
        x2, lead = flatten_rows(x.to(torch.float32))
        lat2 = None if latent is None else latent.reshape(-1, latent.shape[-1]).to(torch.float32)
        from .flows.linear import MatrixExponential, derive_dense_batched
        timed = [] if t is None else [f for f in self.transforms if isinstance(f, MatrixExponential)]
        dense = derive_dense_batched([f for f in self.transforms if f not in timed], x2.device, reverse)   # batched fp64 ops
        cur, total = x2, None
        pre = self._spline_forward_once(x2) if (reverse and lat2 is None and t is None) else None
        for li, f in enumerate(reversed(self.transforms) if reverse else self.transforms):
            if pre is not None:
                # (the whole flow already ran as one launch: each layer's op takes its saved tensors from it and launches nothing)
                cur, ldj = f._autograd_inverse(cur, None, pre=pre[li])
                total = ldj if total is None else total + ldj
                continue
            if isinstance(f, _ColumnShuffle) and not f._feature_only():
                # Flip over other axes moves whole rows (permute.py:35,38): torch.flip on the unflattened state, which is
                # differentiable and its own inverse -- NOT the column reversal f._perm describes
                cur = torch.flip(cur.reshape(*lead, cur.shape[1]), f.dims).reshape(-1, cur.shape[1])
                continue
            if isinstance(f, _ColumnShuffle):
                def build_idx(f=f, d=cur.shape[1], dev=cur.device):
                    perm = f._perm(d).to(dev).long()
                    if not reverse:
                        return perm                                           # permute.py:71 (forward: x[..., perm])
                    inv = torch.empty_like(perm)
                    inv[perm] = torch.arange(perm.numel(), device=perm.device)
                    return inv
                idx = self._cached(('perm_idx', reverse, cur.shape[1], str(cur.device), id(f)), build_idx)
                cur = cur.index_select(1, idx)                                # permute.py:71,75
                continue
            if any(f is g for g in timed):                                    # an explicit time (number or per-row tensor)
                cur, ldj = f._autograd_time(cur, t, reverse)
                total = ldj if total is None else total + ldj
                continue
            if getattr(f, 'set_data', False):                                 # the mask runs over the set axis (coupling.py:48-53)
                if len(lead) < 1:
                    raise ValueError('set_data=True needs inputs of shape (..., N, dim)')
                cur, ldj = f._autograd_set(cur, lat2, lead[-1], reverse)
                total = ldj if total is None else total + ldj
                continue
            step = f._autograd_inverse if reverse else f._autograd_forward
            cur, ldj = step(cur, lat2, dense[id(f)]) if id(f) in dense else step(cur, lat2)
            total = ldj if total is None else total + ldj
        if total is None:
            total = torch.zeros(cur.shape[0], dtype=torch.float32, device=cur.device)
        return cur, total, lead

    def _spline_forward_once(self, x2):
        """Training forward of a flow of spline couplings as ONE launch: the whole-flow fused program (the one inference uses) with
        side outputs -- tanh h of every layer's conditioner and the state every layer but the first received -- instead of one
        program launch per layer.  -> per layer, in evaluation order, (output rows, log-det share, tanh h), or None when the flow
        is not of that kind.  (The log-det is additive: the last layer carries the flow's whole sum, the others zeros -- every
        layer's op still receives dL/dlog-det, which is all its backward needs.)"""
        from .flows.coupling import Coupling
        if debug.on('STRIBOR_SPLINE_FORWARD_PER_LAYER') or x2.dtype != torch.float32 or x2.shape[0] == 0:
            return None
        fs = list(reversed(self.transforms))
        n, d = x2.shape
        if not fs or not all(isinstance(f, Coupling) and f._slab_l1_ok(d) for f in fs):
            return None
        Hs = {f.transform.latent_net.linears()[0][0].shape[0] for f in fs}
        if len(Hs) != 1:
            return None
        prog = self._fused_program(True, d, 0, x2.device)
        if prog is None or not prog.prog.identity_cols or prog.prog.pad_:
            return None
        steps = [prog.prog.steps[i] for i in range(prog.prog.n_steps)]
        hidden = [s_ for s_ in steps if s_.kind == _hip.STEP_RQS_HIDDEN]
        if len(hidden) != len(fs) or [(s_.pad_ >> 8) & 0xff for s_ in hidden] != list(range(len(fs))) \
                or any(s_.kind not in (_hip.STEP_RQS_HIDDEN, _hip.STEP_RQS_PHASE) for s_ in steps):
            return None
        H, L = Hs.pop(), len(fs)
        hbuf = torch.empty(L * n, H, dtype=torch.float32, device=x2.device)
        side = torch.empty(max(L - 1, 1) * n, d, dtype=torch.float32, device=x2.device)
        with torch.no_grad():
            y, ldj, _ = prog.run(x2.contiguous(), None, True, True, False, mlp_out=hbuf, side=side if L > 1 else None)
        zeros = torch.zeros(max(L - 1, 1), n, dtype=torch.float32, device=x2.device)      # (one tensor OBJECT per op output)
        return [(side[l * n:(l + 1) * n] if l < L - 1 else y, ldj if l == L - 1 else zeros[l], hbuf[l * n:(l + 1) * n]) for l in range(L)]

    def _log_prob_layerwise_autograd(self, y, latent=None, t=None):
        """log_prob with a graph for spline-coupling / conditional / mixed flows (the layer-wise training path)."""
        cur, total, lead = self._layerwise_autograd(y, latent, True, t)
        d = cur.shape[1]
        lp = -0.5 * (cur * cur).sum(-1) - d * HALF_LOG_2PI + total            # dist/normal.py:37,52-54
        return lp.reshape(*lead, 1)

    def _transform_with_graph(self, x, latent, reverse: bool, what: str, kwargs):
        """forward / inverse (+ log-det) as differentiable tensors when a graph is wanted, or None.  The reference's
        methods are all differentiable (VI losses built from rsample + log-det); here the layer-wise ops provide that for
        every layer with a backward in the wanted direction (forward: affine couplings, element-wise Affine, the point-wise
        flows, AffineLU / MatrixExponential with the default t, Permute / Flip; inverse: also spline couplings).  Flows
        outside that set evaluate without a graph and say so."""
        if not self._wants_grad(x):
            return None
        if kwargs and set(kwargs) != {'t'}:
            self._warn_detached(what, x)
            return None
        from .flows.permute import _ColumnShuffle
        ok = self._layerwise_autograd_ok() and (reverse or all(isinstance(f, _ColumnShuffle) or hasattr(f, '_autograd_forward')
                                                               for f in self.transforms))
        if ok:
            try:
                cur, total, lead = self._layerwise_autograd(x, latent, reverse, kwargs.get('t'))
                return cur.reshape(*lead, cur.shape[1]).to(x.dtype if x.dtype != torch.bfloat16 else torch.float32), total.reshape(*lead, 1)
            except NotImplementedError:
                pass
        self._warn_detached(what, x)
        return None

    def _wants_grad(self, y) -> bool:
        return torch.is_grad_enabled() and (y.requires_grad or any(p.requires_grad for p in self.parameters()))

    def _warn_detached(self, what: str, x) -> None:
        """A graph was wanted (grad enabled and the input or a parameter requires grad) but this flow has a layer without a
        backward in the wanted direction (or extra keyword arguments): the result is computed by the fused kernels WITHOUT
        a graph.  In the reference every call is differentiable, so a loss built from such a result would train with the
        flow silently frozen: say so -- always when the input itself requires grad, once per method otherwise."""
        import warnings
        seen = self.__dict__.setdefault('_warned_detached', set())
        if x.requires_grad or what not in seen:
            seen.add(what)
            warnings.warn(f'stribor_amd: NormalizingFlow.{what} returns tensors WITHOUT an autograd graph for this flow '
                          f'(a layer has no backward in that direction, or keyword arguments were given); wrap inference in '
                          f'torch.no_grad() to silence this -- see INTEGRATION.md "Differentiable surface"',
                          RuntimeWarning, stacklevel=4)

    def _run(self, x, reverse: bool, latent, want_y, want_ldj, want_logp, sum_out=None, **kwargs):
        """Returns (y, ldj[..., 1], logp[..., 1]) (None where not requested) via the fused kernel, or None
        when the flow cannot be fused."""
        _hip.require_device(x, 'x')
        t = kwargs.pop('t', None)
        if kwargs:          # unknown keyword: let the per-layer path hand it to every transform
            return None
        t_kind = 'tensor' if torch.is_tensor(t) else (None if t is None else float(t))
        x2, lead = flatten_rows(x)
        lat2 = None
        if latent is not None:
            lat2 = latent.reshape(-1, latent.shape[-1])
        ld = 0 if lat2 is None else lat2.shape[1]
        row_t = t.reshape(-1) if t_kind == 'tensor' else None
        shp = lambda t, d: None if t is None else t.reshape(*lead, d)
        prog = self._fused_program(reverse, x2.shape[1], ld, x.device, t_kind)
        if prog is not None:
            y, ldj, logp = prog.run(x2, lat2, want_y, want_ldj, want_logp, sum_out, row_t=row_t)
            return shp(y, x2.shape[1]), shp(ldj, 1), shp(logp, 1)
        segs = self._fused_segments(reverse, x2.shape[1], ld, x.device, t_kind)
        if segs is None:
            return None
        # several launches: fp32 state between them (a bf16 batch is rounded once, on the way out, like the one-launch path whose
        # state never leaves registers), log-dets added on the device; sum_out collects every segment's block sums
        cur = x2.to(torch.float32) if x2.dtype == torch.bfloat16 else x2
        need_l = want_ldj or want_logp
        acc = None
        for i, p in enumerate(segs):
            last = i == len(segs) - 1
            y, ldj, logp = p.run(cur, lat2, want_y or not last, need_l and not (last and want_logp and not want_ldj),
                                 want_logp and last, sum_out, row_t=row_t)
            if not last or want_y:
                cur = y
            if ldj is not None:
                acc = ldj if acc is None else acc + ldj
            if last and logp is not None and not want_ldj:
                logp = logp if acc is None else logp + acc
            elif last and logp is not None:
                # (both asked for: the last segment's logp misses the earlier segments' log-dets, its ldj is part of `acc`)
                logp = logp + (acc - ldj)
        y = cur.to(x2.dtype) if want_y else None
        return shp(y, x2.shape[1]), shp(acc if want_ldj else None, 1), shp(logp if want_logp else None, 1)

    # ---- reference method set -----------------------------------------------------------------------------
    @_errors_leave_the_call
    def forward(self, x, latent=None, **kwargs):
        g = self._transform_with_graph(x, latent, False, 'forward', kwargs)
        if g is not None:
            return g[0]
        r = self._run(x, False, latent, True, False, False, **kwargs)
        if r is not None:
            return r[0]
        kw = dict(kwargs) if latent is None else dict(kwargs, latent=latent)
        x, back = _fp32_between_layers(x)
        for f in self.transforms:                                   # flow.py:99-102
            x = f(x, **kw)
        return back(x)

    @_errors_leave_the_call
    def inverse(self, y, latent=None, **kwargs):
        g = self._transform_with_graph(y, latent, True, 'inverse', kwargs)
        if g is not None:
            return g[0]
        r = self._run(y, True, latent, True, False, False, **kwargs)
        if r is not None:
            return r[0]
        kw = dict(kwargs) if latent is None else dict(kwargs, latent=latent)
        y, back = _fp32_between_layers(y)
        for f in reversed(self.transforms):                         # flow.py:104-107
            y = f.inverse(y, **kw)
        return back(y)

    @_errors_leave_the_call
    def forward_and_log_det_jacobian(self, x, latent=None, **kwargs):
        g = self._transform_with_graph(x, latent, False, 'forward_and_log_det_jacobian', kwargs)
        if g is not None:
            return g
        r = self._run(x, False, latent, True, True, False, **kwargs)
        if r is not None:
            return r[0], r[1]
        kw = dict(kwargs) if latent is None else dict(kwargs, latent=latent)
        acc = 0
        x, back = _fp32_between_layers(x)
        for f in self.transforms:                                   # flow.py:109-116
            x, ldj = f.forward_and_log_det_jacobian(x, **kw)
            acc = acc + ldj
        return back(x), acc

    @_errors_leave_the_call
    def inverse_and_log_det_jacobian(self, y, latent=None, **kwargs):
        g = self._transform_with_graph(y, latent, True, 'inverse_and_log_det_jacobian', kwargs)
        if g is not None:
            return g
        r = self._run(y, True, latent, True, True, False, **kwargs)
        if r is not None:
            return r[0], r[1]
        kw = dict(kwargs) if latent is None else dict(kwargs, latent=latent)
        acc = 0
        y, back = _fp32_between_layers(y)
        for f in reversed(self.transforms):                         # flow.py:118-125
            y, ldj = f.inverse_and_log_det_jacobian(y, **kw)
            acc = acc + ldj
        return back(y), acc

    @_errors_leave_the_call
    def log_prob(self, y, latent=None, **kwargs):
        """[..., D] -> [..., 1]   (flow.py:127-130)."""
        from .dist.normal import UnitNormal
        if isinstance(self.base_dist, UnitNormal):
            _hip.require_device(y, 'y')
            if self._wants_grad(y) and latent is None and not kwargs and self._can_backward(y):
                # differentiable path: fp32 state, hand-written backward (flows of affine couplings); every other
                # flow evaluates without a graph, as before
                y2, lead = flatten_rows(y.to(torch.float32))
                return _FusedLogProb.apply(self, y2, *self._grad_params()).reshape(*lead, 1)
            if self._wants_grad(y) and (not kwargs or set(kwargs) == {'t'}) and self._layerwise_autograd_ok():
                return self._log_prob_layerwise_autograd(y, latent, kwargs.get('t'))
            if self._wants_grad(y) and not getattr(self, '_warned_no_graph', False):
                import warnings
                self._warned_no_graph = True
                warnings.warn('stribor_amd: log_prob of this flow has no backward in this build (a layer outside the '
                              'differentiable set, or extra keyword arguments): it is evaluated WITHOUT an autograd graph. '
                              'Wrap inference in torch.no_grad() to silence this.', RuntimeWarning, stacklevel=2)
            r = self._run(y, True, latent, False, False, True, **kwargs)
            if r is not None:
                return r[2]
        # (bf16 storage: the latent stays fp32 up to the base density, as in the fused kernel, where the state never leaves registers)
        x, acc = self.inverse_and_log_det_jacobian(y.to(torch.float32) if y.dtype == torch.bfloat16 else y, latent=latent, **kwargs)
        if isinstance(self.base_dist, UnitNormal):
            x2, lead = flatten_rows(x)
            ldj = acc.reshape(-1).to(torch.float32).contiguous() if torch.is_tensor(acc) else None
            out = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
            _hip.call('sx_unit_normal_logprob', x2, x2.data_ptr(), _hip.ptr(ldj), out.data_ptr(), x2.shape[0],
                                                   x2.shape[1], _hip.dtype_code(x2))
            return out.reshape(*lead, 1)
        return self.base_dist.log_prob(x).unsqueeze(-1) + acc      # foreign base density: torch ops

    def log_prob_sum(self, y, out: Optional[torch.Tensor] = None, latent=None) -> torch.Tensor:
        """Sum over the batch of log_prob as ONE fp64 scalar accumulated on the device (the operand of the multi-GPU
        all-reduce).  The per-sample log_prob [N] is still written (fp32, 4 B per row: `CompiledProgram.run` allocates it and the
        kernel's epilogue stores it beside the fp64 block sums) -- the bench's algorithmic bytes count it."""
        if out is None:
            out = torch.zeros(1, dtype=torch.float64, device=y.device)
        from .dist.normal import UnitNormal
        if isinstance(self.base_dist, UnitNormal):
            r = self._run(y, True, latent, False, False, True, sum_out=out)
            if r is not None:
                return out
        lp = self.log_prob(y, latent=latent).reshape(-1).contiguous()
        _hip.call('sx_sum_f64', lp, lp.data_ptr(), lp.numel(), out.data_ptr())
        return out

    def sample(self, num_samples: Union[Tuple[int], int], *, rsample: bool = False, **kwargs):
        if isinstance(num_samples, int):
            num_samples = (num_samples,)
        x = self.base_dist.rsample(num_samples) if rsample else self.base_dist.sample(num_samples)
        return self.forward(x, **kwargs)                            # flow.py:132-143

    def rsample(self, num_samples, **kwargs):
        return self.sample(num_samples, **kwargs)                   # flow.py:145-146

    def log_det_jacobian(self, x, y=None, **kwargs):
        _, ldj = self.forward_and_log_det_jacobian(x, **kwargs)     # flow.py:148-152
        return ldj


class NeuralFlow(nn.Module):
    """flow.py:155-184: transforms that are the identity at t = 0; ``forward(x, t, t0)`` first inverts at t0."""

    def __init__(self, transforms: List[Transform]) -> None:
        super().__init__()
        self.transforms = nn.ModuleList(transforms)

    def _fused(self, dim: int, latent_dim: int, with_t0: bool, device):
        """The whole flow -- the inverse pass at t0 (when given) and the forward pass at t -- as ONE program of
        SX_STEP_COUPLING_TIME steps, or None when a transform is not a fusable ContinuousAffineCoupling."""
        from .flows.coupling import ContinuousAffineCoupling
        cache = self.__dict__.setdefault('_programs', ProgramCache())
        key = (dim, latent_dim, bool(with_t0), str(device))

        def build():
            fs = list(self.transforms)
            if not fs or not all(isinstance(f, ContinuousAffineCoupling) and f._fusable() for f in fs):
                return None
            try:
                b = ProgramBuilder(dim, latent_dim, max(f.latent_net.hidden_width for f in fs), time_slots=2 if with_t0 else 1)
                if with_t0:
                    for f in reversed(fs):                                  # flow.py:178-180: x = f.inverse(x, t=t0)
                        if not f._plan_time(b, True, 0.0, 1):
                            return None
                for f in fs:                                                # flow.py:181-182: x = f(x, t=t)
                    if not f._plan_time(b, False, 0.0, 0):
                        return None
                return b.build(device)
            except NotImplementedError:
                return None
        return cache.get(key, build, fingerprint=tuple(map(id, self.transforms)))

    def forward(self, x, t, t0=None, **kwargs):
        latent = kwargs.get('latent')
        plain = set(kwargs) <= {'latent'} and torch.is_tensor(t) and (t0 is None or torch.is_tensor(t0))
        if plain and x.is_cuda and not graph_wanted(self, x, t, t0, latent) and x.numel() > 0:
            x2, lead = flatten_rows(x)
            n, d = x2.shape
            lat2 = None if latent is None else latent.reshape(n, -1).to(torch.float32).contiguous()
            prog = self._fused(d, 0 if lat2 is None else lat2.shape[1], t0 is not None, x.device)
            if prog is not None:
                def rows(v):
                    v = v.to(device=x.device, dtype=torch.float32)
                    return (v.expand(*lead, 1) if v.numel() != n else v).reshape(-1).contiguous()
                y, _, _ = prog.run(x2, lat2, True, False, False, row_t=rows(t), side=None if t0 is None else rows(t0))
                return y.reshape(*lead, d)
        if t0 is not None:
            for transform in reversed(self.transforms):
                x = transform.inverse(x, t=t0, **kwargs)
        for transform in self.transforms:
            x = transform(x, t=t, **kwargs)
        return x
