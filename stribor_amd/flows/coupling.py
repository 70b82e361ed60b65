"""Coupling layer (reference: stribor/flows/coupling.py:10-95).

``Coupling(transform, mask, set_data=False)`` wraps an elementwise transform whose ``latent_net`` maps the
masked input (optionally concatenated with ``latent``) to the transform's parameters.

Three execution tiers, fastest first (the first that applies is used; all give the reference's values):
  1. fused: ``latent_net`` is a ``stribor_amd.net.MLP`` within the fused kernel's tiles (D + latent <= 128 columns,
     hidden <= 128, n_bins <= 16) -- the layer (or the whole flow it sits in) is ONE launch;
  2. MLP program + element-wise kernel: same conditioner class, shapes the one-step program cannot hold;
  3. generic: ANY ``nn.Module`` conditioner (the reference accepts one: flows/affine.py:59-67, flows/spline.py:76-87),
     any width, any n_bins, ``set_data=True`` -- the conditioner is simply called on ``cat[x * mask, latent]``
     (coupling.py:61-65; a wide ``net.MLP`` runs its Linear layers as library GEMMs) and the transform, the blend and
     the masked log-det are one pass of ``sx_affine_coupling`` / ``sx_rqs_coupling`` / ``sx_cubic_coupling``.

* ``Coupling(Affine(latent_net=MLP))`` is ONE launch of the fused MFMA kernel per call: masked GEMM-1,
  tanh, pruned GEMM-2, affine, blend and per-sample log-det all stay in registers; the conditioner runs
  once even for ``*_and_log_det_jacobian`` (the reference runs it twice, quirk Q2).
* ``Coupling(Spline(quadratic, latent_net=MLP))`` runs the conditioner with the MFMA kernel (pruned to the
  transformed columns) and the spline in ``sx_rqs_coupling`` (parameters staged through LDS per wavefront).
"""
import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _hip, debug
from ..flow import Transform, flatten_rows, graph_rows, graph_wanted
from ..fused import ProgramBuilder, ProgramCache, _STRUCT_EPOCH
from ..net.mlp import MLP, _chunk_mlp_program
from ..util.mask import get_mask
from .affine import Affine

__all__ = ['Coupling', 'ContinuousAffineCoupling']


def _ceil32(n: int) -> int:
    return -(-n // 32)


class Coupling(Transform):
    def __init__(self, transform, mask: str, set_data: bool = False, **kwargs):
        super().__init__()
        self.transform = transform
        self.mask_name = mask
        self.mask_func = get_mask(mask)                       # raises NotImplementedError like mask.py:20
        self.set_data = bool(set_data)                        # coupling.py:49-51: the mask runs over the set axis
        self._masks = {}
        self._masks_epoch = -1
        self._mask_tensors = {}                               # (length, device, dtype) -> mask on the device (wrapped tier)
        self._mask_tensors_epoch = -1
        self._programs = ProgramCache()

    # ---- mask: built once per width (the reference rebuilds it from numpy every call, quirk Q4) --------
    def mask_vector(self, dim: int) -> np.ndarray:
        if self._masks_epoch != _STRUCT_EPOCH[0]:          # mask_func / mask_name may have been re-assigned
            self._masks, self._masks_epoch = {}, _STRUCT_EPOCH[0]
        if dim not in self._masks:
            m = self.mask_func(dim).numpy().astype(np.float64).reshape(-1)
            self._masks[dim] = np.full(dim, m[0]) if m.size == 1 else m
        return self._masks[dim]

    def _get_mask(self, x: torch.Tensor) -> torch.Tensor:
        def vec(n):                          # one host-to-device copy per (length, device, dtype), not per call
            key = (n, x.device, x.dtype)
            if self._mask_tensors_epoch != _STRUCT_EPOCH[0]:
                self._mask_tensors, self._mask_tensors_epoch = {}, _STRUCT_EPOCH[0]
            m = self._mask_tensors.get(key)
            if m is None:
                # (built outside inference mode: an inference tensor cached by a first no-grad call could not be saved for the
                #  backward of a later training call -- "Inference tensors cannot be saved for backward")
                with torch.inference_mode(False):
                    m = self._mask_tensors[key] = torch.from_numpy(self.mask_vector(n)).to(device=x.device, dtype=x.dtype)
            return m
        if self.set_data:                                                              # coupling.py:49-51
            *rest, N, D = x.shape
            return vec(N).unsqueeze(-1).expand(*rest, N, D)
        return vec(x.shape[-1]).expand_as(x)                                           # coupling.py:52-53

    def _net(self) -> MLP:
        """The conditioner, when it is this package's MLP (what the fused tiers consume weight by weight)."""
        net = getattr(self.transform, 'latent_net', None)
        if not isinstance(net, MLP):
            raise NotImplementedError('this coupling\'s conditioner is not a stribor_amd.net.MLP: generic tier')
        return net

    def _has_mlp(self) -> bool:
        net = getattr(self.transform, 'latent_net', None)
        return isinstance(net, MLP) and net.fusable()

    # ---- affine: one fused single-step program per (direction, width, latent width, device) -----------
    def _affine_program(self, reverse: bool, ldj_scale: float, dim: int, latent_dim: int, device):
        key = ('affine', reverse, ldj_scale, dim, latent_dim, str(device))

        def build():
            try:
                b = ProgramBuilder(dim, latent_dim, self._net().hidden_width)
                return b.build(device) if self._plan(b, reverse, ldj_scale) else None
            except NotImplementedError:
                return None
        prog = self._programs.get(key, build)
        if prog is None:
            raise NotImplementedError('this coupling cannot run on the fused kernel')
        return prog

    def _run(self, x, latent, reverse, want_y, want_ldj, ldj_scale=1.0):
        _hip.require_device(x, 'x')
        x2, lead = flatten_rows(x)
        d = x2.shape[1]
        lat2 = None if latent is None else latent.reshape(-1, latent.shape[-1])
        ld = 0 if lat2 is None else lat2.shape[1]
        from .spline import Spline
        assert isinstance(self.transform, (Affine, Spline))            # (anything else: the `_wrapped` tier, see forward)
        if self.set_data:
            if x.dim() < 2:
                raise ValueError('set_data=True needs inputs of shape (..., N, dim)')
            y, ldj = self._run_set(x2, lat2, x.shape[-2], reverse, want_ldj, ldj_scale)
        elif not self._has_mlp():
            y, ldj = self._run_generic(x2, lat2, reverse, want_ldj, ldj_scale)
        elif isinstance(self.transform, Affine):
            try:
                prog = self._affine_program(reverse, ldj_scale, d, ld, x.device)
            except NotImplementedError:
                prog = None            # e.g. a conditioner with several hidden layers: MLP kernel + element-wise kernel
            if prog is not None:
                y, ldj, _ = prog.run(x2, lat2, want_y, want_ldj, False)
            else:
                try:
                    y, ldj = self._run_affine_unfused(x2, lat2, reverse, want_ldj, ldj_scale)
                except NotImplementedError:        # wider than the MLP program's tiles
                    y, ldj = self._run_generic(x2, lat2, reverse, want_ldj, ldj_scale)
        else:
            try:
                y, ldj = self._run_spline(x2, lat2, reverse, want_ldj, ldj_scale)
            except NotImplementedError:
                y, ldj = self._run_generic(x2, lat2, reverse, want_ldj, ldj_scale)
        return (None if y is None else y.reshape(*lead, d)), (None if ldj is None else ldj.reshape(*lead, 1))

    # ---- generic tier: any conditioner module, any width --------------------------------------------------------
    def _transform_rows(self, x2, params_full, live, reverse, want_ldj, ldj_scale):
        """The element-wise transform of the `live` columns of [N, D] rows given the conditioner's FULL-width output
        ([N, 2D] affine: log_scale | shift, affine.py:66; [N, D * P] splines, spline.py:82-86) -> (y, ldj | None)."""
        from .affine import run_affine_kernel
        from .spline import Spline, run_cubic_kernel, run_rqs_kernel
        sp = self.transform
        n, d = x2.shape
        dev = x2.device
        if len(live) == 0:
            return x2.clone(), (torch.zeros(n, dtype=torch.float32, device=dev) if want_ldj else None)
        contiguous = np.array_equal(live, np.arange(live[0], live[0] + len(live)))
        live_idx = None if contiguous else torch.from_numpy(live.astype(np.int32)).to(dev)
        p = params_full.to(torch.float32)
        if isinstance(sp, Affine):
            if p.shape[-1] != 2 * d:
                raise ValueError(f'latent_net returned {p.shape[-1]} values per row, expected {2 * d}')
            if len(live) == d:
                params = p.contiguous()
            else:                                       # (log_scale | shift) of the transformed columns, adjacent
                cols = torch.from_numpy(np.concatenate([live, d + live]).astype(np.int64)).to(dev)
                params = p.index_select(1, cols)
            return run_affine_kernel(x2, params, params.stride(0), live_idx, int(live[0]), len(live), reverse, True,
                                     want_ldj, ldj_scale)
        assert isinstance(sp, Spline)
        P = sp.params_per_element
        if p.shape[-1] != d * P:
            raise ValueError(f'latent_net returned {p.shape[-1]} values per row, expected {d * P}')
        if contiguous:
            p = p.contiguous()
            params, stride = p[:, int(live[0]) * P:], p.stride(0)      # the live block of every row, in place
        else:
            cols = torch.from_numpy((live[:, None] * P + np.arange(P)[None, :]).reshape(-1).astype(np.int64)).to(dev)
            params = p.index_select(1, cols)
            stride = params.stride(0)
        if sp.spline_type == 'cubic':
            # a coupling's inverse log-det is MINUS the FORWARD log-det at the inverted point (flow.py:42-47): reverse = 2
            y, ldj, _ = run_cubic_kernel(x2, params, stride, live_idx, int(live[0]), len(live), sp.n_bins, sp.lower, sp.upper,
                                         2 if (reverse and want_ldj) else reverse, want_ldj, False, ldj_scale)
        else:
            y, ldj, _ = run_rqs_kernel(x2, params, stride, live_idx, int(live[0]), len(live), sp.n_bins, sp.lower, sp.upper,
                                       sp.lower, sp.upper, reverse, want_ldj, False, ldj_scale)
        return y, ldj

    def _run_generic(self, x2, lat2, reverse, want_ldj, ldj_scale):
        n, d = x2.shape
        m = self.mask_vector(d)
        live = np.nonzero(m <= 0.5)[0]
        if len(live) == 0:
            return x2.clone(), (torch.zeros(n, dtype=torch.float32, device=x2.device) if want_ldj else None)
        mask_t = self._programs.get(('mask', d, str(x2.device)),
                                    lambda: torch.from_numpy(m.astype(np.float32)).to(x2.device))
        z = x2.to(torch.float32) * mask_t                                              # coupling.py:61
        if d == 1:
            z = z * 0                                                                  # coupling.py:62-63
        if lat2 is not None:
            z = torch.cat([z, lat2.to(torch.float32)], -1)                             # coupling.py:64-65
        params = self.transform.latent_net(z)                                          # affine.py:66 / spline.py:82
        return self._transform_rows(x2, params, live, reverse, want_ldj, ldj_scale)

    # ---- set_data=True: the mask selects ELEMENTS OF THE SET (rows), all columns of a selected row are transformed ------
    def _run_set(self, x2, lat2, set_size, reverse, want_ldj, ldj_scale):
        """x2: [B * N, D] rows of B sets of N elements.  mask[n] = 1: element n passes through (and is all its
        conditioner would see); mask[n] = 0: z = 0 (+ latent), every column transformed (coupling.py:49-51,61,78,95)."""
        rows, d = x2.shape
        N = set_size
        m = self.mask_vector(N)
        dev = x2.device
        y = x2.clone()
        ldj = torch.zeros(rows, dtype=torch.float32, device=dev) if want_ldj else None
        live_n = np.nonzero(m <= 0.5)[0]
        if len(live_n) == 0 or rows == 0:
            return y, ldj
        sel = self._programs.get(('set-rows', rows, N, str(dev)), lambda: torch.from_numpy(
            (np.arange(rows // N)[:, None] * N + live_n[None, :]).reshape(-1).astype(np.int64)).to(dev))
        xt = x2.index_select(0, sel)                                                   # the transformed elements, compact
        params = self._set_params(x2, lat2, N, m, sel)
        yt, lt = self._transform_rows(xt, params, np.arange(d), reverse, want_ldj, ldj_scale)
        y.index_copy_(0, sel, yt)
        if want_ldj:
            ldj.index_copy_(0, sel, lt)
        return y, ldj

    def _set_params(self, x2, lat2, N: int, m: np.ndarray, sel: torch.Tensor) -> torch.Tensor:
        """The conditioner's output for the transformed set elements (rows `sel` of the [B * N, .] rows).  A row-wise conditioner
        (net.MLP) only ever sees z = cat[0, latent] on those rows, so it is called on the compact rows.  Any other module may look
        ACROSS the set (DeepSets / attention conditioners pool over the N axis, which is what set_data is for): it gets exactly
        the reference's z = cat[x * mask, latent] of shape (B, N, .) -- pass-through elements included -- in ONE call
        (coupling.py:49-51,61-65), and the transformed elements' rows of its output are selected afterwards."""
        rows, d = x2.shape
        dev = x2.device
        net = self.transform.latent_net
        if isinstance(net, MLP):
            z = torch.zeros(sel.numel(), d, dtype=torch.float32, device=dev)           # x * mask = 0 on these rows
            if lat2 is not None:
                z = torch.cat([z, lat2.index_select(0, sel).to(torch.float32)], -1)
            return net(z)
        mask_t = self._programs.get(('set-mask', N, str(dev)), lambda: torch.from_numpy(m.astype(np.float32)).to(dev).reshape(N, 1))
        z = x2.to(torch.float32).reshape(rows // N, N, d) * mask_t                     # coupling.py:49-51,61
        if d == 1:
            z = z * 0                                                                  # coupling.py:62-63
        if lat2 is not None:
            z = torch.cat([z, lat2.to(torch.float32).reshape(rows // N, N, -1)], -1)   # coupling.py:64-65
        out = net(z)
        return out.reshape(rows, out.shape[-1]).index_select(0, sel)

    def _conditioner_programs(self, dim: int, latent_dim: int, device, cond: np.ndarray, out_rows: np.ndarray):
        """The conditioner (net.MLP) as MFMA programs writing the selected output rows [N, len(out_rows)]: one program (chunked
        over output windows) when inputs and hidden layers fit the kernel's tiles; for single-hidden-layer conditioners also
        (round 3, the tier that flattens the width cliffs)
          * couplings wider than 128 columns: the program reads only the CONDITIONING columns of the wide rows (a column-subset
            program: <= 128 of them incl. the latent), and
          * hidden layers wider than 128: one program per chunk of 128 hidden units, later chunks accumulating into the output
            (W2 tanh(W1 z + b1) is a sum over hidden-unit chunks).
        -> list of CompiledProgram, to be run in order with the same mlp_out."""
        net = self._net()
        lin = net.linears()
        H = net.hidden_width
        wide_in = _ceil32(dim) + _ceil32(latent_dim) > 4
        if len(lin) != 2 or (H <= 128 and not wide_in):
            b = ProgramBuilder(dim, latent_dim, H)
            b.add_mlp(lin, net.act_code, cond, out_rows)
            return _chunk_mlp_program(b, device)
        cols = np.nonzero(cond)[0]
        if wide_in and (len(cols) == 0 or _ceil32(len(cols)) + _ceil32(latent_dim) > 4):
            raise NotImplementedError('more than 128 conditioning columns: generic tier')
        progs = []
        for h0 in range(0, H, 128):
            hsel = np.arange(h0, min(h0 + 128, H))
            if wide_in:
                b = ProgramBuilder(len(cols), latent_dim, len(hsel))
                b.x_cols, b.x_stride = cols, dim
                b.add_mlp(lin, net.act_code, None, out_rows, hidden_rows=hsel, accumulate=h0 > 0, w1_cols=cols, w1_latent_base=dim)
            else:
                b = ProgramBuilder(dim, latent_dim, len(hsel))
                b.add_mlp(lin, net.act_code, cond, out_rows, hidden_rows=hsel, accumulate=h0 > 0)
            progs += _chunk_mlp_program(b, device)
        return progs

    # ---- affine, unfused: pruned conditioner (MFMA program) + HBM-bound element-wise kernel -----------------
    def _affine_unfused_program(self, dim: int, latent_dim: int, device):
        key = ('affine-unfused', dim, latent_dim, str(device))

        def build():
            net = self._net()
            m = self.mask_vector(dim)
            live = np.nonzero(m <= 0.5)[0]
            cond = m > 0.5
            if dim == 1:
                cond = np.zeros(1, dtype=bool)
            out_rows = np.concatenate([live, dim + live])                           # (log_scale | shift) of live columns
            progs = self._conditioner_programs(dim, latent_dim, device, cond, out_rows)
            contiguous = len(live) > 0 and np.array_equal(live, np.arange(live[0], live[0] + len(live)))
            live_idx = None if contiguous else torch.from_numpy(live.astype(np.int32)).to(device)
            return (progs, live_idx, int(live[0]) if len(live) else 0, len(live))
        return self._programs.get(key, build)

    def _run_affine_unfused(self, x2, lat2, reverse, want_ldj, ldj_scale):
        from .affine import run_affine_kernel
        n, d = x2.shape
        if not (self.mask_vector(d) <= 0.5).any():
            return x2.clone(), (torch.zeros(n, dtype=torch.float32, device=x2.device) if want_ldj else None)
        progs, live_idx, live_start, n_live = self._affine_unfused_program(d, 0 if lat2 is None else lat2.shape[1],
                                                                           x2.device)
        params = torch.empty(n, 2 * n_live, dtype=torch.float32, device=x2.device)
        for p in progs:
            p.run(x2, lat2, mlp_out=params)
        return run_affine_kernel(x2, params, params.stride(0), live_idx, live_start, n_live, reverse, True, want_ldj,
                                 ldj_scale)

    # ---- spline: pruned conditioner (MFMA) + LDS-staged spline kernel --------------------------------------
    def _spline_program(self, dim: int, latent_dim: int, device):
        key = ('spline', dim, latent_dim, str(device))

        def build():
            net, sp = self._net(), self.transform
            m = self.mask_vector(dim)
            live = np.nonzero(m <= 0.5)[0]
            cond = m > 0.5
            if dim == 1:
                cond = np.zeros(1, dtype=bool)                                       # coupling.py:62-63
            P = sp.params_per_element                                                # 3K-1 quadratic, 2K+2 cubic
            out_rows = (live[:, None] * P + np.arange(P)[None, :]).reshape(-1)       # spline.py:82-86, pruned
            progs = self._conditioner_programs(dim, latent_dim, device, cond, out_rows)
            contiguous = len(live) > 0 and np.array_equal(live, np.arange(live[0], live[0] + len(live)))
            live_idx = None if contiguous else torch.from_numpy(live.astype(np.int32)).to(device)
            return (progs, live_idx, int(live[0]) if len(live) else 0, len(live), len(out_rows))
        return self._programs.get(key, build)

    def _run_spline(self, x2, lat2, reverse, want_ldj, ldj_scale):
        from .spline import run_rqs_kernel
        sp = self.transform
        n, d = x2.shape
        if not (self.mask_vector(d) <= 0.5).any():      # dim == 1: mask = [1], nothing is transformed (mask.py:37-38)
            return x2.clone(), (torch.zeros(n, dtype=torch.float32, device=x2.device) if want_ldj else None)
        try:
            return self._run_spline_slab(x2, lat2, reverse, want_ldj, ldj_scale)
        except NotImplementedError:
            pass                                        # > 16 bins, > 256 hidden units, bf16 rows, 'exact' (and 'auto' after a range flag): below
        progs, live_idx, live_start, n_live, width = self._spline_program(d, 0 if lat2 is None else lat2.shape[1],
                                                                          x2.device)
        params = torch.empty(n, width, dtype=torch.float32, device=x2.device)
        for p in progs:
            p.run(x2, lat2, mlp_out=params)
        if sp.spline_type == 'cubic':
            from .spline import run_cubic_kernel
            # one pass; in the inverse direction the kernel's reference mode (reverse = 2) returns what the reference's
            # Transform.inverse_and_log_det_jacobian does: MINUS the FORWARD log-det re-evaluated at the inverted point
            # (flow.py:42-47) -- so elements that round onto / across a domain bound or a knot get the reference's value
            y, ldj, _ = run_cubic_kernel(x2, params, params.stride(0), live_idx, live_start, n_live, sp.n_bins,
                                         sp.lower, sp.upper, 2 if (reverse and want_ldj) else reverse, want_ldj, False, ldj_scale)
        else:
            y, ldj, _ = run_rqs_kernel(x2, params, params.stride(0), live_idx, live_start, n_live, sp.n_bins,
                                       sp.lower, sp.upper, sp.lower, sp.upper, reverse, want_ldj, False, ldj_scale)
        return y, ldj

    # ---- spline, slab tier: hidden activation (MFMA program) + sx_rqs_slab_fwd; the parameter tensor never exists ---------------
    def _spline_slab_plan(self, dim: int, latent_dim: int, device):
        """Rational-quadratic couplings whose conditioner the one-launch program cannot hold (hidden layers of 129 .. 256 units;
        spline.py:76-87 and net/mlp.py:48-58 take any width): everything before the last Linear leaves the last hidden activation h
        in HBM (640 B per row at H = 160 -- against 6 KB per row of spline parameters; one hidden layer: sx_rqs_slab_hidden writes
        it as MFMA fragments, deeper conditioners: MLP programs, row-major), and sx_rqs_slab_fwd evaluates  params = h W2^T + b2
        and the spline slab by slab on the matrix pipe.
        -> (MLP programs [(program, first column of h)], input slots of the hidden kernel | None, slot -> row of the last Linear,
            hidden slots, live_idx, live_start, n_live, H, pack cache)."""
        key = ('spline-slab', dim, latent_dim, str(device))

        def build():
            from .spline import slab_slot_rows
            net, sp = self._net(), self.transform
            lin = net.linears()
            H = lin[-1][0].shape[1]
            widths = [w.shape[0] for (w, _) in lin[:-1]]
            cubic = sp.spline_type == 'cubic'
            if sp.n_bins > 16 or H > 256:
                raise NotImplementedError('slab tier: splines of up to 16 bins behind up to 256 hidden units')
            one_hidden = len(lin) == 2 and lin[0][1] is not None
            if dim + latent_dim > 128 if one_hidden else (_ceil32(dim) + _ceil32(latent_dim) > 4 or max(widths) > 128):
                raise NotImplementedError('slab tier: conditioner inputs of up to 128 columns; deep conditioners of up to 128 units')
            m = self.mask_vector(dim)
            live = np.nonzero(m <= 0.5)[0]
            cond = m > 0.5
            if dim == 1:
                cond = np.zeros(1, dtype=bool)                                       # coupling.py:62-63
            P = sp.params_per_element
            rel = slab_slot_rows(len(live), sp.n_bins, cubic)                        # slot -> row of the compact (live) parameter block
            rows_np = (live[:, None] * P + np.arange(P)[None, :]).reshape(-1)        # spline.py:82-86
            glob = np.where(rel >= 0, rows_np[np.clip(rel, 0, len(rows_np) - 1)], -1).astype(np.int32)
            hid = np.full(_ceil32(H) * 32, -1, dtype=np.int32)
            hid[:H] = np.arange(H)
            progs, in_slots, cond_words = [], None, None
            if one_hidden:
                # one hidden layer: sx_rqs_slab_hidden writes h as the fp16 fragments the slab kernel consumes.  Input slot q = column
                # q of cat[x, latent]; the masked columns have no slot (coupling.py:61: x * mask)
                q = np.full(_ceil32(dim + latent_dim) * 32, -1, dtype=np.int32)
                q[:dim] = np.where(cond, np.arange(dim), -1)
                q[dim:dim + latent_dim] = dim + np.arange(latent_dim)
                in_slots = torch.from_numpy(q).to(device)
                # the same map as four host words for the kernel: inputs without a slot are zeroed before the fp16 split (a transformed
                # column's magnitude must not matter: the reference multiplies it by mask = 0)
                words = [0, 0, 0, 0]
                for i in np.nonzero(q >= 0)[0]:
                    words[int(i) >> 5] |= 1 << (int(i) & 31)
                cond_words = tuple(words)
            else:
                # deeper conditioners: everything before the last Linear as the conditioner's own MLP programs with the identity as
                # last layer -> h [N, H] fp32, row-major
                eye = torch.eye(H, dtype=torch.float32, device=device)
                hlin = list(lin[:-1]) + [(eye, None)]
                b = ProgramBuilder(dim, latent_dim, max(widths))
                b.add_mlp(hlin, net.act_code, cond, np.arange(H))
                progs = [(p, 0) for p in _chunk_mlp_program(b, device)]
            contiguous = np.array_equal(live, np.arange(live[0], live[0] + len(live)))
            live_idx = None if contiguous else torch.from_numpy(live.astype(np.int32)).to(device)
            passthru = np.nonzero(m > 0.5)[0].astype(np.int32)               # y = x there: copied by the slab kernel
            cache = {'pass_idx': torch.from_numpy(passthru).to(device) if len(passthru) else None, 'n_pass': len(passthru), 'cond_words': cond_words}
            return (progs, in_slots, torch.from_numpy(glob).to(device), torch.from_numpy(hid).to(device), live_idx, int(live[0]), len(live), H, cache)
        return self._programs.get(key, build)

    def _run_spline_slab(self, x2, lat2, reverse, want_ldj, ldj_scale):
        sp = self.transform
        mode = _hip.get_gemm_precision()
        if debug.on('STRIBOR_SPLINE_NO_SLAB_FWD') or mode == 'exact' or x2.dtype != torch.float32:
            raise NotImplementedError('slab tier: fp32 rows, the fp16 x 3 arithmetic')
        if mode == 'auto':                          # a flag an EARLIER call left behind is that call's: raise it, do not swallow it
            torch.cuda.current_stream(x2.device).synchronize()
            _hip.poll_errors(device=x2.device)      # the flag word of x's device (and its current stream), not of whichever is current
        cubic = sp.spline_type == 'cubic'
        if (1e-2 if cubic else 1e-3) * sp.n_bins > 1.0:
            raise ValueError('Minimal bin width too large for the number of bins')      # rational_quadratic_spline.py:96-97, cubic_spline.py:93-96
        n, d = x2.shape
        dev = x2.device
        ld = 0 if lat2 is None else lat2.shape[1]
        progs, in_slots, slot_rows, hid_idx, live_idx, live_start, n_live, H, cache = self._spline_slab_plan(d, ld, dev)
        net = self._net()
        lin = net.linears()
        W2, b2 = lin[-1]
        if b2 is None:
            raise NotImplementedError('slab tier: the last Linear carries a bias')
        x2 = x2.contiguous()
        lat2 = None if lat2 is None else lat2.to(torch.float32).contiguous()
        lib = _hip.lib()
        flag = _hip.err_flag(dev)
        # the fragment packs of the conditioner's Linear layers, re-made when the parameters change
        ps = [W2, b2] + ([lin[0][0], lin[0][1]] if in_slots is not None else [])
        stamp = tuple((p.data_ptr(), p._version) for p in ps)
        if cache.get('stamp') != stamp:
            mt, ht = lib.sx_rqs_slab_slots(n_live) // 32, _ceil32(H)
            packs = torch.empty(_hip.packed_linear_floats(mt, ht), dtype=torch.float32, device=dev)
            Wc, bc = W2.detach().contiguous(), b2.detach().contiguous()
            _hip.call('sx_pack_linear', x2, Wc.data_ptr(), bc.data_ptr(), Wc.shape[0], H, slot_rows.data_ptr(), hid_idx.data_ptr(),
                      mt, ht, None, None, 0.0, 0, _hip.GEMM_F16X3, flag, packs.data_ptr())
            cache['packs'] = packs
            if in_slots is not None:
                ct = _ceil32(d + ld)
                w1 = torch.empty(_hip.packed_linear_floats(ht, ct), dtype=torch.float32, device=dev)
                W1c, b1c = lin[0][0].detach().contiguous(), lin[0][1].detach().contiguous()
                _hip.call('sx_pack_linear', x2, W1c.data_ptr(), b1c.data_ptr(), H, d + ld, hid_idx.data_ptr(), in_slots.data_ptr(),
                          ht, ct, None, None, 0.0, 0, _hip.GEMM_F16X3, flag, w1.data_ptr())
                cache['w1'] = w1
            cache['stamp'] = stamp
        packs = cache['packs']
        if in_slots is not None:
            h = torch.empty(lib.sx_rqs_slab_hidden_floats(n, H), dtype=torch.float32, device=dev)
            import ctypes as C
            words = (C.c_uint32 * 4)(*cache['cond_words'])
            _hip.call('sx_rqs_slab_hidden', x2, x2.data_ptr(), _hip.ptr(lat2), cache['w1'].data_ptr(), C.cast(words, C.c_void_p), h.data_ptr(),
                      n, d, ld, H, net.act_code, flag)
            ld_h, frag = 0, 1
        else:
            h = torch.empty(n, H, dtype=torch.float32, device=dev)
            for p, h0 in progs:
                p.run(x2, lat2, mlp_out=h[:, h0:])
            ld_h, frag = h.stride(0), 0
        # every column is written by the slab kernel: the transformed ones, and y = x in the rest (pass_idx).  (A copy made by the
        # hidden-layer kernel, whose registers the rows pass through, measured slower than a streaming clone -- 16-byte pieces 256 B apart:
        # 102 vs 70 + 19 us; the slab kernel's lanes copy one more element each beside the one they transform: no measurable cost.)
        y = torch.empty_like(x2)
        ldj = torch.empty(n, dtype=torch.float32, device=dev) if want_ldj else None
        with _hip.device_of(x2):
            sc = _hip.scratch(dev, lib.sx_rqs_slab_fwd_scratch_floats(n, n_live)) if want_ldj else None
        # cubic splines: a coupling's inverse log-det is MINUS the FORWARD log-det at the inverted point (flow.py:42-47): reverse = 2
        rev = (2 if (cubic and want_ldj) else 1) if reverse else 0
        _hip.call('sx_rqs_slab_fwd', x2, x2.data_ptr(), h.data_ptr(), ld_h, H, packs.data_ptr(), y.data_ptr(), _hip.ptr(ldj),
                  _hip.ptr(live_idx), live_start, n_live, _hip.ptr(cache['pass_idx']), cache['n_pass'], sp.n_bins, float(sp.lower),
                  float(sp.upper), float(sp.lower),
                  float(sp.upper), n, d, rev, float(ldj_scale), 0, frag, int(cubic), _hip.ptr(sc), flag)
        if mode == 'auto':                          # never hand back a NaN-poisoned result: the tier below re-runs the layer exactly
            torch.cuda.current_stream(dev).synchronize()
            with _hip.device_of(x2):
                if _hip.take_flag(dev, _hip.FLAG_F16_RANGE):
                    raise NotImplementedError('slab tier: an operand left the fp16 x 3 range')
        return y, ldj

    # ---- training (autograd): spline couplings, inverse direction --------------------------------------------------
    def _autograd_supported(self) -> bool:
        if self._wraps_other():            # the wrapped transform's own ops carry the graph (see _wrapped)
            return isinstance(self.transform, torch.nn.Module) and not self.set_data
        return True

    def _autograd_set(self, x2: torch.Tensor, lat2, set_size: int, reverse: bool):
        """set_data=True with a graph (coupling.py:48-53, 61, 78, 95): the transformed set elements are gathered into compact rows,
        transformed -- every column -- by the element-wise op of the transform with the conditioner's output for cat[0, latent]
        (torch's graph through the conditioner, the HIP kernels behind AffineCouplingOp / RQSInverse / CubicInverse and their
        forward-direction twins for the transform), and scattered back.  -> (rows [B * N, D], log-det [B * N])."""
        rows, d = x2.shape
        N = int(set_size)
        dev = x2.device
        live_n = np.nonzero(self.mask_vector(N) <= 0.5)[0]
        zero = torch.zeros(rows, dtype=torch.float32, device=dev)
        if len(live_n) == 0 or rows == 0:
            return x2, zero
        sel = self._programs.get(('set-rows', rows, N, str(dev)), lambda: torch.from_numpy(
            (np.arange(rows // N)[:, None] * N + live_n[None, :]).reshape(-1).astype(np.int64)).to(dev))
        xt = x2.index_select(0, sel)
        net = self.transform.latent_net
        if isinstance(net, MLP):
            z = torch.zeros(xt.shape[0], d, dtype=torch.float32, device=dev)           # x * mask = 0 on these rows
            if lat2 is not None:
                z = torch.cat([z, lat2.index_select(0, sel).to(torch.float32)], -1)
            yt, lt = self.transform._autograd_inverse(xt.contiguous(), z, reverse=reverse)
        else:                       # a set-aware conditioner sees the whole set (coupling.py:61-65): torch's graph through it
            params = self._set_params(x2, lat2, N, self.mask_vector(N), sel)
            yt, lt = self.transform._autograd_from_params(xt.contiguous(), params, reverse)
        return x2.index_copy(0, sel, yt), zero.index_copy(0, sel, lt.reshape(-1))

    def _autograd_forward(self, x2: torch.Tensor, lat2=None):
        """forward_and_log_det_jacobian with a graph (affine and spline couplings)."""
        return self._autograd_inverse(x2, lat2, reverse=False)

    def _autograd_inverse(self, x2: torch.Tensor, lat2=None, reverse: bool = True, pre=None):
        """inverse_and_log_det_jacobian on fp32 rows [N, D] with a graph: the conditioner runs through torch's own
        Linear layers (rocBLAS; only the rows of the last layer that parameterise transformed columns), the transform
        and its backward are the HIP kernels behind ``RQSInverse`` / ``AffineCouplingOp``.
        `pre`: this layer's (output rows, log-det share, tanh h) from a one-launch forward of the whole flow
        (NormalizingFlow._spline_forward_once) -- an argument of the call, not state on the module: the same Coupling may sit in two
        flows, or twice in one.  Only the slab path can use it; any other routing with `pre` given is a planning error.
        Returns (x_out [N, D], ldj [N])."""
        from .spline import CubicForward, CubicInverse, RQSForward, RQSInverse, Spline
        from .affine import AffineCouplingOp
        if self._wraps_other():
            y, ldj = self._wrapped(x2, lat2, reverse, True, True, reverse)
            return y, ldj.reshape(-1)
        sp, net = self.transform, self.transform.latent_net
        is_spline = isinstance(sp, Spline)
        n, d = x2.shape
        m = self.mask_vector(d)
        live = np.nonzero(m <= 0.5)[0]
        if len(live) == 0:
            # nothing is transformed (dim 1: mask = [1], mask.py:37-38).  The reference still evaluates the conditioner and
            # multiplies its transform by (1 - mask) = 0 (coupling.py:71-78,94-95): the outputs carry a graph in which every
            # conditioner parameter has a ZERO gradient (base.py:76-81 calls backward on exactly this) -- keep that
            z = x2 * 0 if d == 1 else x2 * torch.from_numpy(m.astype(np.float32)).to(x2.device)     # coupling.py:61-63
            if lat2 is not None:
                z = torch.cat([z, lat2], -1)
            tie = (net.forward_autograd(z) if isinstance(net, MLP) else net(z)).sum() * 0
            return x2 + tie, torch.zeros(n, dtype=torch.float32, device=x2.device) + tie
        key = ('autograd', d, str(x2.device))

        def build():
            if is_spline:
                P = sp.params_per_element
                rows = (live[:, None] * P + np.arange(P)[None, :]).reshape(-1)      # spline.py:82-86
            else:
                rows = np.concatenate([live, d + live])                               # affine.py:66 (log_scale | shift)
            contiguous = np.array_equal(live, np.arange(live[0], live[0] + len(live)))
            return (torch.from_numpy(m.astype(np.float32)).to(x2.device),
                    torch.from_numpy(rows.astype(np.int64)).to(x2.device),
                    None if contiguous else torch.from_numpy(live.astype(np.int32)).to(x2.device))
        mask_t, rows_t, live_idx = self._programs.get(key, build)
        if is_spline and reverse and self._slab_backward_ok(net, sp):
            return self._autograd_inverse_slab(x2, lat2, mask_t, rows_t, live, live_idx, pre)
        if pre is not None:
            raise RuntimeError('stribor_amd: a precomputed forward was handed to a coupling that does not train on the slab path')
        z = x2 * mask_t                                                              # coupling.py:61
        if d == 1:
            z = z * 0                                                                # coupling.py:62-63
        if lat2 is not None:
            z = torch.cat([z, lat2], -1)                                             # coupling.py:64-65
        if isinstance(net, MLP):
            params = net.forward_autograd(z, rows_t)
        else:                                                                        # any nn.Module: torch's own graph
            params = net(z).index_select(1, rows_t)
        if is_spline:
            if sp.spline_type == 'cubic':
                op = CubicInverse if reverse else CubicForward
            else:
                op = RQSInverse if reverse else RQSForward
            return op.apply(x2, params, live_idx, int(live[0]), len(live), sp.n_bins, sp.lower, sp.upper, 1.0)
        # Transform.inverse_and_log_det_jacobian: minus the forward log-det (flow.py:47)
        return AffineCouplingOp.apply(x2, params, live_idx, int(live[0]), len(live), bool(reverse), -1.0 if reverse else 1.0)

    # ---- spline couplings: backward fused with the conditioner's last layer (sx_rqs_slab_bwd) ----------------------------
    def _slab_backward_ok(self, net, sp) -> bool:
        from .spline import RQSCouplingSlab
        if debug.on('STRIBOR_SPLINE_UNFUSED'):                         # A/B switch: the per-row parameter path
            return False
        if not isinstance(net, MLP) or net._wrapped or net.final_activation_name is not None:
            return False
        lin = net.linears()
        return len(lin) >= 2 and lin[-1][1] is not None and RQSCouplingSlab.eligible(lin[-1][0].shape[1], sp.n_bins)

    def _slab_l1_ok(self, d: int) -> bool:
        """Does log_prob's graph path run this layer through RQSCouplingSlabL1 (Linear - Tanh - Linear conditioner without a latent
        input)?  NormalizingFlow._layerwise_autograd uses it to decide on the one-launch forward."""
        from .spline import RQSCouplingSlabL1, Spline
        sp = self.transform
        net = getattr(sp, 'latent_net', None)
        if self.set_data or not isinstance(sp, Spline) or net is None or not self._slab_backward_ok(net, sp):
            return False
        if debug.on('STRIBOR_SPLINE_L1_TORCH') or d < 2 or not np.any(self.mask_vector(d) <= 0.5):
            return False
        lin = net.linears()
        return (len(lin) == 2 and net.activation_name == 'Tanh' and lin[0][1] is not None and lin[0][0].shape[1] == d
                and RQSCouplingSlabL1.eligible(d, lin[-1][0].shape[1], sp.n_bins))

    def _inverse_rows_nograd(self, x2, lat2, h_out=None):
        """inverse_and_log_det_jacobian of [N, D] fp32 rows without a graph: the one-layer fused program when the conditioner
        fits it (parameters never in HBM), else the MLP program + spline kernel.  h_out [N, H] (optional): receives the
        conditioner's last hidden activation from the fused program -> (y, ldj, True); (y, ldj) / (y, ldj, False) otherwise."""
        d, ld = x2.shape[1], 0 if lat2 is None else lat2.shape[1]
        try:
            # planner convention: the coefficient of the FORWARD log-det (flow.py:47: the inverse returns minus it)
            prog = self._affine_program(True, -1.0, d, ld, x2.device)
        except NotImplementedError:
            prog = None
        if prog is not None:
            if h_out is not None and _hip.get_gemm_precision() != 'auto':
                y, ldj, _ = prog.run(x2, lat2, True, True, False, mlp_out=h_out)
                return y, ldj, True
            y, ldj, _ = prog.run(x2, lat2, True, True, False)
            return y, ldj
        try:
            return self._run_spline(x2, lat2, True, True, 1.0)
        except NotImplementedError:          # beyond the MLP program's tiles too (dim + latent > 4 tiles of 32): generic tier, as _run
            return self._run_generic(x2, lat2, True, True, 1.0)

    def _autograd_inverse_slab(self, x2, lat2, mask_t, rows_t, live, live_idx, pre=None):
        from .spline import RQSCouplingSlab, RQSCouplingSlabL1, slab_slot_rows
        from ..net.mlp import SelectRows
        sp, net = self.transform, self.transform.latent_net
        d = x2.shape[1]
        lin = net.linears()
        H = lin[-1][0].shape[1]
        cubic = sp.spline_type == 'cubic'

        def build():
            hid = np.full(((H + 31) // 32) * 32, -1, dtype=np.int32)
            hid[:H] = np.arange(H)
            m = self.mask_vector(d)
            if d == 1:
                m = m * 0                                                            # coupling.py:62-63
            cols = np.full(((d + 31) // 32) * 32, -1, dtype=np.int32)
            cols[:d] = np.arange(d)
            cmap = cols.copy()
            cmap[:d][m <= 0.5] = -1                                                  # transformed columns: no weight in W1 * mask
            words = [int(sum(1 << c for c in range(32) if 32 * t + c < d and m[32 * t + c] > 0.5)) for t in range(2)]
            dev = x2.device
            return (torch.from_numpy(slab_slot_rows(len(live), sp.n_bins, cubic)).to(dev), torch.from_numpy(hid).to(dev),
                    torch.from_numpy(cols).to(dev), torch.from_numpy(cmap).to(dev), words)
        plan = self._programs.get(('slab', d, H, cubic, str(x2.device)), build)
        evaluate = lambda xx, h_out=None: self._inverse_rows_nograd(xx, lat2, h_out)
        evaluate.precomputed = pre                                                   # (one-launch forward of a whole spline flow)
        col_mask = mask_t * 0 if d == 1 else mask_t                                  # coupling.py:62-63
        if (lat2 is None and len(lin) == 2 and net.activation_name == 'Tanh' and lin[0][1] is not None
                and RQSCouplingSlabL1.eligible(d, H, sp.n_bins) and not debug.on('STRIBOR_SPLINE_L1_TORCH')):
            # Linear - Tanh - Linear conditioner: the first layer's backward is part of the op too.  The op takes the WHOLE last
            # layer and a slot map into its rows (no gather of the live parameters' rows on the way in, no zero-fill + scatter of
            # their gradients on the way out: 4 launches per layer and step)
            def build_full():
                rel = slab_slot_rows(len(live), sp.n_bins, cubic)
                P = sp.params_per_element
                rows_np = (np.asarray(live)[:, None] * P + np.arange(P)[None, :]).reshape(-1)
                glob = np.where(rel >= 0, rows_np[np.clip(rel, 0, len(rows_np) - 1)], -1).astype(np.int32)
                return (torch.from_numpy(glob).to(x2.device),) + tuple(plan[1:]) + (True, {})      # {}: the layer's pack cache (RQSCouplingSlabL1.backward)
            plan_full = self._programs.get(('slab_full', d, H, cubic, str(x2.device)), build_full)
            return RQSCouplingSlabL1.apply(x2, lin[0][0], lin[0][1], lin[-1][0], lin[-1][1], col_mask, evaluate, plan_full, live_idx,
                                           int(live[0]), len(live), sp.n_bins, sp.lower, sp.upper, cubic)
        if evaluate.precomputed is not None:
            raise RuntimeError('stribor_amd: a precomputed forward reached a coupling outside the Linear-Tanh-Linear slab path')
        W2, b2 = SelectRows.apply(lin[-1][0], rows_t), SelectRows.apply(lin[-1][1], rows_t)
        # conditioner input cat[x * mask, latent] (coupling.py:61-65) with the mask folded into the first layer's weight
        if lat2 is not None:
            col_mask = torch.cat([col_mask, torch.ones(lat2.shape[1], dtype=torch.float32, device=x2.device)])
        # (a Tanh right before the last Linear is applied inside the op: its backward rides on the kernel that reduces dL/dh)
        h, last, _, pre_tanh = net.hidden_autograd(x2 if lat2 is None else torch.cat([x2, lat2], -1), col_mask, pre_tanh=True,
                                                   want_flag=True)
        return RQSCouplingSlab.apply(x2, h, W2, b2, evaluate, plan[:2], live_idx, int(live[0]), len(live), sp.n_bins, sp.lower,
                                     sp.upper, pre_tanh, cubic)

    # ---- reference method set (coupling.py:69-95) -----------------------------------------------------------
    # Every method is differentiable like the reference's: when a graph is wanted (grad mode and the input, the latent or a
    # parameter requires grad) the call runs through the layer's autograd ops (`_graph`), otherwise through the no-graph tiers.
    def _graph(self, x, latent, reverse: bool):
        """(out [..., D], log-det [..., 1]) WITH a graph: forward_and_log_det_jacobian, or -- reverse -- what
        inverse_and_log_det_jacobian returns (flow.py:42-47)."""
        _hip.require_device(x, 'x')
        from .spline import Spline
        if not isinstance(self.transform, (Affine, Spline)) or getattr(self.transform, 'latent_net', None) is None:
            raise NotImplementedError(f'Coupling({type(self.transform).__name__}) is not on the hot path')
        x2, lat2, lead = graph_rows(x, latent)
        if self.set_data:
            if x.dim() < 2:
                raise ValueError('set_data=True needs inputs of shape (..., N, dim)')
            y, ldj = self._autograd_set(x2, lat2, x.shape[-2], reverse)
        else:
            y, ldj = self._autograd_inverse(x2, lat2, reverse=reverse)
        return y.reshape(*lead, x2.shape[1]), ldj.reshape(*lead, 1)

    # ---- any other ElementwiseTransform (coupling.py:10-46,74-76,94): the reference's op sequence around the wrapped transform --
    def _wraps_other(self) -> bool:
        """The wrapped transform is not an Affine / Spline driven by a conditioner: e.g. Coupling(st.Sigmoid(), mask) or
        Coupling(st.Affine(dim), mask) without a latent_net.  The reference calls transform(x, latent=z), transform.inverse(x,
        latent=z) and transform.log_diag_jacobian(x, y, latent=z) on whatever it wraps; so does this tier: the wrapped transform runs
        its own kernels (sx_pointwise, sx_affine_coupling, ...), the mask blend is three element-wise torch ops, and the whole of
        it is differentiable through the wrapped transform's own autograd ops."""
        from .spline import Spline
        return not isinstance(self.transform, (Affine, Spline)) or getattr(self.transform, 'latent_net', None) is None

    def _wrapped(self, x, latent, reverse: bool, want_y: bool, want_ldj: bool, negate: bool, y=None, **kwargs):
        _hip.require_device(x, 'x')
        if self.set_data and x.dim() < 2:
            raise ValueError('set_data=True needs inputs of shape (..., N, dim)')
        mask = self._get_mask(x)                                                       # coupling.py:48-53

        def conditioning(v):
            z = v * mask                                                               # coupling.py:61
            if v.shape[-1] == 1:
                z = z * 0                                                              # coupling.py:62-63
            return z if latent is None else torch.cat([z, latent.to(z.dtype)], -1)     # coupling.py:64-65
        z = conditioning(x)
        t = self.transform
        if not want_y:                                                                 # log_det_jacobian(x, y): the caller's y goes
            ld = t.log_diag_jacobian(x, y, latent=z, **kwargs)                         # to the wrapped transform (coupling.py:94)
            return None, (ld * (1 - mask)).sum(-1, keepdim=True)
        y_ = t.inverse(x, latent=z, **kwargs) if reverse else t(x, latent=z, **kwargs)     # coupling.py:74-76
        y = y_ * (1 - mask) + x * mask                                                 # coupling.py:78
        if not want_ldj:
            return y, None
        # forward: log_det_jacobian(x, y); inverse: flow.py:42-47 -- x = inverse(y), then log_det_jacobian(x, y), negated
        a, b = (y, x) if reverse else (x, y)
        ld = t.log_diag_jacobian(a, b, latent=conditioning(a) if reverse else z, **kwargs)
        ldj = (ld * (1 - mask)).sum(-1, keepdim=True)
        return y, (-ldj if negate else ldj)

    def forward(self, x, latent=None, reverse: bool = False, **kwargs):
        if self._wraps_other():
            return self._wrapped(x, latent, reverse, True, False, False, **kwargs)[0]
        if graph_wanted(self, x, latent):
            return self._graph(x, latent, reverse)[0]
        return self._run(x, latent, reverse, True, False)[0]

    def inverse(self, y, latent=None, **kwargs):
        if self._wraps_other():
            return self._wrapped(y, latent, True, True, False, False, **kwargs)[0]
        if graph_wanted(self, y, latent):
            return self._graph(y, latent, True)[0]
        return self._run(y, latent, True, True, False)[0]                            # coupling.py:81-82 (Q3)

    def log_det_jacobian(self, x, y=None, latent=None, **kwargs):
        if self._wraps_other():
            return self._wrapped(x, latent, False, False, True, False, y=y, **kwargs)[1]
        if graph_wanted(self, x, latent):
            return self._graph(x, latent, False)[1]
        return self._run(x, latent, False, False, True)[1]

    def forward_and_log_det_jacobian(self, x, latent=None, **kwargs):
        if self._wraps_other():
            return self._wrapped(x, latent, False, True, True, False, **kwargs)
        if graph_wanted(self, x, latent):
            return self._graph(x, latent, False)
        return self._run(x, latent, False, True, True)

    def inverse_and_log_det_jacobian(self, y, latent=None, **kwargs):
        if self._wraps_other():
            return self._wrapped(y, latent, True, True, True, True, **kwargs)
        if graph_wanted(self, y, latent):
            return self._graph(y, latent, True)
        if isinstance(self.transform, Affine):
            # the log-scales that invert y are the forward log-det at x (same conditioner input): one launch
            return self._run(y, latent, True, True, True, ldj_scale=-1.0)
        # spline: inverse kernel returns the already-negated log-diag-Jacobian (rational_quadratic_spline.py:234)
        return self._run(y, latent, True, True, True, ldj_scale=1.0)

    # ---- fused-program hooks --------------------------------------------------------------------------------
    def _plan_hidden_width(self):
        return self._net().hidden_width if self._has_mlp() else 0

    def _plan_spline(self, builder: ProgramBuilder, reverse: bool, ldj_scale: float) -> bool:
        if self.set_data:
            return False
        sp = self.transform
        net = getattr(sp, 'latent_net', None)
        if not isinstance(net, MLP) or net.activation_name != 'Tanh' or sp.n_bins > (32 if sp.spline_type == 'quadratic' else 16) or \
                sp.spline_type not in ('quadratic', 'cubic') or debug.on('STRIBOR_CUBIC_UNFUSED') and sp.spline_type == 'cubic':
            return False
        lin = net.linears()
        if len(lin) < 2:
            return False
        (W1, b1), (W2, b2) = lin[0], lin[-1]
        P = sp.params_per_element                                                        # 3K-1 quadratic, 2K+2 cubic
        if W1.shape[1] != builder.dim + builder.latent_dim or W2.shape[0] != builder.dim * P:
            raise ValueError(f'latent_net maps {W1.shape[1]} -> {W2.shape[0]}, expected '
                             f'{builder.dim + builder.latent_dim} -> {builder.dim * P}')
        builder.add_coupling_rqs(W1, b1, W2, b2, self.mask_vector(builder.dim), reverse, ldj_scale, W1.shape[0],
                                 sp.n_bins, sp.lower, sp.upper, sp.lower, sp.upper, middle=lin[1:-1],
                                 cubic=sp.spline_type == 'cubic')
        return True

    def _plan_first_mask(self, dim):
        return None if self.set_data else self.mask_vector(dim)

    def _plan(self, builder: ProgramBuilder, reverse: bool, ldj_scale: float) -> bool:
        from .spline import Spline
        if isinstance(self.transform, Spline):
            return self._plan_spline(builder, reverse, ldj_scale)
        if self.set_data or not isinstance(self.transform, Affine) or not self._has_mlp():
            return False                # generic tier: the flow runs layer by layer
        net = self._net()
        lin = net.linears()
        if lin[0][0].shape[1] != builder.dim + builder.latent_dim or lin[-1][0].shape[0] != 2 * builder.dim:
            raise ValueError(f'latent_net maps {lin[0][0].shape[1]} -> {lin[-1][0].shape[0]}, expected '
                             f'{builder.dim + builder.latent_dim} -> {2 * builder.dim}')
        if len(lin) >= 3:                 # two or more hidden layers: hidden state kept in registers between steps
            builder.add_coupling_affine_deep(lin, self.mask_vector(builder.dim), net.act_code, reverse, ldj_scale)
            return True
        (W1, b1), (W2, b2) = lin
        if W1.shape[1] != builder.dim + builder.latent_dim or W2.shape[0] != 2 * builder.dim:
            raise ValueError(f'latent_net maps {W1.shape[1]} -> {W2.shape[0]}, expected '
                             f'{builder.dim + builder.latent_dim} -> {2 * builder.dim}')
        builder.add_coupling_affine(W1, b1, W2, b2, self.mask_vector(builder.dim), net.act_code, reverse, ldj_scale,
                                    W1.shape[0])
        return True


class ContinuousAffineCoupling(Transform):
    """Time-conditioned affine coupling (reference: stribor/flows/coupling.py:98-213): identity at t = 0.

    ``latent_net`` (a ``net.MLP``) sees ``cat[x * mask, latent, t]`` (``concatenate_time``) and outputs ``2 * dim``
    values, ``time_net`` (``net.TimeIdentity / TimeLinear / TimeTanh / TimeLog``) embeds t; the conditioner runs as a pruned
    MFMA program, the time embedding, the affine map, the blend and the masked log-det are one pass of
    ``sx_time_affine_coupling``."""

    def __init__(self, latent_net: nn.Module, time_net: nn.Module, mask: str, concatenate_time: Optional[bool] = True,
                 **kwargs):
        super().__init__()
        # fast tier: net.MLP conditioner (pruned MFMA program) + an in-kernel time net (TimeIdentity / TimeLinear /
        # TimeTanh / TimeLog); anything else -- a user-written conditioner, TimeFourier(Bounded), a user-written time
        # net -- is called as a module and its output multiplied into the parameters before the HIP affine kernel
        self.latent_net = latent_net
        self.time_net = time_net
        self.mask_func = get_mask(mask)
        self.concatenate_time = concatenate_time
        self._masks = {}
        self._masks_epoch = -1
        self._programs = ProgramCache()

    def mask_vector(self, dim: int) -> np.ndarray:
        if self._masks_epoch != _STRUCT_EPOCH[0]:
            self._masks, self._masks_epoch = {}, _STRUCT_EPOCH[0]
        if dim not in self._masks:
            m = self.mask_func(dim).numpy().astype(np.float64).reshape(-1)
            self._masks[dim] = np.full(dim, m[0]) if m.size == 1 else m
        return self._masks[dim]

    def _program(self, dim: int, extra: int, device):
        key = (dim, extra, str(device))

        def build():
            net = self.latent_net
            m = self.mask_vector(dim)
            live = np.nonzero(m <= 0.5)[0]
            cond = m > 0.5
            if dim == 1:
                cond = np.zeros(1, dtype=bool)                                       # coupling.py:151-152
            b = ProgramBuilder(dim, extra, net.hidden_width)
            b.add_mlp(net.linears(), net.act_code, cond, np.concatenate([live, dim + live]))   # chunk(2): (ls | sh)
            contiguous = len(live) > 0 and np.array_equal(live, np.arange(live[0], live[0] + len(live)))
            live_idx = None if contiguous else torch.from_numpy(live.astype(np.int32)).to(device)
            return (_chunk_mlp_program(b, device), live_idx, int(live[0]) if len(live) else 0, live)
        return self._programs.get(key, build)

    # ---- fused tier (round 3): the whole layer -- conditioner, time embedding, affine map, log-det -- is ONE step ---------------
    def _fusable(self) -> bool:
        net, tn = self.latent_net, self.time_net
        return (isinstance(net, MLP) and net.fusable() and len(net.linears()) == 2 and net.hidden_width <= 128
                and getattr(tn, 'kind', None) in (0, 1, 2, 3, 4) and (tn.kind != 4 or tn.hidden_dim <= 64))

    def _plan_time(self, builder: ProgramBuilder, reverse: bool, ldj_scale: float, time_sel: int) -> bool:
        """One SX_STEP_COUPLING_TIME step; `time_sel`: which of the program's time slots (0: t, 1: t0) this pass reads."""
        if not self._fusable():
            return False
        net = self.latent_net
        (W1, b1), (W2, b2) = net.linears()
        D, L = builder.dim, builder.latent_dim
        want_in = D + L + (1 if self.concatenate_time else 0)
        if W1.shape[1] != want_in or W2.shape[0] != 2 * D:
            raise ValueError(f'latent_net maps {W1.shape[1]} -> {W2.shape[0]}, expected {want_in} -> {2 * D}')
        builder.add_coupling_time(W1, b1, W2, b2, self.mask_vector(D), net.act_code, reverse, ldj_scale, W1.shape[0],
                                  D + L if self.concatenate_time else None, time_sel, self.time_net)
        return True

    def _fused_program(self, reverse: bool, ldj_scale: float, dim: int, latent_dim: int, device):
        key = ('time-fused', bool(reverse), float(ldj_scale), dim, latent_dim, str(device))

        def build():
            try:
                b = ProgramBuilder(dim, latent_dim, self.latent_net.hidden_width, time_slots=1)
                return b.build(device) if self._plan_time(b, reverse, ldj_scale, 0) else None
            except NotImplementedError:
                return None
        return self._programs.get(key, build)

    def _time_scales(self, dim: int, live: np.ndarray, device):
        """Per live column: the time net's scale for its log_scale and for its shift (chunk(2) of the embedding,
        broadcast against [.., dim] like the reference: half-width dim or 1)."""
        tn = self.time_net
        if tn.kind == 0:
            return None
        sc = tn.scale.detach().reshape(-1).to(device=device, dtype=torch.float32)
        half = sc.numel() // 2
        if half not in (1, dim):
            raise ValueError(f'time_net width {sc.numel()} does not broadcast against 2 x {dim}')
        idx = torch.from_numpy(live.astype(np.int64)).to(device) if half == dim else torch.zeros(len(live), dtype=torch.long, device=device)
        return torch.cat([sc[:half].index_select(0, idx), sc[half:2 * half].index_select(0, idx)]).contiguous()

    def _run(self, x, t, latent, reverse, want_ldj, ldj_scale):
        _hip.require_device(x, 'x')
        x2, lead = flatten_rows(x)
        n, d = x2.shape
        t2 = t.reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
        if t2.numel() != n:
            t2 = t.expand(*lead, 1).reshape(-1).to(torch.float32).contiguous()
        needs_graph = torch.is_grad_enabled() and (x.requires_grad or t.requires_grad or
                                                    (latent is not None and latent.requires_grad) or
                                                    any(p.requires_grad for p in self.parameters()))
        if not needs_graph and self._fusable() and n > 0:
            # ONE launch, no torch op: the kernel reads t straight into the conditioner's time slot and the embedding
            lat_only = None if latent is None else latent.reshape(n, -1).to(torch.float32).contiguous()
            prog = self._fused_program(reverse, ldj_scale, d, 0 if lat_only is None else lat_only.shape[1], x.device)
            if prog is not None:
                y, ldj, _ = prog.run(x2, lat_only, True, want_ldj, False, row_t=t2)
                return y.reshape(*lead, d), (None if ldj is None else ldj.reshape(*lead, 1))
        parts = [] if latent is None else [latent.reshape(n, -1).to(torch.float32)]
        if self.concatenate_time:
            parts.append(t2.reshape(n, 1))                                           # coupling.py:155-156
        lat2 = torch.cat(parts, -1).contiguous() if parts else None
        extra = 0 if lat2 is None else lat2.shape[1]
        fast = isinstance(self.latent_net, MLP) and self.latent_net._fits_program() and getattr(self.time_net, 'kind', 9) <= 3
        if fast:
            try:
                progs, live_idx, live_start, live = self._program(d, extra, x.device)
            except NotImplementedError:
                fast = False
        if not fast:
            live = np.nonzero(self.mask_vector(d) <= 0.5)[0]
            contiguous = len(live) > 0 and np.array_equal(live, np.arange(live[0], live[0] + len(live)))
            live_idx = None if contiguous else torch.from_numpy(live.astype(np.int32)).to(x.device)
            live_start = int(live[0]) if len(live) else 0
        if (needs_graph or not fast) and len(live):
            # training: conditioner and time net through torch (library GEMMs / tiny element-wise ops), the affine map and
            # its backward through the HIP op; same values as the kernel path
            from .affine import AffineCouplingOp
            m = torch.from_numpy(self.mask_vector(d).astype(np.float32)).to(x.device)
            z = x2.to(torch.float32) * m
            if d == 1:
                z = z * 0
            if lat2 is not None:
                z = torch.cat([z, lat2], -1)
            rows = torch.from_numpy(np.concatenate([live, d + live]).astype(np.int64)).to(x.device)
            if isinstance(self.latent_net, MLP):
                params = self.latent_net.forward_autograd(z, rows)
            else:
                params = self.latent_net(z).index_select(1, rows)                    # coupling.py:194
            emb = self.time_net(t2.reshape(n, 1))                                    # [n, out]  coupling.py:195
            half = emb.shape[1] // 2
            cols = torch.from_numpy(live.astype(np.int64)).to(x.device) if half == d else torch.zeros(len(live), dtype=torch.long, device=x.device)
            scale = torch.cat([emb[:, :half].index_select(1, cols), emb[:, half:2 * half].index_select(1, cols)], -1)
            y, ldj = AffineCouplingOp.apply(x2.to(torch.float32), params * scale, live_idx, live_start, len(live), bool(reverse),
                                            float(ldj_scale))
            return y.reshape(*lead, d), (ldj.reshape(*lead, 1) if want_ldj else None)
        y = torch.empty_like(x2)
        ldj = torch.empty(n, dtype=torch.float32, device=x.device) if want_ldj else None
        if len(live) == 0:
            return x2.clone().reshape(*lead, d), (torch.zeros(*lead, 1, device=x.device) if want_ldj else None)
        params = torch.empty(n, 2 * len(live), dtype=torch.float32, device=x.device)
        for p in progs:
            p.run(x2, lat2, mlp_out=params)
        tscale = self._time_scales(d, live, x.device)
        _hip.call('sx_time_affine_coupling', x2, x2.data_ptr(), y.data_ptr(), _hip.ptr(ldj), params.data_ptr(),
                                                params.stride(0), t2.data_ptr(), _hip.ptr(tscale), self.time_net.kind,
                                                _hip.ptr(live_idx), live_start, len(live), n, d, _hip.dtype_code(x2),
                                                int(reverse), 0, float(ldj_scale))
        return y.reshape(*lead, d), (None if ldj is None else ldj.reshape(*lead, 1))

    # ---- reference method set (coupling.py:159-213) ----------------------------------------------------------------
    def forward(self, x, t, latent=None, **kwargs):
        return self._run(x, t, latent, False, False, 1.0)[0]

    def inverse(self, y, t, latent=None, **kwargs):
        return self._run(y, t, latent, True, False, 1.0)[0]

    def log_det_jacobian(self, x, y=None, *, t, latent=None, **kwargs):
        return self._run(x, t, latent, False, True, 1.0)[1]

    def forward_and_log_det_jacobian(self, x, t, latent=None, *, reverse: bool = False, **kwargs):
        return self._run(x, t, latent, reverse, True, 1.0)

    def inverse_and_log_det_jacobian(self, y, t, latent=None, **kwargs):
        return self._run(y, t, latent, True, True, -1.0)                             # coupling.py:211-213
