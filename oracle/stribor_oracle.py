"""CPU oracle for the stribor coupling-flow hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional restatement (torch CPU ops, fp32 or fp64 depending on the dtype of
the tensors handed in) of the reference algorithm for the path named by BASELINE.json.  It is
NOT part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product path (``stribor_amd``) never does.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py`` against vectors
captured from the unmodified reference imported in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and against the reference's own
known-answer test (stribor/test/test_normalizing_flow.py:45-55) and exact mask vectors
(stribor/test/test_mask.py:4-30).

The reference is a class hierarchy of nn.Modules; the oracle is deliberately a set of pure
functions over a *flow spec*: a list of dicts, one per transform, holding plain tensors.

    {'kind': 'coupling_affine', 'mask': 'ordered_right_half', 'net': NET}
    {'kind': 'coupling_rqs',    'mask': ..., 'net': NET, 'n_bins': K, 'lower': a, 'upper': b,
                                'spline_type': 'quadratic' (default here) | 'cubic'}
    {'kind': 'affine',          'log_scale': [1,D] or [D], 'shift': same}       (no latent_net)
    {'kind': 'affine_lu',       'weight': [D,D], 'log_diag': [1,D], 'bias': [1,D]}
    {'kind': 'matrix_exp',      'weight': [D,D], 'diag': [D], 'bias': [D] or None, 'log_time': bool}
    {'kind': 'permute',         'perm': int64[D]}
    {'kind': 'flip'}
    NET = {'weights': [W0, W1, ...], 'biases': [b0, b1, ...], 'activation': 'Tanh'}  |  {'module': nn.Module}
    couplings may carry 'set_data': True (mask over the set axis, flows/coupling.py:49-51)

All ``file:line`` citations are into /root/reference/stribor/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

MIN_BIN_WIDTH = 1e-3      # util/rational_quadratic_spline.py:22
MIN_BIN_HEIGHT = 1e-3     # util/rational_quadratic_spline.py:23
MIN_DERIVATIVE = 1e-3     # util/rational_quadratic_spline.py:24
SEARCH_EPS = 1e-6         # util/search_sorted.py:3


# ----------------------------------------------------------------------------------------------
# masks                                                                          util/mask.py:6-57
# ----------------------------------------------------------------------------------------------
def mask_vector(name: str, dim: int) -> Tensor:
    """0/1 float vector; 1 = pass-through + conditioner input, 0 = transformed.

    util/mask.py:6-20 (name dispatch), :22-23 (none), :35-45 (ordered), :47-57 (parity).
    'random_half' (:25-33) is redrawn on every call in the reference (quirk Q5) and is out of
    parity scope.
    """
    if name == 'none':
        return torch.tensor([0.0])                                   # mask.py:22-23
    if name in ('ordered_right_half', 'ordered_0', 'ordered_left_half', 'ordered_1'):
        if dim == 1:
            return torch.tensor([1.0])                               # mask.py:37-38
        n_zero = int(np.clip(int(dim * 0.5), 1, dim - 1))            # mask.py:40
        m = torch.ones(dim)
        m[:n_zero] = 0.0                                             # mask.py:41
        if name in ('ordered_left_half', 'ordered_1'):               # right_zero=True, mask.py:12
            m = 1.0 - m                                              # mask.py:42-43
        return m
    if name in ('parity_even', 'parity_odd'):
        if dim == 1:
            return torch.tensor([1.0])                               # mask.py:50-51
        m = torch.ones(dim)
        m[::2] = 0.0                                                 # mask.py:53
        if name == 'parity_odd':                                     # even_zero=True, mask.py:17
            m = 1.0 - m                                              # mask.py:54-55
        return m
    raise NotImplementedError(name)                                  # mask.py:20


# ----------------------------------------------------------------------------------------------
# conditioner MLP                                                               net/mlp.py:48-65
# ----------------------------------------------------------------------------------------------
def mlp_forward(net: Dict, z: Tensor) -> Tensor:
    """Linear -> (act -> Linear)*; weights are torch [out, in].  net/mlp.py:48-58, :65.
    A conditioner that is not a stribor MLP (any nn.Module is allowed as latent_net, affine.py:59-67, spline.py:76-87)
    rides along as {'module': <the torch module on the CPU>} and is simply called."""
    if 'module' in net:
        return net['module'](z)
    act = getattr(torch.nn, net.get('activation', 'Tanh'))()         # mlp.py:38-39
    n = len(net['weights'])
    h = z
    for i in range(n):
        h = F.linear(h, net['weights'][i], net['biases'][i])
        if i + 1 < n:
            h = act(h)
    return h


# ----------------------------------------------------------------------------------------------
# coupling glue                                                           flows/coupling.py:48-95
# ----------------------------------------------------------------------------------------------
def _coupling_mask(layer: Dict, x: Tensor) -> Tensor:
    if layer.get('set_data', False):                                 # coupling.py:49-51: mask over the set axis N
        *rest, N, D = x.shape
        return mask_vector(layer['mask'], N).unsqueeze(-1).expand(*rest, N, D).to(x)
    return mask_vector(layer['mask'], x.shape[-1]).to(x).expand_as(x)   # coupling.py:53


def _conditioning(x: Tensor, m: Tensor, latent: Optional[Tensor]) -> Tensor:
    z = x * m                                                        # coupling.py:61
    if x.shape[-1] == 1:
        z = z * 0                                                    # coupling.py:62-63
    if latent is not None:
        z = torch.cat([z, latent], -1)                               # coupling.py:64-65
    return z


# ----------------------------------------------------------------------------------------------
# elementwise affine                                                   flows/affine.py:59-123
# ----------------------------------------------------------------------------------------------
def affine_params(layer: Dict, z: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    if 'net' in layer and layer['net'] is not None:
        p = mlp_forward(layer['net'], z)
        log_scale, shift = p.chunk(2, dim=-1)                        # affine.py:66
        return log_scale, shift
    return layer['log_scale'], layer['shift']                        # affine.py:63-64


def affine_apply(x: Tensor, log_scale: Tensor, shift: Tensor, reverse: bool) -> Tensor:
    if reverse:
        return (x - shift) * torch.exp(-log_scale)                   # affine.py:106
    return x * torch.exp(log_scale) + shift                          # affine.py:108


# ----------------------------------------------------------------------------------------------
# rational-quadratic spline            util/rational_quadratic_spline.py:11-251, search_sorted.py
# ----------------------------------------------------------------------------------------------
def rqs_params_from_net(p: Tensor, dim: int, n_bins: int) -> Tuple[Tensor, Tensor, Tensor]:
    """[..., D*(3K-1)] -> w[..., D, K], h[..., D, K], d[..., D, K-1].  flows/spline.py:82-86."""
    p = p.view(*p.shape[:-1], dim, 3 * n_bins - 1)
    return p[..., :n_bins], p[..., n_bins:2 * n_bins], p[..., 2 * n_bins:]


def rqs_unconstrained(x: Tensor, uw: Tensor, uh: Tensor, ud: Tensor, inverse: bool,
                      lower: float, upper: float,
                      left=None, right=None, bottom=None, top=None) -> Tuple[Tensor, Tensor]:
    """Per-element restatement of unconstrained_rational_quadratic_spline + rational_quadratic_spline.

    Differences from the reference, all value-preserving:
      * no boolean compaction (:161-164) / scatter (:250): every element is evaluated with its
        input clamped into the domain and the tails are selected at the end (:86-87);
      * the domain check (:167-178, quirk Q1: an accidental [M] x [M,1] broadcast, O(M^2)) is
        vacuous after the `inside` selection and is not reproduced;
      * `assert discriminant >= 0` (:223) is kept as a check on the inside elements.
    Returns (outputs, log-diag-Jacobian); the inverse ljd is already negated (:234).
    """
    if all(v is not None for v in (left, right, bottom, top)):       # :55-61
        lo_in, hi_in = (bottom, top) if inverse else (left, right)
    else:
        left = bottom = lower                                        # :63-64
        right = top = upper
        lo_in, hi_in = lower, upper
    # the reference turns scalar bounds into fp32 tensors (:167-172) before using them in arithmetic
    as_t = lambda v: v.to(x) if torch.is_tensor(v) else torch.tensor(float(v), dtype=x.dtype)
    left, right, bottom, top = as_t(left), as_t(right), as_t(bottom), as_t(top)
    K = uw.shape[-1]
    if MIN_BIN_WIDTH * K > 1.0:
        raise ValueError('Minimal bin width too large for the number of bins')     # :96-97
    if MIN_BIN_HEIGHT * K > 1.0:
        raise ValueError('Minimal bin height too large for the number of bins')    # :98-99

    uw = uw.expand(*x.shape, -1)                                     # :67-69
    uh = uh.expand(*x.shape, -1)
    ud = ud.expand(*x.shape, -1)
    inside = (x >= lo_in) & (x <= hi_in)                             # :71 (closed interval)

    if ud.shape[-1] == K - 1:                                        # :79-83
        const = float(np.log(np.exp(1 - MIN_DERIVATIVE) - 1))
        ud = F.pad(ud, pad=(1, 1))
        ud = ud.clone()
        ud[..., 0] = const
        ud[..., -1] = const

    w = MIN_BIN_WIDTH + (1 - MIN_BIN_WIDTH * K) * F.softmax(uw, dim=-1)     # :101-102
    h = MIN_BIN_HEIGHT + (1 - MIN_BIN_HEIGHT * K) * F.softmax(uh, dim=-1)   # :104-105
    d = MIN_DERIVATIVE + F.softplus(ud)                                      # :107

    # knots: cumsum -> pad -> rescale -> pin ends -> re-difference             :180-192
    cw = F.pad(torch.cumsum(w, dim=-1), pad=(1, 0), value=0.0)
    cw = (right - left) * cw + left
    cw[..., 0] = left
    cw[..., -1] = right
    w = cw[..., 1:] - cw[..., :-1]
    ch = F.pad(torch.cumsum(h, dim=-1), pad=(1, 0), value=0.0)
    ch = (top - bottom) * ch + bottom
    ch[..., 0] = bottom
    ch[..., -1] = top
    h = ch[..., 1:] - ch[..., :-1]

    xin = torch.where(inside, x, torch.full_like(x, float(lo_in)))   # keep tails finite
    edges = (ch if inverse else cw).clone()                          # :194-197
    edges[..., -1] += SEARCH_EPS                                     # search_sorted.py:4 (in place there)
    b = (torch.sum(xin[..., None] >= edges, dim=-1) - 1)[..., None]  # search_sorted.py:5
    b = b.clamp(0, K - 1)                                            # only hit by discarded tail lanes

    g = lambda t: t.gather(-1, b)[..., 0]
    cw_b, w_b = g(cw), g(w)                                          # :199-200
    ch_b = g(ch)                                                     # :202
    delta = h / w                                                    # :203
    s_b = g(delta)                                                   # :204
    d_b = g(d)                                                       # :206
    d_b1 = g(d[..., 1:])                                             # :207
    h_b = g(h)                                                       # :209

    if inverse:
        dy = xin - ch_b
        q = d_b + d_b1 - 2 * s_b
        a = dy * q + h_b * (s_b - d_b)                               # :212-215
        bb = h_b * d_b - dy * q                                      # :216-219
        c = -s_b * dy                                                # :220
        disc = bb.pow(2) - 4 * a * c                                 # :222
        if not bool((disc[inside] >= 0).all()):                      # :223
            raise AssertionError('negative discriminant')
        root = (2 * c) / (-bb - torch.sqrt(disc))                    # :225
        out = root * w_b + cw_b                                      # :226
        tomt = root * (1 - root)                                     # :228
        den = s_b + q * tomt                                         # :229-230
        dnum = s_b.pow(2) * (d_b1 * root.pow(2) + 2 * s_b * tomt + d_b * (1 - root).pow(2))  # :231-233
        ljd = -torch.log(dnum) + 2 * torch.log(den)                  # :234 (sign already flipped)
    else:
        theta = (xin - cw_b) / w_b                                   # :236
        tomt = theta * (1 - theta)                                   # :237
        num = h_b * (s_b * theta.pow(2) + d_b * tomt)                # :239-240
        den = s_b + (d_b + d_b1 - 2 * s_b) * tomt                    # :241-242
        out = ch_b + num / den                                       # :243
        dnum = s_b.pow(2) * (d_b1 * theta.pow(2) + 2 * s_b * tomt + d_b * (1 - theta).pow(2))  # :245-247
        ljd = torch.log(dnum) - 2 * torch.log(den)                   # :248

    out = torch.where(inside, out, x)                                # :86
    ljd = torch.where(inside, ljd, torch.zeros_like(ljd))            # :87
    return out, ljd


# ----------------------------------------------------------------------------------------------
# monotone cubic spline (spline_type='cubic', the reference's default)    util/cubic_spline.py:21-251
# ----------------------------------------------------------------------------------------------
CUBIC_MIN_BIN_WIDTH = 1e-2       # util/cubic_spline.py:13
CUBIC_MIN_BIN_HEIGHT = 1e-2      # util/cubic_spline.py:14
CUBIC_EPS = 1e-5                 # util/cubic_spline.py:15
CUBIC_QUADRATIC_THRESHOLD = 1e-3  # util/cubic_spline.py:16


def cubic_params_from_net(p: Tensor, dim: int, n_bins: int) -> Tuple[Tensor, Tensor, Tensor]:
    """[..., D*(2K+2)] -> w[..., D, K], h[..., D, K], d[..., D, 2].  flows/spline.py:82-86."""
    p = p.view(*p.shape[:-1], dim, 2 * n_bins + 2)
    return p[..., :n_bins], p[..., n_bins:2 * n_bins], p[..., 2 * n_bins:]


def _cbrt(x: Tensor) -> Tensor:
    return torch.sign(x) * torch.exp(torch.log(torch.abs(x)) / 3.0)  # cubic_spline.py:18-20


def cubic_unconstrained(x: Tensor, uw: Tensor, uh: Tensor, ud: Tensor, inverse: bool,
                        lower: float, upper: float) -> Tuple[Tensor, Tensor]:
    """Per-element restatement of unconstrained_cubic_spline + cubic_spline (cubic_spline.py:21-251).

    The in-domain elements are compacted first, as in the reference (which makes its domain check :86-89 vacuous).
    The inverse's root selection keeps the reference's precedence (:170-222): one-root formula where the
    discriminant is <= 0, else the first of the three trigonometric roots (order 1, 2, 3) that lies in the bin
    (what torch.argsort(descending=True)[..., 0] returns on CPU; root 1 if none does), both overwritten by the
    quadratic formula where |a| < 1e-3.
    """
    K = uw.shape[-1]
    if CUBIC_MIN_BIN_WIDTH * K > 1.0:
        raise ValueError('Minimal bin width too large for the number of bins')     # :93-94
    if CUBIC_MIN_BIN_HEIGHT * K > 1.0:
        raise ValueError('Minimal bin height too large for the number of bins')    # :95-96
    left = bottom = lower
    right = top = upper
    uw = uw.expand(*x.shape, -1)                                     # :35-36
    uh = uh.expand(*x.shape, -1)
    udl = ud[..., 0, None].expand(*x.shape, -1)                      # :37-38
    udr = ud[..., 1, None].expand(*x.shape, -1)
    inside = (x >= lower) & (x <= upper)                             # :40
    out_full = x.clone()                                             # :43-48 linear tails: identity, ldj 0
    ljd_full = torch.zeros_like(x)
    if not bool(inside.any()):                                       # :52-53
        return out_full, ljd_full
    # boolean compaction as in the reference (:55-60): torch's vectorised CPU kernels are not position-independent
    # to the last bit and the golden vectors pin this function bit for bit
    xin, uw, uh, udl, udr = x[inside], uw[inside, :], uh[inside, :], udl[inside, :], udr[inside, :]
    if inverse:
        xin = (xin - bottom) / (top - bottom)                        # :98-99
    else:
        xin = (xin - left) / (right - left)                          # :100-101

    w = F.softmax(uw, dim=-1)                                        # :103-104
    w = CUBIC_MIN_BIN_WIDTH + (1 - CUBIC_MIN_BIN_WIDTH * K) * w
    cw = torch.cumsum(w, dim=-1)                                     # :106-108
    cw[..., -1] = 1
    cw = F.pad(cw, pad=(1, 0), mode='constant', value=0.0)
    h = F.softmax(uh, dim=-1)                                        # :110-111
    h = CUBIC_MIN_BIN_HEIGHT + (1 - CUBIC_MIN_BIN_HEIGHT * K) * h
    ch = torch.cumsum(h, dim=-1)                                     # :113-115
    ch[..., -1] = 1
    ch = F.pad(ch, pad=(1, 0), mode='constant', value=0.0)

    slopes = h / w                                                   # :117
    m1 = torch.min(torch.abs(slopes[..., :-1]), torch.abs(slopes[..., 1:]))            # :118-119
    m2 = (0.5 * (w[..., 1:] * slopes[..., :-1] + w[..., :-1] * slopes[..., 1:])
          / (w[..., :-1] + w[..., 1:]))                              # :120-123
    ms = torch.min(m1, m2)                                           # :124
    dl = torch.sigmoid(udl) * 3 * slopes[..., 0][..., None]          # :126
    dr = torch.sigmoid(udr) * 3 * slopes[..., -1][..., None]         # :127
    dv = ms * (torch.sign(slopes[..., :-1]) + torch.sign(slopes[..., 1:]))              # :129
    dv = torch.cat([dl, dv, dr], dim=-1)                             # :130-132

    a = (dv[..., :-1] + dv[..., 1:] - 2 * slopes) / w.pow(2)         # :134
    b = (3 * slopes - 2 * dv[..., :-1] - dv[..., 1:]) / w            # :135
    c = dv[..., :-1]                                                 # :136
    d = ch[..., :-1]                                                 # :137

    edges = (ch if inverse else cw).clone()                          # :139-142
    edges[..., -1] += SEARCH_EPS                                     # search_sorted.py:4
    idx = (torch.sum(xin[..., None] >= edges, dim=-1) - 1)[..., None]
    g = lambda t: t.gather(-1, idx)[..., 0]
    ia, ib, ic, id_ = g(a), g(b), g(c), g(d)                         # :144-147
    lcw = g(cw)                                                      # :149
    rcw = cw.gather(-1, idx + 1)[..., 0]                             # :150

    if inverse:
        b_ = (ib / ia) / 3.                                          # :154-156
        c_ = (ic / ia) / 3.
        d_ = (id_ - xin) / ia
        delta_1 = -b_.pow(2) + c_                                    # :158-160
        delta_2 = -c_ * b_ + d_
        delta_3 = b_ * d_ - c_.pow(2)
        disc = 4. * delta_1 * delta_3 - delta_2.pow(2)               # :162
        dep1 = -2. * b_ * delta_1 + delta_2                          # :164-165
        dep2 = delta_1
        three = disc > 0                                             # :167-168
        one = ~three
        # Subsets are compacted as in the reference: torch's vectorised CPU kernels are not position-independent to
        # the last bit, and the golden vectors pin this function bit for bit.
        out = torch.zeros_like(xin)                                  # :170
        sq = torch.sqrt(-disc[one])                                  # :174-175
        pp = _cbrt((-dep1[one] + sq) / 2.)
        qq = _cbrt((-dep1[one] - sq) / 2.)
        out[one] = (pp + qq) - b_[one] + lcw[one]                    # :177-179
        theta = torch.atan2(torch.sqrt(disc[three]), -dep1[three])   # :183-184
        theta = theta / 3.
        cr1, cr2 = torch.cos(theta), torch.sin(theta)                # :186-187
        r1 = cr1                                                     # :189-191
        r2 = -0.5 * cr1 - 0.5 * math.sqrt(3) * cr2
        r3 = -0.5 * cr1 + 0.5 * math.sqrt(3) * cr2
        scale = 2 * torch.sqrt(-dep2[three])                         # :193
        shift = (-b_[three] + lcw[three])                            # :194
        r1, r2, r3 = r1 * scale + shift, r2 * scale + shift, r3 * scale + shift        # :196-198
        lo3, hi3 = lcw[three] - CUBIC_EPS, rcw[three] + CUBIC_EPS
        k1, k2, k3 = (lo3 < r1) & (r1 < hi3), (lo3 < r2) & (r2 < hi3), (lo3 < r3) & (r3 < hi3)   # :200-207
        # argsort(masks, descending)[..., 0] on CPU = the first root whose mask is 1, root 1 if none (:209-212)
        out[three] = torch.where(k1, r1, torch.where(k2, r2, torch.where(k3, r3, r1)))
        quad = ia.abs() < CUBIC_QUADRATIC_THRESHOLD                  # :216-222
        qa, qb, qc = ib[quad], ic[quad], (id_[quad] - xin[quad])
        alpha = (-qb + torch.sqrt(qb.pow(2) - 4 * qa * qc)) / (2 * qa)
        out[quad] = alpha + lcw[quad]
        sh_out = out - lcw                                           # :218
        ljd = -torch.log(3 * ia * sh_out.pow(2) + 2 * ib * sh_out + ic)                 # :219-221
        out = out * (right - left) + left                            # :235
        ljd = ljd - math.log(top - bottom) + math.log(right - left)  # :236
    else:
        t = xin - lcw                                                # :223
        out = ia * t.pow(3) + ib * t.pow(2) + ic * t + id_           # :224-227
        ljd = torch.log(3 * ia * t.pow(2) + 2 * ib * t + ic)         # :229-231
        out = out * (top - bottom) + bottom                          # :238
        ljd = ljd + math.log(top - bottom) - math.log(right - left)  # :239
    out_full[inside] = out                                           # :55
    ljd_full[inside] = ljd
    return out_full, ljd_full


def rqs_from_layer(layer: Dict, x: Tensor, z: Optional[Tensor], reverse: bool) -> Tuple[Tensor, Tensor]:
    """Spline.forward_and_log_diag_jacobian, flows/spline.py:101-105."""
    D, K = x.shape[-1], layer['n_bins']
    cubic = layer.get('spline_type', 'quadratic') == 'cubic'         # spline.py:56-61
    if layer.get('net') is not None:
        uw, uh, ud = (cubic_params_from_net if cubic else rqs_params_from_net)(mlp_forward(layer['net'], z), D, K)
    else:
        uw, uh, ud = layer['width'], layer['height'], layer['derivative']        # spline.py:78-79
    if cubic:
        return cubic_unconstrained(x, uw, uh, ud, reverse, layer.get('lower', 0), layer.get('upper', 1))
    return rqs_unconstrained(x, uw, uh, ud, reverse, layer.get('lower', 0), layer.get('upper', 1))


# ----------------------------------------------------------------------------------------------
# time-conditioned affine coupling                        flows/coupling.py:98-213, net/time_net.py:6-47
#   {'kind': 'continuous_affine_coupling', 'mask': ..., 'net': NET, 'time_kind': 'identity'|'linear'|'tanh'|'log'|'fourier'|'fourier_bounded',
#    'time_scale': [1, out] | None, 'concatenate_time': bool}
# ----------------------------------------------------------------------------------------------
def time_embed(layer: Dict, t: Tensor) -> Tensor:
    kind, sc = layer['time_kind'], layer.get('time_scale')
    if kind == 'identity':
        return t.repeat_interleave(layer['time_out'], dim=-1)       # time_net.py:11
    if kind == 'linear':
        return sc * t                                                # time_net.py:23
    if kind == 'tanh':
        return torch.tanh(sc * t)                                    # time_net.py:30
    if kind == 'log':
        return torch.log(sc.exp() * t + 1)                           # time_net.py:38
    if kind in ('fourier', 'fourier_bounded'):                       # time_net.py:49-91
        w, sh = layer['time_weight'], layer['time_shift']
        scale = F.softmax(w, -1) / 2 if kind == 'fourier_bounded' else w / w.shape[-1]      # :67-71
        return (scale * torch.sin(sh * t.unsqueeze(-1))).sum(-1)     # :73-78
    raise ValueError(kind)


def continuous_affine_coupling(layer: Dict, x: Tensor, t: Tensor, latent: Optional[Tensor], reverse: bool
                               ) -> Tuple[Tensor, Tensor]:
    """ContinuousAffineCoupling.forward_and_log_det_jacobian, coupling.py:184-205."""
    m = _coupling_mask(layer, x)                                     # :192
    z = x * m                                                        # :149-156
    if x.shape[-1] == 1:
        z = z * 0
    if latent is not None:
        z = torch.cat([z, latent], -1)
    if layer.get('concatenate_time', True):
        z = torch.cat([z, t], -1)
    log_scale, shift = mlp_forward(layer['net'], z).chunk(2, dim=-1)               # :194
    t_log_scale, t_shift = time_embed(layer, t).chunk(2, dim=-1)                   # :195
    if reverse:
        y = (x - shift * t_shift) * torch.exp(-log_scale * t_log_scale)            # :197
    else:
        y = x * torch.exp(log_scale * t_log_scale) + shift * t_shift               # :199
    ldiag = log_scale * t_log_scale * (1 - m)                                      # :201
    y = y * (1 - m) + x * m                                                        # :203
    return y, ldiag.sum(-1, keepdim=True)


def neural_flow_forward(spec: Sequence[Dict], x: Tensor, t: Tensor, t0: Optional[Tensor] = None, latent=None) -> Tensor:
    """NeuralFlow.forward, flow.py:170-184."""
    if t0 is not None:
        for layer in reversed(spec):
            x = continuous_affine_coupling(layer, x, t0, latent, True)[0]           # transform.inverse(x, t=t0)
    for layer in spec:
        x = continuous_affine_coupling(layer, x, t, latent, False)[0]
    return x


# ----------------------------------------------------------------------------------------------
# parameter-free element-wise flows       flows/sigmoid.py, flows/activations.py:11-101, flows/cumsum.py:9-92
#   {'kind': 'sigmoid' | 'logit' | 'elu' | 'leaky_relu' (+ 'negative_slope') | 'cumsum' | 'diff' | 'identity'}
# ----------------------------------------------------------------------------------------------
POINTWISE = ('sigmoid', 'logit', 'elu', 'leaky_relu', 'cumsum', 'diff', 'identity')


def _sigmoid_fwd(x: Tensor) -> Tensor:
    finfo = torch.finfo(x.dtype)
    return torch.clamp(torch.sigmoid(x), min=finfo.tiny, max=1. - finfo.eps)      # sigmoid.py:21-23


def _sigmoid_inv(y: Tensor) -> Tensor:
    finfo = torch.finfo(y.dtype)
    y = y.clamp(min=finfo.tiny, max=1.0 - finfo.eps)                # sigmoid.py:28-30
    return y.log() - (-y).log1p()


def _leaky(x: Tensor, slope: float) -> Tensor:
    zeros = torch.zeros_like(x)
    return torch.max(zeros, x) + slope * torch.min(zeros, x)         # activations.py:84-86


def pointwise_apply(layer: Dict, x: Tensor, reverse: bool) -> Tensor:
    kind = layer['kind']
    if kind == 'identity':
        return x                                                     # identity.py:14-18
    if kind in ('sigmoid', 'logit'):
        return _sigmoid_fwd(x) if (kind == 'sigmoid') != reverse else _sigmoid_inv(x)     # sigmoid.py:46-53
    if kind == 'elu':
        if not reverse:
            return F.elu(x)                                          # activations.py:27
        zero = torch.zeros_like(x)
        return torch.max(x, zero) + torch.min(torch.log1p(x), zero)  # activations.py:34-37
    if kind == 'leaky_relu':
        s = layer.get('negative_slope', 0.01)
        return _leaky(x, 1 / s if reverse else s)                    # activations.py:88-92
    if kind in ('cumsum', 'diff'):
        if (kind == 'cumsum') != reverse:
            return x.cumsum(-1)                                      # cumsum.py:62
        return x - F.pad(x, (1, 0))[..., :-1]                        # cumsum.py:33
    raise ValueError(kind)


def pointwise_log_diag(layer: Dict, x: Tensor, y: Optional[Tensor] = None) -> Tensor:
    """log_diag_jacobian(x, y) of the forward transform at x (Logit evaluates it at the y it is handed)."""
    kind = layer['kind']
    if kind == 'sigmoid':
        return -F.softplus(-x) - F.softplus(x)                       # sigmoid.py:44
    if kind == 'logit':
        if y is None:
            y = _sigmoid_inv(x)                                      # Logit.forward
        return -(-F.softplus(-y) - F.softplus(y))                    # sigmoid.py:56 (sign and order flipped)
    if kind == 'elu':
        return -F.relu(-x)                                           # activations.py:63
    if kind == 'leaky_relu':
        s = layer.get('negative_slope', 0.01)
        return torch.where(x >= 0., torch.zeros_like(x), torch.ones_like(x) * math.log(s))   # activations.py:99-101
    return torch.zeros_like(x)                                       # cumsum.py:74, identity.py:24


# ----------------------------------------------------------------------------------------------
# per-transform forward / inverse / log_det_jacobian                       flow.py:8-47 protocol
# ----------------------------------------------------------------------------------------------
def _get_time(layer: Dict, t, shape) -> Tensor:
    if not torch.is_tensor(t):
        t = torch.ones(*shape[:-1], 1) * t                           # affine.py:237-238
    if layer.get('log_time', False):
        t = torch.log1p(t.abs())                                     # affine.py:239-240
    return t


def _matexp_lu(layer: Dict) -> Tuple[Tensor, Tensor]:
    W = layer['weight']
    eye = torch.eye(W.shape[0]).to(W)
    return torch.tril(W, diagonal=-1) + eye, torch.triu(W) + eye     # affine.py:222-226


def transform_apply(layer: Dict, x: Tensor, reverse: bool, latent: Optional[Tensor] = None,
                    t=1.0) -> Tensor:
    """f(x) (reverse=False) or f.inverse(x) (reverse=True) for one transform."""
    kind = layer['kind']
    if kind in ('coupling_affine', 'coupling_rqs'):
        m = _coupling_mask(layer, x)                                 # coupling.py:70
        z = _conditioning(x, m, latent)                              # coupling.py:71
        if kind == 'coupling_affine':
            ls, sh = affine_params(layer, z)
            y_ = affine_apply(x, ls, sh, reverse)                    # coupling.py:73-76
        else:
            y_, _ = rqs_from_layer(layer, x, z, reverse)
        return y_ * (1 - m) + x * m                                  # coupling.py:78
    if kind == 'affine':
        ls, sh = affine_params(layer, latent)
        return affine_apply(x, ls, sh, reverse)
    if kind == 'rqs':
        y, _ = rqs_from_layer(layer, x, latent, reverse)
        return y
    if kind == 'affine_lu':
        W = layer['weight']
        eye = torch.eye(W.shape[0]).to(W)
        L = torch.tril(W, -1) + eye                                  # affine.py:148-150
        U = torch.triu(W, 1) + eye * layer['log_diag'].exp()         # affine.py:152-154
        if reverse:
            v = x - layer['bias']                                    # affine.py:160
            v = torch.linalg.solve_triangular(U, v, upper=True, left=False)     # :161
            return torch.linalg.solve_triangular(L, v, upper=False, left=False)  # :162
        return x @ (L @ U) + layer['bias']                           # affine.py:157
    if kind == 'matrix_exp':
        tt = _get_time(layer, t, x.shape).to(x)                      # affine.py:251
        bias = layer.get('bias', None)
        if reverse:
            tt = -tt                                                 # affine.py:254
            if bias is not None:
                x = x - bias                                         # affine.py:255-256
        L, U = _matexp_lu(layer)
        v = torch.linalg.solve_triangular(L, x.unsqueeze(-1), upper=False, unitriangular=True).squeeze(-1)  # :260
        v = torch.linalg.solve_triangular(U, v.unsqueeze(-1), upper=True, unitriangular=False).squeeze(-1)  # :261
        v = v * (layer['diag'] * tt).exp()                           # :263
        v = F.linear(v, U)                                           # :265
        v = F.linear(v, L)                                           # :266
        if not reverse and bias is not None:
            v = v + bias                                             # :268-269
        return v
    if kind == 'permute':
        perm = layer['perm']
        if reverse:
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel())                   # permute.py:67-68
            return x[..., inv]                                       # permute.py:75
        return x[..., perm]                                          # permute.py:71
    if kind == 'flip':
        return torch.flip(x, [-1])                                   # permute.py:35,38
    if kind in POINTWISE:
        return pointwise_apply(layer, x, reverse)
    raise ValueError(kind)


def transform_ldj(layer: Dict, x: Tensor, latent: Optional[Tensor] = None, t=1.0, y: Optional[Tensor] = None) -> Tensor:
    """f.log_det_jacobian(x, y): forward-direction log|det J| at x, shape [..., 1] (only Logit reads y)."""
    kind = layer['kind']
    if kind in ('coupling_affine', 'coupling_rqs'):
        m = _coupling_mask(layer, x)                                 # coupling.py:91
        z = _conditioning(x, m, latent)                              # coupling.py:92 (2nd MLP call, Q2)
        if kind == 'coupling_affine':
            ls, _ = affine_params(layer, z)                          # affine.py:122
            diag = ls.expand_as(x)                                   # affine.py:123
        else:
            _, diag = rqs_from_layer(layer, x, z, False)             # spline.py:142-143
        return (diag * (1 - m)).sum(-1, keepdim=True)                # coupling.py:95
    if kind == 'affine':
        ls, _ = affine_params(layer, latent)
        return ls.expand_as(x).sum(-1, keepdim=True)                 # affine.py:109
    if kind == 'rqs':
        _, diag = rqs_from_layer(layer, x, latent, False)
        return diag.sum(-1, keepdim=True)                            # spline.py:117
    if kind == 'affine_lu':
        return layer['log_diag'].expand_as(x).sum(-1, keepdim=True)  # affine.py:171
    if kind == 'matrix_exp':
        tt = _get_time(layer, t, x.shape).to(x)
        return layer['diag'].sum() * tt                              # affine.py:287-288
    if kind in ('permute', 'flip'):
        return torch.zeros_like(x[..., :1])                          # permute.py:41,78
    if kind in POINTWISE:
        return pointwise_log_diag(layer, x, y).sum(-1, keepdim=True)  # sigmoid.py:36, activations.py:44,93, cumsum.py:70
    raise ValueError(kind)


def transform_inverse_and_ldj(layer: Dict, y: Tensor, latent=None, t=1.0) -> Tuple[Tensor, Tensor]:
    """Transform.inverse_and_log_det_jacobian default, flow.py:42-47 (conditioner runs twice)."""
    x = transform_apply(layer, y, True, latent, t)
    return x, -transform_ldj(layer, x, latent, t, y)


def transform_forward_and_ldj(layer: Dict, x: Tensor, latent=None, t=1.0) -> Tuple[Tensor, Tensor]:
    """Transform.forward_and_log_det_jacobian default, flow.py:35-40."""
    y = transform_apply(layer, x, False, latent, t)
    return y, transform_ldj(layer, x, latent, t, y)


# ----------------------------------------------------------------------------------------------
# base density + flow container                                 dist/normal.py:37,52-54; flow.py
# ----------------------------------------------------------------------------------------------
def unit_normal_log_prob(x: Tensor) -> Tensor:
    """Independent(Normal(0,1),1).log_prob: sum_d [-(x-0)^2/(2*1) - log(1) - log(sqrt(2pi))]."""
    var = 1.0
    lp = -((x - 0.0) ** 2) / (2 * var) - math.log(1.0) - math.log(math.sqrt(2 * math.pi))
    return lp.sum(-1)


def flow_inverse_and_ldj(spec: Sequence[Dict], y: Tensor, latent=None, t=1.0,
                         trace: Optional[List] = None) -> Tuple[Tensor, Tensor]:
    """NormalizingFlow.inverse_and_log_det_jacobian, flow.py:118-125."""
    acc = 0
    for layer in reversed(list(spec)):
        y, ldj = transform_inverse_and_ldj(layer, y, latent, t)
        acc = acc + ldj
        if trace is not None:
            trace.append((y, ldj))
    return y, acc


def flow_forward_and_ldj(spec: Sequence[Dict], x: Tensor, latent=None, t=1.0) -> Tuple[Tensor, Tensor]:
    """NormalizingFlow.forward_and_log_det_jacobian, flow.py:109-116."""
    acc = 0
    for layer in spec:
        x, ldj = transform_forward_and_ldj(layer, x, latent, t)
        acc = acc + ldj
    return x, acc


def flow_forward(spec: Sequence[Dict], x: Tensor, latent=None, t=1.0) -> Tensor:
    for layer in spec:                                               # flow.py:99-102
        x = transform_apply(layer, x, False, latent, t)
    return x


def flow_inverse(spec: Sequence[Dict], y: Tensor, latent=None, t=1.0) -> Tensor:
    for layer in reversed(list(spec)):                               # flow.py:104-107
        y = transform_apply(layer, y, True, latent, t)
    return y


def flow_log_prob(spec: Sequence[Dict], y: Tensor, latent=None, t=1.0) -> Tensor:
    """NormalizingFlow.log_prob with a UnitNormal base, flow.py:127-130 -> [..., 1]."""
    x, acc = flow_inverse_and_ldj(spec, y, latent, t)
    return unit_normal_log_prob(x).unsqueeze(-1) + acc


def spec_to(spec: Sequence[Dict], dtype: torch.dtype) -> List[Dict]:
    """Deep-copy a spec casting floating tensors (used for the fp64 'truth' runs)."""
    def cv(v):
        if torch.is_tensor(v):
            return v.to(dtype) if v.is_floating_point() else v.clone()
        if isinstance(v, dict):
            return {k: cv(u) for k, u in v.items()}
        if isinstance(v, list):
            return [cv(u) for u in v]
        return v
    return [cv(l) for l in spec]
