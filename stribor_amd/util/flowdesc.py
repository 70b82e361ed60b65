"""Shared test helper: a tiny JSON-able description of a flow, from which we can build

  * the module tree of any package exposing stribor's constructor surface
    (the reference, in tests/golden/make_golden.py only, or ``stribor_amd``), and
  * the oracle's functional spec, given a ``state_dict`` with stribor's keys.

A description is a list of dicts, e.g.
    {"kind": "coupling_affine", "dim": 64, "hidden": [64], "mask": "ordered_right_half", "latent_dim": 0}
    {"kind": "coupling_rqs", "dim": 64, "hidden": [64], "mask": "...", "n_bins": 16, "lower": -3, "upper": 3}
    {"kind": "affine", "dim": 2}                      learnable elementwise affine (no latent_net)
    {"kind": "affine_latent", "dim": 5, "hidden": [32], "latent_dim": 13}
    {"kind": "rqs", "dim": 5, "n_bins": 3, "lower": 0, "upper": 2, "hidden": [12], "latent_dim": 0}
    {"kind": "affine_lu", "dim": 128}
    {"kind": "matrix_exp", "dim": 128, "bias": false, "log_time": false}
    {"kind": "permute", "dim": 64}
    {"kind": "flip"}
"""
from __future__ import annotations

from typing import Dict, List

import torch


POINTWISE = ('sigmoid', 'logit', 'elu', 'leaky_relu', 'cumsum', 'diff', 'identity')


class HandNet(torch.nn.Module):
    """A conditioner that is NOT a stribor MLP (residual branch, LayerNorm, two activations): stands for "any nn.Module
    latent_net" (the reference accepts one: flows/affine.py:59-67, flows/spline.py:76-87).  Plain torch, so the same
    class serves the reference, stribor_amd and the oracle."""

    def __init__(self, in_dim: int, hidden: int, out_dim: int):
        super().__init__()
        self.a = torch.nn.Linear(in_dim, hidden)
        self.n = torch.nn.LayerNorm(hidden)
        self.b = torch.nn.Linear(hidden, hidden)
        self.c = torch.nn.Linear(hidden, out_dim)

    def forward(self, z):
        u = torch.nn.functional.silu(self.a(z))
        u = u + torch.sin(self.b(self.n(u)))
        return self.c(u)


def _make_net(st, d: Dict, in_dim: int, out_dim: int):
    if d.get('net', 'mlp') == 'hand':
        return HandNet(in_dim, d['hidden'][0], out_dim)
    return st.net.MLP(in_dim, list(d['hidden']), out_dim)


def _spline_params(d: Dict) -> int:
    """parameters per element: 3K-1 (quadratic) or 2K+2 (cubic), flows/spline.py:56-61."""
    return 2 * d['n_bins'] + 2 if d.get('spline_type', 'quadratic') == 'cubic' else 3 * d['n_bins'] - 1


def build_transform(st, d: Dict):
    k = d['kind']
    if k == 'coupling_affine':
        dim, ld = d['dim'], d.get('latent_dim', 0)
        net = _make_net(st, d, dim + ld, 2 * dim)
        return st.Coupling(transform=st.Affine(dim, latent_net=net), mask=d['mask'], set_data=d.get('set_data', False))
    if k == 'coupling_rqs':
        dim, ld, K = d['dim'], d.get('latent_dim', 0), d['n_bins']
        net = _make_net(st, d, dim + ld, dim * _spline_params(d))
        sp = st.Spline(dim, K, latent_net=net, lower=d['lower'], upper=d['upper'],
                       spline_type=d.get('spline_type', 'quadratic'))
        return st.Coupling(transform=sp, mask=d['mask'], set_data=d.get('set_data', False))
    if k == 'affine':
        return st.Affine(d['dim'])
    if k == 'affine_latent':
        return st.Affine(d['dim'], latent_net=st.net.MLP(d['latent_dim'], list(d['hidden']), 2 * d['dim']))
    if k == 'rqs':
        dim, ld, K = d['dim'], d.get('latent_dim', 0), d['n_bins']
        net = st.net.MLP(ld, list(d['hidden']), dim * _spline_params(d)) if ld else None
        return st.Spline(dim=dim, n_bins=K, latent_net=net, lower=d['lower'], upper=d['upper'],
                         spline_type=d.get('spline_type', 'quadratic'))
    if k == 'affine_lu':
        return st.AffineLU(d['dim'])
    if k == 'matrix_exp':
        return st.MatrixExponential(d['dim'], bias=d.get('bias', False), log_time=d.get('log_time', False))
    if k == 'permute':
        return st.Permute(d['dim'])
    if k == 'flip':
        return st.Flip([-1])
    if k == 'continuous_affine_coupling':
        dim, ld = d['dim'], d.get('latent_dim', 0)
        cat = d.get('concatenate_time', True)
        net = _make_net(st, d, dim + ld + (1 if cat else 0), 2 * dim)
        if d['time_kind'] in ('fourier', 'fourier_bounded'):
            cls = st.net.TimeFourier if d['time_kind'] == 'fourier' else st.net.TimeFourierBounded
            tn = cls(d.get('time_out', 2 * dim), d.get('time_hidden', 5))
        else:
            tn = {'identity': st.net.TimeIdentity, 'linear': st.net.TimeLinear, 'tanh': st.net.TimeTanh,
                  'log': st.net.TimeLog}[d['time_kind']](d.get('time_out', 2 * dim))
        return st.ContinuousAffineCoupling(latent_net=net, time_net=tn, mask=d['mask'], concatenate_time=cat)
    if k in POINTWISE:
        return {'sigmoid': st.Sigmoid, 'logit': st.Logit, 'elu': st.ELU, 'identity': st.Identity,
                'leaky_relu': lambda: st.LeakyReLU(d.get('negative_slope', 0.01)),
                'cumsum': lambda: st.Cumsum(-1), 'diff': lambda: st.Diff(-1)}[k]()
    raise ValueError(k)


def build_flow(st, desc: List[Dict], dim: int):
    return st.NormalizingFlow(st.UnitNormal(dim), [build_transform(st, d) for d in desc])


def _net_spec(state: Dict[str, torch.Tensor], prefix: str, d: Dict = None, in_dim: int = 0, out_dim: int = 0) -> Dict:
    """MLP state_dict keys are '<prefix>net.{0,2,4,...}.{weight,bias}' (net/mlp.py:48-58)."""
    if d is not None and d.get('net', 'mlp') == 'hand':
        m = HandNet(in_dim, d['hidden'][0], out_dim)
        m.load_state_dict({k[len(prefix):]: v.detach().cpu() for k, v in state.items() if k.startswith(prefix)})
        return {'module': m.eval()}
    ws, bs, i = [], [], 0
    while f'{prefix}net.{i}.weight' in state:
        ws.append(state[f'{prefix}net.{i}.weight'])
        bs.append(state[f'{prefix}net.{i}.bias'])
        i += 2
    assert ws, f'no MLP under {prefix}'
    return {'weights': ws, 'biases': bs, 'activation': 'Tanh'}


def transform_spec(d: Dict, state: Dict[str, torch.Tensor], prefix: str) -> Dict:
    """Oracle spec of one transform from stribor-keyed state ('<prefix>...')."""
    k = d['kind']
    if k == 'coupling_affine':
        return {'kind': k, 'mask': d['mask'], 'set_data': d.get('set_data', False),
                'net': _net_spec(state, prefix + 'transform.latent_net.', d, d['dim'] + d.get('latent_dim', 0), 2 * d['dim'])}
    if k == 'coupling_rqs':
        return {'kind': k, 'mask': d['mask'], 'set_data': d.get('set_data', False),
                'net': _net_spec(state, prefix + 'transform.latent_net.', d, d['dim'] + d.get('latent_dim', 0),
                                 d['dim'] * _spline_params(d)),
                'n_bins': d['n_bins'], 'lower': d['lower'], 'upper': d['upper'],
                'spline_type': d.get('spline_type', 'quadratic')}
    if k == 'affine':
        return {'kind': 'affine', 'log_scale': state[prefix + 'log_scale'], 'shift': state[prefix + 'shift']}
    if k == 'affine_latent':
        return {'kind': 'affine', 'net': _net_spec(state, prefix + 'latent_net.')}
    if k == 'rqs':
        s = {'kind': 'rqs', 'n_bins': d['n_bins'], 'lower': d['lower'], 'upper': d['upper'],
             'spline_type': d.get('spline_type', 'quadratic')}
        if d.get('latent_dim', 0):
            s['net'] = _net_spec(state, prefix + 'latent_net.')
        else:
            s['net'] = None
            s['width'], s['height'], s['derivative'] = (state[prefix + n] for n in ('width', 'height', 'derivative'))
        return s
    if k == 'affine_lu':
        return {'kind': k, 'weight': state[prefix + 'weight'], 'log_diag': state[prefix + 'log_diag'],
                'bias': state[prefix + 'bias']}
    if k == 'matrix_exp':
        return {'kind': k, 'weight': state[prefix + '_weight'], 'diag': state[prefix + 'diag'],
                'bias': state.get(prefix + 'bias', None), 'log_time': d.get('log_time', False)}
    if k == 'permute':
        return {'kind': k, 'perm': state[prefix + 'permutation'].long()}
    if k == 'flip':
        return {'kind': k}
    if k in POINTWISE:
        return dict(d)
    if k == 'continuous_affine_coupling':
        cat = d.get('concatenate_time', True)
        return {'kind': k, 'mask': d['mask'],
                'net': _net_spec(state, prefix + 'latent_net.', d, d['dim'] + d.get('latent_dim', 0) + (1 if cat else 0), 2 * d['dim']),
                'time_kind': d['time_kind'], 'time_scale': state.get(prefix + 'time_net.scale'),
                'time_weight': state.get(prefix + 'time_net.weight'), 'time_shift': state.get(prefix + 'time_net.shift'),
                'time_out': d.get('time_out', 2 * d['dim']), 'concatenate_time': d.get('concatenate_time', True)}
    raise ValueError(k)


def flow_spec(desc: List[Dict], state: Dict[str, torch.Tensor]) -> List[Dict]:
    return [transform_spec(d, state, f'transforms.{i}.') for i, d in enumerate(desc)]


def cfg2_desc(n_layers: int = 8, dim: int = 64, hidden: int = 64) -> List[Dict]:
    """BASELINE cfg 2: alternating ordered_right_half / ordered_left_half."""
    return [{'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden],
             'mask': 'ordered_right_half' if i % 2 == 0 else 'ordered_left_half', 'latent_dim': 0}
            for i in range(n_layers)]


def cfg3_desc(n_layers: int = 8, dim: int = 64, hidden: int = 64, n_bins: int = 16) -> List[Dict]:
    return [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': n_bins, 'lower': -3, 'upper': 3,
             'mask': 'ordered_right_half' if i % 2 == 0 else 'ordered_left_half', 'latent_dim': 0}
            for i in range(n_layers)]


def cfg4_desc(n_blocks: int = 4, dim: int = 128, hidden: int = 64) -> List[Dict]:
    out = []
    for b in range(n_blocks):
        out += [{'kind': 'affine_lu', 'dim': dim},
                {'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': 'ordered_right_half', 'latent_dim': 0},
                {'kind': 'matrix_exp', 'dim': dim, 'bias': False, 'log_time': False},
                {'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': 'ordered_left_half', 'latent_dim': 0}]
    return out
