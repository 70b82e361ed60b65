"""Experiment / A-B switches of the host side, in ONE place (six ``STRIBOR_*`` switches were read ad hoc in
product code paths).  Product code asks ``debug.on(NAME)``; every name is declared here with what it does.  None of them is needed to
use the package; they exist for the measurements in ``tools/`` and the A/B tests in ``tests/``.  The environment is read at the time
of the call (tests flip them with ``monkeypatch.setenv``).

The library's own knobs (``SX_*``, read once per process through ``sx_debug_knob`` in csrc/sx_build_id.cpp) are listed there.
"""
import os

SWITCHES = {
    'STRIBOR_BWD_FACTORS': 'affine flows: train through the single-launch backward program with per-row factors in HBM + sx_wgrad_layer '
                           '(the round-1 form) instead of the layer-major backward with in-kernel weight gradients',
    'STRIBOR_SPLINE_FORWARD_PER_LAYER': 'spline flows: the training forward runs one program per layer instead of the whole-flow program '
                                        'with side outputs (kernel MODE 18 / 19)',
    'STRIBOR_SPLINE_UNFUSED': 'spline couplings: train on the per-row parameter path ([N, D(3K-1)] tensor through HBM, library GEMMs) '
                              'instead of the slab backward',
    'STRIBOR_SPLINE_L1_TORCH': 'spline couplings: the first conditioner layer\'s backward through torch instead of sx_rqs_slab_l1_bwd',
    'STRIBOR_SPLINE_NO_SLAB_FWD': 'rational-quadratic couplings beyond the one-launch tier (hidden layers of 129 .. 256 units) evaluate through '
                                  'the [N, n_live (3K-1)] parameter tensor in HBM (MLP programs + sx_rqs_coupling) instead of the slab '
                                  'forward pass (sx_rqs_slab_fwd)',
    'STRIBOR_CUBIC_UNFUSED': 'cubic-spline couplings stay out of fused programs (conditioner program + cubic_kernel through HBM)',
}


def on(name: str) -> bool:
    """Is the switch set (to anything but '' / '0')?  Unknown names are a programming error."""
    if name not in SWITCHES:
        raise KeyError(f'{name} is not a declared debug switch (stribor_amd/debug.py)')
    return os.environ.get(name, '0') not in ('', '0')
