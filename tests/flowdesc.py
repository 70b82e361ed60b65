"""Test-side alias of the flow-description helper (it lives in the package so that bench.py and
__graft_entry__.smoke() do not import the test tree)."""
from stribor_amd.util.flowdesc import *          # noqa: F401,F403
from stribor_amd.util.flowdesc import _net_spec, _spline_params          # noqa: F401
