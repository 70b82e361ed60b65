"""Is the fused kernel's clock data-dependent (power-limited)?  cfg 2 / cfg 3 / cfg 4 with their random parameters and inputs, and the
same programs with every parameter and input zero (MI355X_MICROARCH.md, DVFS give-back: zero operands hold ~2.3 GHz where random
ones hold 1.9): equal instruction streams, different wall time = the chip lowers its clock under this kernel's load."""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'tools'))
import torch
import flowdesc as fd
import stribor_amd as st
from bench_configs import timed
for name, mk, dim in (('cfg2', fd.cfg2_desc, 64), ('cfg3', fd.cfg3_desc, 64), ('cfg4', fd.cfg4_desc, 128)):
    out = {'config': name}
    for zero in (False, True, False, True):
        torch.manual_seed(0)
        flow = fd.build_flow(st, mk(), dim)
        if zero:
            with torch.no_grad():
                for p in flow.parameters():
                    p.zero_()
        flow = flow.to('cuda')
        x = torch.zeros(1 << 20, dim, device='cuda') if zero else torch.randn(1 << 20, dim, device='cuda')
        with torch.no_grad():
            ms = timed(lambda: flow.log_prob(x))
        out.setdefault('zero' if zero else 'random', []).append(round(ms, 4))
    print(json.dumps(out))
