"""cfg 4 taken apart: flows of dense layers only, of couplings only, and the mixed stack, at D = 128 and 2^20 rows (ms per batch)."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'tools'))
import torch
import flowdesc as fd
import stribor_amd as st
from bench_configs import timed
dim, rows = 128, 1 << 20
lu = {'kind': 'affine_lu', 'dim': dim}
mx = {'kind': 'matrix_exp', 'dim': dim, 'bias': False, 'log_time': False}
def cp(i, hidden=64):
    return {'kind': 'coupling_affine', 'dim': dim, 'hidden': [hidden], 'mask': ('ordered_right_half', 'ordered_left_half')[i % 2], 'latent_dim': 0}
cases = {
    'dense16': [lu] * 16,
    'dense8': [lu] * 8,
    'coupling16': [cp(i) for i in range(16)],
    'coupling8': [cp(i) for i in range(8)],
    'cfg4': fd.cfg4_desc(),
    'lu_cp_x8': sum(([lu, cp(i)] for i in range(8)), []),
}
names = [a for a in sys.argv[1:]] or list(cases)
dev = torch.device('cuda', 0)
x = torch.randn(rows, dim, device=dev)
for name in names:
    torch.manual_seed(0)
    flow = fd.build_flow(st, cases[name], dim)
    with torch.no_grad():
        for n_, p_ in flow.named_parameters():
            if p_.dim() == 2 and p_.shape[0] == dim and p_.shape[1] == dim:
                p_.mul_(0.02)       # (dense layers near the identity: sixteen random ones overflow the fp16 range)
    flow = flow.to(dev)
    fused = flow._fused_program(True, dim, 0, dev)
    with torch.no_grad():
        ms = timed(lambda: flow.log_prob(x))
    print(json.dumps({'case': name, 'ms': round(ms, 4), 'fused': fused is not None, 'mode': getattr(fused, 'mode', None) if fused else None}))
