import sys, json, torch
sys.path.insert(0, '.')
import stribor_amd as st
from stribor_amd.util import flowdesc as fd
dev = torch.device('cuda', 0)
torch.manual_seed(0)
res = {}
for H in (64, 160):
    desc = [{'kind': 'coupling_rqs', 'dim': 64, 'hidden': [H], 'mask': m, 'latent_dim': 0, 'n_bins': 16, 'lower': -3, 'upper': 3}
            for m in ['ordered_right_half', 'ordered_left_half'] * 4]
    flow = fd.build_flow(st, desc, 64).to(dev)
    x = torch.randn(1 << 20, 64, device=dev)
    with torch.no_grad():
        for _ in range(3): flow.log_prob(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): flow.log_prob(x)
        b.record(); torch.cuda.synchronize()
    res[H] = a.elapsed_time(b) / 10
    del flow, x
print(json.dumps({'workload': 'cfg-3 shape (D=64, 8 rqs couplings, K=16, 2^20 rows) at hidden 64 (one launch) and hidden 160 (slab forward tier)', 'ms': res,
                  'rows_per_s': {k: (1 << 20) / v * 1e3 for k, v in res.items()}}))
