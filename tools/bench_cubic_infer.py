import sys, os, json, torch
sys.path.insert(0, '/root/repo')
import stribor_amd as st
from stribor_amd.util import flowdesc as fd
dev = torch.device('cuda:0')
torch.manual_seed(0)
desc = [dict(d, spline_type='cubic') for d in fd.cfg3_desc()]
flow = fd.build_flow(st, desc, 64).to(dev)
x = torch.randn(1 << 20, 64, device=dev)
with torch.no_grad():
    for _ in range(2): flow.log_prob(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): flow.log_prob(x)
    b.record(); torch.cuda.synchronize()
print(json.dumps({'cubic cfg3-shaped log_prob 2^20 rows ms': a.elapsed_time(b) / 5}))
