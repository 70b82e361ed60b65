"""torch.profiler view of one training step of a BASELINE config: python tools/profile_ops.py cfg4 [rows]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stribor_amd as st  # noqa: E402
from stribor_amd.util import flowdesc as fd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
mk, dim = {'cfg2': (fd.cfg2_desc, 64), 'cfg3': (fd.cfg3_desc, 64), 'cfg4': (fd.cfg4_desc, 128)}[name]
dev = torch.device('cuda:0')
torch.manual_seed(0)
flow = fd.build_flow(st, mk(), dim).to(dev)
x = torch.randn(rows, dim, device=dev)


def step():
    for p in flow.parameters():
        p.grad = None
    (-flow.log_prob(x).mean()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=40, max_name_column_width=70))
