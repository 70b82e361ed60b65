"""torch.profiler view of one cfg-3 training step (2^18 rows): which aten ops / autograd nodes launch the small kernels."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stribor_amd as st  # noqa: E402
from stribor_amd.util import flowdesc as fd  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
flow = fd.build_flow(st, fd.cfg3_desc(), 64).to(dev)
x = torch.randn(1 << 18, 64, device=dev)


def step():
    for p in flow.parameters():
        p.grad = None
    (-flow.log_prob(x).mean()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=60,
                                                    max_src_column_width=110))
