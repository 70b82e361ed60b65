"""Case 35 of `fuzz_train.py 60 412 --fat` (weight gradients of a cubic K = 16 coupling 7e-3 off fp64 autograd of the oracle): the same
loss through the reference's op sequence in fp32 (the oracle in fp32), to tell conditioning from a kernel fault."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools')); sys.path.insert(0, os.path.join(R, 'oracle'))
import numpy as np, torch
sys.argv = [sys.argv[0], '--fat']
import fuzz_train as ft
import stribor_amd as st
from stribor_amd.util import flowdesc as fd
import stribor_oracle as orc
seed, idx = 412, 35
rng = np.random.default_rng(seed)
for i in range(idx + 1):
    desc, dim, latent, n = ft.case(rng)
print([(d['kind'], d.get('spline_type'), d.get('n_bins'), d.get('hidden')) for d in desc], dim, latent, n)
torch.manual_seed(seed * 1000 + idx)
flow = fd.build_flow(st, desc, dim)
with torch.no_grad():
    for p in flow.parameters():
        p.add_(torch.randn_like(p) * 0.03)
state = {k: v.clone() for k, v in flow.state_dict().items()}
flow = flow.to('cuda:0')
x = torch.randn(n, dim) * 1.4
def ograd(dtype):
    leaves = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in state.items()}
    xin = x.to(dtype).clone().requires_grad_(True)
    loss = -orc.flow_log_prob(fd.flow_spec(desc, leaves), xin).mean()
    loss.backward()
    return loss.item(), {k: v.grad.double() for k, v in leaves.items()}, xin.grad.double()
l64, g64, gx64 = ograd(torch.float64)
l32, g32, gx32 = ograd(torch.float32)
xg = x.to('cuda:0').requires_grad_(True)
loss = -flow.log_prob(xg).mean()
loss.backward()
print('loss fp64 %.9f fp32-oracle %.9f product %.9f' % (l64, l32, loss.item()))
for name, p in flow.named_parameters():
    ref = g64[name]; s = ref.abs().max().item() + 1e-300
    print('%-50s product %.2e   fp32 oracle %.2e' % (name, (p.grad.cpu().double() - ref).abs().max().item() / s, (g32[name] - ref).abs().max().item() / s))
# per-row contribution: which rows carry the difference in the worst parameter
