"""One case of a fuzz_train.py campaign again, beside the reference's own op sequence in fp32 (the oracle in fp32): tells
conditioning (fp32 itself is that far from fp64) from a kernel fault.
    python tools/experiments/fuzz_case_vs_fp32.py SEED INDEX [--infer] [the campaign's flags: --fat --long --k16 --wide ...]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools')); sys.path.insert(0, os.path.join(R, 'oracle'))
import numpy as np, torch
seed, idx = int(sys.argv[1]), int(sys.argv[2])
infer = '--infer' in sys.argv
sys.argv = [sys.argv[0]] + [a for a in sys.argv[3:]]
import fuzz_train as ft
import stribor_amd as st
from stribor_amd.util import flowdesc as fd
import stribor_oracle as orc
rng = np.random.default_rng(seed)
for i in range(idx + 1):
    desc, dim, latent, n = ft.case(rng)
print(len(desc), 'layers, dim', dim, 'latent', latent, 'rows', n)
torch.manual_seed(seed * 1000 + idx)
flow = fd.build_flow(st, desc, dim)
with torch.no_grad():
    for p in flow.parameters():
        p.add_(torch.randn_like(p) * 0.03)
state = {k: v.clone() for k, v in flow.state_dict().items()}
flow = flow.to('cuda:0')
x = torch.randn(n, dim) * 1.4
lat = torch.randn(n, latent) if latent else None
kw = {} if lat is None else {'latent': lat.to('cuda:0')}
rel = lambda a, b: ((a.double() - b.double()).abs() / (1.0 + b.double().abs())).max().item()
if infer:
    s64 = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
    s32 = fd.flow_spec(desc, {k: v.clone() for k, v in state.items()})
    l64 = None if lat is None else lat.double()
    with torch.no_grad():
        lp = flow.log_prob(x.to('cuda:0'), **kw).cpu()
        y, ldj = flow.forward_and_log_det_jacobian(x.to('cuda:0'), **kw)
        xr = flow.inverse(y, **kw).cpu()
    w64 = orc.flow_log_prob(s64, x.double(), l64); w32 = orc.flow_log_prob(s32, x, lat)
    y64, d64 = orc.flow_forward_and_ldj(s64, x.double(), l64); y32, d32 = orc.flow_forward_and_ldj(s32, x, lat)
    x32 = orc.flow_inverse(s32, y32, lat) if hasattr(orc, 'flow_inverse') else None
    print('log_prob   product %.1e   fp32 oracle %.1e' % (rel(lp, w64), rel(w32, w64)))
    print('y          product %.1e   fp32 oracle %.1e' % (rel(y.cpu(), y64), rel(y32, y64)))
    print('ldj        product %.1e   fp32 oracle %.1e' % (rel(ldj.cpu(), d64), rel(d32, d64)))
    print('round trip product %.1e' % (xr.double() - x.double()).abs().max().item() + ('   fp32 oracle %.1e' % (x32.double() - x.double()).abs().max().item() if x32 is not None else ''))
else:
    def ograd(dtype):
        leaves = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in state.items()}
        xin = x.to(dtype).clone().requires_grad_(True)
        loss = -orc.flow_log_prob(fd.flow_spec(desc, leaves), xin, None if lat is None else lat.to(dtype)).mean()
        loss.backward()
        return loss.item(), {k: v.grad.double() for k, v in leaves.items()}, xin.grad.double()
    l64, g64, gx64 = ograd(torch.float64)
    l32, g32, gx32 = ograd(torch.float32)
    xg = x.to('cuda:0').requires_grad_(True)
    loss = -flow.log_prob(xg, **kw).mean()
    loss.backward()
    print('loss fp64 %.9f fp32-oracle %.9f product %.9f' % (l64, l32, loss.item()))
    sx = gx64.abs().max().item()
    print('%-52s product %.2e   fp32 oracle %.2e' % ('x', (xg.grad.cpu().double() - gx64).abs().max().item() / sx, (gx32 - gx64).abs().max().item() / sx))
    rows = []
    for name, p in flow.named_parameters():
        ref = g64[name]; s = ref.abs().max().item() + 1e-300
        rows.append(((p.grad.cpu().double() - ref).abs().max().item() / s, (g32[name] - ref).abs().max().item() / s, name))
    rows.sort(reverse=True)
    for a, b, name in rows[:8]:
        print('%-52s product %.2e   fp32 oracle %.2e' % (name, a, b))
