"""Case 3 of `fuzz_train.py 120 402 --infer --mix` (forward log-det 0.23 off the fp64 oracle): which layer, which element, and what
the reference's own fp32 op sequence gives there."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools')); sys.path.insert(0, os.path.join(R, 'oracle'))
import numpy as np, torch
sys.argv = [sys.argv[0], '--mix', '--infer']
import fuzz_train as ft
import stribor_amd as st
from stribor_amd.util import flowdesc as fd
import stribor_oracle as orc
seed, idx = 402, 3
rng = np.random.default_rng(seed)
for i in range(idx + 1):
    desc, dim, latent, n = ft.case_mix(rng)
print(desc, dim, latent, n)
torch.manual_seed(seed * 1000 + idx)
flow = fd.build_flow(st, desc, dim)
with torch.no_grad():
    for p in flow.parameters():
        p.add_(torch.randn_like(p) * 0.03)
state = {k: v.clone() for k, v in flow.state_dict().items()}
flow = flow.to('cuda:0')
lead = (n // 3, 3) if n % 3 == 0 else (n,)
x = torch.randn(*lead, dim) * 1.4
spec64 = fd.flow_spec(desc, {k: v.double() for k, v in state.items()})
spec32 = fd.flow_spec(desc, {k: v.clone() for k, v in state.items()})
with torch.no_grad():
    y, ldj = flow.forward_and_log_det_jacobian(x.to('cuda:0'))
w64y, w64l = orc.flow_forward_and_ldj(spec64, x.double())
w32y, w32l = orc.flow_forward_and_ldj(spec32, x)
e = (ldj.cpu().double() - w64l).abs().reshape(-1)
e32 = (w32l.double() - w64l).abs().reshape(-1)
print('product vs fp64: max', e.max().item(), 'rows > 1e-3:', (e > 1e-3).nonzero().reshape(-1).tolist())
print('fp32 oracle vs fp64: max', e32.max().item(), 'rows > 1e-3:', (e32 > 1e-3).nonzero().reshape(-1).tolist())
r = int(e.argmax())
# layer by layer on that row, fp64 oracle, product (per-layer API), fp32 oracle
xr = x.reshape(-1, dim)[r:r + 1]
c64, c32, cp = xr.double(), xr.clone(), xr.to('cuda:0')
for li, (layer, l64, l32) in enumerate(zip(flow.transforms, spec64, spec32)):
    n64, d64 = orc.transform_forward_and_ldj(l64, c64) if hasattr(orc, 'transform_forward_and_ldj') else (None, None)
    with torch.no_grad():
        npd, dpd = layer.forward_and_log_det_jacobian(cp)
    if n64 is not None:
        n32, d32 = orc.transform_forward_and_ldj(l32, c32)
        dl = dpd.cpu().double()
        if dl.shape[-1] != 1:
            dl = dl.sum(-1, keepdim=True)
        print(li, desc[li]['kind'], 'ldj fp64', float(d64.sum()), 'product', float(dl.sum()), 'fp32', float(d32.sum()),
              '| max |y - y64|', float((npd.cpu().double() - n64).abs().max()))
        if abs(float(d64.sum()) - float(dl.sum())) > 1e-3:
            col = int((npd.cpu().double() - n64).abs().argmax())
            print('   input range of the layer (fp64):', float(c64.min()), float(c64.max()))
            # per element: the layer's log-det is a sum over columns -- evaluate the product column by column through y's finite
            # differences is overkill; compare the element-wise oracle terms with a product run on the fp64 input rounded to fp32
            t64 = orc.transform_ldj(l64, c64) if False else None
            from stribor_amd.flows import spline as _sp
            with torch.no_grad():
                yb, lb = layer.forward_and_log_det_jacobian(c64.float().to('cuda:0'))
            print('   product on the fp64 input rounded to fp32: ldj', float(lb.sum()))
            xin = c64.reshape(-1)
            near = (xin.abs() - 3.0).abs()
            k = int(near.argmin())
            print('   element closest to a bound: col', k, 'value fp64 %.9f' % float(xin[k]), 'product input %.9f' % float(cp.reshape(-1)[k].cpu()),
                  'fp32-oracle input %.9f' % float(c32.reshape(-1)[k]), '| distance to the bound %.3e' % float(near[k]))
        c64, c32, cp = n64, n32, npd
