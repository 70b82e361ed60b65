"""Locate the NaN of the 40x-scaled spline flow: layer by layer through one-layer fused programs, then the element's parameters."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import flowdesc as fd
import stribor_amd as st
from oracle import stribor_oracle as orc
dim, hidden, K = 64, 64, 16
desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -3.0, 'upper': 3.0,
         'mask': ('ordered_right_half', 'ordered_left_half')[i % 2], 'latent_dim': 0} for i in range(2)]
torch.manual_seed(5)
flow = fd.build_flow(st, desc, dim)
sd = flow.state_dict()
for k in sd:
    if k.endswith('net.2.weight') or k.endswith('net.2.bias'):
        sd[k] = sd[k] * 40.0
flow.load_state_dict(sd)
x = torch.randn(500, dim) * 1.5
cur = x
with torch.no_grad():
    for li in (1, 0):
        layer = flow.transforms[li]
        one = st.NormalizingFlow(st.UnitNormal(dim), [layer]).to('cuda')
        d1 = [desc[li]]
        spec = fd.flow_spec(d1, {k.replace(f'transforms.{li}.', 'transforms.0.'): v.detach().cpu().clone() for k, v in flow.state_dict().items() if k.startswith(f'transforms.{li}.')})
        z, ldj = one.inverse_and_log_det_jacobian(cur.cuda())
        wz, wl = orc.flow_inverse_and_ldj(spec, cur)
        z, ldj = z.cpu(), ldj.cpu()
        bad = torch.isnan(z) | torch.isnan(ldj.expand_as(z) * 0 + z * 0)
        print('layer', li, 'nan z', int(torch.isnan(z).sum()), 'nan ldj', int(torch.isnan(ldj).sum()), 'oracle nan', int(torch.isnan(wz).sum()), int(torch.isnan(wl).sum()))
        rows = torch.isnan(ldj).reshape(-1).nonzero().reshape(-1).tolist()
        for r in rows[:3]:
            cols = torch.isnan(z[r]).nonzero().reshape(-1).tolist()
            print('  row', r, 'nan cols', cols, 'ldj', ldj[r].item(), 'oracle ldj', wl[r].item())
            for c in cols[:2]:
                print('    x', cur[r, c].item(), 'oracle z', wz[r, c].item())
            if not cols:
                # which column's log-derivative is NaN?  evaluate the oracle's per-column terms
                print('    (values finite: the NaN is in the log-det only)')
            # conditioner output of the oracle for this row
            import copy
            net = copy.deepcopy(layer.transform.latent_net).cpu()
            m = torch.from_numpy(layer.mask_vector(dim)).float()
            p = net((cur[r:r + 1] * m)).reshape(dim, 3 * K - 1)
            live = (m <= 0.5).nonzero().reshape(-1).tolist()
            for c in (cols[:2] if cols else live[:0]):
                print('    w logits', p[c, :K].tolist())
                print('    h logits', p[c, K:2 * K].tolist())
                print('    d logits', p[c, 2 * K:].tolist())
        cur = wz
