"""Accuracy of spline-coupling flows as the conditioner's output layer is scaled up (sharply peaked bins): |got - f64| of the fused
kernel (fast = fp16 x 3 GEMMs, exact = fp32 MFMA) beside |oracle fp32 - f64| (the reference's own op sequence in fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
import flowdesc as fd
import stribor_amd as st
from oracle import stribor_oracle as orc
dim, hidden, K = 64, 64, 16
desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [hidden], 'n_bins': K, 'lower': -3.0, 'upper': 3.0,
         'mask': ('ordered_right_half', 'ordered_left_half')[i % 2], 'latent_dim': 0} for i in range(2)]
for scale in (1.0, 3.0, 6.0, 10.0, 15.0, 20.0, 40.0):
    torch.manual_seed(5)
    flow = fd.build_flow(st, desc, dim)
    sd = flow.state_dict()
    for k in sd:
        if k.endswith('net.2.weight') or k.endswith('net.2.bias'):
            sd[k] = sd[k] * scale
    flow.load_state_dict(sd)
    spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
    spec64 = fd.flow_spec(desc, {k: v.double() for k, v in flow.state_dict().items()})
    flow = flow.to('cuda')
    x = torch.randn(500, dim) * 1.5
    with torch.no_grad():
        got = flow.log_prob(x.cuda()).cpu().double()
        ref = orc.flow_log_prob(spec, x).double()
        f64 = orc.flow_log_prob(spec64, x.double())
        old = st.set_gemm_precision('exact')
        gote = flow.log_prob(x.cuda()).cpu().double()
        st.set_gemm_precision(old)
    def stats(a):
        e = (a - f64).abs()
        e = e[torch.isfinite(e)]
        return 'max %.2e p99 %.2e med %.2e nan %d' % (e.max(), e.quantile(0.99), e.median(), int((~torch.isfinite(a)).sum()))
    print('scale %5.1f | fast: %s | exact: %s | oracle fp32: %s' % (scale, stats(got), stats(gote), stats(ref)))
    try:
        st.check_errors()
    except Exception as e:
        print('  flags:', type(e).__name__)
