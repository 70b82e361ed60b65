import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import flowdesc as fd, stribor_amd as st
from oracle import stribor_oracle as orc
torch.manual_seed(0)
desc = fd.cfg2_desc(4, 64, 64)
flow = fd.build_flow(st, desc, 64)
spec = fd.flow_spec(desc, {k: v.clone() for k, v in flow.state_dict().items()})
flow = flow.to('cuda')
torch.manual_seed(1)
x = torch.randn(777, 64)
x[3, 5] = 1.0e5; x[100, 40] = -2.5e5; x[101] *= 3.0e4; x[101, 0] = 9.0e4; x[500, 63] = 3.0e5; x[600, 17] = 1.0e6
x[601] *= 1.0e6 / x[601].abs().max(); x[776, 31] = 65505.0
spec64 = orc.spec_to(spec, torch.float64)
rows = [3, 100, 101, 500, 600, 601, 776]
with torch.no_grad():
    for name, f32, f64, fn in [('lp', orc.flow_log_prob(spec, x), orc.flow_log_prob(spec64, x.double()), flow.log_prob),
                               ('z', orc.flow_inverse(spec, x), orc.flow_inverse(spec64, x.double()), flow.inverse),
                               ('y', orc.flow_forward(spec, x), orc.flow_forward(spec64, x.double()), flow.forward)]:
        for mode in ('fast', 'exact'):
            st.set_gemm_precision(mode)
            g = fn(x.cuda()).cpu().double().reshape(777, -1)
            t = f64.reshape(777, -1); r = f32.double().reshape(777, -1)
            for i in rows:
                print(name, mode, i, 'ours %.3e ref %.3e rowmax %.3e' % ((g[i]-t[i]).abs().max(), (r[i]-t[i]).abs().max(), t[i].abs().max()))
