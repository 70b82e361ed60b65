"""Error of the default ('fast', fp16 x 3) arithmetic against the fp64 oracle on the three bench flows, beside the reference's own fp32 op
sequence (the oracle in fp32) -- run once per library variant (STRIBOR_HIP_LIB) to compare operand-split roundings.
    python tools/experiments/split_accuracy.py [rows]"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'oracle'))
import stribor_amd as st  # noqa: E402
from stribor_amd.util import flowdesc as fd  # noqa: E402
import stribor_oracle as orc  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, desc, dim in (('cfg2', fd.cfg2_desc(), 64), ('cfg4', fd.cfg4_desc(), 128), ('cfg3', fd.cfg3_desc(), 64)):
    torch.manual_seed(11)
    flow = fd.build_flow(st, desc, dim)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to('cuda:0')
    x = torch.randn(rows, dim) * 1.2
    s32 = fd.flow_spec(desc, state)
    s64 = orc.spec_to(s32, torch.float64)
    with torch.no_grad():
        lp = flow.log_prob(x.to('cuda:0')).cpu().double().reshape(-1)
        z = flow.inverse(x.to('cuda:0')).cpu().double()
    st.check_errors()
    w_lp = orc.flow_log_prob(s64, x.double()).reshape(-1)
    w_z, _ = orc.flow_inverse_and_ldj(s64, x.double())
    r_lp = orc.flow_log_prob(s32, x).double().reshape(-1)
    r_z, _ = orc.flow_inverse_and_ldj(s32, x)

    def stats(g, t):
        e = (g - t).abs() / (1.0 + t.abs())
        return f'max {e.max().item():.2e} rms {e.pow(2).mean().sqrt().item():.2e} mean signed {((g - t) / (1.0 + t.abs())).mean().item():+.2e}'
    print(f'{name}: log_prob product {stats(lp, w_lp)} | fp32 oracle {stats(r_lp, w_lp)}')
    print(f'{name}: z        product {stats(z, w_z)} | fp32 oracle {stats(r_z.double(), w_z)}', flush=True)
