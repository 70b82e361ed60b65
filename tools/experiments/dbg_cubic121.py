"""tests/test_gpu_backward.py::test_spline_training_beyond_the_program_tiles[cubic] beside the reference's own fp32 autograd: is the worst
x-gradient element a knot-adjacent row (fp32 itself differs there) or a product fault?  Prints the worst rows in 'fast' and 'exact'."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'oracle'))
import torch
import stribor_amd as st
from stribor_amd.util import flowdesc as fd
import stribor_oracle as orc
DEV = 'cuda:0'
stype = sys.argv[1] if len(sys.argv) > 1 else 'cubic'
torch.manual_seed(17)
dim, latent, n = 121, 3, 150
desc = [{'kind': 'coupling_rqs', 'dim': dim, 'hidden': [40], 'n_bins': 6, 'lower': -3.0, 'upper': 3.0, 'mask': 'ordered_left_half',
         'latent_dim': latent, 'spline_type': stype}]
flow = fd.build_flow(st, desc, dim)
state = {k: v.clone() for k, v in flow.state_dict().items()}
flow = flow.to(DEV)
x, lat = torch.randn(n, dim) * 1.3, torch.randn(n, latent)


def ograd(dt):
    leaves = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in state.items()}
    xin = x.to(dt).clone().requires_grad_(True)
    loss = -orc.flow_log_prob(fd.flow_spec(desc, leaves), xin, lat.to(dt)).mean()
    loss.backward()
    return xin.grad.double(), {k: v.grad.double() for k, v in leaves.items()}


g64, p64 = ograd(torch.float64)
g32, p32 = ograd(torch.float32)
sx = g64.abs().max().item()
print('fp32 oracle vs fp64: max x-grad err %.3e (scale %.3e)' % ((g32 - g64).abs().max().item(), sx))
for mode in ('fast', 'exact'):
    st.set_gemm_precision(mode)
    for p in flow.parameters():
        p.grad = None
    xg = x.to(DEV).requires_grad_(True)
    loss = -flow.log_prob(xg, latent=lat.to(DEV)).mean()
    loss.backward()
    e = (xg.grad.cpu().double() - g64).abs()
    r, c = divmod(int(e.argmax()), dim)
    print(f'{mode}: max x-grad err {e.max().item():.3e} at row {r} col {c}: ours {xg.grad[r, c].item():+.6e} fp64 {g64[r, c].item():+.6e} fp32 {g32[r, c].item():+.6e}  x {x[r, c].item():+.7f}')
    rows = e.amax(1)
    top = rows.topk(4)
    for v, i in zip(top.values.tolist(), top.indices.tolist()):
        print(f'   row {i}: ours-fp64 {v:.3e}   fp32-fp64 {(g32[i] - g64[i]).abs().max().item():.3e}')
    for name, p in flow.named_parameters():
        ref = p64[name]; s = ref.abs().max().item() + 1e-300
        print(f'   {name:50s} ours {((p.grad.cpu().double() - ref).abs().max().item() / s):.2e}  fp32 oracle {((p32[name] - ref).abs().max().item() / s):.2e}')
st.set_gemm_precision('fast')
