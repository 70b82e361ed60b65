"""One case of `tools/fuzz_dense.py SEED --big`, layer by layer: the state of the worst out-of-range row after each inverse (or forward)
step, product ('fast' and 'exact') and the reference's fp32 op sequence against the fp64 oracle -- where does a row of 1e5 lose its digits?
    python tools/experiments/dbg_big_layers.py SEED INDEX [--forward]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools')); sys.path.insert(0, os.path.join(R, 'oracle'))
import numpy as np, torch
seed, idx = int(sys.argv[1]), int(sys.argv[2])
fwd = '--forward' in sys.argv
sys.argv = [sys.argv[0], '--big']
import fuzz_dense as fz
import stribor_amd as st
from stribor_amd.util import flowdesc as fd
import stribor_oracle as orc
rng = np.random.default_rng(seed)
# replay the campaign's random stream up to the case (fuzz_dense.main draws the big rows from the same generator)
for i in range(idx + 1):
    desc, dim, n = fz.case(rng)
    torch.manual_seed(seed * 1000 + i)
    flow = fd.build_flow(st, desc, dim)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if p.dim() == 2 and p.shape == (dim, dim):
                p.mul_(0.08)
            else:
                p.add_(torch.randn_like(p) * 0.03)
    x = torch.randn(n, dim) * 1.3
    big = []
    for r in rng.choice(n, size=min(n, int(rng.integers(1, 5))), replace=False):
        if rng.random() < 0.5:
            x[r] *= float(10.0 ** rng.uniform(4.0, 6.0)) / x[r].abs().max()
        else:
            x[r, int(rng.integers(0, dim))] = float(10.0 ** rng.uniform(4.8, 6.0)) * (1 if rng.random() < 0.5 else -1)
        big.append(int(r))
state = {k: v.clone() for k, v in flow.state_dict().items()}
kinds = [d['kind'] for d in desc]
print('case', idx, 'dim', dim, 'rows', n, kinds, 'big rows', big)
L = len(desc)
for k in range(1, L + 1):
    sub = desc[:k] if fwd else desc[L - k:]
    # sub-flow with the same weights: keys are transforms.<i>.*
    off = 0 if fwd else L - k
    sub_state = {}
    for key, v in state.items():
        parts = key.split('.')
        if parts[0] == 'transforms' and off <= int(parts[1]) < off + k:
            sub_state['.'.join([parts[0], str(int(parts[1]) - off)] + parts[2:])] = v
        elif parts[0] != 'transforms':
            sub_state[key] = v
    s32 = fd.flow_spec(sub, sub_state)
    s64 = orc.spec_to(s32, torch.float64)
    f = fd.build_flow(st, sub, dim)
    f.load_state_dict(sub_state, strict=False)
    f = f.to('cuda:0')
    if fwd:
        w, _ = orc.flow_forward_and_ldj(s64, x.double()); r32, _ = orc.flow_forward_and_ldj(s32, x)
    else:
        w, _ = orc.flow_inverse_and_ldj(s64, x.double()); r32, _ = orc.flow_inverse_and_ldj(s32, x)
    out = {}
    for mode in ('fast', 'exact'):
        st.set_gemm_precision(mode)
        with torch.no_grad():
            g = (f.forward(x.to('cuda:0')) if fwd else f.inverse(x.to('cuda:0'))).cpu().double()
        out[mode] = g
    st.set_gemm_precision('fast')
    line = f'after {k} steps ({sub[-1]["kind"] if fwd else sub[0]["kind"]}):'
    for r in big:
        m = w[r].abs().max().item()
        line += f'  row {r}: max {m:.2e} fast {(out["fast"][r] - w[r]).abs().max().item() / m:.1e} exact {(out["exact"][r] - w[r]).abs().max().item() / m:.1e} fp32 {(r32[r].double() - w[r]).abs().max().item() / m:.1e} |'
    print(line, flush=True)

# ---- the conditioner of the step where 'fast' leaves 'exact' behind: its hidden pre-activations in fp64 and its output in the three
# arithmetics (the stand-alone MLP program: the same GEMM arithmetic and rescale as the coupling's in-kernel conditioner)
if '--cond' in sys.argv or True:
    st.set_gemm_precision('fast')
    order = list(range(L)) if fwd else list(range(L - 1, -1, -1))
    cur64 = x.double()
    for step, li in enumerate(order):
        d1 = [desc[li]]
        sub_state = {}
        for key, v in state.items():
            parts = key.split('.')
            if parts[0] == 'transforms' and int(parts[1]) == li:
                sub_state['.'.join([parts[0], '0'] + parts[2:])] = v
        s64 = orc.spec_to(fd.flow_spec(d1, sub_state), torch.float64)
        if desc[li]['kind'] == 'coupling_affine':
            f1 = fd.build_flow(st, d1, dim)
            f1.load_state_dict(sub_state, strict=False)
            cpl = f1.transforms[0].to('cuda:0')
            mask = torch.from_numpy(cpl.mask_vector(dim)).double()
            zin = cur64 * mask                                  # coupling.py:80: the conditioner sees the masked state
            net = cpl.transform.latent_net
            W1, b1 = net.net[0].weight.detach().cpu().double(), net.net[0].bias.detach().cpu().double()
            pre = zin @ W1.T + b1
            want = None
            outs = {}
            for mode in ('fast', 'exact'):
                st.set_gemm_precision(mode)
                with torch.no_grad():
                    outs[mode] = net(zin.float().to('cuda:0')).cpu().double()
            st.set_gemm_precision('fast')
            lin2 = [m for m in net.net if isinstance(m, torch.nn.Linear)][-1]
            W2, b2 = lin2.weight.detach().cpu().double(), lin2.bias.detach().cpu().double()
            want = torch.tanh(pre) @ W2.T + b2
            with torch.no_grad():
                o32 = torch.tanh(zin.float() @ W1.float().T + b1.float()) @ W2.float().T + b2.float()
            for r in big:
                a = pre[r].abs()
                print(f'step {step + 1} coupling {li} row {r}: min |pre| {a.min().item():.2f}, units with |pre| < 8: {int((a < 8).sum())} of {a.numel()}; '
                      f'conditioner output error (abs, max): fast {(outs["fast"][r] - want[r]).abs().max().item():.2e} exact {(outs["exact"][r] - want[r]).abs().max().item():.2e} '
                      f'torch fp32 {(o32[r].double() - want[r]).abs().max().item():.2e}', flush=True)
        cur64 = (orc.flow_forward_and_ldj(s64, cur64)[0] if fwd else orc.flow_inverse_and_ldj(s64, cur64)[0])
