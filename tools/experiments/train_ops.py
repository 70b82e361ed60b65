"""Which Python lines launch the small kernels of a training step?  torch.profiler over ONE step of a bench config
(tools/bench_configs.py's flows), device kernels grouped by the innermost stribor_amd frame that issued them:

    python tools/experiments/train_ops.py cfg3 [--rows N]

Prints per (file:line, kernel) the launch count and device time of one step -- the list the launch-count work of a round starts
from (rocprofv3's kernel trace has the kernels but not who asked for them)."""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
import flowdesc as fd  # noqa: E402
import stribor_amd as st  # noqa: E402
from bench_configs import CONFIGS  # noqa: E402


def main():
    name = [a for a in sys.argv[1:] if not a.startswith('--') and not a.isdigit()][0]
    mk, dim, rows, dt = CONFIGS[name]
    if '--rows' in sys.argv:
        rows = int(sys.argv[sys.argv.index('--rows') + 1])
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    flow = fd.build_flow(st, mk(), dim).to(dev)
    x = torch.randn(rows, dim, device=dev).to(dt)

    def step():
        for p_ in flow.parameters():
            p_.grad = None
        (-flow.log_prob(x).mean()).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    by_site = defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        site = 'autograd engine / other'
        for fr in ev.stack or []:
            if 'stribor_amd' in fr or 'bench_configs' in fr or 'train_ops' in fr:
                site = fr.split('stribor_amd/')[-1] if 'stribor_amd/' in fr else fr
                break
        for k in ev.kernels:
            e = by_site[(site[:70], ev.name[:28], k.name[:60])]
            e[0] += 1
            e[1] += k.duration
    tot_n = sum(v[0] for v in by_site.values())
    tot_t = sum(v[1] for v in by_site.values())
    print('%s: %d device launches, %.3f ms of device time in one step (rows = %d)' % (name, tot_n, tot_t / 1e3, rows))
    for (site, op, kern), (n, t) in sorted(by_site.items(), key=lambda kv: -kv[1][0]):
        print('%4d  %9.1f us  %-70s %-28s %s' % (n, t, site, op, kern))


if __name__ == '__main__':
    main()
