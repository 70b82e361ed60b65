"""Times the reference's flagship stack (affine coupling -> Flip -> Sigmoid -> cubic-spline coupling -> Logit, D = 64, K = 16) as one
fused launch (kernel MODE 14) and layer by layer; 2^20 rows.  python tools/bench_stack.py"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stribor_amd as st
import bench


def main():
    dev = torch.device('cuda', 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    from stribor_amd.util import flowdesc as fd
    with torch.no_grad():
        print(json.dumps(bench.extra_flow_entries(st, fd, dev, gen, 20)))


if __name__ == '__main__' and '--rq' not in sys.argv:
    main()


def rq_mixed():
    """A mixed flow with rational-quadratic splines (kernel MODE 17): affine coupling -> rq-spline coupling -> LeakyReLU -> affine coupling
    -> rq-spline coupling, D = 64, K = 16, 2^20 rows: one launch vs layer by layer."""
    dev = torch.device('cuda', 0)
    from stribor_amd.util import flowdesc as fd
    torch.manual_seed(0)
    D, H, K = 64, 64, 16
    aff = lambda m: {'kind': 'coupling_affine', 'dim': D, 'hidden': [H], 'mask': m, 'latent_dim': 0}
    rq = lambda m: {'kind': 'coupling_rqs', 'dim': D, 'hidden': [H], 'mask': m, 'latent_dim': 0, 'n_bins': K, 'lower': -3, 'upper': 3}
    desc = [aff('ordered_0'), rq('ordered_1'), {'kind': 'leaky_relu', 'negative_slope': 0.3}, aff('ordered_0'), rq('ordered_1')]
    flow = fd.build_flow(st, desc, D).to(dev)
    x = torch.randn(1 << 20, D, device=dev)

    def timed(fn, reps=10):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    with torch.no_grad():
        fused = flow._fused_program(True, D, 0, dev) is not None
        ms = timed(lambda: flow.log_prob(x))

        def layerwise():
            cur, acc = x, 0
            for f in reversed(flow.transforms):
                cur, l = f.inverse_and_log_det_jacobian(cur)
                acc = acc + l
            return flow.base_dist.log_prob(cur).unsqueeze(-1) + acc
        ms_l = timed(layerwise, 4)
    print(json.dumps({'name': 'rq_mixed', 'one_fused_launch': fused, 'ms_per_step': ms, 'ms_per_step_layer_by_layer': ms_l}))


if __name__ == '__main__' and '--rq' in sys.argv:
    rq_mixed()
