"""Small-batch training step (forward + backward of -log_prob.mean(), gradients into static .grad tensors) eager vs.
captured once into a HIP graph and replayed.  Run on an MI355X:  python tools/bench_graph_training.py [cfg2|cfg3|cfg4] [rows]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import flowdesc as fd
import stribor_amd as st
from bench_configs import timed


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    desc, dim = {'cfg2': (fd.cfg2_desc(), 64), 'cfg3': (fd.cfg3_desc(), 64), 'cfg4': (fd.cfg4_desc(), 128)}[name]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    flow = fd.build_flow(st, desc, dim).to(dev)
    x = torch.randn(rows, dim, device=dev)
    params = list(flow.parameters())

    def step():
        loss = -flow.log_prob(x).mean()
        grads = torch.autograd.grad(loss, params)
        return loss, grads

    eager_ms = timed(step, reps=10, inner=5)
    loss_e, grads_e = step()
    loss_e, grads_e = loss_e.detach().clone(), [g_.detach().clone() for g_ in grads_e]   # no eager autograd graph alive during capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        loss_g, grads_g = step()
    graph.replay()
    torch.cuda.synchronize()
    err = max(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(grads_g, grads_e))
    graph_ms = timed(graph.replay, reps=10, inner=20)
    print(json.dumps({'config': name, 'rows': rows, 'eager_ms_per_step': eager_ms, 'graph_ms_per_step': graph_ms,
                      'loss_equal': bool(torch.equal(loss_g.detach(), loss_e)), 'max_rel_grad_diff_vs_eager': err}))


if __name__ == '__main__':
    main()
