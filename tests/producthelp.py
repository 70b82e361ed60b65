"""Test helper: build the product flow / transform for a golden case and move it to the GPU."""
import torch

import flowdesc as fd
import stribor_amd as st


def product_flow(g, case, device='cuda'):
    m = g.meta[case]
    flow = fd.build_flow(st, m['desc'], m['dim'])
    flow.load_state_dict(g.state(case))
    return flow.to(device)


def product_transform(g, case, device='cuda'):
    m = g.meta[case]
    d = m['desc'][0]
    f = fd.build_transform(st, d)
    state = {k[len('transforms.0.'):]: v for k, v in g.state(case).items()}
    f.load_state_dict(state)
    return f.to(device)


def close(a, b, rtol=1e-5, atol=1e-5):
    """|a-b| <= atol + rtol*|b| — the north_star's 1e-5 relative bound with a 1e-5 floor at |b| < 1."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    ok = torch.allclose(a, b, rtol=rtol, atol=atol)
    if not ok:
        err = (a - b).abs()
        i = err.argmax()
        raise AssertionError(f'max abs err {err.max().item():.3e} at {i.item()} '
                             f'(got {a.flatten()[i].item():.7g}, want {b.flatten()[i].item():.7g})')
    return True
