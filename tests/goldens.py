"""Loader for tests/golden/*.npz (written by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        self.meta = json.loads(bytes(z['meta']).decode())
        self.arrays = {k: z[k] for k in z.files if k != 'meta'}

    def t(self, key):
        return torch.from_numpy(np.array(self.arrays[key]))

    def has(self, key):
        return key in self.arrays

    def state(self, prefix):
        """All '<prefix>/state/<k>' arrays as a {k: tensor} state_dict."""
        p = prefix + '/state/' if prefix else 'state/'
        return {k[len(p):]: self.t(k) for k in self.arrays if k.startswith(p)}

    def cases(self, head):
        return sorted(k for k in self.meta if k.startswith(head))

    def seeded_state(self, case):
        """State of a fixture that stores sha256 hashes instead of weights (F13): the reference's default init under the
        fixture's seed, rebuilt through the host classes (same construction order, same RNG draws) and held, tensor by tensor,
        to the hashes captured from the reference.  Pure host code: runs without a GPU."""
        import hashlib
        import stribor_amd as st
        from stribor_amd.util import flowdesc as fd
        m = self.meta[case]
        torch.manual_seed(m['seed'])
        flow = fd.build_flow(st, m['desc'], m['dim'])
        state = {k: v.clone() for k, v in flow.state_dict().items()}
        for i, d in enumerate(m['desc']):
            if d['kind'] == 'permute':
                state[f'transforms.{i}.permutation'] = flow.transforms[i].permutation.clone()
        want = m['state_sha256']
        assert set(state) == set(want), (sorted(set(state) ^ set(want)))
        for k, v in state.items():
            a = np.ascontiguousarray(v.detach().numpy())
            h = hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()
            assert h == want[k], f'{case}: state tensor {k} differs from the reference\'s'
        return state
