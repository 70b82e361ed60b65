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
