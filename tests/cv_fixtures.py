"""Helpers for the cost-volume / regulariser golden fixture (tests/golden/cv_fixtures.npz)."""
import math
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load():
    z = np.load(os.path.join(GOLDEN, 'cv_fixtures.npz'))
    out = {}
    for k in z.files:
        grp, name = k.split('.', 1)
        v = z[k]
        out.setdefault(grp, {})[name] = v if name in ('keys', 'k_list') else torch.from_numpy(v)
    return out


def key_shapes(blob):
    lines = bytes(blob).decode().split('\n')
    res = []
    for ln in lines:
        k, s = ln.split(':')
        res.append((k, tuple(int(t) for t in s.split(',')) if s else ()))
    return res


def seeded_params(shapes, seed):
    """Must stay identical to tests/golden/make_cv_fixtures.py:seeded_params (the fixtures were made
    with these weights)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in shapes:
        if name.endswith('num_batches_tracked'):
            out[name] = torch.zeros(shape, dtype=torch.long)
        elif name.endswith('running_var'):
            out[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith('running_mean'):
            out[name] = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 5:
            fan_in = shape[1] * 27
            out[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif name.endswith('bn.weight'):
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            out[name] = 0.1 * torch.randn(shape, generator=g)
    return out
