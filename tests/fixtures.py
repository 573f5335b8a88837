"""Loading of the committed golden fixtures (tests/golden/*.npz)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    frame, weights, groups = {}, {}, {}
    cfg = None
    for k in z.files:
        if k == 'cfg':
            cfg = json.loads(bytes(z[k]).decode())
            continue
        head, rest = k.split('.', 1)
        t = torch.from_numpy(z[k])
        if head == 'in':
            frame[rest] = t
        elif head == 'w':
            weights[rest] = t
        else:
            groups.setdefault(head, {})[rest] = t
    return frame, weights, cfg, groups


def rel_l2(a, b):
    return float(torch.linalg.norm((a - b).double()) / (torch.linalg.norm(b.double()) + 1e-30))


def max_abs(a, b):
    return float((a - b).abs().max())


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 10 * np.log10(1.0 / max(mse, 1e-30))


def seeded_renderer_state(shapes, seed):
    """Deterministic values for every entry of a renderer state_dict given as [(name, shape)], in order.
    Used by tests/golden/make_fixtures.py (to overwrite the reference's parameters) and by the tests
    (to rebuild the same weights): the fixture then needs inputs and outputs only."""
    import math
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in shapes:
        if name == 'sph_fitter.regs':
            out[name] = torch.tensor([0.0] + [0.001] * 3 + [0.005] * 5 + [0.05] * 7)
        elif len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            out[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif name.endswith('weight'):               # norm scales
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            out[name] = 0.1 * torch.randn(shape, generator=g)
    for pre in ('agg_net', 'fine_agg_net'):         # density boost so the image is not black
        k = pre + '.agg_impl.out_geometry_fc.2.bias'
        if k in out:
            out[k] = out[k] + 0.6
    for pre in ('dist_decoder', 'fine_dist_decoder'):
        k = pre + '.var_decoder.4.bias'
        if k in out:
            out[k] = out[k] + 2.0
    return out
