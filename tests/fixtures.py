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
