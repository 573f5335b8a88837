"""Shared by tests/golden/make_mvs_fixtures.py (runs the reference) and the tests: the reference-style args of the MVS
stage, the seeded weights and the stand-in mono-depth net of the fixture (tests/golden/mvs_fixtures.npz)."""
import math
import os

import numpy as np
import torch
import torch.nn as nn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# the keys FullPipeline / initialize_cost_volume_network / CostVolumeInitNet read on this path, shipped values
# (configs/train/gen_hr_1.0/neuray_gen_cv_erp_mono_stereo_uniform_512x1024.yaml)
MVS_ARGS = dict(
    device='cpu', dataset_name='m3d', net='Equi', num_layers=18, imagenet_pretrained=False, with_sin=False, fusion='cee',
    se_in_fusion=True, contain_dnet=True, mono_net='UniFuse', mono_uncertainty=False, mono_uncert_tune=False, change_input=False,
    dnet=True, mvs_uncertainty=False, wo_hdh=False, use_depth_sampling=True, revise_range=False, group_wise=False,
    use_new_reg3dnet=False, cnn3d_num_layer=3, interpolation_mode='bilinear', out_type='depth', wo_mono_feat=False,
    MAGNET_sampling_range=3, MAGNET_num_samples=5, MAGNET_mvs_weighting='CW5', fixed_sigma=0.5, min_depth=0.1, max_depth=10.0,
    mvs_min_depth=0.1, mvs_max_depth=10.0, verbose=False, debug=False)


class StubMono(nn.Module):
    """Stand-in for the UniFuse mono-depth net behind the same call interface
    (`d_net(erp, cube, dnet=True) -> {'pred_depth', 'mono_feat'}`, pipeline3_model.py:664-671): a seeded 2-layer conv net
    on the (normalised) ERP image; the cube-map input is ignored."""

    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, 1, 1)
        self.c2 = nn.Conv2d(8, 32, 3, 2, 1)
        self.c3 = nn.Conv2d(8, 1, 3, 1, 1)

    def forward(self, erp, cube=None, dnet=True):
        h = torch.tanh(self.c1(erp))
        return {'pred_depth': 1.0 + 6.0 * torch.sigmoid(self.c3(h)), 'mono_feat': self.c2(h)}


class StubMonoUncert(StubMono):
    """The same stand-in with the output of a mono net trained with uncertainty (args mono_uncertainty):
    'pred' = [depth, sigma] (pipeline3_model.py:660-661, :718-719)."""

    def __init__(self):
        super().__init__()
        self.c4 = nn.Conv2d(8, 1, 3, 1, 1)

    def forward(self, erp, cube=None, dnet=True):
        h = torch.tanh(self.c1(erp))
        depth = 1.0 + 6.0 * torch.sigmoid(self.c3(h))
        return {'pred_depth': depth, 'pred': torch.cat([depth, 0.02 + 0.6 * torch.sigmoid(self.c4(h))], 1), 'mono_feat': self.c2(h)}


def seeded_state(shapes, seed):
    """Deterministic values for every entry of a state_dict given as [(name, shape)], in order."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in shapes:
        if name.endswith('num_batches_tracked'):
            out[name] = torch.zeros(shape, dtype=torch.long)
        elif name.endswith('running_var'):
            out[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith('running_mean'):
            out[name] = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            out[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif name.endswith('weight'):           # norm scales
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            out[name] = 0.1 * torch.randn(shape, generator=g)
    return out


def key_shapes(blob):
    res = []
    for ln in bytes(blob).decode().split('\n'):
        k, s = ln.split(':')
        res.append((k, tuple(int(t) for t in s.split(',')) if s else ()))
    return res


def load(name='mvs_fixtures'):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    out = {}
    for k in z.files:
        grp, name = k.split('.', 1)
        out.setdefault(grp, {})[name] = z[k] if name == 'keys' else torch.from_numpy(z[k])
    return out
