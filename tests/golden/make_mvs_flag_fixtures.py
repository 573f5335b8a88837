"""Golden vectors for three MVS-stage switches of SURVEY.md 8(b)'s key list that no shipped configuration sets, produced by
running the REFERENCE's own FullPipeline.estimate_depth_using_cost_volume_v3_erp (build container only):

  python tests/golden/make_mvs_flag_fixtures.py        ->  tests/golden/mvs_flag_fixtures.npz

  gw.*   group_wise + group_num = 8 with use_new_reg3dnet (pipeline3_model.py:849-853, models/test_models.py:95-100): the
         32-channel cost volume is averaged in 8 groups of 4 and regularised by CostRegNet(in_channels = 64 // 8)
  rr.*   revise_range + fixed_dist (pipeline3_model.py:781-796): the uniform planes span [mu - fixed_dist, mu + fixed_dist]
         per pixel (with the reference's own off-by-one: d_min twice, d_max never reached)
  mu.*   mono_uncertainty + basic_sigma (pipeline3_model.py:660-661, :718-731): the mono net returns 'pred' = [depth, sigma]
         and the guided planes are mu + max(sigma, basic_sigma) * k

Same construction as make_mvs_fixtures.py (seeded weights regenerated from the stored (name, shape) lists, stand-in mono net)."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_import as R  # noqa: E402
from make_fixtures import smooth_pano  # noqa: E402
from make_cv_fixtures import poses  # noqa: E402
from mvs_fixtures import seeded_state, StubMono, StubMonoUncert, MVS_ARGS  # noqa: E402


def build(args, mono):
    from models.test_models import initialize_cost_volume_network
    from network.omni_mvsnet.pipeline3_model import FullPipeline
    torch.manual_seed(0)
    unet, unet3d, dec1, dec2 = initialize_cost_volume_network(args, layers=5, size=4, use_wrap_padding=True, use_v_input=False,
                                                             out_channels=1, cost_volume_channels=64)
    fp = FullPipeline.__new__(FullPipeline)
    nn.Module.__init__(fp)
    fp.args = args
    fp.contain_dnet = True
    fp.unet, fp.unet3d, fp.decoders1, fp.decoders2 = unet, unet3d, dec1, dec2
    fp.d_net = mono
    fp.sampling_range, fp.n_samples = args['MAGNET_sampling_range'], args['MAGNET_num_samples']
    fp.k_list = fp.depth_sampling()
    fp.weighting = args['MAGNET_mvs_weighting']
    fp.cost_volume_channels = 64
    fp.G = args['group_num'] if args.get('group_num', 1) > 1 else 1
    fp.use_cost_volume = 'v3_erp'
    fp.e2c_mono = fp.e2c = None
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    fp.normalize = lambda x: (x - mean) / std
    fp.e2c_process = lambda panos, e2c: torch.zeros(panos.shape[0], panos.shape[1], 4, 24, 3)
    fp.eval()
    shapes = [(k, tuple(v.shape)) for k, v in fp.state_dict().items()]
    return fp, shapes


def main():
    torch.set_num_threads(1)
    R.setup()
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    H, W = 32, 64
    base = dict(MVS_ARGS, height=H, width=W, mono_height=H, mono_width=W, depth_height=H, depth_width=W, inplace_abn=False,
                use_wrap_padding=True)
    g = torch.Generator().manual_seed(15)
    B = 2
    panos = torch.stack([torch.stack([smooth_pano(g, 3, H, W) for _ in range(2)]) for _ in range(B)]).permute(0, 1, 3, 4, 2).contiguous()
    rots, trans = poses(73, B, 2)
    arrs = {'in.panos': panos.numpy(), 'in.rots': rots.numpy(), 'in.trans': trans.numpy()}
    enc = lambda shapes: np.frombuffer('\n'.join(f'{k}:{",".join(map(str, s))}' for k, s in shapes).encode(), np.uint8)
    cases = {
        'gw': (dict(base, group_wise=True, group_num=8, use_new_reg3dnet=True), StubMono(), 161),
        'rr': (dict(base, revise_range=True, fixed_dist=1.0), StubMono(), 163),
        'mu': (dict(base, mono_uncertainty=True, basic_sigma=0.1, relaxation_factor=1.0), StubMonoUncert(), 165),   # (relaxation_factor must exist: the reference evaluates args['relaxation_factor'] in self.args -- a VALUE looked up as a key, False -- at :725)
    }
    for tag, (args, mono, seed) in cases.items():
        fp, shapes = build(args, mono)
        fp.load_state_dict(seeded_state(shapes, seed))
        with torch.no_grad():
            ret = fp.estimate_depth_using_cost_volume_v3_erp(panos, rots, trans, args['mvs_min_depth'], args['mvs_max_depth'])
        arrs[f'{tag}.depth'] = ret['depth'].numpy()
        arrs[f'{tag}.cost_reg'] = ret['cost_reg'].numpy()
        arrs[f'{tag}.mono_depth_ref'] = ret['mono_depth_ref'].numpy()
        arrs[f'{tag}.keys'] = enc(shapes)
        arrs[f'{tag}.seed'] = np.array([seed])
        print(tag, 'depth', float(ret['depth'].min()), float(ret['depth'].max()), 'cost_reg', tuple(ret['cost_reg'].shape), len(shapes), 'keys')
    path = os.path.join(HERE, 'mvs_flag_fixtures.npz')
    np.savez_compressed(path, **arrs)
    print('mvs_flag_fixtures', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
