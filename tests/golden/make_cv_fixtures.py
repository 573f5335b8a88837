"""Golden vectors for the spherical cost volume, hypothesis generation and the two 3-D regularisers,
produced by running the REFERENCE (build container only).  python tests/golden/make_cv_fixtures.py

The regulariser weights are NOT stored (UNet2 has 16.9 M parameters): every parameter/buffer is
overwritten, in state_dict order, by `seeded_params` below, which the tests re-run to rebuild the
identical weights; the fixtures hold inputs and expected outputs only.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402
from make_fixtures import rot_xyz  # noqa: E402


def seeded_params(shapes, seed):
    """Deterministic weights for a list of (name, shape): conv weights ~ N(0, 2/fan_in), biases and BN
    affine small, BN running stats positive.  Shared by the fixture generator and the tests."""
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


def poses(seed, B, S):
    g = torch.Generator().manual_seed(seed)
    rots = torch.stack([torch.stack([torch.from_numpy(rot_xyz(*((torch.rand(3, generator=g) - 0.5) * 0.4).tolist()))
                                     for _ in range(S)]) for _ in range(B)])
    trans = (torch.rand(B, S, 3, generator=g) - 0.5) * 0.8
    return rots, trans


def main():
    torch.set_num_threads(1)
    R.setup()
    from models.spherical_cost_volume import calculate_cost_volume_erp
    from models.spherical_cost_volume_mv import calculate_cost_volume_erp_multiview
    from models.cost_reg import CostRegNet
    from models.common_blocks import Conv3DBlockv2, UNet2
    import torch.nn as nn
    args = dict(dataset_name='m3d', contain_dnet=True, mono_uncertainty=False)
    arrs = {}

    # ---- hypotheses: restated from pipeline3_model.py:537-545,723-733,774-815 with the reference's k_list
    from network.omni_mvsnet.pipeline3_model import FullPipeline
    stub = type('S', (), {'sampling_range': 3, 'n_samples': 5})()
    k_list = FullPipeline.depth_sampling(stub)
    g = torch.Generator().manual_seed(11)
    B, h, w = 2, 8, 16
    ref_mu = 0.3 + 9.9 * torch.rand(B, 1, h, w, generator=g)
    ref_mu[0, 0, 0, 0] = 0.05       # clamps at both ends
    ref_mu[0, 0, 0, 1] = 11.0
    guided = [torch.clamp(ref_mu + k * 0.5, min=0.1, max=10) for k in k_list]
    uni = torch.linspace(0.1, 10, 59).reshape(1, 59, 1, 1).repeat(B, 1, h, w)
    dv = torch.sort(torch.cat(guided + [uni], 1), dim=1)[0]
    arrs.update({'hyp.k_list': np.asarray(k_list, np.float64), 'hyp.ref_mu': ref_mu.numpy(), 'hyp.out': dv.numpy()})

    # ---- 2-view cost volume (a-11)
    C, D = 32, 16
    feats = torch.randn(B, 2, h, w, C, generator=g)
    rots, trans = poses(21, B, 2)
    dsub = dv[:, ::4].contiguous()          # 16 of the 64 planes
    cv = calculate_cost_volume_erp(args, feats, None, trans, rots, depth_volume=dsub, cost_type='abs_diff',
                                   nghbr_gmms=ref_mu)
    cvd = calculate_cost_volume_erp(args, feats, None, trans, rots, depth_volume=dsub[:, :4], cost_type='dot',
                                    nghbr_gmms=ref_mu)
    arrs.update({'cv2.feats': feats.numpy(), 'cv2.rots': rots.numpy(), 'cv2.trans': trans.numpy(),
                 'cv2.depth_volume': dsub.numpy(), 'cv2.out': cv.numpy(), 'cv2.out_dot': cvd.numpy()})
    # ---- multi-view (S=4)
    S = 4
    featm = torch.randn(1, S, h, w, C, generator=g)
    rotm, tranm = poses(31, 1, S)
    dm = dv[:1, ::8].contiguous()
    cvm = calculate_cost_volume_erp_multiview(args, featm, None, tranm, rotm, depth_volume=dm, cost_type='abs_diff',
                                              nghbr_gmms=ref_mu[:1], curr_idx=0)
    cvm1 = calculate_cost_volume_erp_multiview(args, featm, None, tranm, rotm, depth_volume=dm[:, :2],
                                               cost_type='abs_diff', nghbr_gmms=ref_mu[:1], curr_idx=1)
    arrs.update({'cvm.feats': featm.numpy(), 'cvm.rots': rotm.numpy(), 'cvm.trans': tranm.numpy(),
                 'cvm.depth_volume': dm.numpy(), 'cvm.out': cvm.numpy(), 'cvm.out_idx1': cvm1.numpy()})

    # ---- CostRegNet (a-12)
    net = CostRegNet(dict(inplace_abn=False, use_wrap_padding=True), in_channels=32).eval()
    sd = net.state_dict()
    net.load_state_dict(seeded_params([(k, tuple(v.shape)) for k, v in sd.items()], 41))
    x = torch.randn(1, 32, 8, 8, 16, generator=g)
    with torch.no_grad():
        y = net(x)
    arrs.update({'crn.x': x.numpy(), 'crn.out': y.numpy(),
                 'crn.keys': np.frombuffer('\n'.join(f'{k}:{",".join(map(str, v.shape))}' for k, v in sd.items()).encode(), np.uint8)})

    # ---- UNet2 (a-13), built like models/test_models.py:84-146 (size=4, cnn3d_num_layer=3)
    def blk(cin, cout, pool=True):
        kw = {} if pool else dict(pooling=nn.Identity())
        return Conv3DBlockv2(in_channels=cin, out_channels=cout, kernel_size=(3, 3, 3), stride=(1, 1, 1),
                             padding=(1, 1, 1), use_batch_norm=False, use_wrap_padding=True, use_v_input=False, **kw)
    enc = [blk(32, 64), blk(64, 128), blk(128, 256), blk(256, 512, pool=False)]
    dec = [blk(128, 1, pool=False), blk(256, 64, pool=False), blk(512, 128, pool=False)]
    unet = UNet2(nn.ModuleList(enc), nn.ModuleList(dec), interpolation='trilinear', name='unet3d').eval()
    sd = unet.state_dict()
    unet.load_state_dict(seeded_params([(k, tuple(v.shape)) for k, v in sd.items()], 51))
    xu = torch.randn(1, 32, 8, 8, 16, generator=g)
    with torch.no_grad():
        yu = unet(xu)
    arrs.update({'unet.x': xu.numpy(), 'unet.out': yu.numpy(),
                 'unet.keys': np.frombuffer('\n'.join(f'{k}:{",".join(map(str, v.shape))}' for k, v in sd.items()).encode(), np.uint8)})
    path = os.path.join(HERE, 'cv_fixtures.npz')
    np.savez_compressed(path, **arrs)
    print('cv_fixtures', os.path.getsize(path) // 1024, 'KiB', 'unet out std', float(yu.std()), 'crn out std', float(y.std()),
          'cv mean', float(cv.mean()))


if __name__ == '__main__':
    main()
