"""Golden vectors for the MVS stage around the cost volume, produced by running the
REFERENCE's own code (build container only):  python tests/golden/make_mvs_fixtures.py

What runs: `FullPipeline.estimate_depth_using_cost_volume_v3_erp` / `..._multiview`
(network/omni_mvsnet/pipeline3_model.py:602-949, :951-1300) on an object assembled from the reference's own factory
(`models/test_models.py:initialize_cost_volume_network`: Equi feature net, UNet2, decoders1, decoders2), and the tail
of `CostVolumeInitNet.forward` (network/init_net.py:578-650: res_net, depth_conv, out_conv).  What is replaced: the
mono-depth net (UniFuse needs its checkpoint and the cv2-based cube-map conversion; SURVEY marks it out of scope): a
small seeded conv net stands in, called through the same `self.d_net(erp, cube, dnet=True)` interface with the same
ImageNet normalisation in front.  `.cuda()` calls of the reference are no-ops here (CPU run).

Weights are not stored: every parameter is overwritten in state_dict order by `seeded_state` (shared with the tests),
the fixture holds inputs, key lists and outputs only.
"""
import math
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
from mvs_fixtures import seeded_state, StubMono, MVS_ARGS  # noqa: E402  (tests/mvs_fixtures.py)


def build_pipeline(H=32, W=64):
    """The reference's FullPipeline assembled from its own factory, the stand-in mono net in place of UniFuse, seeded weights.
    -> (fp, args, [(key, shape)])   (shared with make_uncert_fixtures.py)"""
    torch.nn.Module.cuda = lambda self, *a, **k: self          # the reference moves new layers to the GPU at build time
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models.test_models import initialize_cost_volume_network
    from network.omni_mvsnet.pipeline3_model import FullPipeline

    args = dict(MVS_ARGS, height=H, width=W, mono_height=H, mono_width=W, depth_height=H, depth_width=W)
    torch.manual_seed(0)
    unet, unet3d, dec1, dec2 = initialize_cost_volume_network(args, layers=5, size=4, use_wrap_padding=True, use_v_input=False,
                                                             out_channels=1, cost_volume_channels=64)
    fp = FullPipeline.__new__(FullPipeline)
    nn.Module.__init__(fp)
    fp.args = args
    fp.contain_dnet = True
    fp.unet, fp.unet3d, fp.decoders1, fp.decoders2 = unet, unet3d, dec1, dec2
    fp.d_net = StubMono()
    fp.sampling_range, fp.n_samples = args['MAGNET_sampling_range'], args['MAGNET_num_samples']
    fp.k_list = fp.depth_sampling()
    fp.weighting = args['MAGNET_mvs_weighting']
    fp.cost_volume_channels = 64
    fp.G = 1
    fp.use_cost_volume = 'v3_erp'
    fp.e2c_mono = fp.e2c = None
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    fp.normalize = lambda x: (x - mean) / std
    fp.e2c_process = lambda panos, e2c: torch.zeros(panos.shape[0], panos.shape[1], 4, 24, 3)   # cube maps: unused by the stub
    fp.eval()
    sd = fp.state_dict()
    shapes = [(k, tuple(v.shape)) for k, v in sd.items()]
    fp.load_state_dict(seeded_state(shapes, 61))
    return fp, args, shapes


def main():
    torch.set_num_threads(1)
    R.setup()
    H, W = 32, 64
    fp, args, shapes = build_pipeline(H, W)
    import network.init_net as init_net

    g = torch.Generator().manual_seed(5)
    B = 2
    panos = torch.stack([torch.stack([smooth_pano(g, 3, H, W) for _ in range(2)]) for _ in range(B)])     # [B,2,3,H,W]
    panos_cl = panos.permute(0, 1, 3, 4, 2).contiguous()
    rots, trans = poses(71, B, 2)
    with torch.no_grad():
        ret = fp.estimate_depth_using_cost_volume_v3_erp(panos_cl, rots, trans, args['mvs_min_depth'], args['mvs_max_depth'])
    arrs = {'two.panos': panos_cl.numpy(), 'two.rots': rots.numpy(), 'two.trans': trans.numpy(),
            'two.depth': ret['depth'].numpy(), 'two.cost_reg': ret['cost_reg'].numpy(),
            'two.rectified_depth_d1': ret['rectified_depth_d1'].numpy(), 'two.mono_depth_ref': ret['mono_depth_ref'].numpy(),
            'two.cost_volume': ret['mid_outputs']['cost_volume'].numpy()}
    # feature net alone (for the CPU test of the torch mirror)
    with torch.no_grad():
        feat = fp.unet(panos[:, 1], None)
    arrs['equi.x'] = panos[:, 1].numpy()
    arrs['equi.out'] = feat.numpy()

    # multi-view: S = 4 stacked views [ref, s1, s2, s3]
    S = 4
    panom = torch.stack([torch.stack([smooth_pano(g, 3, H, W) for _ in range(S)])]).permute(0, 1, 3, 4, 2).contiguous()
    rotm, tranm = poses(81, 1, S)
    with torch.no_grad():
        retm = fp.estimate_depth_using_cost_volume_v3_erp_multiview(panom, rotm, tranm, 0, args['mvs_min_depth'], args['mvs_max_depth'])
    arrs.update({'mv.panos': panom.numpy(), 'mv.rots': rotm.numpy(), 'mv.trans': tranm.numpy(),
                 'mv.depth': retm['depth'].numpy(), 'mv.cost_reg': retm['cost_reg'].numpy()})
    arrs['pipe.keys'] = np.frombuffer('\n'.join(f'{k}:{",".join(map(str, s))}' for k, s in shapes).encode(), np.uint8)

    # ---- CostVolumeInitNet tail: depth -> ray_feats (network/init_net.py:578-650), on the two-view depth above
    cfg = dict(args, use_wrap_padding=True, handle_distort=False, handle_distort_all=False, handle_distort_input_all=False,
               autoencoder=False, mvs_min_depth=args['mvs_min_depth'],
               mvs_max_depth=args['mvs_max_depth'], uncert_tune=False, seq_len=3, min_depth=0.5, max_depth=15.0)
    inet = init_net.CostVolumeInitNet.__new__(init_net.CostVolumeInitNet)
    nn.Module.__init__(inet)
    inet.cfg = {**init_net.CostVolumeInitNet.default_cfg, **cfg}
    from network.ops import ResUNetLight, conv3x3, conv1x1, ResidualBlock
    norm_layer = lambda dim: nn.InstanceNorm2d(dim, track_running_stats=False, affine=True)
    inet.res_net = ResUNetLight(inet.cfg, out_dim=32, use_wrap_padding=True)
    inet.depth_conv = nn.Sequential(conv3x3(1, 32, use_wrap_padding=True),
                                    ResidualBlock(32, 32, norm_layer=norm_layer, use_wrap_padding=True),
                                    conv1x1(32, 32, use_wrap_padding=True))
    inet.out_conv = nn.Sequential(conv3x3(64, 32, use_wrap_padding=True),
                                  ResidualBlock(32, 32, norm_layer=norm_layer, use_wrap_padding=True),
                                  conv1x1(32, 32, use_wrap_padding=True))
    inet.depth_net = fp
    inet.eval()
    isd = {k: v for k, v in inet.state_dict().items() if not k.startswith('depth_net.')}
    ishapes = [(k, tuple(v.shape)) for k, v in isd.items()]
    inet.load_state_dict({**seeded_state(ishapes, 91), **{'depth_net.' + k: v for k, v in fp.state_dict().items()}})
    Hf, Wf = 64, 128                                   # full-resolution source images; depth resolution stays H x W
    imgs_ref = torch.stack([smooth_pano(g, 3, Hf, Wf) for _ in range(B)])
    imgs_src = torch.stack([smooth_pano(g, 3, Hf, Wf) for _ in range(B)])
    ref_info = {'imgs': imgs_ref, 'rots': rots[:, 1], 'trans': trans[:, 1]}
    src_info = {'imgs': imgs_src, 'rots': rots[:, 0], 'trans': trans[:, 0]}
    with torch.no_grad():
        out = inet(ref_info, src_info, True)            # is_train=True only skips torch.cuda.synchronize()/empty_cache()
    arrs.update({'init.ref_imgs': imgs_ref.numpy(), 'init.src_imgs': imgs_src.numpy(), 'init.ref_rots': rots[:, 1].numpy(),
                 'init.ref_trans': trans[:, 1].numpy(), 'init.src_rots': rots[:, 0].numpy(), 'init.src_trans': trans[:, 0].numpy(),
                 'init.ray_feats': out['ray_feats'].numpy(), 'init.mvs_depth': out['mvs_depth'].numpy(),
                 'init.keys': np.frombuffer('\n'.join(f'{k}:{",".join(map(str, s))}' for k, s in ishapes).encode(), np.uint8)})
    path = os.path.join(HERE, 'mvs_fixtures.npz')
    np.savez_compressed(path, **arrs)
    print('mvs_fixtures', os.path.getsize(path) // 1024, 'KiB; depth range', float(ret['depth'].min()), float(ret['depth'].max()),
          'ray_feats std', float(out['ray_feats'].std()), 'n keys', len(shapes), len(ishapes))
    for k, s in shapes:
        if not k.startswith('unet3d'):
            print(k, s)
    for k, s in ishapes[:6] + ishapes[-12:]:
        print('init_net', k, s)


if __name__ == '__main__':
    main()
