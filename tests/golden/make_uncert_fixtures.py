"""Golden vectors for cfg['uncert_tune'] (the MVS depth net inside the reference's UncertWrapper), produced by running the REFERENCE's
own code (build container only):  python tests/golden/make_uncert_fixtures.py

What runs: `UncertWrapper.estimate_depth_using_cost_volume` (network/omni_mvsnet/uncert_wrapper.py:95-150) around the FullPipeline of
make_mvs_fixtures.py (same factory, same seeded weights, same stand-in mono net), for both variance activations (`new_uncert_tune` off /
on), and `CostVolumeInitNet.forward` with `uncert_tune` (network/init_net.py:578-650 through `construct_cost_volume_with_src`, :233-329)
-> 'mvs_uncert'.  What is replaced: the wrapper's constructor loads a checkpoint from disk (`load_checkpoint`) -- the layers are built by
the same three statements, then every parameter and BatchNorm running statistic is overwritten by `seeded_state` (shared with the tests).
`.cuda()` calls are no-ops here (CPU run).  The fixture holds inputs, key lists and outputs only.
"""
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
from make_mvs_fixtures import build_pipeline  # noqa: E402
from mvs_fixtures import seeded_state  # noqa: E402  (tests/mvs_fixtures.py)


def main():
    torch.set_num_threads(1)
    R.setup()
    H, W = 32, 64
    fp, args, shapes = build_pipeline(H, W)
    import network.init_net as init_net
    from network.omni_mvsnet.uncert_wrapper import UncertWrapper
    from network.ops import ResUNetLight, conv3x3, conv1x1, ResidualBlock
    from models.common_blocks import Upscale

    cfg = dict(args, use_wrap_padding=True, handle_distort=False, handle_distort_all=False, handle_distort_input_all=False, autoencoder=False,
               uncert_tune=True, seq_len=3, min_depth=0.5, max_depth=15.0)

    def wrapper(c):
        w = UncertWrapper.__new__(UncertWrapper)       # __init__ minus load_checkpoint(args['mvs_checkpoints_dir'], ...): uncert_wrapper.py:53-93
        nn.Module.__init__(w)
        w.cfg = c
        w.mvs_net = fp
        w.volume_conv2d = nn.Sequential(conv3x3(64, 32, use_wrap_padding=True), ResidualBlock(32, 32, use_wrap_padding=True),
                                        conv1x1(32, 32, use_wrap_padding=True))
        w.depth_conv = nn.Sequential(conv3x3(1, 32, use_wrap_padding=True), ResidualBlock(32, 32, use_wrap_padding=True),
                                     conv1x1(32, 32, use_wrap_padding=True))
        w.out_conv = nn.Sequential(conv3x3(64, 32, use_wrap_padding=True), ResidualBlock(32, 32, use_wrap_padding=True), Upscale(),
                                   conv3x3(32, 16, use_wrap_padding=True), Upscale(), conv1x1(16, 1, use_wrap_padding=True))
        w.eval()
        head = [(k, tuple(v.shape)) for k, v in w.state_dict().items() if not k.startswith('mvs_net.')]
        w.load_state_dict({**seeded_state(head, 131), **{'mvs_net.' + k: v for k, v in fp.state_dict().items()}})
        return w, head

    g = torch.Generator().manual_seed(17)
    B = 2
    panos = torch.stack([torch.stack([smooth_pano(g, 3, H, W) for _ in range(2)]) for _ in range(B)]).permute(0, 1, 3, 4, 2).contiguous()
    rots, trans = poses(73, B, 2)
    arrs = {'in.panos': panos.numpy(), 'in.rots': rots.numpy(), 'in.trans': trans.numpy()}
    for tag, extra in (('elu', {}), ('softplus', {'new_uncert_tune': True})):
        w, head = wrapper(dict(cfg, **extra))
        with torch.no_grad():
            ret = w.estimate_depth_using_cost_volume(panos, rots, trans, args['mvs_min_depth'], args['mvs_max_depth'])
        arrs[tag + '.depth'] = ret['depth'].numpy()
        arrs[tag + '.var'] = ret['var'].numpy()
        arrs[tag + '.pred_final'] = ret['pred_final'].numpy()
    arrs['head.keys'] = np.frombuffer('\n'.join(f'{k}:{",".join(map(str, s))}' for k, s in head).encode(), np.uint8)
    arrs['pipe.keys'] = np.frombuffer('\n'.join(f'{k}:{",".join(map(str, s))}' for k, s in shapes).encode(), np.uint8)

    # ---- the init net with uncert_tune: forward -> {'mvs_depth', 'mvs_uncert', 'ray_feats'}
    inet = init_net.CostVolumeInitNet.__new__(init_net.CostVolumeInitNet)
    nn.Module.__init__(inet)
    inet.cfg = {**init_net.CostVolumeInitNet.default_cfg, **cfg}
    norm_layer = lambda dim: nn.InstanceNorm2d(dim, track_running_stats=False, affine=True)
    inet.res_net = ResUNetLight(inet.cfg, out_dim=32, use_wrap_padding=True)
    inet.depth_conv = nn.Sequential(conv3x3(1, 32, use_wrap_padding=True), ResidualBlock(32, 32, norm_layer=norm_layer, use_wrap_padding=True),
                                    conv1x1(32, 32, use_wrap_padding=True))
    inet.out_conv = nn.Sequential(conv3x3(64, 32, use_wrap_padding=True), ResidualBlock(32, 32, norm_layer=norm_layer, use_wrap_padding=True),
                                  conv1x1(32, 32, use_wrap_padding=True))
    w, head = wrapper(cfg)
    inet.depth_net = w
    inet.eval()
    ishapes = [(k, tuple(v.shape)) for k, v in inet.state_dict().items() if not k.startswith('depth_net.')]
    inet.load_state_dict({**seeded_state(ishapes, 91), **{'depth_net.' + k: v for k, v in w.state_dict().items()}})
    Hf, Wf = 64, 128
    imgs_ref = torch.stack([smooth_pano(g, 3, Hf, Wf) for _ in range(B)])
    imgs_src = torch.stack([smooth_pano(g, 3, Hf, Wf) for _ in range(B)])
    ref_info = {'imgs': imgs_ref, 'rots': rots[:, 1], 'trans': trans[:, 1]}
    src_info = {'imgs': imgs_src, 'rots': rots[:, 0], 'trans': trans[:, 0]}
    with torch.no_grad():
        out = inet(ref_info, src_info, True)            # is_train=True only skips torch.cuda.synchronize()/empty_cache()
    arrs.update({'init.ref_imgs': imgs_ref.numpy(), 'init.src_imgs': imgs_src.numpy(), 'init.ref_rots': rots[:, 1].numpy(),
                 'init.ref_trans': trans[:, 1].numpy(), 'init.src_rots': rots[:, 0].numpy(), 'init.src_trans': trans[:, 0].numpy(),
                 'init.ray_feats': out['ray_feats'].numpy(), 'init.mvs_depth': out['mvs_depth'].numpy(), 'init.mvs_uncert': out['mvs_uncert'].numpy(),
                 'init.keys': np.frombuffer('\n'.join(f'{k}:{",".join(map(str, s))}' for k, s in ishapes).encode(), np.uint8)})
    path = os.path.join(HERE, 'uncert_fixtures.npz')
    np.savez_compressed(path, **arrs)
    print('uncert_fixtures', os.path.getsize(path) // 1024, 'KiB; var range', float(arrs['elu.var'].min()), float(arrs['elu.var'].max()),
          '| softplus', float(arrs['softplus.var'].min()), float(arrs['softplus.var'].max()), '| head keys', len(head))


if __name__ == '__main__':
    main()
