#!/usr/bin/env python3
"""Host script with the plumbing of the reference's render.py (render_video_gen, :97-305) on the HIP path:
cfg merge -> build renderer -> (optional) load `network_state_dict` -> source imgs_info -> list of query poses ->
`renderer(data)` per pose -> save colour / depth images.  The reference reads Matterport3D through Habitat; here
the source panoramas are synthetic (or any .npy stack given with --imgs / --poses) since data acquisition is out
of scope.  Needs an MI355X.

  python render.py --height 256 --width 512 --samples 32 --num-poses 4 --out out_render
"""
import argparse
import math
import os

import numpy as np
import torch

from bench import synthetic_case, synthetic_weights, _smooth_pano
from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, imgs_info_to_torch
from panogrf_amd.renderer import name2network

DEFAULT_CFG = dict(network='neuray_gen', dataset_name='m3d', use_wrap_padding=True, use_disp=True, min_depth=0.5,
                   max_depth=15.0, use_hierarchical_sampling=True, render_depth=True, ray_batch_num=4096,
                   dist_decoder_cfg={'use_vis': False}, fine_dist_decoder_cfg={'use_vis': False})
# the MVS stage of the shipped yaml (configs/train/gen_hr_1.0/neuray_gen_cv_erp_mono_stereo_uniform_512x1024.yaml)
MVS_CFG = dict(init_net_type='cost_volume', net='Equi', contain_dnet=True, mono_net='UniFuse', seq_len=3, mvs_min_depth=0.1,
               mvs_max_depth=10.0, MAGNET_sampling_range=3, MAGNET_num_samples=5, fixed_sigma=0.5, use_depth_sampling=True,
               out_type='depth', interpolation_mode='bilinear', cnn3d_num_layer=3)


class ConstantMono(torch.nn.Module):
    """Stand-in for the mono-depth net (UniFuse, not part of this repository): a constant depth prior and zero mono
    features behind the call interface of network/omni_mvsnet/pipeline3_model.py:664-671."""

    def __init__(self, depth):
        super().__init__()
        self.depth = float(depth)

    def forward(self, erp, cube=None, dnet=True):
        n, _, h, w = erp.shape
        return {'pred_depth': torch.full((n, 1, h, w), self.depth, device=erp.device),
                'mono_feat': torch.zeros(n, 32, h // 2, w // 2, device=erp.device)}


def save_ppm(path, rgb):                      # rgb uint8 [h,w,3]; no image library needed
    with open(path, 'wb') as f:
        f.write(b'P6\n%d %d\n255\n' % (rgb.shape[1], rgb.shape[0]))
        f.write(rgb.tobytes())


def check_loaded(result):
    """A reference checkpoint must fill every parameter of the render path; only the init net (MVS stage, passed to the
    renderer separately) and the direct-rendering solver may be left over (render.py:150-153 of the reference loads
    strictly)."""
    need = ('dist_decoder.', 'agg_net.', 'fine_dist_decoder.', 'fine_agg_net.', 'image_encoder.', 'vis_encoder.')
    missing = [k for k in result.missing_keys if k.startswith(need)]
    extra = [k for k in result.unexpected_keys if not k.startswith(('init_net.', 'sph_fitter.'))]
    if missing or extra:
        raise SystemExit(f'checkpoint does not match the renderer configuration: {len(missing)} missing '
                         f'(e.g. {missing[:3]}), {len(extra)} unexpected (e.g. {extra[:3]})')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--height', type=int, default=256)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--samples', type=int, default=32)
    ap.add_argument('--views', type=int, default=2)
    ap.add_argument('--num-poses', type=int, default=4)
    ap.add_argument('--ckpt', default=None, help="torch checkpoint with 'network_state_dict' (reference format)")
    ap.add_argument('--imgs', default=None, help='.npy [n,3,H,W] source panoramas in [0,1]')
    ap.add_argument('--poses', default=None, help='.npy [n,3,4] world->camera poses of the source panoramas')
    ap.add_argument('--out', default='out_render')
    ap.add_argument('--mvs', action='store_true',
                    help='run the MVS stage (init net: cost volume + 3-D regulariser, render.py:264-278 flow with src_imgs_info) '
                         'instead of random ray features; the mono prior is a constant depth (--mono-depth)')
    ap.add_argument('--mono-depth', type=float, default=3.0)
    args = ap.parse_args()
    H, W, dn = args.height, args.width, args.samples
    cfg = {**DEFAULT_CFG, 'height': H, 'width': W, 'depth_sample_num': dn, 'fine_depth_sample_num': dn, 'sample_num': dn}
    if args.mvs:
        cfg.update(MVS_CFG, depth_height=H // 2, depth_width=W // 2)
        renderer = name2network[cfg['network']](cfg, d_net=ConstantMono(args.mono_depth))
    else:
        renderer = name2network[cfg['network']](cfg)
    if args.ckpt:
        ckpt = torch.load(args.ckpt, map_location='cpu')
        check_loaded(renderer.load_state_dict(ckpt['network_state_dict'], strict=False))
    else:   # random-init encoders + seeded per-ray nets with a density boost (no pretrained files offline)
        renderer.load_state_dict(synthetic_weights(0), strict=False)
    renderer = renderer.cuda().eval()

    if args.imgs:
        imgs, w2c = np.load(args.imgs), np.load(args.poses)
    else:
        case = synthetic_case(H, W, args.views, dn)
        imgs, w2c = case['frame']['imgs'].numpy(), case['frame']['w2c'].numpy()
    n = imgs.shape[0]
    ref = imgs_info_to_torch(build_imgs_info(imgs, w2c, [[cfg['min_depth'], cfg['max_depth']]] * n), 'cuda')
    src = None
    if args.mvs:       # the reference pairs every reference view with its neighbour as MVS source (render.py:266-268: ids [2,0])
        order = list(range(n))[::-1]
        src = imgs_info_to_torch(build_imgs_info(imgs[order], w2c[order], [[cfg['min_depth'], cfg['max_depth']]] * n), 'cuda')
    else:
        g = torch.Generator().manual_seed(99)
        ref['ray_feats'] = torch.randn(n, 32, H // 8, W // 8, generator=g).cuda()    # stands in for the init net (MVS) output
    os.makedirs(args.out, exist_ok=True)
    for qi in range(args.num_poses):          # query poses on a small circle between the source cameras
        a = 2 * math.pi * qi / max(args.num_poses, 1)
        pose = np.concatenate([np.eye(3), np.array([[0.1 * math.cos(a)], [0.0], [0.1 * math.sin(a)]])], 1)
        que = imgs_info_to_torch(build_render_imgs_info(pose, (H, W), [cfg['min_depth'], cfg['max_depth']]), 'cuda')
        with torch.no_grad():
            data = {'que_imgs_info': que, 'ref_imgs_info': dict(ref), 'eval': True}
            if src is not None:
                data['src_imgs_info'] = dict(src)
            info = renderer(data)
        rgb = info['pixel_colors_nr_fine'].reshape(H, W, 3).clamp(0, 1).mul(255).round().byte().cpu().numpy()
        save_ppm(os.path.join(args.out, f'{qi}-nr_fine.ppm'), rgb)
        np.save(os.path.join(args.out, f'{qi}-depth.npy'), info['render_depth_fine'].reshape(H, W).cpu().numpy())
        print(f'pose {qi}: mean rgb {float(info["pixel_colors_nr_fine"].mean()):.4f}, '
              f'mean depth {float(info["render_depth_fine"].mean()):.3f}')


if __name__ == '__main__':
    main()
