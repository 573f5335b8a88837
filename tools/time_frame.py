"""Whole `renderer(data)` frame at the shipped configuration (512x1024 query and sources, 2 views, 64 + 64 samples, MVS
stage at depth 256x512, a constant mono-depth prior standing in for UniFuse): ms per frame with and without the
encoder / MVS-stage overlap and the source-map reuse.  python tools/time_frame.py [H W]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from render import DEFAULT_CFG, MVS_CFG, ConstantMono
from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, imgs_info_to_torch
from panogrf_amd.renderer import name2network

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
cfg = dict(DEFAULT_CFG, height=H, width=W, depth_sample_num=64, fine_depth_sample_num=64, sample_num=64)
cfg.update(MVS_CFG, depth_height=H // 2, depth_width=W // 2)
torch.manual_seed(0)
net = name2network['neuray_gen'](cfg, d_net=ConstantMono(3.0)).cuda().eval()
g = torch.Generator().manual_seed(1)
imgs = torch.rand(2, 3, H, W, generator=g).numpy()
pose = lambda tx: np.concatenate([np.eye(3), np.array([[tx], [0.0], [0.0]])], 1)
ref = imgs_info_to_torch(build_imgs_info(imgs, np.stack([pose(0.5), pose(-0.5)]), [[0.5, 15.0]] * 2), 'cuda')
src = imgs_info_to_torch(build_imgs_info(imgs[::-1].copy(), np.stack([pose(-0.5), pose(0.5)]), [[0.5, 15.0]] * 2), 'cuda')
que = imgs_info_to_torch(build_render_imgs_info(pose(0.0), (H, W), [0.5, 15.0]), 'cuda')


def frame():
    with torch.no_grad():
        return net({'que_imgs_info': dict(que), 'ref_imgs_info': dict(ref), 'src_imgs_info': dict(src), 'eval': True})


outs = {}
MODES = ((True, True),) if os.environ.get('PGRF_FRAME_MODES') == 'reuse' else ((True, False), (False, False), (True, False), (True, True))
for overlap, reuse in MODES:
    net.cfg['hip_overlap_encoders'] = overlap
    net.cfg['hip_reuse_source_maps'] = reuse
    for _ in range(3):
        out = frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        out = frame()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    outs[(overlap, reuse)] = out['pixel_colors_nr_fine'].clone()
    print(f'renderer(data) {H}x{W}, 64+64 samples, MVS stage {H//2}x{W//2}, encoder overlap {overlap}, source-map reuse {reuse}: {ms:.2f} ms per frame '
          f'({H*W/ms/1e3:.2f} M rays/s end to end)')
if len(MODES) > 1:
    print('max |overlap - no overlap| =', float((outs[(True, False)] - outs[(False, False)]).abs().max()),
          ' max |reuse - no reuse| =', float((outs[(True, True)] - outs[(True, False)]).abs().max()))
