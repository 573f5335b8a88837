"""Determinism soak: N whole `renderer(data)` frames at the shipped configuration, every frame's outputs compared bit for bit
with the first one's (a race between the side streams / graph replays / kernels would show up as a differing frame), then N
frames of the multi-view (4 source views) path.  python tools/soak.py [frames]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from render import DEFAULT_CFG, MVS_CFG, ConstantMono
from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, imgs_info_to_torch
from panogrf_amd.renderer import name2network

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
H, W = 512, 1024
pose = lambda tx: np.concatenate([np.eye(3), np.array([[tx], [0.0], [0.0]])], 1)
for V, seq_len in ((2, 3), (4, 5)):
    cfg = dict(DEFAULT_CFG, height=H, width=W, depth_sample_num=64, fine_depth_sample_num=64, sample_num=64)
    cfg.update(MVS_CFG, depth_height=H // 2, depth_width=W // 2, seq_len=seq_len)
    torch.manual_seed(0)
    net = name2network['neuray_gen'](cfg, d_net=ConstantMono(3.0)).cuda().eval()
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(V, 3, H, W, generator=g).numpy()
    w2c = np.stack([pose(0.5 - v / (V - 1)) for v in range(V)])
    ref = imgs_info_to_torch(build_imgs_info(imgs, w2c, [[0.5, 15.0]] * V), 'cuda')
    src = imgs_info_to_torch(build_imgs_info(imgs[::-1].copy(), w2c[::-1].copy(), [[0.5, 15.0]] * V), 'cuda')
    if V > 2:
        ref['nn_ids'] = np.array([[(i + 1) % V, (i + 2) % V, (i + 3) % V] for i in range(V)])
    que = imgs_info_to_torch(build_render_imgs_info(pose(0.0), (H, W), [0.5, 15.0]), 'cuda')
    first, bad = None, 0
    n = N if V == 2 else max(N // 5, 5)
    with torch.no_grad():
        for i in range(n):
            net.cfg['hip_reuse_source_maps'] = bool(i % 3 == 2)          # mix of recomputed and reused frames
            out = net({'que_imgs_info': dict(que), 'ref_imgs_info': dict(ref), 'src_imgs_info': dict(src), 'eval': True})
            cur = {k: out[k] for k in ('pixel_colors_nr', 'pixel_colors_nr_fine', 'render_depth_fine')}
            if first is None:
                first = {k: v.clone() for k, v in cur.items()}
                assert all(torch.isfinite(v).all() for v in first.values())
            elif not all(torch.equal(first[k], cur[k]) for k in first):
                bad += 1
    torch.cuda.synchronize()
    print(f'{V} source views: {n} frames, {bad} differ from the first')
    assert bad == 0
    del net
    torch.cuda.empty_cache()
print('soak ok')
