"""Coarse + fine frame with and without render_c2f_all (renderer.py:484-518: one composite over the union of both passes'
samples) at 512x1024: what the merge kernel costs.  python tools/time_c2f.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from panogrf_amd import engine as E

H, W = 512, 1024
dev = torch.device('cuda:0')
for c2f in (False, True):
    case = B.synthetic_case(H, W, 2, 64, fine=True)
    case['cfg']['render_c2f_all'] = c2f
    f = case['frame']
    fr = E.Frame(f['imgs'], f['img_feats'], f['ray_feats'], f['w2c'], f['ref_depth_range'], f['que_c2w'], f['que_depth_range'], device=dev)
    rr = E.RayRenderer(case['cfg'], case['weights'], device=dev)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    coords = torch.stack([xs, ys], -1).reshape(-1, 2).to(dev)
    for _ in range(2):
        rr.render_rays(fr, coords)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(3):
        rr.render_rays(fr, coords)
    b.record()
    torch.cuda.synchronize()
    print(f'render_c2f_all={c2f}: {a.elapsed_time(b) / 3:.2f} ms per frame')
