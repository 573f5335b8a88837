"""Randomised shapes/poses against the oracle (script, not collected by pytest): python tests/gpu_fuzz.py [n] [seed]"""
import sys
import torch
sys.path.insert(0, '.')
from oracle import render_oracle as O
from bench import synthetic_case
from panogrf_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for it in range(n):
    rfn = int(torch.randint(1, 5, (1,), generator=g))
    dn = [32, 64][int(torch.randint(0, 2, (1,), generator=g))]
    H = 8 * int(torch.randint(1, 5, (1,), generator=g))
    W = 8 * int(torch.randint(2, 9, (1,), generator=g))
    fine = bool(torch.randint(0, 2, (1,), generator=g))
    use_vis = bool(torch.randint(0, 2, (1,), generator=g))
    rn = int(torch.randint(1, 200, (1,), generator=g))
    case = synthetic_case(H, W, rfn, dn, seed=100 + it, fine=fine, use_vis=use_vis)
    fr = case['frame']
    frame = E.Frame(fr['imgs'], fr['img_feats'], fr['ray_feats'], fr['w2c'], fr['ref_depth_range'], fr['que_c2w'], fr['que_depth_range'])
    rr = E.RayRenderer(case['cfg'], case['weights'])
    coords = torch.stack([torch.randint(0, W, (rn,), generator=g), torch.randint(0, H, (rn,), generator=g)], -1).float()
    out = rr.render_rays(frame, coords)
    ref = O.render_rays(case['weights'], fr, coords, dict(case['cfg'], use_vis=use_vis))
    e = float((out['pixel_colors_nr'].cpu() - ref['pixel_colors_nr']).abs().max())
    ed = float((out['render_depth'].cpu() - ref['render_depth']).abs().max())
    worst = max(worst, e)
    flag = '' if (e < 1e-4 and ed < 5e-3 and not torch.isnan(out['pixel_colors_nr']).any()) else '  <-- CHECK'
    print(f'{it:3d} rfn={rfn} dn={dn} {H}x{W} fine={int(fine)} vis={int(use_vis)} rn={rn:3d}  pixel err {e:.2e}  depth err {ed:.2e}{flag}')
print('worst pixel err', worst)
