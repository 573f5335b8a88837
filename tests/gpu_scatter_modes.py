"""Helper of tests/test_gpu_train_backward.py (run as a script in a fresh process, because the library reads PGRF_TRAIN_SCATTER_RAY once):
feature-map gradients of one network pass on a synthetic frame for the (rn, dn, views) cases given in argv[2:], saved to argv[1]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                          # noqa: E402
from panogrf_amd import engine as E                   # noqa: E402


def main(out_path, cases):
    res = {}
    for spec in cases:
        rn, dn, V = (int(x) for x in spec.split('x'))
        case = bench.synthetic_case(64, 128, V, dn, seed=3)
        f = case['frame']
        fr = E.Frame(f['imgs'], f['img_feats'], f['ray_feats'], f['w2c'], f['ref_depth_range'], f['que_c2w'], f['que_depth_range'])
        g = torch.Generator().manual_seed(7 + rn)
        idx = torch.randperm(64 * 128, generator=g)[:rn]
        coords = torch.stack([(idx % 128).float(), (idx // 128).float()], -1)
        depth = E.sample_coarse_depth(0.5, 15.0, rn, dn, device='cuda')
        up = {'pixel_colors_nr': torch.randn(rn, 3, generator=g), 'render_depth': 0.1 * torch.randn(rn, generator=g),
              'hit_prob_nr': torch.randn(rn, dn, generator=g), 'colors_nr': 0.05 * torch.randn(rn, dn, 3, generator=g),
              'density_nr': 0.05 * torch.randn(rn, dn, generator=g)}
        raw = E.raw_weight_tensors(case['weights'], 'dist_decoder', 'agg_net', False, fr.device)
        _, g_if, g_rf, _ = E.render_pass_backward(fr, raw, False, coords, depth, up)
        res[spec + '.img_feats'] = g_if.cpu()
        res[spec + '.ray_feats'] = g_rf.cpu()
    torch.save(res, out_path)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
