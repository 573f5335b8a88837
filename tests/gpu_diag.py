"""Diagnostic (not a pytest): per-stage max-abs error of the HIP render pass vs golden fixtures.
Run on the GPU box:  python tests/gpu_diag.py [fixture ...]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import fixtures as FX
from panogrf_amd import engine as E

names = sys.argv[1:] or ['render_stages_v2_dn32', 'render_stages_v2_dn64', 'render_stages_v4_dn32',
                         'render_stages_v3_vis', 'render_stages_v2_randinit']
for name in names:
    frame, w, cfg, g = FX.load(name)
    fr = E.Frame(frame['imgs'], frame['img_feats'], frame['ray_feats'], frame['w2c'], frame['ref_depth_range'],
                 frame['que_c2w'], frame['que_depth_range'])
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    print(f'== {name} rfn={fr.rfn} dn={cfg["depth_sample_num"]} use_vis={use_vis}')
    for tag, dd, ag in [('coarse', 'dist_decoder', 'agg_net'), ('fine', 'fine_dist_decoder', 'fine_agg_net')]:
        blob = E.pack_pass_weights(w, dd, ag, use_vis).cuda()
        gold = g[tag]
        out = E.render_pass(fr, blob, use_vis, frame['coords'], gold['depth'], debug=True)
        torch.cuda.synchronize()
        flat = {**out['dbg'], **{k: v for k, v in out.items() if k != 'dbg'}}
        for k in ['px', 'depth_v', 'ddiff', 'ray_feats', 'rgb', 'img_feats', 'vis', 'hit', 'neuray', 'base_x', 'geo',
                  'attn', 'density_nr', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr', 'render_depth']:
            if k in gold and k in flat:
                a, b = flat[k].cpu(), gold[k]
                if k == 'px':
                    W = frame['imgs'].shape[-1]
                    dx = (a[..., 0] - b[..., 0]).abs(); dx = torch.minimum(dx, (W - 1) - dx)
                    err = max(float(dx.max()), float((a[..., 1] - b[..., 1]).abs().max()))
                else:
                    err = float((a - b).abs().max())
                print(f'  {tag:6s} {k:16s} max_abs_err {err:.3e}   ref_max {float(b.abs().max()):.3e}  nan={bool(torch.isnan(a).any())}')
    # samplers
    dn = cfg['depth_sample_num']
    cd = E.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], frame['coords'].shape[0], dn)
    print('  coarse depth err', float((cd.cpu() - g['coarse']['depth']).abs().max()))
    fd = E.sample_fine_depth(g['coarse']['depth'].cuda(), g['coarse']['hit_prob_nr'].cuda(), frame['que_depth_range'], dn)
    print('  fine depth err  ', float((fd.cpu() - g['fine']['depth']).abs().max()))
