"""Helper of tests/test_grad_fixtures.py (run as a script in a fresh process, because the library reads PGRF_TRAIN_DETERMINISTIC once):
the coarse + fine + merged composite graph of `render_grad_c2fall_v2_dn32` through the HIP backward; saves every weight gradient to argv[1]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import fixtures as FX                      # noqa: E402
from tests import test_grad_fixtures as T             # noqa: E402
from panogrf_amd import engine as E                   # noqa: E402


def main(out_path):
    frame, w, cfg, _ = FX.load('render_frame_v2_c2fall')
    G = T._load('render_grad_c2fall_v2_dn32')
    dev = 'cuda'
    params = {k: v.to(dev).clone().requires_grad_(True) for k, v in w.items()}
    img_feats = frame['img_feats'].to(dev).clone().requires_grad_(True)
    ray_feats = frame['ray_feats'].to(dev).clone().requires_grad_(True)
    coords = G['coords'].to(dev)
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    common = (frame['imgs'].to(dev), img_feats, ray_feats, frame['w2c'].to(dev), frame['ref_depth_range'].to(dev), frame['que_c2w'],
              frame['que_depth_range'], params)
    depth, fd = G['depth.coarse'].to(dev), G['depth.fine'].to(dev)
    oc = E.render_pass_autograd(*common, 'dist_decoder', 'agg_net', use_vis, coords, depth)
    of = E.render_pass_autograd(*common, 'fine_dist_decoder', 'fine_agg_net', use_vis, coords, fd)
    m = E.merge_c2f_autograd(depth, oc['colors_nr'], oc['density_nr'], fd, of['colors_nr'], of['density_nr'])
    out = {'pixel_colors_nr': oc['pixel_colors_nr'], 'render_depth': oc['render_depth'], 'pixel_colors_nr_fine': m['pixel_colors_nr'],
           'render_depth_fine': m['render_depth']}
    up = {k[3:]: v.to(dev) for k, v in G.items() if k.startswith('up.')}
    T._c2f_loss(out, up).backward()
    grads = {k: (torch.zeros_like(p_) if p_.grad is None else p_.grad).cpu() for k, p_ in params.items()
             if k.startswith(('dist_decoder.', 'agg_net.', 'fine_dist_decoder.', 'fine_agg_net.'))}
    grads['img_feats'] = img_feats.grad.cpu()
    grads['ray_feats'] = ray_feats.grad.cpu()
    torch.save(grads, out_path)


if __name__ == '__main__':
    main(sys.argv[1])
