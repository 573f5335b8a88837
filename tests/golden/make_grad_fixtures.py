"""Gradient golden vectors from the REFERENCE's own autograd graph (round 4).

The backward of a network pass had been checked against torch.autograd through the CPU restatement only, and the
restatement is pinned to the reference in the FORWARD direction.  Here the reference itself is differentiated: its own modules and
functions in `render_by_depth` order (network/renderer.py:223-304) -- resp. its whole `render_impl` (:567-633, with
`render_c2f_all`: :484-518) -- under autograd, a fixed linear functional of the outputs as the loss, and d loss / d every per-ray
weight and d loss / d `img_feats`, `ray_feats` stored.

  render_grad_v2_dn32.npz, render_grad_v4_dn32.npz   coarse and fine pass of the stage fixtures render_stages_v{2,4}_dn32
                                                      (same inputs, weights and sample depths; upstream gradients stored here)
  render_grad_c2fall_v2_dn32.npz                     render_impl with render_c2f_all on the inputs of render_frame_v2_c2fall

Runs only in the build container (needs /root/reference; see _ref_import.py):   python tests/golden/make_grad_fixtures.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _ref_import as R  # noqa: E402

UP_KEYS = ('pixel_colors_nr', 'render_depth', 'hit_prob_nr', 'colors_nr', 'density_nr')


def upstream(rn, dn, seed):
    """d loss / d output of one pass (the loss is sum_k <out_k, up_k>): the same draws as tests/test_gpu_train_backward.py"""
    g = torch.Generator().manual_seed(seed)
    return {'pixel_colors_nr': torch.randn(rn, 3, generator=g), 'render_depth': 0.1 * torch.randn(rn, generator=g),
            'hit_prob_nr': torch.randn(rn, dn, generator=g), 'colors_nr': 0.05 * torch.randn(rn, dn, 3, generator=g),
            'density_nr': 0.05 * torch.randn(rn, dn, generator=g)}


def load_fixture(name):
    z = np.load(os.path.join(HERE, name + '.npz'))
    cfg = json.loads(bytes(z['cfg']).decode())
    frame = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in.')}
    w = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}
    return z, cfg, frame, w


def build_reference(cfg, w):
    from network.renderer import NeuralRayBaseRenderer
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer(dict(cfg)).eval()
    missing, unexpected = r.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    assert all(not k.startswith(('dist_decoder', 'agg_net', 'fine_dist_decoder', 'fine_agg_net')) for k in missing), missing
    return r


def ref_pass_outputs(r, frame, ref, coords, depth, is_fine):
    """the reference's functions in render_by_depth order; returns the five differentiable outputs of a pass"""
    from network import render_ops as ro
    que = {'c2w': frame['que_c2w'][None], 'w2c': frame['que_w2c'][None], 'coords': coords[None],
           'depth_range': frame['que_depth_range'][None]}
    qd = ro.depth2inv_dists(depth[None], que['depth_range'])
    pts, qdir = ro.depth2points_spherical(que, depth[None], r.spt_utils)
    prj = ro.project_points_dict(ref, pts, r.spt_utils)
    prj = r.predict_proj_ray_prob(prj, ref, qd, is_fine)
    prj = r.get_img_feats(ref, prj)
    hit, colors, pixel, density = r.network_rendering(prj, qdir, is_fine)
    return {'pixel_colors_nr': pixel[0], 'render_depth': torch.sum(hit[0] * depth, -1), 'hit_prob_nr': hit[0],
            'colors_nr': colors[0], 'density_nr': density[0]}


def pass_fixture(src, out_name):
    z, cfg, frame, w = load_fixture(src)
    r = build_reference(cfg, w)
    coords = frame['coords']
    arrs = {'source': np.frombuffer(src.encode(), dtype=np.uint8)}
    for tag, dd, ag, seed, is_fine in (('coarse', 'dist_decoder', 'agg_net', 1, False), ('fine', 'fine_dist_decoder', 'fine_agg_net', 2, True)):
        depth = torch.from_numpy(z[tag + '.depth'])
        rn, dn = depth.shape
        up = upstream(rn, dn, seed)
        r.zero_grad(set_to_none=True)
        img_feats = frame['img_feats'].clone().requires_grad_(True)
        ray_feats = frame['ray_feats'].clone().requires_grad_(True)
        ref = {'imgs': frame['imgs'], 'w2c': frame['w2c'], 'depth_range': frame['ref_depth_range'], 'ray_feats': ray_feats,
               'img_feats': img_feats}
        out = ref_pass_outputs(r, frame, ref, coords, depth, is_fine)
        # the graph is the one the stage fixture recorded (forward equality, bit for bit)
        for k in ('pixel_colors_nr', 'hit_prob_nr', 'density_nr', 'colors_nr'):
            assert torch.equal(out[k].detach(), torch.from_numpy(z[f'{tag}.{k}'])), (src, tag, k)
        loss = sum((out[k] * up[k]).sum() for k in UP_KEYS)
        loss.backward()
        n = 0
        for k, prm in r.named_parameters():
            if k.startswith((dd + '.', ag + '.')):
                g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
                arrs[f'grad.{tag}.{k}'] = g.detach().numpy().copy()
                n += 1
            elif k.startswith(('dist_decoder', 'agg_net', 'fine_dist_decoder', 'fine_agg_net')):
                assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, (tag, k)     # the other pass's networks are untouched
        arrs[f'grad.{tag}.img_feats'] = img_feats.grad.numpy().copy()
        arrs[f'grad.{tag}.ray_feats'] = ray_feats.grad.numpy().copy()
        arrs[f'loss.{tag}'] = np.float64(loss.item())
        for k in UP_KEYS:
            arrs[f'up.{tag}.{k}'] = up[k].numpy()
        print(src, tag, 'loss', float(loss), 'weight tensors', n, 'max |d img_feats|', float(img_feats.grad.abs().max()),
              'max |d ray_feats|', float(ray_feats.grad.abs().max()))
    path = os.path.join(HERE, out_name + '.npz')
    np.savez_compressed(path, **arrs)
    print(out_name, os.path.getsize(path) // 1024, 'KiB')


def c2f_fixture(src, out_name, rn=40):
    """render_impl (coarse pass -> importance sampling on detached weights -> fine pass -> render_c2f_all merge) under autograd."""
    z, cfg, frame, w = load_fixture(src)
    assert cfg.get('render_c2f_all')
    r = build_reference(cfg, w)
    H, W = frame['imgs'].shape[-2:]
    g = torch.Generator().manual_seed(77)
    idx = torch.randperm(H * W, generator=g)[:rn]
    idx[0], idx[1], idx[2], idx[3] = 0, W - 1, (H - 1) * W, H * W - 1
    coords = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    dn = cfg['depth_sample_num']
    up = {'pixel_colors_nr': torch.randn(rn, 3, generator=g), 'pixel_colors_nr_fine': torch.randn(rn, 3, generator=g),
          'render_depth_fine': 0.1 * torch.randn(rn, generator=g), 'render_depth': 0.1 * torch.randn(rn, generator=g)}
    img_feats = frame['img_feats'].clone().requires_grad_(True)
    ray_feats = frame['ray_feats'].clone().requires_grad_(True)
    que = {'c2w': frame['que_c2w'][None], 'w2c': frame['que_w2c'][None], 'coords': coords[None], 'depth_range': frame['que_depth_range'][None]}
    ref = {'imgs': frame['imgs'], 'w2c': frame['w2c'], 'depth_range': frame['ref_depth_range'], 'ray_feats': ray_feats, 'img_feats': img_feats}
    cap = {}
    rbd = r.render_by_depth

    def spy(que_depth, *a, **k):               # the sample depths of the two passes (the fine ones come out of the importance sampler)
        cap['fine' if (a[3] if len(a) > 3 else k.get('is_fine')) else 'coarse'] = que_depth.detach().clone()
        return rbd(que_depth, *a, **k)
    r.render_by_depth = spy
    out = r.render_impl(que, ref, False)       # eval-mode sampler (no random draws), gradients enabled
    r.render_by_depth = rbd
    loss = sum((out[k][0] * up[k]).sum() for k in up)
    loss.backward()
    arrs = {'source': np.frombuffer(src.encode(), dtype=np.uint8), 'coords': coords.numpy(), 'loss': np.float64(loss.item()),
            'depth.coarse': cap['coarse'][0].numpy(), 'depth.fine': cap['fine'][0].numpy()}
    for k in up:
        arrs['up.' + k] = up[k].numpy()
        arrs['out.' + k] = out[k][0].detach().numpy()
    n = 0
    for k, prm in r.named_parameters():
        if k.startswith(('dist_decoder', 'agg_net', 'fine_dist_decoder', 'fine_agg_net')):
            arrs['grad.' + k] = (prm.grad if prm.grad is not None else torch.zeros_like(prm)).detach().numpy().copy()
            n += 1
    arrs['grad.img_feats'] = img_feats.grad.numpy().copy()
    arrs['grad.ray_feats'] = ray_feats.grad.numpy().copy()
    path = os.path.join(HERE, out_name + '.npz')
    np.savez_compressed(path, **arrs)
    print(out_name, os.path.getsize(path) // 1024, 'KiB', 'loss', float(loss), 'weight tensors', n)


def main():
    torch.set_num_threads(1)
    R.setup()
    pass_fixture('render_stages_v2_dn32', 'render_grad_v2_dn32')
    pass_fixture('render_stages_v4_dn32', 'render_grad_v4_dn32')
    c2f_fixture('render_frame_v2_c2fall', 'render_grad_c2fall_v2_dn32')


if __name__ == '__main__':
    main()
