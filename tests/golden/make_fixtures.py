"""Generate golden vectors for the per-ray renderer by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; see _ref_import.py).  Writes small
.npz fixtures next to this file; those fixtures (data only: inputs, seeded weights, expected
outputs) are what travels to the GPU box.  Re-run:  python tests/golden/make_fixtures.py

All tensors are generated with torch.set_num_threads(1) (the reference
differs from itself by ~4e-7 coarse / ~3e-5 fine across thread counts).
"""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402


def rot_xyz(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


def smooth_pano(g, C, H, W, lo=0.0, hi=1.0):
    """Band-limited random texture on the sphere grid (sum of low-frequency sinusoids)."""
    th = torch.linspace(0, 2 * math.pi, W).view(1, 1, W)
    ph = torch.linspace(0, math.pi, H).view(1, H, 1)
    out = torch.zeros(C, H, W)
    for _ in range(8):
        a = torch.randint(1, 5, (C, 1, 1), generator=g).float()
        b = torch.randint(1, 4, (C, 1, 1), generator=g).float()
        p = torch.rand(C, 1, 1, generator=g) * 2 * math.pi
        q = torch.rand(C, 1, 1, generator=g) * 2 * math.pi
        out = out + torch.sin(a * th + p) * torch.cos(b * ph + q)
    out = (out - out.amin()) / (out.amax() - out.amin())
    return out * (hi - lo) + lo


def make_frame(seed, rfn, H, W, rot_scale=0.15):
    g = torch.Generator().manual_seed(seed)
    imgs = torch.stack([smooth_pano(g, 3, H, W) for _ in range(rfn)])
    img_feats = torch.randn(rfn, 32, H // 4, W // 4, generator=g)
    ray_feats = torch.randn(rfn, 32, H // 8, W // 8, generator=g)
    w2c = []
    for v in range(rfn):
        ang = (torch.rand(3, generator=g) - 0.5) * 2 * rot_scale
        Rm = rot_xyz(*ang.tolist())
        ang_t = 2 * math.pi * v / rfn
        t = np.array([0.5 * math.cos(ang_t), 0.05 * (v - 1), 0.5 * math.sin(ang_t)], np.float32)
        w2c.append(np.concatenate([Rm, t[:, None]], 1))
    w2c = torch.from_numpy(np.stack(w2c))
    qa = (torch.rand(3, generator=g) - 0.5) * 2 * rot_scale
    Rq = rot_xyz(*qa.tolist())
    tq = np.array([0.03, -0.02, 0.05], np.float32)
    que_w2c = np.concatenate([Rq, tq[:, None]], 1)
    que_c2w = np.concatenate([Rq.T, (-Rq.T @ tq)[:, None]], 1).astype(np.float32)
    return dict(imgs=imgs, img_feats=img_feats, ray_feats=ray_feats, w2c=w2c,
                ref_depth_range=torch.tensor([[0.5, 15.0]] * rfn), que_w2c=torch.from_numpy(que_w2c),
                que_c2w=torch.from_numpy(que_c2w), que_depth_range=torch.tensor([0.5, 15.0]))


def base_cfg(H, W, dn, **kw):
    cfg = dict(use_wrap_padding=True, dataset_name='m3d', batch_size=1, height=H, width=W,
               handle_distort=False, handle_distort_all=False, handle_distort_input_all=False, debug=False,
               use_disp=True, min_depth=0.5, max_depth=15.0, use_polar_weighted_loss=False,
               render_uncert=False, render_depth=True, use_hierarchical_sampling=True,
               depth_sample_num=dn, fine_depth_sample_num=dn, sample_num=dn, ray_batch_num=2048,
               dist_decoder_cfg={'use_vis': False}, fine_dist_decoder_cfg={'use_vis': False},
               agg_net_cfg={}, fine_agg_net_cfg={})
    cfg.update(kw)
    return cfg


def boost_density(renderer, seed):
    """Seeded 'density boost' (SURVEY.md section 7): default init gives sigma ~ 3e-4 and a black image."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for pre in ['agg_net', 'fine_agg_net']:
            if not hasattr(renderer, pre):
                continue
            impl = getattr(renderer, pre).agg_impl
            impl.out_geometry_fc[2].bias.add_(0.6)
            impl.out_geometry_fc[2].weight.mul_(3.0)
            impl.out_geometry_fc[0].weight.mul_(2.0)
            for m in [impl.geometry_fc[0], impl.geometry_fc[2], impl.rgb_fc[4], impl.vis_fc2[2], impl.neuray_fc[2]]:
                m.bias.add_(torch.randn(m.bias.shape, generator=g) * 0.3)
            for m in [impl.rgb_fc[0], impl.rgb_fc[2], impl.rgb_fc[4]]:
                m.weight.mul_(2.0)
        for pre in ['dist_decoder', 'fine_dist_decoder']:
            if not hasattr(renderer, pre):
                continue
            dd = getattr(renderer, pre)
            for name in ['mean_decoder', 'var_decoder', 'aw_decoder', 'vis_decoder']:
                if hasattr(dd, name):
                    m = getattr(dd, name)[4]
                    m.bias.add_(torch.randn(m.bias.shape, generator=g) * 0.5)
                    m.weight.mul_(3.0)
            # spread variances so the logistic CDFs are not all flat
            dd.var_decoder[4].bias.add_(2.0)


def per_ray_weights(renderer):
    sd = renderer.state_dict()
    keep = ('dist_decoder', 'agg_net', 'fine_dist_decoder', 'fine_agg_net')
    return {k: v.detach().clone() for k, v in sd.items() if k.startswith(keep)}


def ref_pass(renderer, frame, coords, depth, is_fine):
    """One reference pass, calling the reference's own functions in render_by_depth order and
    capturing the intermediate tensors (renderer.py:223-304)."""
    from network import render_ops as ro
    from network.aggregate_net import get_dir_diff
    que = {'c2w': frame['que_c2w'][None], 'w2c': frame['que_w2c'][None], 'coords': coords[None],
           'depth_range': frame['que_depth_range'][None]}
    ref = {'imgs': frame['imgs'], 'w2c': frame['w2c'], 'depth_range': frame['ref_depth_range'],
           'ray_feats': frame['ray_feats'], 'img_feats': frame['img_feats']}
    cap = {}
    impl = (renderer.fine_agg_net if is_fine else renderer.agg_net)
    hooks = [
        impl.prob_embed.register_forward_hook(lambda m, i, o: cap.__setitem__('neuray', o.detach().clone())),
        impl.agg_impl.base_fc.register_forward_hook(lambda m, i, o: cap.__setitem__('base_x', o.detach().clone())),
        impl.agg_impl.geometry_fc.register_forward_hook(lambda m, i, o: cap.__setitem__('geo', o.detach().clone())),
        impl.agg_impl.ray_attention.register_forward_hook(lambda m, i, o: cap.__setitem__('attn', o[0].detach().clone())),
    ]
    qd = ro.depth2inv_dists(depth[None], que['depth_range'])
    pts, qdir = ro.depth2points_spherical(que, depth[None], renderer.spt_utils)
    prj = ro.project_points_dict(ref, pts, renderer.spt_utils)
    prj = renderer.predict_proj_ray_prob(prj, ref, qd, is_fine)
    prj = renderer.get_img_feats(ref, prj)
    hit, colors, pixel, density = renderer.network_rendering(prj, qdir, is_fine)
    for h in hooks:
        h.remove()
    rfn = frame['imgs'].shape[0]
    rn, dn = depth.shape
    dd = get_dir_diff(prj['dir'], qdir)                     # [rn,dn,rfn,4]
    out = dict(
        que_dists=qd[0], pts=pts[0], que_dir=qdir[0, :, 0],
        px=prj['pts'][:, 0], depth_v=prj['depth'][:, 0, ..., 0], ddiff=dd.permute(2, 0, 1, 3),
        ray_feats=prj['ray_feats'][:, 0], rgb=prj['rgb'][:, 0], img_feats=prj['img_feats'][:, 0],
        vis=prj['vis'][:, 0, ..., 0], hit=prj['hit_prob'][:, 0, ..., 0],
        neuray=cap['neuray'][:, 0].permute(1, 2, 0, 3), base_x=cap['base_x'], geo=cap['geo'], attn=cap['attn'],
        hit_prob_nr=hit[0], colors_nr=colors[0], pixel_colors_nr=pixel[0], density_nr=density[0],
        render_depth=torch.sum(hit[0] * depth, -1))
    return out


def to_np(d, prefix=''):
    return {prefix + k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def stage_fixture(name, seed, rfn, H, W, rn, dn, boost=True, use_vis=False, keep_big=True):
    from network.renderer import NeuralRayBaseRenderer
    from network import render_ops as ro
    cfg = base_cfg(H, W, dn, dist_decoder_cfg={'use_vis': use_vis}, fine_dist_decoder_cfg={'use_vis': use_vis})
    torch.manual_seed(seed)
    r = NeuralRayBaseRenderer(dict(cfg)).eval()
    if boost:
        boost_density(r, seed + 1)
    frame = make_frame(seed + 2, rfn, H, W)
    g = torch.Generator().manual_seed(seed + 3)
    idx = torch.randperm(H * W, generator=g)[:rn]
    # include poles and the seam column explicitly
    idx[0], idx[1], idx[2], idx[3] = 0, W - 1, (H - 1) * W, H * W - 1
    coords = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    with torch.no_grad():
        depth, _ = ro.sample_depth(r.cfg, coords[None], dn, False, use_disp=True)
        depth = depth[0]
        c = ref_pass(r, frame, coords, depth, False)
        fd = ro.sample_fine_depth(r.cfg, depth[None], c['hit_prob_nr'][None], frame['que_depth_range'][None], dn, False)
        fd = torch.sort(fd, -1)[0][0]
        f = ref_pass(r, frame, coords, fd, True)
        # cross-check against the reference's own top-level per-batch entry
        que = {'c2w': frame['que_c2w'][None], 'w2c': frame['que_w2c'][None], 'coords': coords[None],
               'depth_range': frame['que_depth_range'][None]}
        ref = {'imgs': frame['imgs'], 'w2c': frame['w2c'], 'depth_range': frame['ref_depth_range'],
               'ray_feats': frame['ray_feats'], 'img_feats': frame['img_feats']}
        top = r.render_impl(que, ref, False)
        assert torch.equal(top['pixel_colors_nr'][0], c['pixel_colors_nr'])
        assert torch.equal(top['pixel_colors_nr_fine'][0], f['pixel_colors_nr'])
        assert torch.equal(top['render_depth_fine'][0], f['render_depth'])
    big = ('ray_feats', 'img_feats', 'neuray', 'base_x', 'pts')
    if not keep_big:
        c = {k: v for k, v in c.items() if k not in big}
    f = {k: v for k, v in f.items() if k not in big}
    arrs = {}
    arrs.update(to_np(frame, 'in.'))
    arrs['in.coords'] = coords.numpy()
    arrs.update(to_np(per_ray_weights(r), 'w.'))
    arrs['coarse.depth'] = depth.numpy()
    arrs.update(to_np(c, 'coarse.'))
    arrs['fine.depth'] = fd.numpy()
    arrs.update(to_np(f, 'fine.'))
    arrs['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(name, os.path.getsize(path) // 1024, 'KiB',
          'mean fine rgb', float(f['pixel_colors_nr'].mean()), 'sigma max', float(c['density_nr'].max()))


def frame_fixture(name, seed, rfn, H, W, dn, boost=True, c2f_all=False):
    """Whole small panorama through the reference's ray-batch loop body (render_impl per batch)."""
    from network.renderer import NeuralRayBaseRenderer
    cfg = base_cfg(H, W, dn, ray_batch_num=512, render_c2f_all=c2f_all)
    torch.manual_seed(seed)
    r = NeuralRayBaseRenderer(dict(cfg)).eval()
    if boost:
        boost_density(r, seed + 1)
    frame = make_frame(seed + 2, rfn, H, W)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    coords = torch.stack([xs, ys], -1).reshape(-1, 2)
    outs = {}
    with torch.no_grad():
        for s in range(0, H * W, 512):
            que = {'c2w': frame['que_c2w'][None], 'w2c': frame['que_w2c'][None], 'coords': coords[None, s:s + 512],
                   'depth_range': frame['que_depth_range'][None]}
            ref = {'imgs': frame['imgs'], 'w2c': frame['w2c'], 'depth_range': frame['ref_depth_range'],
                   'ray_feats': frame['ray_feats'], 'img_feats': frame['img_feats']}
            o = r.render_impl(que, ref, False)
            for k in ['pixel_colors_nr', 'render_depth', 'pixel_colors_nr_fine', 'render_depth_fine',
                      'density_nr', 'density_nr_fine']:
                outs.setdefault(k, []).append(o[k][0])
    outs = {k: torch.cat(v, 0) for k, v in outs.items()}
    arrs = {}
    arrs.update(to_np(frame, 'in.'))
    arrs.update(to_np(per_ray_weights(r), 'w.'))
    arrs.update(to_np(outs, 'out.'))
    arrs['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(name, os.path.getsize(path) // 1024, 'KiB', 'mean fine rgb', float(outs['pixel_colors_nr_fine'].mean()))


def full_render_fixture(name, seed, rfn, H, W, dn):
    """NeuralRayBaseRenderer.render end to end (2-D encoders + ray loop) with fully seeded weights."""
    sys.path.insert(0, os.path.dirname(HERE))
    from fixtures import seeded_renderer_state
    from network.renderer import NeuralRayBaseRenderer
    cfg = base_cfg(H, W, dn, ray_batch_num=700)
    r = NeuralRayBaseRenderer(dict(cfg)).eval()
    sd = r.state_dict()
    shapes = [(k, tuple(v.shape)) for k, v in sd.items()]
    r.load_state_dict(seeded_renderer_state(shapes, seed))
    frame = make_frame(seed + 2, rfn, H, W)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    coords = torch.stack([xs, ys], -1).reshape(1, -1, 2)
    que = {'c2w': frame['que_c2w'][None], 'w2c': frame['que_w2c'][None], 'coords': coords,
           'depth_range': frame['que_depth_range'][None]}
    ref = {'imgs': frame['imgs'], 'w2c': frame['w2c'], 'depth_range': frame['ref_depth_range'],
           'ray_feats': frame['ray_feats']}
    with torch.no_grad():
        out = r.render(dict(que), dict(ref), False)
    arrs = {}
    arrs.update(to_np({k: v for k, v in frame.items() if k != 'img_feats'}, 'in.'))
    keep = ['pixel_colors_nr', 'pixel_colors_nr_fine', 'render_depth', 'render_depth_fine', 'density_nr_fine', 'ray_mask_fine']
    arrs.update(to_np({k: out[k] for k in keep}, 'out.'))
    arrs['out.keys'] = np.frombuffer('\n'.join(sorted(out.keys())).encode(), np.uint8)
    arrs['state.keys'] = np.frombuffer('\n'.join(f'{k}:{",".join(map(str, s))}' for k, s in shapes).encode(), np.uint8)
    arrs['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(name, os.path.getsize(path) // 1024, 'KiB', 'mean fine rgb', float(out['pixel_colors_nr_fine'].mean()),
          sorted(out.keys()))


def train_fixture(name, seed, rfn, H, W, rn, dn):
    """Training-mode forward (is_train=True) of the reference's render_impl: deterministic coarse depths
    (renderer.py:607 passes random_sample=False) and random importance samples (renderer.py:437-438).  The uniform
    draws are reproduced by re-seeding (render_ops.py:449 draws [1,rn,fdn] on the CPU generator; nothing else consumes
    it) and stored with the outputs; the fixture also holds sample_depth(random_sample=True) for the jitter path."""
    from network.renderer import NeuralRayBaseRenderer
    cfg = base_cfg(H, W, dn)
    torch.manual_seed(seed)
    r = NeuralRayBaseRenderer(dict(cfg)).eval()
    boost_density(r, seed + 1)
    frame = make_frame(seed + 2, rfn, H, W)
    g = torch.Generator().manual_seed(seed + 3)
    idx = torch.randperm(H * W, generator=g)[:rn]
    coords = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    que = {'c2w': frame['que_c2w'][None], 'w2c': frame['que_w2c'][None], 'coords': coords[None],
           'depth_range': frame['que_depth_range'][None]}
    ref = {'imgs': frame['imgs'], 'w2c': frame['w2c'], 'depth_range': frame['ref_depth_range'],
           'ray_feats': frame['ray_feats'], 'img_feats': frame['img_feats']}
    from network import render_ops as ro
    torch.manual_seed(seed + 4)
    u = torch.rand([1, rn, dn])[0]
    torch.manual_seed(seed + 4)
    with torch.no_grad():
        top = r.render_impl(que, ref, True)
    torch.manual_seed(seed + 5)
    jitter = torch.rand(1, rn, dn - 2, dtype=torch.float32)[0]
    torch.manual_seed(seed + 5)
    jit_depth = ro.sample_depth(r.cfg, coords[None], dn, True, use_disp=True)[0][0]
    keep = ['pixel_colors_nr', 'pixel_colors_nr_fine', 'render_depth', 'render_depth_fine', 'density_nr', 'density_nr_fine',
            'hit_prob_nr', 'hit_prob_nr_fine']
    arrs = {}
    arrs.update(to_np(frame, 'in.'))
    arrs['in.coords'] = coords.numpy()
    arrs['in.jitter'] = jitter.numpy()
    arrs['in.u'] = u.numpy()
    arrs['out.jittered_depth'] = jit_depth.numpy()
    arrs.update(to_np(per_ray_weights(r), 'w.'))
    arrs.update(to_np({k: top[k][0] for k in keep}, 'out.'))
    arrs['out.keys'] = np.frombuffer('\n'.join(sorted(top.keys())).encode(), np.uint8)
    arrs['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(name, os.path.getsize(path) // 1024, 'KiB', sorted(top.keys()))


def main():
    torch.set_num_threads(1)
    R.setup()
    if '--train-only' in sys.argv:  # added in round 2: the other fixtures are not regenerated
        train_fixture('render_train_v2_dn32', seed=1000, rfn=2, H=32, W=64, rn=48, dn=32)
        return
    if '--full-only' in sys.argv:
        full_render_fixture('render_full_v2_dn32', seed=800, rfn=2, H=32, W=64, dn=32)
        return
    if '--v1-only' in sys.argv:     # added in round 2: the other fixtures are not regenerated
        stage_fixture('render_stages_v1_dn32', seed=900, rfn=1, H=32, W=64, rn=12, dn=32, keep_big=False)
        return
    stage_fixture('render_stages_v2_dn32', seed=100, rfn=2, H=32, W=64, rn=24, dn=32)
    stage_fixture('render_stages_v2_dn64', seed=200, rfn=2, H=32, W=64, rn=12, dn=64, keep_big=False)
    stage_fixture('render_stages_v4_dn32', seed=300, rfn=4, H=32, W=64, rn=12, dn=32, keep_big=False)
    stage_fixture('render_stages_v3_vis', seed=400, rfn=3, H=32, W=64, rn=12, dn=32, use_vis=True, keep_big=False)
    stage_fixture('render_stages_v2_randinit', seed=500, rfn=2, H=32, W=64, rn=12, dn=32, boost=False, keep_big=False)
    frame_fixture('render_frame_v2_dn32', seed=600, rfn=2, H=32, W=64, dn=32)
    frame_fixture('render_frame_v2_c2fall', seed=700, rfn=2, H=16, W=32, dn=32, c2f_all=True)
    full_render_fixture('render_full_v2_dn32', seed=800, rfn=2, H=32, W=64, dn=32)
    # single source view: ray attention is fully masked there (ibrnet.py:358-360) -> uniform weights
    stage_fixture('render_stages_v1_dn32', seed=900, rfn=1, H=32, W=64, rn=12, dn=32, keep_big=False)
    train_fixture('render_train_v2_dn32', seed=1000, rfn=2, H=32, W=64, rn=48, dn=32)


if __name__ == '__main__':
    main()
