"""Round-3 golden vectors, produced by running the REFERENCE itself (build container only; see _ref_import.py):

  python tests/golden/make_round3_fixtures.py

  render_r3_uncert.npz       render_impl with cfg render_uncert=True (network/renderer.py:305-312): 'render_uncert',
                             'render_uncert_fine' next to the usual outputs
  render_r3_uncert_c2f.npz   the same with render_c2f_all=True (renderer.py:484-514: uncertainty over the merged samples)
  render_r3_nodisp.npz       cfg use_disp=False: sample_depth's linear-depth branch (render_ops.py:303-318) and
                             sample_fine_depth with inv_mode off (render_ops.py:423-424), alone and through render_impl
  enc_fixtures.npz           ResUNetLight (network/ops.py:235-455) and DefaultVisEncoder (network/vis_encoder.py:6-32)
                             with seeded weights: the tensor after every stage of the image encoder (forward hooks) and the
                             visibility encoder's output, at two frame sizes

Fixtures hold inputs, seeded per-ray weights (or the (name, shape) list to regenerate them) and expected outputs only.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_import as R  # noqa: E402
from make_fixtures import base_cfg, boost_density, make_frame, per_ray_weights, to_np, smooth_pano  # noqa: E402
from mvs_fixtures import seeded_state  # noqa: E402  (tests/mvs_fixtures.py)


def impl_fixture(name, seed, rfn, H, W, rn, dn, **cfg_kw):
    """render_impl on a seeded ray sample; every output key the reference returns that is a float tensor is kept."""
    from network.renderer import NeuralRayBaseRenderer
    from network import render_ops as ro
    cfg = base_cfg(H, W, dn, **cfg_kw)
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
    with torch.no_grad():
        top = r.render_impl(que, ref, False)
        depth_c, _ = ro.sample_depth(r.cfg, coords[None], dn, False, use_disp=r.cfg['use_disp'])
        depth_f = ro.sample_fine_depth(r.cfg, depth_c, top['hit_prob_nr'], que['depth_range'], cfg['fine_depth_sample_num'], False)
    arrs = {}
    arrs.update(to_np(frame, 'in.'))
    arrs['in.coords'] = coords.numpy()
    arrs.update(to_np(per_ray_weights(r), 'w.'))
    arrs.update(to_np({k: v[0] for k, v in top.items() if v.dtype == torch.float32}, 'out.'))
    arrs['out.coarse_depth'] = depth_c[0].numpy()
    arrs['out.fine_depth_unsorted'] = depth_f[0].numpy()
    arrs['out.keys'] = np.frombuffer('\n'.join(sorted(top.keys())).encode(), np.uint8)
    arrs['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(name, os.path.getsize(path) // 1024, 'KiB', sorted(top.keys()))


def enc_fixture():
    from network.ops import ResUNetLight
    from network.vis_encoder import DefaultVisEncoder
    cfg = base_cfg(32, 64, 32)
    arrs = {}
    for tag, (H, W), seed in (('a', (32, 64), 41), ('b', (48, 80), 43)):
        torch.manual_seed(seed)
        ie = ResUNetLight(dict(cfg), 3, [1, 2, 6, 4], 32, inplanes=16, use_wrap_padding=True, autoencoder=False).eval()
        ve = DefaultVisEncoder(dict(cfg)).eval()
        ishapes = [(k, tuple(v.shape)) for k, v in ie.state_dict().items()]
        vshapes = [(k, tuple(v.shape)) for k, v in ve.state_dict().items()]
        ie.load_state_dict(seeded_state(ishapes, seed + 1))
        ve.load_state_dict(seeded_state(vshapes, seed + 2))
        g = torch.Generator().manual_seed(seed + 3)
        imgs = torch.stack([smooth_pano(g, 3, H, W) for _ in range(2)])
        ray_feats = torch.randn(2, 32, H // 8, W // 8, generator=g)
        stages = {}
        hooks = [m.register_forward_hook(lambda mod, i, o, n=n: stages.__setitem__(n, o.detach().clone()))
                 for n, m in (('bn1', ie.bn1), ('layer1', ie.layer1), ('layer2', ie.layer2), ('layer3', ie.layer3),
                              ('upconv3', ie.upconv3), ('iconv3', ie.iconv3), ('upconv2', ie.upconv2), ('iconv2', ie.iconv2))]
        with torch.no_grad():
            feats = ie(imgs)
            vis = ve(ray_feats, feats)
        for h in hooks:
            h.remove()
        arrs[f'{tag}.imgs'], arrs[f'{tag}.ray_feats'] = imgs.numpy(), ray_feats.numpy()
        arrs[f'{tag}.img_feats'], arrs[f'{tag}.vis_feats'] = feats.numpy(), vis.numpy()
        for n, t in stages.items():
            arrs[f'{tag}.stage.{n}'] = t.numpy()
        enc = lambda shapes: np.frombuffer('\n'.join(f'{k}:{",".join(map(str, s))}' for k, s in shapes).encode(), np.uint8)
        arrs[f'{tag}.ie_keys'], arrs[f'{tag}.ve_keys'] = enc(ishapes), enc(vshapes)
        arrs[f'{tag}.seed'] = np.array([seed])
    path = os.path.join(HERE, 'enc_fixtures.npz')
    np.savez_compressed(path, **arrs)
    print('enc_fixtures', os.path.getsize(path) // 1024, 'KiB', sorted(k for k in arrs if k.startswith('a.'))[:20])


def main():
    torch.set_num_threads(1)
    R.setup()
    only = [a for a in sys.argv[1:] if not a.startswith('-')]
    jobs = {
        'uncert': lambda: impl_fixture('render_r3_uncert', seed=1100, rfn=2, H=32, W=64, rn=40, dn=32, render_uncert=True),
        'uncert_c2f': lambda: impl_fixture('render_r3_uncert_c2f', seed=1200, rfn=2, H=32, W=64, rn=40, dn=32, render_uncert=True,
                                           render_c2f_all=True),
        'nodisp': lambda: impl_fixture('render_r3_nodisp', seed=1300, rfn=2, H=32, W=64, rn=40, dn=32, use_disp=False),
        'enc': enc_fixture,
    }
    for k, fn in jobs.items():
        if not only or k in only:
            fn()


if __name__ == '__main__':
    main()
