"""The drop-in nn.Module (panogrf_amd.renderer) against a full NeuralRayBaseRenderer.render() run of the
reference (2-D encoders + ray loop), weights rebuilt from a seed (tests/fixtures.seeded_renderer_state)."""
import numpy as np
import pytest
import torch

from tests import fixtures as FX


def _load():
    frame, _, cfg, g = FX.load('render_full_v2_dn32')
    z = np.load(FX.GOLDEN + '/render_full_v2_dn32.npz')
    shapes = []
    for ln in bytes(z['state.keys']).decode().split('\n'):
        k, s = ln.split(':')
        shapes.append((k, tuple(int(t) for t in s.split(',')) if s else ()))
    return frame, cfg, g['out'], shapes, bytes(z['out.keys']).decode().split('\n')


def test_state_dict_keys_match_reference():
    """CPU: a reference checkpoint's network_state_dict must load with strict=True."""
    from panogrf_amd.renderer import name2network
    frame, cfg, gold, shapes, out_keys = _load()
    net = name2network['neuray_base'](cfg)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == dict(shapes)
    net.load_state_dict(FX.seeded_renderer_state(shapes, 800), strict=True)
    assert 'neuray_gen' in name2network


@pytest.mark.gpu
def test_render_end_to_end_vs_reference():
    from panogrf_amd.renderer import name2network
    frame, cfg, gold, shapes, out_keys = _load()
    net = name2network['neuray_gen'](cfg)
    net.load_state_dict(FX.seeded_renderer_state(shapes, 800))
    net = net.cuda().eval()
    H, W = frame['imgs'].shape[-2:]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    coords = torch.stack([xs, ys], -1).reshape(1, -1, 2)
    que = {'c2w': frame['que_c2w'][None].cuda(), 'w2c': frame['que_w2c'][None].cuda(), 'coords': coords.cuda(),
           'depth_range': frame['que_depth_range'][None].cuda()}
    ref = {'imgs': frame['imgs'].cuda(), 'w2c': frame['w2c'].cuda(), 'depth_range': frame['ref_depth_range'].cuda(),
           'ray_feats': frame['ray_feats'].cuda()}
    with torch.no_grad():
        out = net({'que_imgs_info': que, 'ref_imgs_info': ref, 'eval': True})
    assert sorted(out.keys()) == out_keys                       # same dict keys as the reference returns in eval
    for k in ['pixel_colors_nr', 'render_depth']:
        assert out[k].shape == gold[k].shape
        assert FX.max_abs(out[k].cpu(), gold[k]) < 5e-4 * max(1.0, float(gold[k].abs().max())), k
    assert FX.psnr(out['pixel_colors_nr'].cpu(), gold['pixel_colors_nr']) > 80
    assert FX.psnr(out['pixel_colors_nr_fine'].cpu(), gold['pixel_colors_nr_fine']) > 60
    assert torch.equal(out['ray_mask_fine'].cpu(), gold['ray_mask_fine'])
    # training mode with autograd on runs the per-ray passes as autograd Functions (tests/test_gpu_train_backward.py) ...
    # ... and under no_grad it returns the reference's training outputs: hit_prob_* kept (renderer.py:675), random
    # importance samples -> a different fine image, the same deterministic coarse image
    ref2 = {'imgs': frame['imgs'].cuda(), 'w2c': frame['w2c'].cuda(), 'depth_range': frame['ref_depth_range'].cuda(),
            'ray_feats': frame['ray_feats'].cuda()}
    with torch.no_grad():
        tr = net({'que_imgs_info': que, 'ref_imgs_info': ref2})
    assert sorted(tr.keys()) == sorted(out_keys + ['hit_prob_nr', 'hit_prob_nr_fine'])
    assert FX.max_abs(tr['pixel_colors_nr'], out['pixel_colors_nr']) < 1e-6
    assert FX.max_abs(tr['pixel_colors_nr_fine'], out['pixel_colors_nr_fine']) > 1e-5


def test_imgs_info_builders_shapes():
    import numpy as np
    from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, build_render_cube_imgs_info
    cq = build_render_cube_imgs_info(np.eye(4)[:3], np.eye(3), (3, 5), [0.5, 15.0])
    assert cq['poses'].shape == (1, 3, 4) and cq['Ks'].shape == (1, 3, 3) and cq['coords'].shape == (1, 15, 2) and cq['shape'] == (3, 5)
    pose = np.concatenate([np.eye(3), np.array([[0.1], [0.2], [0.3]])], 1)
    q = build_render_imgs_info(pose, (4, 6), [0.5, 15.0])
    assert q['coords'].shape == (1, 24, 2) and q['coords'][0, 1].tolist() == [1.0, 0.0] and q['coords'][0, 6].tolist() == [0.0, 1.0]
    assert np.allclose(q['c2w'][0, :, 3], [-0.1, -0.2, -0.3]) and q['depth_range'].shape == (1, 2)
    r = build_imgs_info(np.zeros((2, 3, 4, 6), np.float32), np.stack([pose, pose]), [[0.5, 15.0]] * 2)
    assert set(r) >= {'imgs', 'rots', 'trans', 'c2w', 'w2c', 'depth_range', 'masks', 'polar_weights'}
    assert r['polar_weights'].shape == (2, 1, 4, 6)


@pytest.mark.gpu
def test_render_script_runs(tmp_path):
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'render.py'), '--height', '64', '--width', '128', '--samples', '32',
                        '--num-poses', '2', '--out', str(tmp_path)], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-1500:]
    assert os.path.exists(tmp_path / '1-nr_fine.ppm') and os.path.exists(tmp_path / '1-depth.npy')
    # the whole flow incl. the MVS stage (init net) in front of the renderer
    r = subprocess.run([sys.executable, os.path.join(root, 'render.py'), '--height', '128', '--width', '256', '--samples', '32',
                        '--num-poses', '1', '--mvs', '--out', str(tmp_path / 'mvs')], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-1500:]
    assert os.path.exists(tmp_path / 'mvs' / '0-nr_fine.ppm')


@pytest.mark.gpu
def test_render_pinhole_query_through_the_module():
    """renderer.render(..., is_perspec=True) (render_cubes.py): same numbers as the oracle's pinhole rays through the
    same per-ray weights and the module's own encoders."""
    from oracle import render_oracle as O
    from panogrf_amd.renderer import name2network
    frame, cfg, gold, shapes, out_keys = _load()
    net = name2network['neuray_base'](cfg)
    sd = FX.seeded_renderer_state(shapes, 800)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    Hp = Wp = 10
    K = torch.tensor([[Wp / 2.0, 0.0, Wp / 2.0], [0.0, Hp / 2.0, Hp / 2.0], [0.0, 0.0, 1.0]])
    ys, xs = torch.meshgrid(torch.arange(Hp, dtype=torch.float32), torch.arange(Wp, dtype=torch.float32), indexing='ij')
    coords = torch.stack([xs, ys], -1).reshape(1, -1, 2)
    que = {'poses': frame['que_w2c'][None].cuda(), 'Ks': K[None].cuda(), 'coords': coords.cuda(),
           'depth_range': frame['que_depth_range'][None].cuda()}
    ref = {'imgs': frame['imgs'].cuda(), 'w2c': frame['w2c'].cuda(), 'depth_range': frame['ref_depth_range'].cuda(),
           'ray_feats': frame['ray_feats'].cuda()}
    with torch.no_grad():
        out = net.render(que, ref, False, is_perspec=True)
    # oracle on the encoder outputs the module left in `ref` (render() stores them like the reference does)
    fr = dict(imgs=frame['imgs'], img_feats=ref['img_feats'].cpu(), ray_feats=ref['ray_feats'].cpu(), w2c=frame['w2c'],
              ref_depth_range=frame['ref_depth_range'], que_depth_range=frame['que_depth_range'],
              que_pose=frame['que_w2c'], que_K=K)
    dn = cfg['depth_sample_num']
    depth = O.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], coords.shape[1], dn, cfg.get('use_disp', True))
    w = {k: v for k, v in sd.items() if k.startswith(('dist_decoder', 'agg_net'))}
    want = O.render_pass(w, 'dist_decoder', 'agg_net', fr, coords[0], depth, dict(use_vis=cfg['dist_decoder_cfg']['use_vis']))
    assert FX.max_abs(out['pixel_colors_nr'][0].cpu(), want['pixel_colors_nr']) < 5e-5
    assert FX.max_abs(out['render_depth'][0].cpu(), want['render_depth']) < 2e-4 * 15.0


@pytest.mark.gpu
def test_encoder_graph_replay_equals_eager():
    """The per-frame encoders run as a replayed hipGraph: same numbers as eager launches (to MIOpen's own run-to-run
    round-off), also for a second frame with different pixels, and re-captured for a new resolution."""
    from panogrf_amd.renderer import name2network
    frame, cfg, gold, shapes, out_keys = _load()
    net = name2network['neuray_base'](cfg)
    net.load_state_dict(FX.seeded_renderer_state(shapes, 800))
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for scale in (1, 2):
            for rep in range(2):
                imgs = torch.rand(2, 3, 32 * scale, 64 * scale, generator=g).cuda()
                rf = torch.randn(2, 32, 4 * scale, 8 * scale, generator=g).cuda()
                a_f, a_r = net._encode(imgs, rf)
                e_f = net.image_encoder(imgs)
                e_r = net.vis_encoder(rf, e_f)
                e_f2 = net.image_encoder(imgs)
                noise = max(float((e_f - e_f2).abs().max()), 1e-6)        # eager vs eager
                assert float((a_f - e_f).abs().max()) <= 4 * noise + 1e-5 * float(e_f.abs().max())
                assert float((a_r - e_r).abs().max()) <= 1e-4 * max(1.0, float(e_r.abs().max()))
    assert len(net._graphs) == 2
    # a weight update after the capture (load_state_dict / optimiser step) must reach the replayed path: the HIP encoders
    # hold packed copies, so the capture is retaken
    with torch.no_grad():
        before, _ = net._encode(imgs, rf)
        for p in net.image_encoder.parameters():
            p.mul_(1.25)
        a_f, a_r = net._encode(imgs, rf)
        e_f = net.image_encoder(imgs)
        e_r = net.vis_encoder(rf, e_f)
    assert float((before - e_f).abs().max()) > 1e-3 * float(e_f.abs().max())
    assert float((a_f - e_f).abs().max()) <= 1e-4 * float(e_f.abs().max())
    assert float((a_r - e_r).abs().max()) <= 1e-4 * max(1.0, float(e_r.abs().max()))
    assert len(net._graphs) == 2


def test_pixel_colors_gt_is_bilinear_with_zero_padding():
    """CPU: ground-truth lookup = interpolate_feats(..., align_corners=True), zeros padding (renderer.py:277-288,
    ops.py:32-52), for 'imgs' and for 'cube_imgs' queries; integer coords reproduce the pixel, fractional ones blend."""
    from panogrf_amd.renderer import _interpolate_feats
    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 3, 5, 7, generator=g)
    pts = torch.tensor([[[0.0, 0.0], [6.0, 4.0], [2.0, 3.0], [2.5, 3.0], [1.25, 0.5], [-0.5, 0.0], [6.5, 4.0]]])
    out = _interpolate_feats(img, pts)
    assert out.shape == (1, 7, 3)
    assert torch.allclose(out[0, 0], img[0, :, 0, 0], atol=1e-6) and torch.allclose(out[0, 1], img[0, :, 4, 6], atol=1e-6)
    assert torch.allclose(out[0, 2], img[0, :, 3, 2], atol=1e-6)
    assert torch.allclose(out[0, 3], 0.5 * (img[0, :, 3, 2] + img[0, :, 3, 3]), atol=1e-6)
    want = 0.5 * (0.75 * img[0, :, 0, 1] + 0.25 * img[0, :, 0, 2]) + 0.5 * (0.75 * img[0, :, 1, 1] + 0.25 * img[0, :, 1, 2])
    assert torch.allclose(out[0, 4], want, atol=1e-6)
    assert torch.allclose(out[0, 5], 0.5 * img[0, :, 0, 0], atol=1e-6)       # half outside: zeros padding
    assert torch.allclose(out[0, 6], 0.5 * img[0, :, 4, 6], atol=1e-6)


def test_render_script_rejects_mismatched_checkpoint():
    import importlib, sys, os
    from collections import namedtuple
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    R = importlib.import_module('render')
    Res = namedtuple('Res', 'missing_keys unexpected_keys')
    R.check_loaded(Res([], ['init_net.depth_conv.0.weight', 'sph_fitter.regs']))        # tolerated
    with pytest.raises(SystemExit):
        R.check_loaded(Res(['fine_agg_net.agg_impl.base_fc.0.weight'], []))
    with pytest.raises(SystemExit):
        R.check_loaded(Res([], ['module.agg_net.prob_embed.0.weight']))


def test_bench_refuses_gpu_count_mismatch():
    """A launcher that started a different number of ranks than --gpus says must not silently report the wrong GPU count
    (plain `python bench.py --gpus N` starts its own ranks: tests/test_cabi_and_host.py, tests/test_gpu_bench_multirank.py)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8'], capture_output=True, text=True,
                       timeout=300, cwd=root, env=env)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_neuray_ft_module_surface_on_cpu():
    """name2network['neuray_ft'] (network/renderer.py:788-1200): per-scene `ray_feats` parameters [1, dim, h/8, w/8] per source view
    (:822, :1021-1027), `slice_imgs_info` batches with the reference's keys (:1029-1107), `get_coords_mask` pixel draws
    (utils/base_utils.py:585-604); the scene is passed in, the Habitat readers are out of scope."""
    import numpy as np
    from panogrf_amd.imgs_info import build_imgs_info
    from panogrf_amd.renderer import name2network, get_coords_mask
    H, W, n = 32, 64, 3
    g = np.random.RandomState(0)
    info = build_imgs_info(g.rand(n, 3, H, W).astype(np.float32), np.tile(np.eye(3, 4)[None], (n, 1, 1)), np.tile([[0.5, 15.0]], (n, 1)))
    cfg = dict(use_wrap_padding=True, dataset_name='m3d', height=H, width=W, min_depth=0.5, max_depth=15.0, use_hierarchical_sampling=True,
               depth_sample_num=32, fine_depth_sample_num=32, sample_num=32, train_ray_num=50, use_validation=False,
               dist_decoder_cfg={'use_vis': False}, fine_dist_decoder_cfg={'use_vis': False})
    with pytest.raises(Exception):
        name2network['neuray_ft'](dict(cfg))                                   # no scene given
    with pytest.raises(Exception):
        name2network['neuray_ft'](dict(cfg, use_self_hit_prob=True), ref_imgs_info=info)
    net = name2network['neuray_ft'](dict(cfg), ref_imgs_info=info)
    assert len(net.ray_feats) == n and all(p.shape == (1, 32, H // 8, W // 8) and p.requires_grad for p in net.ray_feats)
    assert any(k.startswith('ray_feats.') for k, _ in net.named_parameters())
    np.random.seed(3)
    ref, que = net.slice_imgs_info([0, 2], 1, True)
    assert ref['imgs'].shape == (2, 3, H, W) and ref['ray_feats'].shape == (2, 32, H // 8, W // 8) and ref['w2c'].shape == (2, 3, 4)
    assert que['coords'].shape == (1, 50, 2) and que['imgs'].shape == (1, 3, H, W) and que['c2w'].shape == (1, 3, 4)
    c = que['coords'][0].numpy()
    assert len({(x, y) for x, y in c.tolist()}) == 50 and c[:, 0].max() < W and c[:, 1].max() < H       # distinct pixels of the view
    mask = np.zeros((H, W), bool); mask[:4] = True
    np.random.seed(1)
    cc = get_coords_mask(mask, 40, 0.5)
    assert cc.shape == (40, 2) and (cc[:20, 1] < 4).all()                      # the foreground share comes from inside the mask first
    with pytest.raises(Exception):
        net.slice_imgs_info([0, 1, 2], 0, False)                               # no validation views were given
