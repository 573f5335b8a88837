"""Pin the CPU oracle (oracle/render_oracle.py) against golden vectors captured from the
reference implementation (tests/golden/make_fixtures.py).  CPU only."""
import pytest
import torch

from oracle import render_oracle as O
from tests import fixtures as FX

STAGE_FIXTURES = ['render_stages_v2_dn32', 'render_stages_v2_dn64', 'render_stages_v4_dn32',
                  'render_stages_v3_vis', 'render_stages_v2_randinit', 'render_stages_v1_dn32']

# tolerances: the reference differs from itself by ~4e-7 (coarse) between thread counts
TOL = dict(px=2e-3, depth_v=2e-5, ddiff=2e-6, vis=2e-5, hit=2e-5, ray_feats=2e-5, rgb=2e-5, img_feats=2e-5,
           neuray=5e-5, base_x=5e-5, geo=5e-5, attn=5e-5, density_nr=5e-5, colors_nr=2e-5,
           pixel_colors_nr=2e-5, hit_prob_nr=2e-5, render_depth=2e-4, que_dists=1e-6)


@pytest.mark.parametrize('name', STAGE_FIXTURES)
def test_stage_parity(name):
    torch.set_num_threads(1)
    frame, w, cfg, g = FX.load(name)
    coords = frame['coords']
    dn = cfg['depth_sample_num']
    depth = O.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], coords.shape[0], dn)
    assert torch.equal(depth, g['coarse']['depth'])
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    for tag, dd, ag in [('coarse', 'dist_decoder', 'agg_net'), ('fine', 'fine_dist_decoder', 'fine_agg_net')]:
        gold = g[tag]
        out = O.render_pass(w, dd, ag, frame, coords, gold['depth'], {'use_vis': use_vis}, want_debug=True)
        flat = {**out['dbg'], **{k: v for k, v in out.items() if k != 'dbg'}}
        checked = 0
        for k, tol in TOL.items():
            if k in gold:
                if k == 'px':
                    # the seam makes x jump by W-1 for angles a hair either side of it; compare on the circle
                    W = frame['imgs'].shape[-1]
                    dx = (flat[k][..., 0] - gold[k][..., 0]).abs()
                    dx = torch.minimum(dx, (W - 1) - dx)
                    assert float(dx.max()) < tol and FX.max_abs(flat[k][..., 1], gold[k][..., 1]) < tol
                else:
                    err = FX.max_abs(flat[k], gold[k])
                    assert err < tol * max(1.0, float(gold[k].abs().max())), (name, tag, k, err)
                checked += 1
        assert checked >= 10
    # importance sampler, fed with the reference's own coarse outputs
    fd = O.sample_fine_depth(g['coarse']['depth'], g['coarse']['hit_prob_nr'], frame['que_depth_range'], dn)
    assert FX.max_abs(fd, g['fine']['depth']) < 1e-4 * 15.0


def test_gather_explicit_matches_grid_sample():
    frame, w, cfg, g = FX.load('render_stages_v2_dn32')
    H, W = frame['imgs'].shape[-2:]
    px = g['coarse']['px'].reshape(2, -1, 2)
    for key in ['imgs', 'img_feats', 'ray_feats']:
        a = O.gather_map(frame[key], px, H, W)
        b = O.gather_map_explicit(frame[key], px, H, W)
        assert FX.max_abs(a, b) < 1e-5


@pytest.mark.parametrize('name', ['render_frame_v2_dn32', 'render_frame_v2_c2fall'])
def test_frame_parity_and_chunk_invariance(name):
    torch.set_num_threads(1)
    frame, w, cfg, g = FX.load(name)
    out = O.render_frame(w, frame, cfg, ray_batch_num=512)
    gold = g['out']
    assert FX.max_abs(out['pixel_colors_nr'], gold['pixel_colors_nr']) < 2e-5
    assert FX.psnr(out['pixel_colors_nr_fine'], gold['pixel_colors_nr_fine']) > 80.0
    assert FX.max_abs(out['render_depth_fine'], gold['render_depth_fine']) < 5e-3
    out2 = O.render_frame(w, frame, cfg, ray_batch_num=173)
    assert FX.max_abs(out2['pixel_colors_nr_fine'], out['pixel_colors_nr_fine']) < 1e-4


# ---- spherical conventions of the other datasets and pinhole queries, pinned to the reference's own
# outputs (tests/golden/make_geometry_fixtures.py -> geometry_conventions.npz)
import os as _os

import numpy as _np
import pytest as _pytest


def _geo():
    z = _np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'golden', 'geometry_conventions.npz'))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else int(z[k])) for k in z.files}


@_pytest.mark.parametrize('ds', ['m3d', 'replica_test', 'residential', 'CoffeeArea'])
def test_oracle_dataset_conventions_vs_reference(ds):
    from oracle import render_oracle as O
    g = _geo()
    pts, qdir = O.ray_points(g['coords'], g['que_c2w'], g['depth'], g['H'], g['W'], ds)
    assert float((pts - g[f'{ds}.pts']).abs().max()) < 2e-6
    assert float((qdir - g[f'{ds}.que_dir']).abs().max()) < 1e-6
    px, dv = O.project_to_views(g[f'{ds}.pts'].reshape(-1, 3), g['w2c'], g['H'], g['W'], ds)
    assert float((dv - g[f'{ds}.depth_v']).abs().max()) < 2e-6
    assert float((px - g[f'{ds}.px']).abs().max()) < 2e-5          # pixels, W = 32


def test_oracle_perspective_rays_vs_reference():
    from oracle import render_oracle as O
    g = _geo()
    pts, qdir = O.ray_points_perspective(g['persp.coords'], g['persp.pose'], g['persp.K'], g['depth'])
    assert float((pts - g['persp.pts']).abs().max()) < 2e-6
    assert float((qdir - g['persp.que_dir']).abs().max()) < 1e-6


def test_training_mode_forward_vs_reference():
    """is_train=True: deterministic coarse depths + random importance samples, fed with the reference's own uniform
    draws (tests/golden/make_fixtures.py train_fixture); and sample_depth(random_sample=True) for the jitter path."""
    torch.set_num_threads(1)
    frame, w, cfg, g = FX.load('render_train_v2_dn32')
    gold = g['out']
    out = O.render_rays(w, frame, frame['coords'], cfg, u=frame['u'])
    assert FX.max_abs(out['pixel_colors_nr'], gold['pixel_colors_nr']) < 2e-5
    assert FX.max_abs(out['hit_prob_nr'], gold['hit_prob_nr']) < 2e-5
    assert FX.max_abs(out['render_depth'], gold['render_depth']) < 2e-4
    assert FX.psnr(out['pixel_colors_nr_fine'], gold['pixel_colors_nr_fine']) > 80.0
    assert FX.max_abs(out['render_depth_fine'], gold['render_depth_fine']) < 5e-3
    rn, dn = frame['coords'].shape[0], cfg['depth_sample_num']
    d0 = O.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], rn, dn)
    dj = O.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], rn, dn, jitter=frame['jitter'])
    assert FX.max_abs(dj, gold['jittered_depth']) < 2e-6
    # the jitter really moved the samples, and never past the neighbouring bin
    assert float((dj - d0).abs().max()) > 1e-3 and bool((dj[:, 1:] > dj[:, :-1]).all())


# ---- round 3: cfg keys of SURVEY 8(b) that had no reference vector (tests/golden/make_round3_fixtures.py)
@pytest.mark.parametrize('name', ['render_r3_uncert', 'render_r3_uncert_c2f', 'render_r3_nodisp'])
def test_render_uncert_and_linear_depth_sampling_vs_reference(name):
    """render_uncert (renderer.py:305-312; over the merged samples with render_c2f_all, :507-514) and use_disp=False
    (render_ops.py:303-318 linear-depth coarse samples, :423-424 importance sampling without the inverse-depth map)."""
    torch.set_num_threads(1)
    frame, w, cfg, g = FX.load(name)
    gold = g['out']
    rn, dn = frame['coords'].shape[0], cfg['depth_sample_num']
    cd = O.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], rn, dn, cfg['use_disp'])
    assert FX.max_abs(cd, gold['coarse_depth']) < 2e-6
    fd = O.sample_fine_depth(gold['coarse_depth'], gold['hit_prob_nr'], frame['que_depth_range'], cfg['fine_depth_sample_num'], cfg['use_disp'])
    assert FX.max_abs(fd, torch.sort(gold['fine_depth_unsorted'], -1)[0]) < 1e-4 * 15.0
    out = O.render_rays(w, frame, frame['coords'], cfg)
    assert FX.max_abs(out['pixel_colors_nr'], gold['pixel_colors_nr']) < 2e-5
    assert FX.max_abs(out['render_depth'], gold['render_depth']) < 2e-4
    assert FX.psnr(out['pixel_colors_nr_fine'], gold['pixel_colors_nr_fine']) > 80.0
    assert FX.max_abs(out['render_depth_fine'], gold['render_depth_fine']) < 5e-3
    if cfg.get('render_uncert'):
        assert FX.max_abs(out['render_uncert'], gold['render_uncert']) < 2e-6
        assert FX.max_abs(out['render_uncert_fine'], gold['render_uncert_fine']) < 2e-5
        assert float(gold['render_uncert'].min()) >= 1e-5
    else:
        assert 'render_uncert' not in out and not cfg['use_disp']
        assert float((cd[:, 1:] - cd[:, :-1]).std()) < 1e-5          # linear in depth: equal steps
