"""Backward pass of one network pass (pgrf_render_pass_backward) against torch.autograd THROUGH THE ORACLE
(oracle/render_oracle.py is pinned to the reference's stage tensors by tests/test_oracle_render.py, so its autograd graph is the
reference's: network/renderer.py:223-304 as the trainer differentiates it).  Inputs: the reference stage fixtures -- 1, 2, 3
and 4 source views, 32 and 64 samples, with and without the visibility decoder.  -m gpu"""
import pytest
import torch

from tests import fixtures as FX

pytestmark = pytest.mark.gpu

STAGE_FIXTURES = ['render_stages_v2_dn32', 'render_stages_v2_dn64', 'render_stages_v4_dn32', 'render_stages_v3_vis',
                  'render_stages_v2_randinit', 'render_stages_v1_dn32']


def _frame(E, frame):
    return E.Frame(frame['imgs'], frame['img_feats'], frame['ray_feats'], frame['w2c'], frame['ref_depth_range'],
                   frame['que_c2w'], frame['que_depth_range'])


def _upstream(rn, dn, seed):
    g = torch.Generator().manual_seed(seed)
    return {'pixel_colors_nr': torch.randn(rn, 3, generator=g), 'render_depth': 0.1 * torch.randn(rn, generator=g),
            'hit_prob_nr': torch.randn(rn, dn, generator=g), 'colors_nr': 0.05 * torch.randn(rn, dn, 3, generator=g),
            'density_nr': 0.05 * torch.randn(rn, dn, generator=g)}


def _oracle_grads(w, dd, ag, frame, coords, depth, use_vis, up):
    from oracle import render_oracle as O
    torch.set_num_threads(8)
    keys = [k for k in w if k.startswith(dd + '.') or k.startswith(ag + '.')]
    wl = {k: (w[k].clone().requires_grad_(True) if k in keys else w[k]) for k in w}
    fr = dict(frame)
    fr['img_feats'] = frame['img_feats'].clone().requires_grad_(True)
    fr['ray_feats'] = frame['ray_feats'].clone().requires_grad_(True)
    out = O.render_pass(wl, dd, ag, fr, coords, depth, {'use_vis': use_vis})
    loss = sum((out[k] * up[k]).sum() for k in up)
    leaves = [wl[k] for k in keys] + [fr['img_feats'], fr['ray_feats']]
    grads = torch.autograd.grad(loss, leaves, allow_unused=True)
    gw = {k: g for k, g in zip(keys, grads[:len(keys)])}
    return gw, grads[-2], grads[-1], {k: out[k].detach() for k in up}


def _close(a, b, rel, floor):
    scale = max(float(b.abs().max()), floor)
    return float((a - b).abs().max()) / scale


@pytest.mark.parametrize('name,rn_sub,dn_sub', [(n, None, None) for n in STAGE_FIXTURES] +
                         # ragged batches: (point x view) row counts that are not multiples of the kernels' 128-row workgroup tiles (448),
                         # nor of a wave's 32 rows (310, 435, 203), nor of a 64-lane ray chunk
                         [('render_stages_v2_dn32', 7, 32), ('render_stages_v2_dn32', 5, 31), ('render_stages_v3_vis', 5, 29),
                          ('render_stages_v1_dn32', 7, 29)])
def test_backward_vs_autograd_through_the_oracle(name, rn_sub, dn_sub):
    from panogrf_amd import engine as E
    frame, w, cfg, g = FX.load(name)
    fr = _frame(E, frame)
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    coords = frame['coords'] if rn_sub is None else frame['coords'][:rn_sub].contiguous()
    for tag, dd, ag, seed in [('coarse', 'dist_decoder', 'agg_net', 1), ('fine', 'fine_dist_decoder', 'fine_agg_net', 2)]:
        depth = g[tag]['depth']
        if rn_sub is not None:
            depth = depth[:rn_sub, :dn_sub].contiguous()
        rn, dn = depth.shape
        up = _upstream(rn, dn, seed)
        gw_ref, gif_ref, grf_ref, out_ref = _oracle_grads(w, dd, ag, frame, coords, depth, use_vis, up)
        raw = E.raw_weight_tensors(w, dd, ag, use_vis, fr.device)
        g_raw, g_if, g_rf, fwd = E.render_pass_backward(fr, raw, use_vis, coords, depth, up, want_forward=True)
        # the rematerialised forward is the fused kernel's forward (and the reference's)
        blob = E.pack_pass_weights(w, dd, ag, use_vis).cuda()
        fused = E.render_pass(fr, blob, use_vis, coords, depth) if dn in (32, 64, 128) else None      # (the fused kernel's sample counts)
        for k in ('pixel_colors_nr', 'hit_prob_nr', 'density_nr', 'colors_nr', 'render_depth'):
            if fused is not None:
                assert FX.max_abs(fwd[k].cpu(), fused[k].cpu()) < 2e-5 * max(1.0, float(fused[k].abs().max())), (name, tag, k)
            assert FX.max_abs(fwd[k].cpu(), out_ref[k]) < 1e-4 * max(1.0, float(out_ref[k].abs().max())), (name, tag, k)
        # weight gradients, in the reference's state_dict layout
        names = E.raw_names()
        checked = 0
        for n_, gt in zip(names, g_raw):
            key = (dd + n_[len('dist_decoder'):]) if n_.startswith('dist_decoder.') else (ag + n_[len('agg_net'):])
            if gt is None:
                assert '.vis_decoder.' in key and not use_vis
                continue
            ref = gw_ref[key]
            ref = torch.zeros_like(w[key]) if ref is None else ref
            assert gt.shape == ref.shape, (key, gt.shape, ref.shape)
            # (floor: rgb_fc.4.bias has an exactly zero gradient -- a softmax over the views ignores a common shift -- so both
            #  sides hold round-off of ~1e-7 there)
            err = _close(gt.cpu(), ref, 1e-3, 1e-3)
            assert err < 2e-3, (name, tag, key, err, float(ref.abs().max()))
            checked += 1
        assert checked >= 62
        # feature-map gradients (scatter-add of the bilinear gathers)
        assert _close(g_if.cpu(), gif_ref, 1e-3, 1e-6) < 2e-3, (name, tag, 'img_feats')
        assert _close(g_rf.cpu(), grf_ref, 1e-3, 1e-6) < 2e-3, (name, tag, 'ray_feats')
        assert float(grf_ref.abs().max()) > 0 and float(gif_ref.abs().max()) > 0


@pytest.mark.parametrize('name', ['render_stages_v2_dn32', 'render_stages_v3_vis'])
def test_autograd_function_kept_activations_equal_rematerialised(name, monkeypatch):
    """RenderPassFunction in its two modes: staged forward kept for the reverse sweep (default: pgrf_render_pass_backward_phase 1 / 2)
    against the fused forward + rematerialising backward (PANOGRF_TRAIN_KEEP_ACTS=0).  Outputs equal to round-off (two different forward
    kernels), gradients equal to ~1e-5 of their scale (the same sweep over the same activations, atomics in a different order)."""
    from panogrf_amd import engine as E
    frame, w, cfg, g = FX.load(name)
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    depth = g['coarse']['depth'].cuda()
    coords = frame['coords'].cuda()
    rn, dn = depth.shape
    up = {k: v.cuda() for k, v in _upstream(rn, dn, 5).items()}
    res = []
    for keep in ('1', '0'):
        monkeypatch.setenv('PANOGRF_TRAIN_KEEP_ACTS', keep)
        params = {k: torch.nn.Parameter(v.cuda().clone()) for k, v in w.items()}
        imf = frame['img_feats'].cuda().clone().requires_grad_(True)
        rf = frame['ray_feats'].cuda().clone().requires_grad_(True)
        out = E.render_pass_autograd(frame['imgs'].cuda(), imf, rf, frame['w2c'].cuda(), frame['ref_depth_range'].cuda(), frame['que_c2w'],
                                     frame['que_depth_range'], params, 'dist_decoder', 'agg_net', use_vis, coords, depth)
        loss = sum((out[k] * up[k]).sum() for k in up)
        loss.backward()
        res.append(({k: v.detach().clone() for k, v in out.items()}, {k: p.grad.clone() for k, p in params.items() if p.grad is not None},
                    imf.grad.clone(), rf.grad.clone()))
    (o1, g1, i1, r1), (o0, g0, i0, r0) = res
    for k in o1:
        assert FX.max_abs(o1[k].cpu(), o0[k].cpu()) < 2e-5 * max(1.0, float(o0[k].abs().max())), k
    assert g1.keys() == g0.keys() and len(g1) >= 62
    for k in g1:
        assert FX.max_abs(g1[k].cpu(), g0[k].cpu()) < 2e-4 * max(float(g0[k].abs().max()), 1e-3), k
    assert FX.max_abs(i1.cpu(), i0.cpu()) < 2e-4 * float(i0.abs().max()) and FX.max_abs(r1.cpu(), r0.cpu()) < 2e-4 * float(r0.abs().max())


def test_backward_argument_validation():
    import ctypes as C
    from panogrf_amd import engine as E
    from panogrf_amd._lib import lib, PgrfError, PgrfPassGrad
    frame, w, cfg, g = FX.load('render_stages_v2_dn32')
    fr = _frame(E, frame)
    raw = E.raw_weight_tensors(w, 'dist_decoder', 'agg_net', False, fr.device)
    depth = g['coarse']['depth']
    with pytest.raises(PgrfError):          # a missing non-optional weight
        E.render_pass_backward(fr, [None] + raw[1:], False, frame['coords'], depth, {})
    with pytest.raises(PgrfError):          # unsupported sample count
        E.render_pass_backward(fr, raw, False, frame['coords'], torch.ones(frame['coords'].shape[0], 2), {})
    assert lib.pgrf_render_pass_backward_workspace_bytes(2, 0, 32) == 0
    need = lib.pgrf_render_pass_backward_workspace_bytes(2, 12, 32)
    assert need > 12 * 32 * 2 * 4000
    # empty batch: a no-op
    g_raw, g_if, g_rf, _ = E.render_pass_backward(fr, raw, False, torch.zeros(0, 2), torch.zeros(0, 32), {})
    assert all(float(t.abs().max()) == 0.0 for t in g_raw if t is not None) and float(g_if.abs().max()) == 0.0


def test_module_training_step_matches_autograd_through_the_oracle():
    """renderer(data) WITHOUT 'eval' under autograd (the reference's training call, network/renderer.py:777-786): coarse + fine
    pass as autograd Functions on the HIP kernels, encoders on torch.  One loss, backward, and every gradient -- per-ray
    networks, image encoder, visibility encoder -- compared with the same graph built from the oracle's torch ops on the
    SAME encoder modules and the same importance-sampling draws."""
    import numpy as np
    from oracle import render_oracle as O
    from panogrf_amd.renderer import name2network
    frame, _, cfg, g = FX.load('render_full_v2_dn32')
    z = np.load(FX.GOLDEN + '/render_full_v2_dn32.npz')
    shapes = []
    for ln in bytes(z['state.keys']).decode().split('\n'):
        k, s_ = ln.split(':')
        shapes.append((k, tuple(int(t) for t in s_.split(',')) if s_ else ()))
    net = name2network['neuray_base'](dict(cfg))
    net.load_state_dict(FX.seeded_renderer_state(shapes, 800))
    net = net.cuda().train()
    H, W = frame['imgs'].shape[-2:]
    gsel = torch.Generator().manual_seed(3)
    idx = torch.randperm(H * W, generator=gsel)[:96]
    coords = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    target = torch.rand(96, 3, generator=gsel).cuda()
    que = {'c2w': frame['que_c2w'][None].cuda(), 'coords': coords[None].cuda(), 'depth_range': frame['que_depth_range'][None].cuda()}
    mk_ref = lambda: {'imgs': frame['imgs'].cuda(), 'w2c': frame['w2c'].cuda(), 'depth_range': frame['ref_depth_range'].cuda(),
                      'ray_feats': frame['ray_feats'].cuda()}

    def loss_of(out):
        return ((out['pixel_colors_nr_fine'] - target) ** 2).mean() + 0.5 * ((out['pixel_colors_nr'] - target) ** 2).mean() + \
            0.01 * out['render_depth_fine'].mean() + 0.1 * (out['hit_prob_nr'] ** 2).sum(-1).mean()

    # ---- HIP path
    net.zero_grad()
    torch.manual_seed(11)                      # the importance sampler's torch.rand draws
    out = net({'que_imgs_info': dict(que), 'ref_imgs_info': mk_ref()})          # no 'eval': training mode
    assert out['pixel_colors_nr_fine'].requires_grad and 'hit_prob_nr_fine' in out
    loss = loss_of({k: v[0] for k, v in out.items() if v.dtype == torch.float32})
    loss.backward()
    mine = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    assert any(k.startswith('image_encoder.') for k in mine) and any(k.startswith('vis_encoder.') for k in mine)
    assert all(bool(torch.isfinite(v).all()) for v in mine.values())

    # ---- the same graph from the oracle's torch ops (GPU tensors, same modules, same draws)
    net.zero_grad()
    torch.manual_seed(11)
    u = torch.rand([1, 96, cfg['fine_depth_sample_num']])[0].cuda()
    ref = mk_ref()
    feats, rf = net._torch_encoders(ref['imgs'], ref['ray_feats'])
    w = {k: p for k, p in net.named_parameters()}
    fr = dict(imgs=ref['imgs'], img_feats=feats, ray_feats=rf, w2c=ref['w2c'], ref_depth_range=ref['depth_range'],
              que_c2w=frame['que_c2w'].cuda(), que_depth_range=frame['que_depth_range'].cuda())
    # sample depths are constants of the graph: take the fine depths the HIP sampler drew from the HIP coarse pass (the oracle's
    # own inverse CDF lands ~1e-4 away, which white-noise feature maps turn into per-cent level gradient differences)
    from panogrf_amd import engine as E
    dn = cfg['depth_sample_num']
    depth_c = O.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], 96, dn, cfg.get('use_disp', True), 'cuda')
    fd = E.sample_fine_depth(depth_c, out['hit_prob_nr'][0].detach(), frame['que_depth_range'], cfg['fine_depth_sample_num'],
                             use_disp=cfg.get('use_disp', True), u=u)
    uv = cfg['dist_decoder_cfg']['use_vis']
    oc = O.render_pass(w, 'dist_decoder', 'agg_net', fr, coords.cuda(), depth_c, {'use_vis': uv})
    of = O.render_pass(w, 'fine_dist_decoder', 'fine_agg_net', fr, coords.cuda(), fd, {'use_vis': cfg['fine_dist_decoder_cfg']['use_vis']})
    oout = dict(oc, **{k + '_fine': v for k, v in of.items()})
    oloss = loss_of(oout)
    oloss.backward()
    theirs = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    assert abs(float(loss.detach()) - float(oloss.detach())) < 1e-5 * max(1.0, abs(float(oloss.detach())))
    assert set(mine) == set(theirs)
    worst = 0.0
    for k in theirs:
        a, b = mine[k].double().flatten(), theirs[k].double().flatten()
        nb = float(b.norm())
        if nb < 1e-7:
            assert float(a.norm()) < 1e-5, k
            continue
        rel = float((a - b).norm()) / nb
        worst = max(worst, rel)
        assert rel < 3e-3, (k, rel)
    assert worst > 0.0
    # an optimiser step through the module's own parameters works (and invalidates the packed inference weights)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    for k, p in net.named_parameters():
        p.grad = mine.get(k)
    before = float(loss.detach())
    opt.step()
    torch.manual_seed(11)
    out2 = net({'que_imgs_info': dict(que), 'ref_imgs_info': mk_ref()})
    after = float(loss_of({k: v[0] for k, v in out2.items() if v.dtype == torch.float32}).detach())
    assert after < before, (before, after)


def test_c2f_all_with_render_uncert_backward_vs_autograd_of_the_oracle():
    """cfg render_c2f_all + render_uncert under autograd (network/renderer.py:484-518: the merged composite's uncertainty
    sum_i w_i (z_i - depth)^2 + 1e-5 over the sorted union of both passes' samples).  The module's training path (two HIP passes,
    MergeC2fFunction, the uncertainty in torch on the merged hit probabilities) against the same graph from the oracle's ops
    (render_pass x 2 + merge_c2f_all(uncert=True)) on the SAME sample depths: loss and every gradient."""
    import numpy as np
    from oracle import render_oracle as O
    from panogrf_amd import engine as E
    from panogrf_amd.renderer import name2network
    frame, _, cfg, g = FX.load('render_full_v2_dn32')
    z = np.load(FX.GOLDEN + '/render_full_v2_dn32.npz')
    shapes = []
    for ln in bytes(z['state.keys']).decode().split('\n'):
        k, s_ = ln.split(':')
        shapes.append((k, tuple(int(t) for t in s_.split(',')) if s_ else ()))
    cfg = dict(cfg, render_c2f_all=True, render_uncert=True, render_depth=True, sample_num=64, depth_sample_num=32, fine_depth_sample_num=32)
    # (the merged composite has 64 samples; the per-pass networks keep 32: the positional table is per pass)
    cfg['sample_num'] = 32
    net = name2network['neuray_base'](dict(cfg))
    net.load_state_dict(FX.seeded_renderer_state(shapes, 800))
    net = net.cuda().train()
    H, W = frame['imgs'].shape[-2:]
    gsel = torch.Generator().manual_seed(5)
    idx = torch.randperm(H * W, generator=gsel)[:80]
    coords = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    target = torch.rand(80, 3, generator=gsel).cuda()
    que = {'c2w': frame['que_c2w'][None].cuda(), 'coords': coords[None].cuda(), 'depth_range': frame['que_depth_range'][None].cuda()}
    mk_ref = lambda: {'imgs': frame['imgs'].cuda(), 'w2c': frame['w2c'].cuda(), 'depth_range': frame['ref_depth_range'].cuda(),
                      'ray_feats': frame['ray_feats'].cuda()}

    def loss_of(out):
        return ((out['pixel_colors_nr_fine'] - target) ** 2).mean() + 0.3 * out['render_uncert_fine'].mean() + 0.1 * out['render_uncert'].mean() + \
            0.05 * out['render_depth_fine'].mean()

    net.zero_grad()
    torch.manual_seed(13)
    out = net({'que_imgs_info': dict(que), 'ref_imgs_info': mk_ref()})
    assert out['render_uncert_fine'].requires_grad and out['render_uncert_fine'].shape == (1, 80) and out['hit_prob_nr_fine'].shape == (1, 80, 64)
    loss = loss_of({k: v[0] for k, v in out.items() if v.dtype == torch.float32})
    loss.backward()
    mine = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}

    net.zero_grad()
    torch.manual_seed(13)
    u = torch.rand([1, 80, 32])[0].cuda()
    ref = mk_ref()
    feats, rf = net._torch_encoders(ref['imgs'], ref['ray_feats'])
    w = {k: p for k, p in net.named_parameters()}
    fr = dict(imgs=ref['imgs'], img_feats=feats, ray_feats=rf, w2c=ref['w2c'], ref_depth_range=ref['depth_range'],
              que_c2w=frame['que_c2w'].cuda(), que_depth_range=frame['que_depth_range'].cuda())
    depth_c = O.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], 80, 32, cfg.get('use_disp', True), 'cuda')
    fd = E.sample_fine_depth(depth_c, out['hit_prob_nr'][0].detach(), frame['que_depth_range'], 32, use_disp=cfg.get('use_disp', True), u=u)
    uv = cfg['dist_decoder_cfg']['use_vis']
    oc = O.render_pass(w, 'dist_decoder', 'agg_net', fr, coords.cuda(), depth_c, {'use_vis': uv})
    of = O.render_pass(w, 'fine_dist_decoder', 'fine_agg_net', fr, coords.cuda(), fd, {'use_vis': cfg['fine_dist_decoder_cfg']['use_vis']})
    m = O.merge_c2f_all(depth_c, oc, fd, of, uncert=True)
    oout = {'pixel_colors_nr_fine': m['pixel_colors_nr'], 'render_uncert_fine': m['render_uncert'], 'render_depth_fine': m['render_depth'],
            'render_uncert': O.depth_uncertainty(depth_c, oc['render_depth'], oc['hit_prob_nr'])}
    oloss = loss_of(oout)
    oloss.backward()
    theirs = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    assert abs(float(loss.detach()) - float(oloss.detach())) < 2e-5 * max(1.0, abs(float(oloss.detach())))
    assert FX.max_abs(out['render_uncert_fine'][0].detach().cpu(), m['render_uncert'].detach().cpu()) < 2e-5 * max(1.0, float(m["render_uncert"].detach().abs().max()))
    assert set(mine) == set(theirs)
    num = den = 0.0
    for k in theirs:
        a, b = mine[k].double().flatten(), theirs[k].double().flatten()
        num, den = num + float((a - b).norm()) ** 2, den + float(b.norm()) ** 2
        if float(b.norm()) > 1e-7:
            assert float((a - b).norm()) / float(b.norm()) < (5e-3 if b.numel() >= 512 else 2e-2), (k, float((a - b).norm()) / float(b.norm()))
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5


def test_map_gradient_scatter_per_ray_merge_equals_one_atomic_per_tap():
    """gather32_bwd_ray_kernel (a ray's taps merged in registers before the atomics, round 5) against the plain scatter
    (PGRF_TRAIN_SCATTER_RAY=0) on shapes around its segmentation: the minimum of 3 samples, odd counts, more waves' worth of samples
    than a FIFO holds, 1-4 views.  Same products; only the association of a texel's sum differs."""
    import os
    import subprocess
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    cases = ['5x3x1', '7x5x3', '3x128x4', '9x64x2', '33x31x2']
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for mode in ('0', '1'):
            out = os.path.join(td, 's%s.pt' % mode)
            env = dict(os.environ, PGRF_TRAIN_SCATTER_RAY=mode)
            r = subprocess.run([sys.executable, os.path.join(here, 'gpu_scatter_modes.py'), out] + cases, capture_output=True, text=True,
                               timeout=600, env=env)
            assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2500:])
            res[mode] = torch.load(out)
    assert set(res['0']) == set(res['1']) and len(res['0']) == 2 * len(cases)
    for k in res['0']:
        a, b = res['1'][k], res['0'][k]
        assert float(b.abs().max()) > 0, k
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (k, float((a - b).abs().max()), float(b.abs().max()))
