"""Gradients of a network pass against golden vectors from the REFERENCE's own autograd graph (tests/golden/make_grad_fixtures.py:
network/renderer.py:223-304 differentiated in the build container).  CPU: the oracle restatement under torch.autograd;
`-m gpu`: pgrf_render_pass_backward.  Bar: relative L2 <= 5e-4 per tensor (tensors whose reference gradient is round-off --
rgb_fc.4.bias: a softmax over the views ignores a common shift -- are compared absolutely)."""
import os

import numpy as np
import pytest
import torch

from tests import fixtures as FX

GRAD_FIXTURES = [('render_grad_v2_dn32', 'render_stages_v2_dn32'), ('render_grad_v4_dn32', 'render_stages_v4_dn32')]
REL_L2 = 5e-4
PASSES = [('coarse', 'dist_decoder', 'agg_net'), ('fine', 'fine_dist_decoder', 'fine_agg_net')]


def _err(got, ref):
    """relative L2, or -- where the reference gradient is pure round-off -- the absolute norm scaled by 1e-4"""
    nr = float(torch.linalg.norm(ref.double()))
    d = float(torch.linalg.norm((got.double() - ref.double())))
    return d / nr if nr > 1e-5 else d / 1e-4 * REL_L2


def _load(gname):
    z = np.load(FX.GOLDEN + '/' + gname + '.npz')
    return {k: torch.from_numpy(z[k]) for k in z.files if k not in ('source',)}


@pytest.mark.parametrize('gname,sname', GRAD_FIXTURES)
def test_oracle_autograd_vs_reference_gradients(gname, sname):
    from oracle import render_oracle as O
    frame, w, cfg, g = FX.load(sname)
    G = _load(gname)
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    for tag, dd, ag in PASSES:
        keys = [k for k in w if k.startswith((dd + '.', ag + '.'))]
        wl = {k: (w[k].clone().requires_grad_(True) if k in keys else w[k]) for k in w}
        fr = dict(frame, img_feats=frame['img_feats'].clone().requires_grad_(True), ray_feats=frame['ray_feats'].clone().requires_grad_(True))
        out = O.render_pass(wl, dd, ag, fr, frame['coords'], g[tag]['depth'], {'use_vis': use_vis})
        up = {k[len(f'up.{tag}.'):]: v for k, v in G.items() if k.startswith(f'up.{tag}.')}
        loss = sum((out[k] * up[k]).sum() for k in up)
        assert abs(float(loss.detach()) - float(G[f"loss.{tag}"])) < 1e-4 * max(1.0, abs(float(G[f'loss.{tag}'])))
        grads = torch.autograd.grad(loss, [wl[k] for k in keys] + [fr['img_feats'], fr['ray_feats']], allow_unused=True)
        worst = ('', 0.0)
        for k, gt in zip(keys + ['img_feats', 'ray_feats'], grads):
            ref = G[f'grad.{tag}.{k}']
            gt = torch.zeros_like(ref) if gt is None else gt
            e = _err(gt, ref)
            worst = max(worst, (k, e), key=lambda t: t[1])
            assert e < REL_L2, (gname, tag, k, e)
        assert len(keys) == 62


@pytest.mark.gpu
@pytest.mark.parametrize('gname,sname', GRAD_FIXTURES)
def test_hip_backward_vs_reference_gradients(gname, sname):
    from panogrf_amd import engine as E
    frame, w, cfg, g = FX.load(sname)
    G = _load(gname)
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    fr = E.Frame(frame['imgs'], frame['img_feats'], frame['ray_feats'], frame['w2c'], frame['ref_depth_range'], frame['que_c2w'],
                 frame['que_depth_range'])
    names = E.raw_names()
    for tag, dd, ag in PASSES:
        up = {k[len(f'up.{tag}.'):]: v for k, v in G.items() if k.startswith(f'up.{tag}.')}
        raw = E.raw_weight_tensors(w, dd, ag, use_vis, fr.device)
        g_raw, g_if, g_rf, fwd = E.render_pass_backward(fr, raw, use_vis, frame['coords'], g[tag]['depth'], up, want_forward=True)
        loss = sum(float((fwd[k].cpu().double() * up[k].double()).sum()) for k in up)
        assert abs(loss - float(G[f'loss.{tag}'])) < 2e-4 * max(1.0, abs(float(G[f'loss.{tag}'])))
        checked = 0
        for n_, gt in zip(names, g_raw):
            key = (dd + n_[len('dist_decoder'):]) if n_.startswith('dist_decoder.') else (ag + n_[len('agg_net'):])
            if gt is None:
                assert '.vis_decoder.' in key and not use_vis
                continue
            e = _err(gt.cpu(), G[f'grad.{tag}.{key}'])
            assert e < REL_L2, (gname, tag, key, e)
            checked += 1
        assert checked == 62
        assert _err(g_if.cpu(), G[f'grad.{tag}.img_feats']) < REL_L2, (gname, tag, 'img_feats')
        assert _err(g_rf.cpu(), G[f'grad.{tag}.ray_feats']) < REL_L2, (gname, tag, 'ray_feats')


def _c2f_loss(out, up):
    return sum((out[k] * up[k]).sum() for k in up)


def test_oracle_c2f_all_autograd_vs_reference_gradients():
    """render_impl with render_c2f_all (coarse pass -> importance sampling on detached weights -> fine pass -> sorted merge -> one
    composite, network/renderer.py:484-518) differentiated by the reference; both passes run on the sample depths the reference
    used (stored with the gradients: the importance sampler is not part of the differentiated graph, renderer.py:437)."""
    from oracle import render_oracle as O
    frame, w, cfg, _ = FX.load('render_frame_v2_c2fall')
    G = _load('render_grad_c2fall_v2_dn32')
    keys = [k for k in w if k.startswith(('dist_decoder.', 'agg_net.', 'fine_dist_decoder.', 'fine_agg_net.'))]
    wl = {k: (w[k].clone().requires_grad_(True) if k in keys else w[k]) for k in w}
    fr = dict(frame, img_feats=frame['img_feats'].clone().requires_grad_(True), ray_feats=frame['ray_feats'].clone().requires_grad_(True))
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    oc = O.render_pass(wl, 'dist_decoder', 'agg_net', fr, G['coords'], G['depth.coarse'], {'use_vis': use_vis})
    of = O.render_pass(wl, 'fine_dist_decoder', 'fine_agg_net', fr, G['coords'], G['depth.fine'], {'use_vis': use_vis})
    m = O.merge_c2f_all(G['depth.coarse'], oc, G['depth.fine'], of)
    out = {'pixel_colors_nr': oc['pixel_colors_nr'], 'render_depth': oc['render_depth'], 'pixel_colors_nr_fine': m['pixel_colors_nr'],
           'render_depth_fine': m['render_depth']}
    up = {k[3:]: v for k, v in G.items() if k.startswith('up.')}
    for k in up:
        assert FX.max_abs(out[k].detach(), G['out.' + k]) < 2e-5 * max(1.0, float(G['out.' + k].abs().max())), k
    loss = _c2f_loss(out, up)
    grads = torch.autograd.grad(loss, [wl[k] for k in keys] + [fr['img_feats'], fr['ray_feats']], allow_unused=True)
    for k, gt in zip(keys + ['img_feats', 'ray_feats'], grads):
        ref = G['grad.' + k]
        gt = torch.zeros_like(ref) if gt is None else gt
        assert _err(gt, ref) < REL_L2, (k, _err(gt, ref))
    assert len(keys) == 124


@pytest.mark.gpu
def test_hip_c2f_all_backward_vs_reference_gradients():
    """The same on the HIP path: two RenderPassFunction passes + MergeC2fFunction (pgrf_merge_c2f_backward) under torch.autograd."""
    from panogrf_amd import engine as E
    frame, w, cfg, _ = FX.load('render_frame_v2_c2fall')
    G = _load('render_grad_c2fall_v2_dn32')
    dev = 'cuda'
    params = {k: v.to(dev).clone().requires_grad_(True) for k, v in w.items()}
    img_feats = frame['img_feats'].to(dev).clone().requires_grad_(True)
    ray_feats = frame['ray_feats'].to(dev).clone().requires_grad_(True)
    coords = G['coords'].to(dev)
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    common = (frame['imgs'].to(dev), img_feats, ray_feats, frame['w2c'].to(dev), frame['ref_depth_range'].to(dev), frame['que_c2w'],
              frame['que_depth_range'], params)
    depth, fd = G['depth.coarse'].to(dev), G['depth.fine'].to(dev)
    # (the samplers themselves: HIP against the reference's depths, the usual sampler tolerance)
    assert FX.max_abs(E.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], depth.shape[0], depth.shape[1], device=dev).cpu(), G['depth.coarse']) < 2e-5
    oc = E.render_pass_autograd(*common, 'dist_decoder', 'agg_net', use_vis, coords, depth)
    of = E.render_pass_autograd(*common, 'fine_dist_decoder', 'fine_agg_net', use_vis, coords, fd)
    m = E.merge_c2f_autograd(depth, oc['colors_nr'], oc['density_nr'], fd, of['colors_nr'], of['density_nr'])
    out = {'pixel_colors_nr': oc['pixel_colors_nr'], 'render_depth': oc['render_depth'], 'pixel_colors_nr_fine': m['pixel_colors_nr'],
           'render_depth_fine': m['render_depth']}
    up = {k[3:]: v.to(dev) for k, v in G.items() if k.startswith('up.')}
    for k in up:
        assert FX.max_abs(out[k].detach().cpu(), G['out.' + k]) < 5e-5 * max(1.0, float(G['out.' + k].abs().max())), k
    _c2f_loss(out, up).backward()
    n = 0
    errs = {k: _err((torch.zeros_like(p_) if p_.grad is None else p_.grad).cpu(), G['grad.' + k]) for k, p_ in params.items()
            if k.startswith(('dist_decoder.', 'agg_net.', 'fine_dist_decoder.', 'fine_agg_net.'))}
    print('worst c2f gradient errors:', sorted(errs.items(), key=lambda t: -t[1])[:6])
    for k, p_ in params.items():
        if k.startswith(('dist_decoder.', 'agg_net.', 'fine_dist_decoder.', 'fine_agg_net.')):
            got = torch.zeros_like(p_) if p_.grad is None else p_.grad
            # Bars of this composite graph: the merged composite spreads a ray's weight over 64 + 64 samples, so the fine pass's
            # gradients are sums of ~1e-6-sized terms of both signs and ILL-CONDITIONED: perturbing every weight by 1e-6 relative
            # moves the oracle's own out_geometry_fc.2.bias gradient by 1.1e-4 and the dist decoder's biases by 8e-4 (float32 CPU,
            # measured) -- amplification 100-800 -- and the HIP forward agrees with the reference to ~2e-6.  Measured (round 5):
            # 2.8e-3 on that [1] bias, <= 1.8e-3 on the other small vectors of the fine density branch, 4.8e-4 on geometry_fc.*.weight,
            # <= 3e-5 on every other matrix.  NOT the order of the float atomics: the deterministic reduction mode reproduces
            # these numbers bit for bit (test_hip_c2f_all_backward_deterministic_mode_agrees below).  The merge's own backward is
            # exact to 1e-5 (test_merge_c2f_backward_vs_autograd_of_the_oracle_merge), the per-pass backward is held to 5e-4 above.
            bar = 2 * REL_L2 if p_.numel() >= 512 else 4e-3
            assert _err(got.cpu(), G['grad.' + k]) < bar, (k, _err(got.cpu(), G['grad.' + k]))
            n += 1
    assert n == 124
    assert _err(img_feats.grad.cpu(), G['grad.img_feats']) < REL_L2
    assert _err(ray_feats.grad.cpu(), G['grad.ray_feats']) < REL_L2


@pytest.mark.gpu
def test_module_trains_with_render_c2f_all():
    """renderer(data) without 'eval' under autograd with cfg render_c2f_all: no longer refused; the loss's gradient reaches both
    passes' networks (the merged composite mixes the coarse and the fine samples)."""
    from panogrf_amd.renderer import name2network
    frame, w, cfg, _ = FX.load('render_frame_v2_c2fall')
    cfg = dict(cfg, network='neuray_base', use_wrap_padding=True)
    torch.manual_seed(0)
    net = name2network['neuray_base'](cfg)
    net.load_state_dict(w, strict=False)
    net = net.cuda().train()
    H, W = frame['imgs'].shape[-2:]
    idx = torch.arange(0, H * W, 7)[:48]
    coords = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    que = {'c2w': frame['que_c2w'][None].cuda(), 'coords': coords[None].cuda(), 'depth_range': frame['que_depth_range'][None].cuda()}
    ref = {'imgs': frame['imgs'].cuda(), 'w2c': frame['w2c'].cuda(), 'depth_range': frame['ref_depth_range'].cuda(), 'ray_feats': frame['ray_feats'].cuda()}
    out = net({'que_imgs_info': que, 'ref_imgs_info': ref})
    assert out['hit_prob_nr_fine'].shape[-1] == 2 * cfg['depth_sample_num']          # the merged samples
    (out['pixel_colors_nr_fine'] ** 2).mean().backward()
    gc = net.agg_net.agg_impl.base_fc[0].weight.grad
    gf = net.fine_agg_net.agg_impl.base_fc[0].weight.grad
    assert gc is not None and gf is not None and float(gc.abs().max()) > 0 and float(gf.abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('dn_c,dn_f', [(32, 32), (64, 64), (32, 64)])
def test_merge_c2f_backward_vs_autograd_of_the_oracle_merge(dn_c, dn_f):
    """pgrf_merge_c2f_backward alone against torch.autograd through oracle.merge_c2f_all (renderer.py:484-518) on random samples:
    interleaved depths, some zero densities (the ReLU's kink), every upstream gradient present."""
    from oracle import render_oracle as O
    from panogrf_amd import engine as E
    g = torch.Generator().manual_seed(dn_c * 100 + dn_f)
    rn = 37
    zc = torch.sort(0.5 + 14.5 * torch.rand(rn, dn_c, generator=g), -1)[0]
    zf = torch.sort(0.5 + 14.5 * torch.rand(rn, dn_f, generator=g), -1)[0]
    # (no exact ties between the lists: torch.sort is not stable, the order of tied samples -- and with it the composite -- would
    #  be the CPU sort's choice; the kernel's tie rule is pinned by tests/test_gpu_render.py against the reference frame)
    dc = torch.relu(torch.randn(rn, dn_c, generator=g) * 0.3 + 0.05)
    df = torch.relu(torch.randn(rn, dn_f, generator=g) * 0.3 + 0.05)
    cc, cf = torch.rand(rn, dn_c, 3, generator=g), torch.rand(rn, dn_f, 3, generator=g)
    up = {'pixel_colors_nr': torch.randn(rn, 3, generator=g), 'render_depth': 0.1 * torch.randn(rn, generator=g),
          'hit_prob_nr': torch.randn(rn, dn_c + dn_f, generator=g), 'colors_nr': 0.05 * torch.randn(rn, dn_c + dn_f, 3, generator=g),
          'density_nr': 0.05 * torch.randn(rn, dn_c + dn_f, generator=g)}
    leaves = [t.clone().requires_grad_(True) for t in (cc, dc, cf, df)]
    mo = O.merge_c2f_all(zc, {'colors_nr': leaves[0], 'density_nr': leaves[1]}, zf, {'colors_nr': leaves[2], 'density_nr': leaves[3]})
    want = torch.autograd.grad(sum((mo[k] * up[k]).sum() for k in up), leaves)
    dl = [t.cuda().clone().requires_grad_(True) for t in (cc, dc, cf, df)]
    mh = E.merge_c2f_autograd(zc.cuda(), dl[0], dl[1], zf.cuda(), dl[2], dl[3])
    for k in up:
        assert FX.max_abs(mh[k].detach().cpu(), mo[k].detach()) < 1e-5 * max(1.0, float(mo[k].abs().max())), k
    got = torch.autograd.grad(sum((mh[k] * up[k].cuda()).sum() for k in up), dl)
    for name, a, b in zip(('colors_c', 'density_c', 'colors_f', 'density_f'), got, want):
        assert FX.rel_l2(a.cpu(), b) < 1e-5, (name, FX.rel_l2(a.cpu(), b))


@pytest.mark.gpu
def test_hip_c2f_all_backward_deterministic_mode_agrees():
    """PGRF_TRAIN_DETERMINISTIC=1 (train_pass.hip): every weight-gradient element is one workgroup's sum over all rows in index order,
    one add into a zeroed buffer, no replicas, no side streams.  Same graph as above in two fresh processes: the default (atomic,
    replicated) reduction must agree with the deterministic one to 1e-5 relative L2 on EVERY tensor -- so what is left against the
    reference's gradients (bars above) is the forward's round-off through an ill-conditioned sum, not summation order -- and the
    deterministic gradients meet the same reference bars."""
    import subprocess, sys, tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    G = _load('render_grad_c2fall_v2_dn32')
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for mode in ('0', '1'):
            out = os.path.join(td, 'g%s.pt' % mode)
            env = dict(os.environ, PGRF_TRAIN_DETERMINISTIC=mode)
            r = subprocess.run([sys.executable, os.path.join(here, 'gpu_grad_modes.py'), out], capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2500:])
            res[mode] = torch.load(out)
    assert set(res['0']) == set(res['1']) and len(res['0']) == 126
    for k in res['0']:
        assert _err(res['0'][k], res['1'][k]) < 1e-5, (k, _err(res['0'][k], res['1'][k]))
        ref = G['grad.' + k]
        bar = REL_L2 if k in ('img_feats', 'ray_feats') else (2 * REL_L2 if ref.numel() >= 512 else 4e-3)
        assert _err(res['1'][k], ref) < bar, (k, _err(res['1'][k], ref))
