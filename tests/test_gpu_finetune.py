"""`name2network['neuray_ft']` (network/renderer.py:788-1200) on the HIP fast path: 20 fine-tuning steps -- the per-scene `ray_feats`
parameters, both 2-D encoders and the coarse + fine per-ray networks under Adam, the reference's RenderLoss (network/loss.py:55-165,
panogrf_amd/loss.py) reduced as the trainer does (train/trainer.py:377-394) -- against the SAME training loop built from the oracle's
torch ops (oracle/render_oracle.py + oracle/loss_oracle.py) on an identically initialised copy with the same random draws."""
import copy
import math

import numpy as np
import pytest
import torch

from tests import fixtures as FX

pytestmark = pytest.mark.gpu


def _scene(H, W, n, seed):
    """n panoramas on a small circle around the origin: smooth textures (sums of low-frequency sinusoids), y-up w2c poses"""
    g = np.random.RandomState(seed)
    th, ph = np.meshgrid((np.arange(W) + 0.5) / W * 2 * math.pi, (np.arange(H) + 0.5) / H * math.pi)
    imgs = np.zeros((n, 3, H, W), np.float32)
    for v in range(n):
        for c in range(3):
            acc = np.zeros((H, W))
            for _ in range(6):
                a, b, p0 = g.randint(1, 4), g.randint(1, 3), g.uniform(0, 6.28)
                acc += g.uniform(0.3, 1.0) * np.sin(a * th + b * ph + p0 + 0.2 * v)
            imgs[v, c] = (acc - acc.min()) / (acc.max() - acc.min())
    w2c = np.zeros((n, 3, 4))
    for v in range(n):
        w2c[v, :3, :3] = np.eye(3)
        w2c[v, :, 3] = [-0.4 * math.cos(2 * math.pi * v / n), 0.0, -0.4 * math.sin(2 * math.pi * v / n)]
    return imgs, w2c, np.tile(np.array([[0.5, 15.0]], np.float32), (n, 1))


# Two fp32 implementations of one training graph part ways when both run FREE: measured here, step 0 equal to the last digit, 6e-6
# at step 1, 3e-3 by step 3 under SGD (lr 2e-3) as under Adam, with white-noise or smooth feature maps alike -- the round-off of one
# step is amplified ~50 x per step by the dynamics (Adam's first updates are lr * sign(g) whatever |g| is; the seeded weights are in
# a stiff regime).  So the oracle does not run free: before every step it takes over the fast path's parameters, and what is held to
# the bar is what one step contributes -- the loss (1e-4 relative, the verdict's figure; measured <= 5e-6) and every gradient tensor
# (the whole gradient -- all parameters concatenated --: median step within 1e-3 relative L2, every step within 1e-2) at 20 DIFFERENT points of a real fine-tuning trajectory under the
# shipped optimiser settings (Adam, lr_init 1e-4: configs/train/ft_hr_1.0/*.yaml:103-104) and, for a curve that moves, 1e-3.
@pytest.mark.parametrize('lr', [1e-4, 1e-3])
def test_neuray_ft_twenty_steps_match_the_oracle_training_loop(lr):
    from oracle import loss_oracle as LO
    from oracle import render_oracle as O
    from panogrf_amd import engine as E
    from panogrf_amd import loss as PL
    from panogrf_amd.imgs_info import build_imgs_info
    from panogrf_amd.renderer import name2network

    _, _, cfg, _ = FX.load('render_full_v2_dn32')
    H, W, n_ref, rays = 64, 128, 3, 128
    cfg = dict(cfg, height=H, width=W, train_ray_num=rays, use_nr_fine_loss=True, fix_all=False, fix_coarse=False, network='neuray_ft',
               use_validation=False, ft_include_self=False)
    imgs, w2c, dr = _scene(H, W, n_ref, 5)
    info = build_imgs_info(imgs, w2c, dr)
    torch.manual_seed(0)
    net_h = name2network['neuray_ft'](dict(cfg), ref_imgs_info=info)
    # seeded weights with live densities (default init gives alpha ~ 0: nothing to fit), the same state for both copies
    shapes = [(k, tuple(v.shape)) for k, v in net_h.state_dict().items() if not k.startswith('ray_feats')]
    sd = FX.seeded_renderer_state(shapes, 800)
    # per-scene visibility features: smooth fields (what an init net leaves), not the white noise of the from-scratch branch -- with
    # white-noise maps a 1e-6 shift of a fine sample changes the gathered feature at the per-cent level and two fp32 implementations
    # of one graph part ways within five steps
    gsm = torch.Generator().manual_seed(7)
    for k, v in net_h.state_dict().items():
        if k.startswith('ray_feats'):
            low = torch.randn(1, v.shape[1], 3, 4, generator=gsm)
            sd[k] = torch.nn.functional.interpolate(low, size=v.shape[-2:], mode='bicubic', align_corners=True)
    net_h.load_state_dict(sd)
    net_o = copy.deepcopy(net_h)
    net_h, net_o = net_h.cuda().train(), net_o.cuda().train()
    assert len(net_h.ray_feats) == n_ref and net_h.ray_feats[0].shape == (1, 32, H // 8, W // 8)
    loss_fn = PL.name2loss['render'](cfg)
    opt_h = torch.optim.Adam(net_h.parameters(), lr=lr)
    worst_grad, worst_small, totals, per_tensor_bad, per_tensor_n = 0.0, 0.0, [], 0, 0
    uv = cfg['dist_decoder_cfg']['use_vis']
    dn, fdn = cfg['depth_sample_num'], cfg['fine_depth_sample_num']
    curve_h, curve_o = [], []
    for step in range(20):
        # ---- HIP fast path: renderer(data) without 'eval' = train_step() (network/renderer.py:1116-1157, 1188-1200)
        np.random.seed(1000 + step)
        torch.manual_seed(step)
        opt_h.zero_grad()
        out = net_h({'index': 0})
        assert out['pixel_colors_nr_fine'].requires_grad and out['pixel_colors_gt'].shape == (1, rays, 3) and out['que_id'] in (0, 1, 2)
        loss_h = PL.total_loss(loss_fn(out, None, step))
        loss_h.backward()
        grads_h = {k: p.grad.detach().clone() for k, p in net_h.named_parameters() if p.grad is not None}
        curve_h.append(float(loss_h.detach()))
        # ---- the same step from the oracle's ops: same view / pixel draws, same importance-sampling draws
        np.random.seed(1000 + step)
        torch.manual_seed(step)
        net_o.load_state_dict(net_h.state_dict())          # the oracle starts every step from the fast path's parameters
        net_o.zero_grad()
        que_i = np.random.randint(0, n_ref)
        assert que_i == out['que_id']
        ref, que = net_o.slice_imgs_info([i for i in range(n_ref) if i != que_i], que_i, True)
        assert torch.equal(que['coords'], out['que_imgs_info']['coords'])
        feats, rf = net_o._torch_encoders(ref['imgs'], ref['ray_feats'])
        w = dict(net_o.named_parameters())
        fr = dict(imgs=ref['imgs'], img_feats=feats, ray_feats=rf, w2c=ref['w2c'], ref_depth_range=ref['depth_range'],
                  que_c2w=que['c2w'][0], que_depth_range=que['depth_range'][0])
        coords = que['coords'][0]
        depth_c = O.sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], rays, dn, cfg.get('use_disp', True), 'cuda')
        oc = O.render_pass(w, 'dist_decoder', 'agg_net', fr, coords, depth_c, {'use_vis': uv})
        u = torch.rand([1, rays, fdn])[0].cuda()
        # the fine samples from the FAST path's coarse hit probabilities (its own 'hit_prob_nr' output): the inverse-CDF sampler is piecewise
        # -- a 1e-6 difference between the two coarse passes moves a sample across a bin edge now and then, and the two fine passes then
        # look at different points of that ray (a whole-gradient deviation of 1e-2 .. 4e-1 at such a step, ~1 step in 100)
        assert float((oc['hit_prob_nr'].detach() - out['hit_prob_nr'][0].detach()).abs().max()) < 1e-4
        fd = E.sample_fine_depth(depth_c, out['hit_prob_nr'][0].detach(), que['depth_range'][0].cpu(), fdn, use_disp=cfg.get('use_disp', True), u=u)
        of = O.render_pass(w, 'fine_dist_decoder', 'fine_agg_net', fr, coords, fd, {'use_vis': cfg['fine_dist_decoder_cfg']['use_vis']})
        h, wd = que['imgs'].shape[-2:]
        grid = torch.stack([coords[:, 0] / (wd - 1) * 2 - 1, coords[:, 1] / (h - 1) * 2 - 1], -1)[None, None]
        gt = torch.nn.functional.grid_sample(que['imgs'], grid, mode='bilinear', padding_mode='zeros', align_corners=True)[0, :, 0].t()[None]
        data_pr = {'pixel_colors_nr': oc['pixel_colors_nr'][None], 'pixel_colors_nr_fine': of['pixel_colors_nr'][None], 'pixel_colors_gt': gt,
                   'ray_mask': torch.ones(1, rays, dtype=torch.bool, device='cuda')}
        loss_o = LO.total_loss(LO.render_loss(data_pr, cfg))
        loss_o.backward()
        curve_o.append(float(loss_o.detach()))
        grads_o = {k: p.grad.detach() for k, p in net_o.named_parameters() if p.grad is not None}
        assert set(grads_h) == set(grads_o)
        num = den = 0.0
        for k, go in grads_o.items():
            nb = float(go.double().norm())
            diff = float((grads_h[k].double() - go.double()).norm())
            num, den = num + diff * diff, den + nb * nb
            if nb < 1e-7:
                assert float(grads_h[k].double().norm()) < 1e-5, (step, k)
                continue
            rel = diff / nb
            worst_grad = max(worst_grad, rel) if go.numel() >= 512 else worst_grad
            worst_small = max(worst_small, rel) if go.numel() < 512 else worst_small
            # per tensor: a loose bar -- a tensor whose gradient is a small difference of large per-ray terms (the fine pass's dist
            # decoder, biases, the 1- and 2-row heads: the conditioning of tests/test_grad_fixtures.py, where the deterministic
            # reduction mode reproduces the deviation) reaches 6e-3 in some runs and steps (float atomics: run-to-run different)
            per_tensor_bad += int(rel >= (2e-2 if go.numel() >= 512 else 5e-2))
            per_tensor_n += 1
        # the whole gradient (every parameter concatenated: what the optimiser follows)
        totals.append((num / den) ** 0.5)
        opt_h.step()
    print('HIP   ', ['%.6f' % x for x in curve_h])
    print('oracle', ['%.6f' % x for x in curve_o])
    print('whole-gradient relative L2 error per step:', ['%.1e' % t for t in totals])
    print('per tensor: worst %.2e (>= 512 elements), %.2e (smaller); %d of %d tensor-steps beyond 2e-2 / 5e-2' % (worst_grad, worst_small, per_tensor_bad, per_tensor_n))
    # With the sampler's discontinuity taken out (above) the whole gradient of a step agrees to 1e-6 .. 1.3e-4 (30 runs observed).  ReLU kinks
    # (prob_embed's ReLU, the density's max(., 0), masked softmaxes) remain: where a pre-activation sits within the two forwards' 1e-6 of
    # zero its derivative is 0 on one side and 1 on the other, and which samples do varies from run to run with the order of the float
    # atomics.  Hence: the median step within 3e-4, every step but at most ONE within 2e-3 (none beyond 2e-2), and at most 1 % of the
    # (tensor, step) pairs beyond the per-tensor bars.
    st = sorted(totals)
    assert st[len(st) // 2] < 3e-4 and st[-2] < 2e-3 and st[-1] < 2e-2, totals
    assert per_tensor_bad <= per_tensor_n // 100, (per_tensor_bad, per_tensor_n)
    for a, b in zip(curve_h, curve_o):
        assert abs(a - b) <= 1e-4 * abs(b), (curve_h, curve_o)
    # the fit moves: mean loss of the last five steps below the first five (views differ step to step, so compare windows)
    if lr >= 1e-3:
        assert sum(curve_h[-5:]) < sum(curve_h[:5])
    # every trainable group received updates, the per-scene ray_feats included
    for k, p in net_h.named_parameters():
        if k.startswith(('ray_feats.', 'image_encoder.conv1', 'vis_encoder.', 'agg_net.agg_impl.base_fc.0', 'fine_dist_decoder.mean_decoder.0')):
            assert not torch.equal(p.detach().cpu(), sd[k].cpu()), k
    # validation / free-pose rendering run on the fused inference path with the trained parameters
    pose = {k: torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v for k, v in
            __import__('panogrf_amd.imgs_info', fromlist=['x']).build_render_imgs_info(w2c[0], (H, W), (0.5, 15.0)).items()}
    img = net_h.render_pose(pose)['pixel_colors_nr_fine']
    assert img.shape == (1, H * W, 3) and bool(torch.isfinite(img).all())
