"""Every BASELINE.json configuration at its REAL size under `-m gpu`: size-dependent paths (automatic
2-D ray tiles from 8 Mpx of source maps, the 84 MB pre-activation workspace, 2 M rays x 64 x 3 per-sample outputs, every
regulariser layer on the Winograd kernel) exercised where they are actually taken, with size-independent properties on the
whole result and CPU-oracle parity on strips / planes / slabs small enough for the host.

  configs[4]  1024x2048 ERP, 4 source views (multi-view cost volume), 64 samples
  configs[3]  512x1024 ERP, 2 views, 64 coarse + 64 fine samples
  configs[2]  D=64 cost volume at 128x256 + regulariser (UNet2 / CostRegNet)
  configs[0]  256x512 ERP, 2 views, 32 + 32 samples, random-init nets through render.py's plumbing (round 4)
(configs[1] = tests/test_gpu_render.py::test_full_size_properties_and_oracle_strip.)
"""
import os

import pytest
import torch

from tests import fixtures as FX
from tests.fixtures import max_abs

pytestmark = pytest.mark.gpu


def _coords(H, W):
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    return torch.stack([xs, ys], -1).reshape(-1, 2)


def _strip(H, W, n_random, seed):
    """Ray ids of an oracle strip: random pixels + the places where the spherical geometry is singular -- both pole rows,
    the +-pi seam columns and the four corners."""
    g = torch.Generator().manual_seed(seed)
    idx = torch.randperm(H * W, generator=g)[:n_random]
    xs = torch.tensor([0, 1, W // 2 - 1, W // 2, W - 2, W - 1])
    special = [y * W + xs for y in (0, 1, H // 2, H - 2, H - 1)]
    return torch.unique(torch.cat([idx] + special))


def _frame(E, frame):
    return E.Frame(frame['imgs'], frame['img_feats'], frame['ray_feats'], frame['w2c'], frame['ref_depth_range'],
                   frame['que_c2w'], frame['que_depth_range'])


def test_config5_1024x2048_four_views_properties_tile_order_and_oracle_strip(monkeypatch):
    """BASELINE configs[4]: whole-frame properties (finite, gamut, depth range), bit-identity of the automatically chosen
    2-D ray-tile visiting order (capi.hip: on from 8 Mpx of source maps) against PGRF_RAY_TILE=0, and oracle parity on a
    strip of > 256 rays incl. pole / seam rays.  Reference path: network/renderer.py:223-304 at rfn = 4."""
    from oracle import render_oracle as O
    from panogrf_amd import engine as E
    from bench import synthetic_case
    H, W, V, dn = 1024, 2048, 4, 64
    monkeypatch.delenv('PGRF_RAY_TILE', raising=False)
    case = synthetic_case(H, W, V, dn, seed=0)
    assert V * H * W >= (8 << 20)                      # the size from which the tile order is chosen automatically
    fr = _frame(E, case['frame'])
    rr = E.RayRenderer(case['cfg'], case['weights'])
    coords = _coords(H, W)
    out = rr.render_rays(fr, coords)
    px, dens, dep = out['pixel_colors_nr'], out['density_nr'], out['render_depth']
    assert px.shape == (H * W, 3) and out['colors_nr'].shape == (H * W, dn, 3)
    assert bool(torch.isfinite(px).all()) and bool(torch.isfinite(dens).all()) and bool(torch.isfinite(out['colors_nr']).all())
    assert float(px.min()) >= -1e-5 and float(px.max()) <= 1.0 + 1e-4          # sub-probability blend of source colours
    assert float(dep.min()) >= 0.0 and float(dep.max()) <= case['cfg']['max_depth'] * (1 + 1e-5)
    assert float(dens.min()) >= 0.0
    hp = out['hit_prob_nr']
    assert float(hp.min()) >= 0.0 and float(hp.sum(-1).max()) <= 1.0 + 1e-4
    # the same frame in plain row-major ray order: every output bit-identical
    monkeypatch.setenv('PGRF_RAY_TILE', '0')
    plain = rr.render_rays(fr, coords)
    for k in ('pixel_colors_nr', 'render_depth', 'density_nr', 'colors_nr', 'hit_prob_nr'):
        assert torch.equal(plain[k], out[k]), k
    monkeypatch.delenv('PGRF_RAY_TILE')
    del plain
    # rays are independent: a shuffled subset (ragged count -> plain order) reproduces the frame's pixels
    strip = _strip(H, W, 2048, 5)
    assert strip.numel() >= 256
    sub = rr.render_rays(fr, coords[strip])
    assert max_abs(sub['pixel_colors_nr'], px[strip.cuda()]) < 1e-6
    # oracle strip
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = O.render_rays(case['weights'], case['frame'], coords[strip], case['cfg'])
    mine_px, mine_den = px[strip.cuda()].cpu(), dens[strip.cuda()].cpu()
    e_px = max_abs(mine_px, ref['pixel_colors_nr'])
    e_den = max_abs(mine_den, ref['density_nr']) / max(1.0, float(ref['density_nr'].abs().max()))
    # W = 2048: one fp32 ulp of a projected angle is ~8e-5 px, libm / ocml atan2f, acosf differ by a few ulp, and the
    # synthetic feature maps are white noise (gradient ~2 per texel): twice the bounds of the 1024-wide test
    assert e_px < 1e-4, e_px
    assert e_den < 1e-3, e_den
    assert FX.psnr(mine_px, ref['pixel_colors_nr']) > 85.0
    assert max_abs(dep[strip.cuda()].cpu(), ref['render_depth']) < 5e-4


def test_config5_multiview_cost_volume_256x512_vs_oracle_planes():
    """BASELINE configs[4]'s MVS stage: spherical_cost_volume_mv (models/spherical_cost_volume_mv.py:219-347) with S = 4
    source views at the 1024x2048 configuration's cost-volume size h x w = 256 x 512 (depth 1024x2048 / 4), D = 64, C = 32 --
    1.07 GiB per reference view -- against oracle planes, for two choices of the reference view."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_volume as CV
    args = dict(dataset_name='m3d', contain_dnet=True, mono_uncertainty=False)
    g = torch.Generator().manual_seed(11)
    B, S, h, w, Cc, D = 1, 4, 256, 512, 32, 64
    feats = torch.randn(B, S, h, w, Cc, generator=g)
    ref_mu = 1.0 + 7.0 * torch.rand(B, 1, h, w, generator=g)
    dv = CV.build_depth_volume(ref_mu.cuda(), CV.depth_sampling(5, 3), 0.5, 0.1, 10, D)
    assert dv.shape == (B, D, h, w) and bool((dv[:, 1:] >= dv[:, :-1]).all())
    rots = torch.eye(3).view(1, 1, 3, 3).repeat(B, S, 1, 1).clone()
    c, s_ = 0.9961947, 0.0871557                                   # 5 degrees of yaw on two of the views
    rots[0, 1] = torch.tensor([[c, 0, s_], [0, 1, 0], [-s_, 0, c]])
    rots[0, 3] = torch.tensor([[c, 0, -s_], [0, 1, 0], [s_, 0, c]])
    trans = torch.tensor([[[0.0, 0.0, 0.0], [0.5, 0.0, 0.1], [-0.5, 0.05, 0.0], [0.0, -0.05, 0.5]]])
    for curr in (0, 2):
        cv = CV.calculate_cost_volume_erp_multiview(args, feats.cuda(), None, trans.cuda(), rots.cuda(), depth_volume=dv, curr_idx=curr)
        assert cv.shape == (B, D, h, w, Cc) and bool(torch.isfinite(cv).all())
        sel = [0, 40, 63] if curr == 0 else [17]
        ref = O.cost_volume_erp_multiview(feats, dv[:, sel].cpu(), rots, trans, curr_idx=curr)
        err = (cv[:, sel].cpu() - ref).abs()
        # w = 512: warp coordinates carry ~2e-4 px of fp32 noise, features are white noise (gradient ~2 per texel)
        assert float(err.max()) < 1e-2 and float(err.mean()) < 4e-5, (curr, float(err.max()), float(err.mean()))
        del cv


def test_config4_512x1024_coarse_plus_fine_vs_oracle_strip():
    """BASELINE configs[3] on one GPU: 512x1024, 2 views, 64 coarse + 64 importance-sampled fine samples
    (network/renderer.py:435-518 after :567-633): whole-frame properties + fine-pass PSNR against the oracle on a strip."""
    from oracle import render_oracle as O
    from panogrf_amd import engine as E
    from bench import synthetic_case
    H, W, V, dn = 512, 1024, 2, 64
    case = synthetic_case(H, W, V, dn, seed=0, fine=True)
    fr = _frame(E, case['frame'])
    rr = E.RayRenderer(case['cfg'], case['weights'])
    coords = _coords(H, W)
    out = rr.render_rays(fr, coords)
    for k in ('pixel_colors_nr', 'pixel_colors_nr_fine', 'render_depth_fine', 'density_nr_fine', 'depth_fine'):
        assert bool(torch.isfinite(out[k]).all()), k
    pf = out['pixel_colors_nr_fine']
    assert float(pf.min()) >= -1e-5 and float(pf.max()) <= 1.0 + 1e-4
    fd = out['depth_fine']
    assert fd.shape == (H * W, dn) and bool((fd[:, 1:] >= fd[:, :-1]).all())            # sorted importance samples
    assert float(fd.min()) >= case['cfg']['min_depth'] * (1 - 1e-5) and float(fd.max()) <= case['cfg']['max_depth'] * (1 + 1e-5)
    strip = _strip(H, W, 2048, 9)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = O.render_rays(case['weights'], case['frame'], coords[strip], case['cfg'])
    sc = strip.cuda()
    assert max_abs(out['pixel_colors_nr'][sc].cpu(), ref['pixel_colors_nr']) < 5e-5
    ps = FX.psnr(pf[sc].cpu(), ref['pixel_colors_nr_fine'])
    assert ps > 70.0, ps            # the inverse CDF amplifies round-off of the coarse hit probabilities (same bar as the fixtures)
    d = (fd[sc].cpu() - ref['depth_fine']).abs()
    assert float((d > 2e-3).float().mean()) < 2e-3, float((d > 2e-3).float().mean())
    assert max_abs(out['render_depth_fine'][sc].cpu(), ref['render_depth_fine']) < 2e-2
    # PSNR against a fixed target: |PSNR(HIP) - PSNR(oracle)| <= 0.05 dB (north star)
    gt = case['frame']['imgs'][0].permute(1, 2, 0).reshape(-1, 3)[strip]
    assert abs(FX.psnr(pf[sc].cpu(), gt) - FX.psnr(ref['pixel_colors_nr_fine'], gt)) <= 0.05


@pytest.mark.parametrize('reg', ['unet2', 'costregnet'])
def test_config3_regulariser_128x256x64_winograd_vs_direct_and_oracle_slab(reg, monkeypatch):
    """BASELINE configs[2], regulariser half (models/common_blocks.py:187-242 UNet2, models/cost_reg.py:99-175 CostRegNet) at
    the full 64 x 128 x 256 volume -- every UNet2 layer but the Cout = 1 head takes the Winograd kernel there -- against the
    all-direct-kernel run of the same volume, and against the oracle (F.conv3d) on a D/H-cropped slab that keeps W = 256, i.e.
    the same tile-row geometry (32 x 1 tiles per row, 16 x 2 at half width, ...), small enough for the host."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as CR
    torch.manual_seed(7)
    net = (CR.UNet2Reg() if reg == 'unet2' else CR.CostRegNet(dict(use_wrap_padding=True), 32)).eval()
    gw = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for name, t in net.state_dict().items():          # (state_dict tensors alias the parameters / buffers)
            if t.dim() > 1:
                t.mul_(0.5)
            elif name.endswith('running_var'):
                t.copy_(0.5 + torch.rand(t.shape, generator=gw))
            elif name.endswith('running_mean') or name.endswith('bias'):
                t.copy_(0.1 * torch.randn(t.shape, generator=gw))
            elif name.endswith('bn.weight'):
                t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=gw))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    g = torch.Generator().manual_seed(8)
    D, H, W = 64, 128, 256
    vol = torch.randn(1, D, H, W, 32, generator=g).cuda()                 # channel-last, the cost-volume kernel's layout
    outs = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('PANOGRF_CONV_WINOGRAD', flag)
        with torch.no_grad():
            outs[flag] = net.forward_ndhwc(vol).clone()
    monkeypatch.delenv('PANOGRF_CONV_WINOGRAD')
    a, b = outs['1'], outs['0']
    assert a.shape == b.shape and bool(torch.isfinite(a).all())
    scale = max(1.0, float(b.abs().max()))
    d = max_abs(a, b)
    if reg == 'unet2':
        assert 0.0 < d < 1e-4 * scale, (d, scale)        # different kernels (transforms add ~1e-6 relative round-off), same function
    else:
        assert d == 0.0                                  # CostRegNet's narrow layers never take the Winograd kernel
    del outs
    # oracle slab: same W, cropped D and H
    Ds, Hs = 32, 64
    slab = vol[:, :Ds, :Hs].contiguous()
    with torch.no_grad():
        mine = net.forward_ndhwc(slab)
    x = slab.permute(0, 4, 1, 2, 3).contiguous().cpu()                   # NCDHW for F.conv3d
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.unet2(sd, x) if reg == 'unet2' else O.cost_reg_net(sd, x)
    mine_c = mine.cpu().reshape(ref.shape)                               # one channel: [B,D,H,W,1] and [B,1,D,H,W] coincide
    err = max_abs(mine_c, ref)
    assert err < 5e-5 * max(1.0, float(ref.abs().max())), (err, float(ref.abs().max()))


def test_config0_256x512_random_init_through_render_py_plumbing_vs_oracle_strip():
    """BASELINE configs[0] AS WRITTEN: one 256x512 ERP panorama, 2 source views, 32 + 32 samples, RANDOM-INIT nets (PyTorch's
    default initialisation, no density boost, no checkpoint), through the plumbing of the reference's render.py:97-305 --
    cfg merge, `name2network[cfg['network']](cfg)`, `build_imgs_info` / `build_render_imgs_info`, `renderer(data)` with 'eval' --
    on the HIP path; whole-frame properties, chunking invariance, and a >= 2 048-ray strip (poles, seam, corners) against the CPU
    oracle on the module's own encoder outputs and parameters."""
    import numpy as np
    from oracle import render_oracle as O
    from render import DEFAULT_CFG
    from bench import synthetic_case
    from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, imgs_info_to_torch
    from panogrf_amd.renderer import name2network
    H, W, V, dn = 256, 512, 2, 32
    cfg = {**DEFAULT_CFG, 'height': H, 'width': W, 'depth_sample_num': dn, 'fine_depth_sample_num': dn, 'sample_num': dn}
    torch.manual_seed(0)
    net = name2network[cfg['network']](cfg).cuda().eval()            # random init: nothing is loaded
    case = synthetic_case(H, W, V, dn)
    imgs, w2c = case['frame']['imgs'].numpy(), case['frame']['w2c'].numpy()
    rng = [[cfg['min_depth'], cfg['max_depth']]] * V
    ref = imgs_info_to_torch(build_imgs_info(imgs, w2c, rng), 'cuda')
    ray_feats_in = torch.randn(V, 32, H // 8, W // 8, generator=torch.Generator().manual_seed(99))
    ref['ray_feats'] = ray_feats_in.cuda()
    pose = np.concatenate([np.eye(3), np.array([[0.1], [0.0], [0.02]])], 1)
    que = imgs_info_to_torch(build_render_imgs_info(pose, (H, W), [cfg['min_depth'], cfg['max_depth']]), 'cuda')
    with torch.no_grad():
        ref_run = dict(ref)
        out = net({'que_imgs_info': dict(que), 'ref_imgs_info': ref_run, 'eval': True})
        net.cfg['ray_batch_num'] = net.cfg['hip_ray_batch_num'] = 1536    # chunking invariance (renderer.py:656-683): not a divisor of H*W
        out2 = net({'que_imgs_info': dict(que), 'ref_imgs_info': dict(ref), 'eval': True})
    want_keys = {'pixel_colors_nr', 'pixel_colors_nr_fine', 'render_depth', 'render_depth_fine', 'colors_nr', 'colors_nr_fine',
                 'density_nr', 'density_nr_fine'}
    assert want_keys <= set(out), sorted(want_keys - set(out))
    assert not any(k.startswith('hit_prob') for k in out)             # eval drops them (renderer.py:675)
    pf = out['pixel_colors_nr_fine'][0]
    assert pf.shape == (H * W, 3) and bool(torch.isfinite(pf).all())
    assert float(pf.min()) >= -1e-5 and float(pf.max()) <= 1.0 + 1e-4
    rd = out['render_depth_fine'][0]
    assert bool(torch.isfinite(rd).all()) and float(rd.min()) >= 0.0 and float(rd.max()) <= cfg['max_depth'] * (1 + 1e-4)
    for k in ('pixel_colors_nr', 'pixel_colors_nr_fine', 'render_depth_fine'):
        assert torch.equal(out[k], out2[k]), k
    # CPU oracle on a strip: the module's parameters and the encoder outputs render() left in ref_imgs_info
    strip = _strip(H, W, 2048, 5)
    assert strip.numel() >= 2048
    coords = _coords(H, W)[strip]
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    c2w = que['c2w'][0].cpu()
    # (renderer(data) shallow-copies its dicts, renderer.py:778-781; render() leaves the encoder outputs in the dict it is given)
    ref_run = dict(ref)
    que_few = dict(que, coords=que['coords'][:, :64])
    with torch.no_grad():
        net.render(que_few, ref_run, False)
    fr = dict(imgs=case['frame']['imgs'], img_feats=ref_run['img_feats'].float().cpu().contiguous(), ray_feats=ref_run['ray_feats'].float().cpu().contiguous(),
              w2c=case['frame']['w2c'], ref_depth_range=torch.tensor(rng), que_c2w=c2w, que_depth_range=torch.tensor(rng[0]))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ocfg = dict(min_depth=cfg['min_depth'], max_depth=cfg['max_depth'], use_disp=True, depth_sample_num=dn, fine_depth_sample_num=dn,
                use_hierarchical_sampling=True, dist_decoder_cfg={'use_vis': False}, fine_dist_decoder_cfg={'use_vis': False}, height=H, width=W)
    want = O.render_rays(sd, fr, coords, ocfg)
    sc = strip.cuda()
    assert max_abs(out['pixel_colors_nr'][0][sc].cpu(), want['pixel_colors_nr']) < 2e-5
    assert max_abs(out['density_nr'][0][sc].cpu(), want['density_nr']) < 2e-5 * max(1.0, float(want['density_nr'].abs().max()))
    assert FX.psnr(pf[sc].cpu(), want['pixel_colors_nr_fine']) > 70.0
    assert max_abs(out['render_depth'][0][sc].cpu(), want['render_depth']) < 2e-4 * cfg['max_depth']
