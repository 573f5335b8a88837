"""MVS stage around the cost volume against vectors captured from the reference's own FullPipeline /
CostVolumeInitNet methods (tests/golden/make_mvs_fixtures.py; the mono-depth net is the fixture's stand-in)."""
import pytest
import torch

from tests import mvs_fixtures as MF
from tests.fixtures import max_abs


def _args(H=32, W=64):
    return dict(MF.MVS_ARGS, height=H, width=W, mono_height=H, mono_width=W, depth_height=H, depth_width=W)


def _depth_net(device='cpu'):
    from panogrf_amd.mvs import MVSDepthNet
    fx = MF.load()
    shapes = MF.key_shapes(fx['pipe']['keys'])
    net = MVSDepthNet(_args(), d_net=MF.StubMono()).eval()
    return net, fx, shapes


def test_state_dict_keys_match_reference():
    """CPU: the `init_net.*` entries of a reference checkpoint must load with strict=True."""
    from panogrf_amd.mvs import CostVolumeInitNet
    net, fx, shapes = _depth_net()
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == dict(shapes)
    net.load_state_dict(MF.seeded_state(shapes, 61), strict=True)
    cfg = dict(_args(), use_wrap_padding=True, seq_len=3, min_depth=0.5, max_depth=15.0)
    inet = CostVolumeInitNet(cfg, d_net=MF.StubMono())
    ishapes = MF.key_shapes(fx['init']['keys'])
    mine = {k: tuple(v.shape) for k, v in inet.state_dict().items() if not k.startswith('depth_net.')}
    assert mine == dict(ishapes)
    assert {k[len('depth_net.'):] for k in inet.state_dict() if k.startswith('depth_net.')} == set(dict(shapes))


def test_equi_feature_net_vs_reference():
    """CPU: ERP ResNet-18 encoder + decoder (models/my_equi.py) in torch."""
    torch.set_num_threads(4)
    net, fx, shapes = _depth_net()
    net.load_state_dict(MF.seeded_state(shapes, 61))
    with torch.no_grad():
        out = net.unet(fx['equi']['x'])
    assert out.shape == fx['equi']['out'].shape
    assert max_abs(out, fx['equi']['out']) < 2e-4 * max(1.0, float(fx['equi']['out'].abs().max()))


@pytest.mark.gpu
def test_two_view_depth_vs_reference():
    net, fx, shapes = _depth_net()
    net.load_state_dict(MF.seeded_state(shapes, 61))
    net = net.cuda()
    t = fx['two']
    with torch.no_grad():
        ret = net.estimate_depth_using_cost_volume(t['panos'].cuda(), t['rots'].cuda(), t['trans'].cuda(), 0.1, 10.0)
    assert max_abs(ret['mono_depth_ref'].cpu(), t['mono_depth_ref']) < 1e-5
    scale = max(1.0, float(t['cost_reg'].abs().max()))
    assert max_abs(ret['cost_reg'].cpu(), t['cost_reg']) < 2e-4 * scale
    assert max_abs(ret['mid_outputs']['cost_volume'].cpu(), t['cost_volume']) < 2e-4 * scale
    assert max_abs(ret['rectified_depth_d1'].cpu(), t['rectified_depth_d1']) < 2e-4 * max(1.0, float(t['rectified_depth_d1'].max()))
    assert ret['depth'].shape == t['depth'].shape
    assert max_abs(ret['depth'].cpu(), t['depth']) < 2e-4 * max(1.0, float(t['depth'].max()))


@pytest.mark.gpu
def test_multiview_depth_vs_reference():
    net, fx, shapes = _depth_net()
    net.load_state_dict(MF.seeded_state(shapes, 61))
    net = net.cuda()
    t = fx['mv']
    with torch.no_grad():
        ret = net.estimate_depth_using_cost_volume_v3_erp_multiview(t['panos'].cuda(), t['rots'].cuda(), t['trans'].cuda(), 0, 0.1, 10.0)
    assert max_abs(ret['cost_reg'].cpu(), t['cost_reg']) < 2e-4 * max(1.0, float(t['cost_reg'].abs().max()))
    assert max_abs(ret['depth'].cpu(), t['depth']) < 2e-4 * max(1.0, float(t['depth'].max()))


@pytest.mark.gpu
def test_init_net_vs_reference_and_through_the_renderer():
    """CostVolumeInitNet.forward (depth -> ray_feats) vs the reference, and as `init_net` of NeuralRayGenRenderer."""
    from panogrf_amd.mvs import CostVolumeInitNet
    from panogrf_amd import PgrfError
    fx = MF.load()
    shapes, ishapes = MF.key_shapes(fx['pipe']['keys']), MF.key_shapes(fx['init']['keys'])
    cfg = dict(_args(), use_wrap_padding=True, seq_len=3, min_depth=0.5, max_depth=15.0)
    inet = CostVolumeInitNet(cfg, d_net=MF.StubMono()).eval()
    inet.load_state_dict({**MF.seeded_state(ishapes, 91), **{'depth_net.' + k: v for k, v in MF.seeded_state(shapes, 61).items()}})
    inet = inet.cuda()
    t = fx['init']
    ref = {'imgs': t['ref_imgs'].cuda(), 'rots': t['ref_rots'].cuda(), 'trans': t['ref_trans'].cuda()}
    src = {'imgs': t['src_imgs'].cuda(), 'rots': t['src_rots'].cuda(), 'trans': t['src_trans'].cuda()}
    with torch.no_grad():
        out = inet(ref, src, False)
    assert max_abs(out['mvs_depth'].cpu(), t['mvs_depth']) < 2e-4 * max(1.0, float(t['mvs_depth'].max()))
    assert out['ray_feats'].shape == t['ray_feats'].shape
    assert max_abs(out['ray_feats'].cpu(), t['ray_feats']) < 5e-4 * max(1.0, float(t['ray_feats'].abs().max()))
    # without a mono prior the stage refuses to run instead of inventing one
    bare = CostVolumeInitNet(cfg).cuda().eval()
    with pytest.raises(PgrfError):
        bare(ref, src, False)
    # use_gt_depth_ray (init_net.py:593-595): the ray features come from ref_imgs_info['depth'], 'mvs_depth' stays the network's --
    # fed the network's own depth as "ground truth" the features must be the ones above, fed another depth they must move
    gt = CostVolumeInitNet(dict(cfg, use_gt_depth_ray=True), d_net=MF.StubMono()).eval()
    gt.load_state_dict(inet.state_dict())
    gt = gt.cuda()
    with torch.no_grad():
        o1 = gt(dict(ref, depth=out['mvs_depth']), src, False)
        o2 = gt(dict(ref, depth=out['mvs_depth'] * 0.5 + 0.3), src, False)
    # (two module instances: MIOpen may pick different convolution algorithms for them, so equal to round-off, not to the bit)
    close = lambda a, b: float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    assert close(o1['mvs_depth'], out['mvs_depth']) and close(o1['ray_feats'], out['ray_feats'])
    assert close(o2['mvs_depth'], out['mvs_depth']) and float((o2['ray_feats'] - out['ray_feats']).abs().max()) > 1e-2
    # use_empty_depth_ray (init_net.py:639-646): a fresh trainable N(0, 1) tensor per call in place of the features
    emp = CostVolumeInitNet(dict(cfg, use_empty_depth_ray=True), d_net=MF.StubMono()).eval()
    emp.load_state_dict(inet.state_dict())
    emp = emp.cuda()
    with torch.no_grad():
        e1, e2 = emp(ref, src, False), emp(ref, src, False)
    assert e1['ray_feats'].shape == out['ray_feats'].shape and e1['ray_feats'].requires_grad and e1['ray_feats'].is_cuda
    assert abs(float(e1['ray_feats'].mean())) < 0.05 and abs(float(e1['ray_feats'].std()) - 1.0) < 0.05
    assert not torch.equal(e1['ray_feats'], e2['ray_feats']) and close(e1['mvs_depth'], out['mvs_depth'])
    for key in ('wo_stereo', 'handle_distort'):
        with pytest.raises((PgrfError, NotImplementedError)):
            CostVolumeInitNet(dict(cfg, **{key: True}), d_net=MF.StubMono())


def test_uncert_tune_state_dict_keys_match_reference():
    """CPU: cfg['uncert_tune'] wraps the depth net in UncertWrapper (network/init_net.py:491-497): the wrapper's own layers and the
    wrapped net's (`depth_net.mvs_net.*`) carry the reference's names and shapes, BatchNorm running statistics included."""
    from panogrf_amd.mvs import CostVolumeInitNet
    fx = MF.load('uncert_fixtures')
    cfg = dict(_args(), use_wrap_padding=True, seq_len=3, min_depth=0.5, max_depth=15.0, uncert_tune=True)
    inet = CostVolumeInitNet(cfg, d_net=MF.StubMono())
    sd = {k: tuple(v.shape) for k, v in inet.state_dict().items()}
    head = {k[len('depth_net.'):]: v for k, v in sd.items() if k.startswith('depth_net.') and not k.startswith('depth_net.mvs_net.')}
    assert head == dict(MF.key_shapes(fx['head']['keys'])) and len(head) == 43
    assert {k[len('depth_net.mvs_net.'):]: v for k, v in sd.items() if k.startswith('depth_net.mvs_net.')} == dict(MF.key_shapes(fx['pipe']['keys']))
    assert {k: v for k, v in sd.items() if not k.startswith('depth_net.')} == dict(MF.key_shapes(fx['init']['keys']))


@pytest.mark.gpu
def test_uncert_tune_vs_reference():
    """UncertWrapper.estimate_depth_using_cost_volume (both variance activations) and the init net's 'mvs_uncert' against vectors
    from the reference's own wrapper (tests/golden/make_uncert_fixtures.py)."""
    from panogrf_amd.mvs import CostVolumeInitNet, MVSDepthNet, UncertWrapper
    from panogrf_amd import PgrfError
    fx = MF.load('uncert_fixtures')
    shapes, head, ishapes = (MF.key_shapes(fx[g]['keys']) for g in ('pipe', 'head', 'init'))
    cfg = dict(_args(), use_wrap_padding=True, seq_len=3, min_depth=0.5, max_depth=15.0, uncert_tune=True)
    t = fx['in']
    for tag, extra in (('elu', {}), ('softplus', {'new_uncert_tune': True})):
        w = UncertWrapper(dict(cfg, **extra), MVSDepthNet(_args(), d_net=MF.StubMono())).eval()
        w.load_state_dict({**MF.seeded_state(head, 131), **{'mvs_net.' + k: v for k, v in MF.seeded_state(shapes, 61).items()}})
        w = w.cuda()
        with torch.no_grad():
            ret = w.estimate_depth_using_cost_volume(t['panos'].cuda(), t['rots'].cuda(), t['trans'].cuda(), 0.1, 10.0)
        want = fx[tag]
        assert max_abs(ret['depth'].cpu(), want['depth']) < 2e-4 * max(1.0, float(want['depth'].max()))
        assert ret['var'].shape == want['var'].shape and float(ret['var'].min()) > 0.0
        assert max_abs(ret['var'].cpu(), want['var']) < 5e-4 * max(1.0, float(want['var'].max())), tag
        assert max_abs(ret['pred_final'].cpu(), want['pred_final']) < 5e-4 * max(1.0, float(want['pred_final'].abs().max()))
        with pytest.raises(PgrfError):
            w.estimate_depth_using_cost_volume_v3_erp_multiview(None, None, None, 0, 0.1, 10.0)
    inet = CostVolumeInitNet(cfg, d_net=MF.StubMono()).eval()
    inet.load_state_dict({**MF.seeded_state(ishapes, 91), **{'depth_net.' + k: v for k, v in MF.seeded_state(head, 131).items()},
                          **{'depth_net.mvs_net.' + k: v for k, v in MF.seeded_state(shapes, 61).items()}})
    inet = inet.cuda()
    t = fx['init']
    ref = {'imgs': t['ref_imgs'].cuda(), 'rots': t['ref_rots'].cuda(), 'trans': t['ref_trans'].cuda()}
    src = {'imgs': t['src_imgs'].cuda(), 'rots': t['src_rots'].cuda(), 'trans': t['src_trans'].cuda()}
    with torch.no_grad():
        out = inet(ref, src, False)
    assert max_abs(out['mvs_depth'].cpu(), t['mvs_depth']) < 2e-4 * max(1.0, float(t['mvs_depth'].max()))
    assert max_abs(out['mvs_uncert'].cpu(), t['mvs_uncert']) < 5e-4 * max(1.0, float(t['mvs_uncert'].max()))
    assert max_abs(out['ray_feats'].cpu(), t['ray_feats']) < 5e-4 * max(1.0, float(t['ray_feats'].abs().max()))


@pytest.mark.gpu
def test_gen_renderer_end_to_end_with_the_mvs_stage():
    """name2network['neuray_gen'](cfg) with init_net_type 'cost_volume': init net (MVS stage) -> 2-D encoders -> per-ray
    renderer in one `renderer(data)` call, the reference's render.py flow (render.py:264-278)."""
    import numpy as np
    from panogrf_amd.renderer import name2network
    from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, imgs_info_to_torch
    H, W, dn = 64, 128, 32
    cfg = dict(_args(32, 64), network='neuray_gen', init_net_type='cost_volume', use_wrap_padding=True, seq_len=3, height=H, width=W,
               use_disp=True, use_hierarchical_sampling=True, render_depth=True, depth_sample_num=dn, fine_depth_sample_num=dn,
               sample_num=dn, dist_decoder_cfg={'use_vis': False}, fine_dist_decoder_cfg={'use_vis': False})
    cfg.update(min_depth=0.5, max_depth=15.0)                 # render range; the MVS range stays mvs_min/max_depth
    torch.manual_seed(0)
    net = name2network['neuray_gen'](cfg, d_net=MF.StubMono()).cuda().eval()
    assert any(k.startswith('init_net.depth_net.unet3d') for k in net.state_dict())
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(2, 3, H, W, generator=g).numpy()
    pose = lambda tx: np.concatenate([np.eye(3), np.array([[tx], [0.0], [0.0]])], 1)
    ref = imgs_info_to_torch(build_imgs_info(imgs, np.stack([pose(0.5), pose(-0.5)]), [[0.5, 15.0]] * 2), 'cuda')
    src = imgs_info_to_torch(build_imgs_info(imgs[::-1].copy(), np.stack([pose(-0.5), pose(0.5)]), [[0.5, 15.0]] * 2), 'cuda')
    que = imgs_info_to_torch(build_render_imgs_info(pose(0.0), (H, W), [0.5, 15.0]), 'cuda')
    with torch.no_grad():
        out = net({'que_imgs_info': dict(que), 'ref_imgs_info': dict(ref), 'src_imgs_info': dict(src), 'eval': True})
    assert out['pixel_colors_nr_fine'].shape == (1, H * W, 3) and torch.isfinite(out['pixel_colors_nr_fine']).all()
    assert out['render_depth_fine'].shape == (1, H * W) and float(out['render_depth_fine'].min()) >= 0.0
    # the image encoder runs on a side stream next to the MVS stage (cfg['hip_overlap_encoders']): same launches on the same
    # inputs, so the frame is bit-identical with the overlap off, and a second frame replays the captured graphs
    assert '_early_feats' not in net.__dict__
    with torch.no_grad():
        again = net({'que_imgs_info': dict(que), 'ref_imgs_info': dict(ref), 'src_imgs_info': dict(src), 'eval': True})
        net.cfg['hip_overlap_encoders'] = False
        serial = net({'que_imgs_info': dict(que), 'ref_imgs_info': dict(ref), 'src_imgs_info': dict(src), 'eval': True})
    for k in ('pixel_colors_nr_fine', 'render_depth_fine', 'pixel_colors_nr'):
        assert torch.equal(out[k], again[k]) and torch.equal(out[k], serial[k]), k


@pytest.mark.gpu
def test_source_maps_are_reused_only_while_inputs_and_weights_are_unchanged():
    """cfg['hip_reuse_source_maps']: a trajectory against fixed source views runs the MVS stage and the encoders once.  Same
    frames as without the reuse; any change of a source pixel / pose (new tensor or in-place write) or of a weight recomputes."""
    import numpy as np
    from panogrf_amd.renderer import name2network
    from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, imgs_info_to_torch
    H, W, dn = 64, 128, 32
    cfg = dict(_args(32, 64), network='neuray_gen', init_net_type='cost_volume', use_wrap_padding=True, seq_len=3, height=H, width=W,
               use_disp=True, use_hierarchical_sampling=True, render_depth=True, depth_sample_num=dn, fine_depth_sample_num=dn,
               sample_num=dn, dist_decoder_cfg={'use_vis': False}, fine_dist_decoder_cfg={'use_vis': False})
    cfg.update(min_depth=0.5, max_depth=15.0)
    torch.manual_seed(0)
    net = name2network['neuray_gen'](cfg, d_net=MF.StubMono()).cuda().eval()
    calls = []
    inner = net._run_init_net_sharded
    net.__dict__['_run_init_net_sharded'] = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(2, 3, H, W, generator=g).numpy()
    pose = lambda tx: np.concatenate([np.eye(3), np.array([[tx], [0.0], [0.0]])], 1)
    mk = lambda im: (imgs_info_to_torch(build_imgs_info(im, np.stack([pose(0.5), pose(-0.5)]), [[0.5, 15.0]] * 2), 'cuda'),
                     imgs_info_to_torch(build_imgs_info(im[::-1].copy(), np.stack([pose(-0.5), pose(0.5)]), [[0.5, 15.0]] * 2), 'cuda'))
    ref, src = mk(imgs)
    que = [imgs_info_to_torch(build_render_imgs_info(pose(t), (H, W), [0.5, 15.0]), 'cuda') for t in (0.0, 0.2)]
    frame = lambda q, r, s: net({'que_imgs_info': dict(q), 'ref_imgs_info': dict(r), 'src_imgs_info': dict(s), 'eval': True})
    key = 'pixel_colors_nr_fine'
    with torch.no_grad():
        a0 = frame(que[0], ref, src)
        a1 = frame(que[1], ref, src)                  # same tensor objects: no compare, no MVS stage
        assert len(calls) == 1
        ref2, src2 = mk(imgs)                         # new tensors, equal contents (what the reference's render.py loop builds per frame)
        net.cfg['hip_reuse_source_maps'] = 'equal'    # opt-in: compare contents (one torch.equal = one host sync per tensor)
        a1b = frame(que[1], ref2, src2)
        assert len(calls) == 1 and torch.equal(a1[key], a1b[key])
        net.cfg['hip_reuse_source_maps'] = True       # default: identity + version counter only -> new objects recompute
        ref2b, src2b = mk(imgs)
        a1c = frame(que[1], ref2b, src2b)
        assert len(calls) == 2 and torch.equal(a1[key], a1c[key])
        frame(que[1], ref, src)
        calls.clear(); calls.append(1)
        net.cfg['hip_reuse_source_maps'] = False
        b0, b1 = frame(que[0], ref, src), frame(que[1], ref, src)
        assert len(calls) == 3
        for k in (key, 'render_depth_fine', 'pixel_colors_nr'):
            assert torch.equal(a0[k], b0[k]) and torch.equal(a1[k], b1[k]), k
        assert not torch.equal(a0[key], a1[key])
        net.cfg['hip_reuse_source_maps'] = True
        frame(que[0], ref, src)
        n = len(calls)
        ref['imgs'][0, :, 10:20, 30:50] = 0.25        # in-place write: version counter
        c0 = frame(que[0], ref, src)
        assert len(calls) == n + 1 and not torch.equal(c0[key], a0[key])
        imgs3 = imgs.copy()
        imgs3[0, :, 10:20, 30:50] = 0.25
        ref3, src3 = mk(imgs3)
        src3['imgs'] = src['imgs']                    # sources as in the previous frame, reference = the edited pixels
        net.cfg['hip_reuse_source_maps'] = 'equal'
        c0b = frame(que[0], ref3, src3)               # equal contents again
        assert len(calls) == n + 1 and torch.equal(c0[key], c0b[key])
        src3['trans'] = src3['trans'] * 1.5           # a pose change
        frame(que[0], ref3, src3)
        assert len(calls) == n + 2
        frame(que[0], ref3, src3)
        assert len(calls) == n + 2
        for p in net.init_net.parameters():           # weight update
            p.mul_(1.0)
            break
        frame(que[0], ref3, src3)
        assert len(calls) == n + 3
        d0 = frame(que[1], ref3, src3)
        next(net.image_encoder.parameters()).mul_(1.5)          # encoder weights: MVS result stays, encoders rerun
        d1 = frame(que[1], ref3, src3)
        assert len(calls) == n + 3 and not torch.equal(d0[key], d1[key])
        net.cfg['hip_reuse_source_maps'] = True
    # torch.inference_mode(): inference tensors carry no version counter -> nothing is kept, nothing raises
    with torch.inference_mode():
        ref4, src4 = mk(imgs)
        q4 = imgs_info_to_torch(build_render_imgs_info(pose(0.0), (H, W), [0.5, 15.0]), 'cuda')
        n = len(calls)
        e0 = frame(q4, ref4, src4)
        e1 = frame(q4, ref4, src4)
        assert len(calls) == n + 2 and torch.equal(e0[key], e1[key])
    assert torch.isfinite(e0[key]).all()


@pytest.mark.gpu
def test_hip_init_net_vs_reference():
    """The same stage with its 2-D networks on the HIP library (panogrf_amd/mvs_hip.py: BatchNorm-folded ERP ResNet-18,
    decoders, ResUNetLight, heads) against the reference's outputs and against the torch mirror."""
    from panogrf_amd.mvs import CostVolumeInitNet
    from panogrf_amd.mvs_hip import HipInitNet
    fx = MF.load()
    shapes, ishapes = MF.key_shapes(fx['pipe']['keys']), MF.key_shapes(fx['init']['keys'])
    cfg = dict(_args(), use_wrap_padding=True, seq_len=3, min_depth=0.5, max_depth=15.0)
    inet = CostVolumeInitNet(cfg, d_net=MF.StubMono()).eval()
    inet.load_state_dict({**MF.seeded_state(ishapes, 91), **{'depth_net.' + k: v for k, v in MF.seeded_state(shapes, 61).items()}})
    inet = inet.cuda()
    hip = HipInitNet(inet, 'cuda:0')
    with torch.no_grad():
        f = hip.equi(fx['equi']['x'].cuda()).permute(0, 3, 1, 2)
    assert max_abs(f.cpu(), fx['equi']['out']) < 2e-4 * max(1.0, float(fx['equi']['out'].abs().max()))
    t = fx['init']
    ref = {'imgs': t['ref_imgs'].cuda(), 'rots': t['ref_rots'].cuda(), 'trans': t['ref_trans'].cuda()}
    src = {'imgs': t['src_imgs'].cuda(), 'rots': t['src_rots'].cuda(), 'trans': t['src_trans'].cuda()}
    with torch.no_grad():
        out = hip(ref, src, False)
        tor = inet(ref, src, False)
    assert max_abs(out['mvs_depth'].cpu(), t['mvs_depth']) < 2e-4 * max(1.0, float(t['mvs_depth'].max()))
    assert out['ray_feats'].shape == t['ray_feats'].shape
    assert max_abs(out['ray_feats'].cpu(), t['ray_feats']) < 5e-4 * max(1.0, float(t['ray_feats'].abs().max()))
    assert max_abs(out['ray_feats'], tor['ray_feats']) < 5e-4 * max(1.0, float(t['ray_feats'].abs().max()))


@pytest.mark.gpu
def test_hip_init_net_multiview_vs_reference():
    """seq_len > 3 (the 4-view config): per reference view [ref, src[nn_ids]] through the HIP Equi net and the multi-view
    cost volume; depth against the reference's multiview run, ray features against the torch mirror, and the same through
    the renderer's hipGraph capture (neighbour ids travel as a device tensor)."""
    from panogrf_amd.mvs import CostVolumeInitNet
    from panogrf_amd.mvs_hip import HipInitNet
    from panogrf_amd.renderer import name2network
    fx = MF.load()
    shapes, ishapes = MF.key_shapes(fx['pipe']['keys']), MF.key_shapes(fx['init']['keys'])
    cfg = dict(_args(), use_wrap_padding=True, seq_len=5, min_depth=0.5, max_depth=15.0)
    inet = CostVolumeInitNet(cfg, d_net=MF.StubMono()).eval()
    inet.load_state_dict({**MF.seeded_state(ishapes, 91), **{'depth_net.' + k: v for k, v in MF.seeded_state(shapes, 61).items()}})
    inet = inet.cuda()
    t = fx['mv']
    panos = t['panos'].permute(0, 1, 4, 2, 3).contiguous()            # [1,4,3,H,W], view 0 = reference
    ref = {'imgs': panos[:, 0].cuda(), 'rots': t['rots'][:, 0].cuda(), 'trans': t['trans'][:, 0].cuda(), 'nn_ids': [[0, 1, 2]]}
    src = {'imgs': panos[0, 1:].cuda(), 'rots': t['rots'][0, 1:].cuda(), 'trans': t['trans'][0, 1:].cuda()}
    hip = HipInitNet(inet, 'cuda:0')
    with torch.no_grad():
        out = hip(ref, src, False)
        tor = inet(ref, src, False)
    want = t['depth'].permute(0, 3, 1, 2)
    assert max_abs(out['mvs_depth'].cpu(), want) < 2e-4 * max(1.0, float(want.max()))
    assert max_abs(out['ray_feats'], tor['ray_feats']) < 5e-4 * max(1.0, float(tor['ray_feats'].abs().max()))
    rcfg = dict(cfg, network='neuray_gen', init_net_type='cost_volume', height=32, width=64)
    net = name2network['neuray_gen'](rcfg, init_net=inet).cuda().eval()
    with torch.no_grad():
        for _ in range(2):
            got = net._run_init_net(ref, src, False)
    assert any(k[0] == 'init_net_hip' for k in net._graphs)
    assert max_abs(got['ray_feats'], out['ray_feats']) < 1e-5 * max(1.0, float(tor['ray_feats'].abs().max()))
    assert max_abs(got['mvs_depth'], out['mvs_depth']) < 1e-5 * max(1.0, float(want.max()))


@pytest.mark.gpu
@pytest.mark.parametrize('tag,extra,mono', [('gw', dict(group_wise=True, group_num=8, use_new_reg3dnet=True), 'plain'),
                                            ('rr', dict(revise_range=True, fixed_dist=1.0), 'plain'),
                                            ('mu', dict(mono_uncertainty=True, basic_sigma=0.1, relaxation_factor=1.0), 'uncert')])
def test_unshipped_mvs_switches_vs_reference(tag, extra, mono):
    """group_wise / revise_range / mono_uncertainty (SURVEY.md 8(b) key list; pipeline3_model.py:849-853, :781-796, :718-731)
    against the reference's own estimate_depth_using_cost_volume_v3_erp (tests/golden/make_mvs_flag_fixtures.py)."""
    from panogrf_amd.mvs import MVSDepthNet
    fx = MF.load('mvs_flag_fixtures')
    t, inp = fx[tag], fx['in']
    args = dict(_args(), inplace_abn=False, use_wrap_padding=True, **extra)
    net = MVSDepthNet(args, d_net=MF.StubMonoUncert() if mono == 'uncert' else MF.StubMono()).eval()
    shapes = MF.key_shapes(t['keys'])
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == dict(shapes)       # reference key names and shapes
    net.load_state_dict(MF.seeded_state(shapes, int(t['seed'][0])), strict=True)
    net = net.cuda()
    with torch.no_grad():
        ret = net.estimate_depth_using_cost_volume_v3_erp(inp['panos'].cuda(), inp['rots'].cuda(), inp['trans'].cuda(),
                                                          args['mvs_min_depth'], args['mvs_max_depth'])
    assert max_abs(ret['mono_depth_ref'].cpu(), t['mono_depth_ref']) < 1e-5
    cr = t['cost_reg']
    assert max_abs(ret['cost_reg'].cpu(), cr) < 1e-4 * max(1.0, float(cr.abs().max()))
    assert max_abs(ret['depth'].cpu(), t['depth']) < 1e-3 * max(1.0, float(t['depth'].abs().max()))
