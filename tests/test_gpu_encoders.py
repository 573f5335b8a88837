"""HIP per-frame 2-D encoders against the same layers on PyTorch-ROCm: the torch modules of
panogrf_amd.encoders are the parameter holders AND the comparison (they are pinned to the reference by
tests/test_renderer_module.py::test_render_end_to_end_vs_reference).  -m gpu"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    return float((a - b).abs().max()) / max(1.0, float(b.abs().max()))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('cin,cout,k,stride,wrap,bias', [(32, 32, 3, 1, True, False), (16, 32, 3, 2, True, False),
                                                         (64, 128, 3, 2, True, False), (128, 64, 3, 1, False, True),
                                                         (128, 64, 3, 1, True, True), (16, 32, 1, 2, True, False),
                                                         (32, 32, 1, 1, True, True), (64, 32, 3, 1, True, False)])
def test_conv2d_vs_torch(cin, cout, k, stride, wrap, bias):
    from panogrf_amd import encoders as EN
    from panogrf_amd.encoders_hip import HipEncoders, _Map, _Conv
    g = torch.Generator().manual_seed(cin + cout * 3 + k)
    N, H, W = 2, 12, 40
    x = torch.randn(N, cin, H, W, generator=g).cuda()
    conv = torch.nn.Conv2d(cin, cout, k, stride, 0, bias=bias).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (k * k * cin)) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        xp = x
        if k == 3:
            xp = EN.ErpPad(1)(x) if wrap else F.pad(x, (1, 1, 1, 1))
        ref = conv(xp)
    he = HipEncoders.__new__(HipEncoders)
    he.device = torch.device('cuda', 0)
    pc = _Conv(conv, he.device)
    Ho, Wo = ref.shape[-2:]
    res = torch.randn(N, Ho, Wo, cout, generator=g).cuda()
    out = he._conv(pc, _Map(_nhwc(x)), _Map.empty(N, Ho, Wo, cout, he.device), wrap=wrap, residual=_Map(res))
    assert _relerr(out.buf, _nhwc(ref) + res) < 2e-5


def test_stem_conv_vs_torch():
    from panogrf_amd import encoders as EN
    from panogrf_amd._lib import lib
    from panogrf_amd.engine import _ptr, _stream
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 32, 64, generator=g).cuda()
    stem = EN.erp_conv(3, 16, 7, 2).cuda()
    with torch.no_grad():
        ref = stem(x)
    out = torch.empty(2, 16, 32, 16, device='cuda')
    lib.check(lib.pgrf_conv2d_direct(_ptr(x), 2, 3, 32, 64, _ptr(stem[1].weight.detach().contiguous()), None, 16, 7, 2, 1, 0, _ptr(out), 16,
                                     0, _stream()))
    assert _relerr(out, _nhwc(ref)) < 1e-5


@pytest.mark.parametrize('C,act,with_res', [(16, 'relu', False), (32, 'relu', True), (64, 'elu', False), (128, 'none', False)])
def test_instnorm_vs_torch(C, act, with_res):
    from panogrf_amd.encoders_hip import HipEncoders, _Map
    g = torch.Generator().manual_seed(C)
    N, H, W = 2, 24, 96                        # 2304 pixels: more than one statistics chunk
    x = (torch.randn(N, C, H, W, generator=g) * 2.0 + 0.7).cuda()
    norm = torch.nn.InstanceNorm2d(C, track_running_stats=False, affine=True).cuda()
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.2 * torch.randn(C, generator=g))
        norm.bias.copy_(0.1 * torch.randn(C, generator=g))
        res = torch.randn(N, C, H, W, generator=g).cuda() if with_res else None
        y = norm(x) + (res if with_res else 0.0)
        ref = {'relu': F.relu, 'elu': F.elu, 'none': lambda t: t}[act](y)
    he = HipEncoders.__new__(HipEncoders)
    he.device = torch.device('cuda', 0)
    he._partial = torch.empty(16, device='cuda')
    wide = torch.full((N, H, W, C + 8), 7.0, device='cuda')          # output into a channel slice of a wider buffer
    out = he._norm(_Map(_nhwc(x)), norm, act, out=_Map(wide, 4, C), residual=_Map(_nhwc(res)) if with_res else None)
    assert _relerr(wide[..., 4:4 + C], _nhwc(ref)) < 2e-5
    assert bool((wide[..., :4] == 7.0).all()) and bool((wide[..., 4 + C:] == 7.0).all())


@pytest.mark.parametrize('Ho,Wo,align', [(16, 40, True), (4, 10, False), (16, 40, False), (11, 17, True)])
def test_resize_vs_torch(Ho, Wo, align):
    from panogrf_amd.encoders_hip import HipEncoders, _Map
    g = torch.Generator().manual_seed(Ho)
    x = torch.randn(2, 32, 8, 20, generator=g).cuda()
    ref = F.interpolate(x, (Ho, Wo), mode='bilinear', align_corners=align)
    he = HipEncoders.__new__(HipEncoders)
    he.device = torch.device('cuda', 0)
    out = he._resize(_Map(_nhwc(x)), _Map.empty(2, Ho, Wo, 32, he.device), align)
    assert _relerr(out.buf, _nhwc(ref)) < 1e-5


@pytest.mark.parametrize('H,W', [(64, 128), (96, 160)])
def test_encoders_end_to_end_vs_torch(H, W):
    """Whole ResUNetLight + DefaultVisEncoder: HIP layer sequence vs the torch modules holding the same parameters."""
    from panogrf_amd import encoders as EN
    from panogrf_amd.encoders_hip import HipEncoders
    torch.manual_seed(3)
    ie, ve = EN.ResUNetLight().cuda().eval(), EN.DefaultVisEncoder().cuda().eval()
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for m in list(ie.modules()) + list(ve.modules()):               # non-trivial affine parameters of the norms
            if isinstance(m, torch.nn.InstanceNorm2d):
                m.weight.copy_((1.0 + 0.2 * torch.randn(m.weight.shape, generator=g)).cuda())
                m.bias.copy_((0.1 * torch.randn(m.bias.shape, generator=g)).cuda())
        imgs = torch.rand(2, 3, H, W, generator=g).cuda()
        rf = torch.randn(2, 32, H // 8, W // 8, generator=g).cuda()
        f_ref = ie(imgs)
        r_ref = ve(rf, f_ref)
        he = HipEncoders(ie, ve, imgs.device)
        f, r = he(imgs, rf)
    assert f.shape == f_ref.shape and r.shape == r_ref.shape
    assert _relerr(f, f_ref) < 2e-4
    assert _relerr(r, r_ref) < 2e-4
    # a parameter update is picked up (packed weights are re-made)
    with torch.no_grad():
        ie.out_conv.weight.mul_(2.0)
        f2, _ = he(imgs, rf)
        assert _relerr(f2, ie(imgs)) < 2e-4
    # shapes the layer sequence does not cover are refused, not approximated
    from panogrf_amd import PgrfError
    with pytest.raises(PgrfError):
        he.image_encoder(torch.rand(1, 3, 40, 72).cuda())
