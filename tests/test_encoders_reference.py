"""Per-frame 2-D encoders against STAGE-LEVEL tensors of the reference's own modules
(tests/golden/make_round3_fixtures.py enc_fixture: ResUNetLight network/ops.py:235-455 with forward hooks on bn1, layer1-3,
upconv3/2, iconv3/2, and DefaultVisEncoder network/vis_encoder.py:6-32; seeded weights regenerated from the stored
(name, shape) lists).  CPU: the torch mirror (panogrf_amd/encoders.py) stage by stage.  GPU: the HIP layer sequence."""
import os

import numpy as np
import pytest
import torch

from tests import mvs_fixtures as MF

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(tag):
    z = np.load(os.path.join(GOLDEN, 'enc_fixtures.npz'))
    d = {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + '.')}
    seed = int(d['seed'][0])
    ie_sd = MF.seeded_state(MF.key_shapes(d['ie_keys']), seed + 1)
    ve_sd = MF.seeded_state(MF.key_shapes(d['ve_keys']), seed + 2)
    t = {k: torch.from_numpy(v) for k, v in d.items() if k not in ('seed', 'ie_keys', 've_keys')}
    return t, ie_sd, ve_sd


def _build(ie_sd, ve_sd):
    from panogrf_amd import encoders as EN
    ie, ve = EN.ResUNetLight().eval(), EN.DefaultVisEncoder().eval()
    ie.load_state_dict(ie_sd, strict=True)          # the reference's parameter names load unchanged
    ve.load_state_dict(ve_sd, strict=True)
    return ie, ve


def _rel(a, b):
    return float((a - b).abs().max()) / max(1.0, float(b.abs().max()))


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_torch_mirror_stage_by_stage_vs_reference(tag):
    torch.set_num_threads(1)
    t, ie_sd, ve_sd = _load(tag)
    ie, ve = _build(ie_sd, ve_sd)
    got = {}
    mods = dict(bn1=ie.bn1, layer1=ie.layer1, layer2=ie.layer2, layer3=ie.layer3, upconv3=ie.upconv3, iconv3=ie.iconv3,
                upconv2=ie.upconv2, iconv2=ie.iconv2)
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: got.__setitem__(n, o.detach())) for n, m in mods.items()]
    with torch.no_grad():
        feats = ie(t['imgs'])
        vis = ve(t['ray_feats'], feats)
    for h in hooks:
        h.remove()
    for n in mods:
        assert got[n].shape == t['stage.' + n].shape, n
        assert _rel(got[n], t['stage.' + n]) < 2e-5, (n, _rel(got[n], t['stage.' + n]))
    assert _rel(feats, t['img_feats']) < 2e-5
    assert _rel(vis, t['vis_feats']) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_hip_encoders_vs_reference(tag):
    """HIP layer sequence (pgrf_conv2d / pgrf_instnorm / pgrf_resize_bilinear) against the reference's outputs."""
    from panogrf_amd.encoders_hip import HipEncoders
    t, ie_sd, ve_sd = _load(tag)
    ie, ve = _build(ie_sd, ve_sd)
    ie, ve = ie.cuda(), ve.cuda()
    he = HipEncoders(ie, ve, torch.device('cuda', 0))
    with torch.no_grad():
        f, r = he(t['imgs'].cuda(), t['ray_feats'].cuda())
    assert f.shape == t['img_feats'].shape and r.shape == t['vis_feats'].shape
    assert _rel(f.cpu(), t['img_feats']) < 1e-4, _rel(f.cpu(), t['img_feats'])
    assert _rel(r.cpu(), t['vis_feats']) < 1e-4, _rel(r.cpu(), t['vis_feats'])


@pytest.mark.gpu
def test_hip_encoders_take_parameters_from_any_device_or_dtype():
    """The kernels dereference raw pointers: gamma / beta of a module that is still on the CPU (or in double precision) must
    be copied to fp32 on the compute device, never handed over as they are (a host pointer is a GPU memory fault)."""
    from panogrf_amd.encoders_hip import HipEncoders
    t, ie_sd, ve_sd = _load('a')
    ie, ve = _build(ie_sd, ve_sd)                     # modules stay on the CPU
    he = HipEncoders(ie.double(), ve, torch.device('cuda', 0))
    with torch.no_grad():
        f, r = he(t['imgs'].cuda(), t['ray_feats'].cuda())
    assert _rel(f.cpu(), t['img_feats']) < 1e-4 and _rel(r.cpu(), t['vis_feats']) < 1e-4


@pytest.mark.gpu
def test_instnorm_with_a_large_channel_offset():
    """|mean| >> std (a biased conv output ahead of the norm): E[x^2] - mean^2 on fp32 sums loses most of its digits there;
    the kernel's shifted sums + (x - mean) * rstd must stay at fp32 round-off of a float64 evaluation."""
    from panogrf_amd.encoders_hip import HipEncoders, _Map
    g = torch.Generator().manual_seed(3)
    N, C, H, W = 2, 32, 40, 96
    x = torch.randn(N, C, H, W, generator=g) * 0.37 + torch.linspace(-900.0, 1500.0, C).view(1, C, 1, 1)
    norm = torch.nn.InstanceNorm2d(C, track_running_stats=False, affine=True)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.2 * torch.randn(C, generator=g))
        norm.bias.copy_(0.1 * torch.randn(C, generator=g))
        ref = norm.double()(x.double())
    he = HipEncoders.__new__(HipEncoders)
    he.device = torch.device('cuda', 0)
    he._partial = torch.empty(16, device='cuda')
    out = he._norm(_Map(x.permute(0, 2, 3, 1).contiguous().cuda()), norm.float(), 'none')
    err = float((out.buf.cpu().double() - ref.permute(0, 2, 3, 1)).abs().max())
    assert err < 2e-5, err          # (a single-float mean alone would cost 6e-8 * 1500 / 0.37 = 2.4e-4)
