"""HIP 3-D regularisers (wrap-padded implicit-GEMM convs) vs golden vectors from the reference. -m gpu"""
import pytest
import torch

from tests import cv_fixtures as CF
from tests.fixtures import max_abs

pytestmark = pytest.mark.gpu


def test_cost_reg_net_vs_reference():
    from panogrf_amd.cost_reg import CostRegNet
    fx = CF.load()['crn']
    net = CostRegNet(dict(inplace_abn=False, use_wrap_padding=True), in_channels=32).eval()
    sd = CF.seeded_params(CF.key_shapes(fx['keys']), 41)
    assert set(sd) == set(net.state_dict())              # reference key names load unchanged
    net.load_state_dict(sd)
    net.cuda()
    out = net(fx['x'].cuda())
    assert out.shape == fx['out'].shape
    assert max_abs(out.cpu(), fx['out']) < 5e-5 * max(1.0, float(fx['out'].abs().max()))


def test_unet2_vs_reference():
    from panogrf_amd.cost_reg import UNet2Reg
    fx = CF.load()['unet']
    net = UNet2Reg().eval()
    sd = CF.seeded_params(CF.key_shapes(fx['keys']), 51)
    assert set(sd) == set(net.state_dict())
    net.load_state_dict(sd)
    net.cuda()
    out = net(fx['x'].cuda())
    assert out.shape == fx['out'].shape
    assert max_abs(out.cpu(), fx['out']) < 5e-5 * max(1.0, float(fx['out'].abs().max()))


@pytest.mark.parametrize('cin,cout,stride', [(32, 64, 1), (64, 64, 1), (8, 16, 2), (128, 1, 1), (1, 1, 1), (16, 8, 1)])
def test_single_conv_vs_oracle(cin, cout, stride):
    """One wrap-padded conv against the oracle's F.conv3d restatement, incl. ragged tile counts."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    g = torch.Generator().manual_seed(cin * 131 + cout)
    B, D, H, W = 1, 6, 10, 12
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5,
         'c.bias': torch.randn(cout, generator=g) * 0.1}
    ref = torch.nn.functional.leaky_relu(O._conv3(w, 'c', x, stride), 0.01)
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    if cin % 4 == 0:
        xin = R._Vol(R._to_ndhwc(x.cuda()), 0, cin)
    else:
        xin = R._Vol(x.cuda().permute(0, 2, 3, 4, 1).contiguous(), 0, cin)
    Do, Ho, Wo = ref.shape[2:]
    out = R._conv(pc, xin, R._Vol.empty(B, Do, Ho, Wo, cout, 'cuda'), stride=stride)
    got = out.buf.permute(0, 4, 1, 2, 3).cpu()
    assert max_abs(got, ref) < 2e-5 * max(1.0, float(ref.abs().max()))
