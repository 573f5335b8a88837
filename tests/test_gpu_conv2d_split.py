"""Depth-1 (2-D) wrap-padded convolutions on the split kernel's three decompositions (csrc/conv3d.hip, round 5: 4 waves per (tile, cout tile)
pair below 512 pairs, 2 from 512, 1 from 2 048) against the oracle's convolution -- shapes on both sides of each threshold, ragged last
tiles, an odd tile count (the second wave pair of the last workgroup has no tile), stride 2, residual.  -m gpu"""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # B, cin, cout, H, W, stride, residual     -> pairs = ceil(B * Ho * Wo / 32) * ceil(cout / 32)
    (1, 32, 32, 8, 36, 1, False),       # 9 pairs: 4 waves per pair
    (2, 64, 64, 33, 124, 1, True),      # 256 tiles x 2 = 512 pairs: 2 waves per pair, ragged last tile
    (1, 32, 64, 65, 130, 1, False),     # 265 tiles (odd) x 2 = 530 pairs: the last workgroup's second pair is empty
    (2, 32, 64, 130, 256, 2, False),    # stride 2 -> 520 tiles x 2 = 1 040 pairs
    (2, 32, 64, 128, 256, 1, True),     # 2 048 tiles x 2 = 4 096 pairs: one wave per pair
    (1, 64, 32, 64, 1024, 1, False),    # 2 048 tiles x 1 = 2 048 pairs: exactly the upper threshold
]


@pytest.mark.parametrize('B,cin,cout,H,W,stride,use_res', CASES)
def test_depth1_convolution_vs_oracle(B, cin, cout, H, W, stride, use_res):
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    g = torch.Generator().manual_seed(B * 1000 + cin + cout + H + W)
    x = torch.randn(B, cin, 1, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5, 'c.bias': torch.randn(cout, generator=g) * 0.1}
    conv = O._conv3(w, 'c', x, stride)
    res = torch.randn(conv.shape, generator=g) if use_res else None
    ref = torch.nn.functional.leaky_relu(conv, 0.01) + (res if use_res else 0)
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    xin = R._Vol(R._to_ndhwc(x.cuda()), 0, cin)
    Do, Ho, Wo = ref.shape[2:]
    resv = R._Vol(res.cuda().permute(0, 2, 3, 4, 1).contiguous(), 0, cout) if use_res else None
    out = R._conv(pc, xin, R._Vol.empty(B, Do, Ho, Wo, cout, 'cuda'), stride=stride, residual=resv)
    got = out.buf.permute(0, 4, 1, 2, 3).cpu()
    err = float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    assert err < 2e-5, err
    # the same call again: bit-identical (fixed summation order per shape)
    out2 = R._conv(pc, xin, R._Vol.empty(B, Do, Ho, Wo, cout, 'cuda'), stride=stride, residual=resv)
    assert torch.equal(out.buf, out2.buf)
