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


@pytest.mark.parametrize('D,H,W', [(24, 40, 72), (8, 24, 296)])
def test_cost_reg_net_ragged_volume_vs_oracle(D, H, W):
    """CostRegNet on volumes whose sizes are not multiples of the 16-row kernels' bricks (2 x 4 x 32 voxels; the up-sampling-fused
    brick kernel needs a low-res W >= 33: W = 72 takes it with a ragged last brick, W = 296 with five bricks per row) against the oracle's F.conv3d restatement of
    models/cost_reg.py:99-175: ragged bricks in x and y on every level, the 1/4-resolution layers on groups of 16 output channels,
    the Cout = 1 head from its LDS brick."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd.cost_reg import CostRegNet
    torch.manual_seed(11)
    net = CostRegNet(dict(use_wrap_padding=True), 32).eval()
    gw = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for name, p in net.state_dict().items():
            if p.dim() > 1:
                p.mul_(0.5)
            elif name.endswith('running_var'):
                p.copy_(0.5 + torch.rand(p.shape, generator=gw))
            elif name.endswith('running_mean') or name.endswith('bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=gw))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    x = torch.randn(1, 32, D, H, W, generator=torch.Generator().manual_seed(13))
    with torch.no_grad():
        got = net(x.cuda()).cpu()
        ref = O.cost_reg_net(sd, x)
    assert got.shape == ref.shape
    assert max_abs(got, ref) < 5e-5 * max(1.0, float(ref.abs().max()))


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


@pytest.mark.parametrize('cin,cout,stride', [(32, 64, 1), (64, 64, 1), (8, 16, 2), (128, 1, 1), (1, 1, 1), (16, 8, 1), (8, 1, 1)])
def test_single_conv_vs_oracle(cin, cout, stride, monkeypatch):
    """One wrap-padded conv against the oracle's F.conv3d restatement, incl. ragged tile counts."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    monkeypatch.setenv('PANOGRF_CONV_SMALL16_WIDE', '0')          # the general kernel (the 16-row kernels have their own tests)
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


@pytest.mark.parametrize('cin,cout,stride,D,ns', [(32, 8, 1, 8, 4), (16, 8, 1, 12, 4), (8, 8, 1, 4, 4), (16, 16, 1, 6, 2),
                                                  (32, 16, 1, 6, 2), (8, 16, 2, 6, 1), (16, 32, 2, 7, 1), (16, 8, 1, 6, 2),
                                                  (8, 4, 1, 5, 1)])
def test_narrow_layer_kernel_vs_oracle(cin, cout, stride, D, ns, monkeypatch):
    """pgrf_conv3d_small (taps packed along K, depth shifts packed along M) against F.conv3d of the oracle, with a
    residual, a channel-sliced output and a voxel count that is not a multiple of the tile; the dispatcher must
    pick the depth-shift count the layer shape allows."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    monkeypatch.setenv('PANOGRF_CONV_SMALL16', '0')               # this test is about the 32-row kernel
    g = torch.Generator().manual_seed(cin * 977 + cout * 13 + stride + D)
    B, H, W = 2, 5, 11
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5,
         'c.bias': torch.randn(cout, generator=g) * 0.1}
    conv = O._conv3(w, 'c', x, stride)
    res = torch.randn(conv.shape, generator=g)
    ref = torch.nn.functional.leaky_relu(conv, 0.01) + res
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    xin = R._Vol(R._to_ndhwc(x.cuda()), 0, cin)
    assert R._small_shift(pc, xin, stride) == ns
    Do, Ho, Wo = ref.shape[2:]
    wide = R._Vol.empty(B, Do, Ho, Wo, cout + 8, 'cuda')          # write into channels [4, 4+cout) of a wider buffer
    wide.buf.fill_(7.0)
    resv = R._Vol(res.cuda().permute(0, 2, 3, 4, 1).contiguous(), 0, cout)
    out = R._conv(pc, xin, wide.slice(4, cout), stride=stride, residual=resv)
    got = out.buf[..., 4:4 + cout].permute(0, 4, 1, 2, 3).cpu()
    assert max_abs(got, ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert bool((out.buf[..., :4] == 7.0).all()) and bool((out.buf[..., 4 + cout:] == 7.0).all())


@pytest.mark.parametrize('cin,cout,stride,D,ns,nt', [(32, 8, 1, 8, 2, 0), (16, 8, 1, 12, 2, 0), (32, 8, 1, 8, 2, 8), (16, 8, 1, 12, 2, 4),
                                                     (8, 8, 1, 4, 0, 0), (16, 16, 1, 6, 1, 0), (32, 16, 1, 5, 1, 0), (32, 16, 1, 5, 1, 8),
                                                     (16, 16, 1, 6, 1, 2), (8, 16, 2, 6, 1, 0), (8, 16, 2, 7, 1, 4), (16, 16, 2, 7, 1, 2),
                                                     (32, 8, 1, 5, 0, 0), (16, 32, 1, 4, 1, 0), (64, 64, 1, 3, 1, 0), (32, 32, 1, 6, 1, 0),
                                                     (64, 32, 2, 4, 0, 0), (64, 16, 1, 2, 1, 0)])
def test_narrow_layer_16_row_kernel_vs_oracle(cin, cout, stride, D, ns, nt, monkeypatch):
    """pgrf_conv3d_small16 (v_mfma_f32_16x16x4_f32 tiles: 8 channels x 2 depth shifts, or 16 channels) against F.conv3d of the
    oracle, with a residual, a channel-sliced output, ragged bricks / tiles (H = 5, W = 11: the brick is wider than the panorama),
    odd depths and the zero-padded reduction tail of Cin = 8 / 16.  nt = 0: the dispatcher's choice (LDS-brick kernel for stride 1
    from 16 / 32 channels); nt > 0: the register-path kernel with that many tiles per wave.  ns == 0: the dispatcher must leave
    the layer to the other kernels."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    if nt:
        monkeypatch.setenv('PGRF_CONV16_NT', str(nt))
        monkeypatch.setenv('PGRF_CONV16_NO_BRICK', '1')
    g = torch.Generator().manual_seed(cin * 877 + cout * 17 + stride + D)
    B, H, W = 2, 5, 11
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5,
         'c.bias': torch.randn(cout, generator=g) * 0.1}
    conv = O._conv3(w, 'c', x, stride)
    res = torch.randn(conv.shape, generator=g)
    ref = torch.nn.functional.leaky_relu(conv, 0.01) + res
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    xin = R._Vol(R._to_ndhwc(x.cuda()), 0, cin)
    assert R._small16_shift(pc, xin, stride) == ns
    Do, Ho, Wo = ref.shape[2:]
    resv = R._Vol(res.cuda().permute(0, 2, 3, 4, 1).contiguous(), 0, cout)
    # the unshifted layers on the LDS-brick kernel: both brick shapes (four rows; two rows = what small volumes take by themselves)
    for yb in (('2', '4') if ns == 1 and nt == 0 and stride == 1 and cin >= 16 else ('',)):
        if yb:
            monkeypatch.setenv('PGRF_CONV16_YB', yb)
        wide = R._Vol.empty(B, Do, Ho, Wo, cout + 8, 'cuda')
        wide.buf.fill_(7.0)
        out = R._conv(pc, xin, wide.slice(4, cout), stride=stride, residual=resv)
        got = out.buf[..., 4:4 + cout].permute(0, 4, 1, 2, 3).cpu()
        assert max_abs(got, ref) < 2e-5 * max(1.0, float(ref.abs().max())), yb
        assert bool((out.buf[..., :4] == 7.0).all()) and bool((out.buf[..., 4 + cout:] == 7.0).all())


@pytest.mark.parametrize('cin,cout,ns,D,H,W', [(32, 8, 2, 6, 9, 70), (16, 16, 1, 5, 6, 40), (32, 16, 1, 4, 8, 64), (8, 8, 2, 4, 5, 19)])
def test_narrow_layer_16_row_kernel_c_abi_shapes(cin, cout, ns, D, H, W):
    """The C entry point itself on shapes with several bricks per axis (interior bricks, a ragged last brick in x and y, the
    x wrap across the seam) and the Cin = 8 two-shift instance no dispatcher path reaches."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    from panogrf_amd._lib import lib
    g = torch.Generator().manual_seed(cin + 3 * cout + D * H * W)
    x = torch.randn(1, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5, 'c.bias': torch.randn(cout, generator=g) * 0.1}
    ref = torch.nn.functional.leaky_relu(O._conv3(w, 'c', x, 1), 0.01)
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    wide_in = torch.full((1, D, H, W, cin + 8), float('nan'), device='cuda')          # a channel slice of a wider buffer
    wide_in[..., 4:4 + cin] = R._to_ndhwc(x.cuda())
    xin = R._Vol(wide_in, 4, cin)
    out = R._Vol.empty(1, D, H, W, cout, 'cuda')
    lib.check(lib.pgrf_conv3d_small16(R._ptr(xin.buf), xin.cs, xin.coff, cin, R._ptr(out.buf), out.cs, out.coff, cout, R._ptr(pc.small16(ns)),
                                      1, ns, 1, D, H, W, 1, 0.01, None, 0, 0, R._stream()))
    got = out.buf.permute(0, 4, 1, 2, 3).cpu()
    assert max_abs(got, ref) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_narrow_layer_16_row_kernel_rejects_what_it_cannot_do():
    from panogrf_amd._lib import lib
    import ctypes
    assert lib.pgrf_conv3d_small16_packed_floats(8, 32, 2) == 72 * 256 + 16
    assert lib.pgrf_conv3d_small16_packed_floats(16, 8, 1) == 14 * 256 + 16          # 13.5 groups padded to a pair
    assert lib.pgrf_conv3d_small16_packed_floats(8, 32, 4) == 0 and lib.pgrf_conv3d_small16_packed_floats(24, 32, 1) == 0
    assert lib.pgrf_conv3d_small16_packed_floats(32, 64, 1) == 2 * 108 * 256 + 32          # two groups of 16 rows
    buf = (ctypes.c_float * 16)()
    assert lib.pgrf_pack_conv3d_small16(ctypes.cast(buf, ctypes.c_void_p), None, 16, 16, 2, ctypes.cast(buf, ctypes.c_void_p)) != 0
    assert b'pgrf_pack_conv3d_small16' in lib.pgrf_last_error()


@pytest.mark.parametrize('B,D,H,W,res', [(2, 5, 19, 45, True), (1, 2, 16, 32, False), (1, 7, 3, 70, True)])
def test_single_channel_head_from_lds_brick_vs_oracle(B, D, H, W, res, monkeypatch):
    """The Cout = 1 head from eight channels (conv3d_cout1_c8_brick_kernel: 4 x 18 x 34 input brick in LDS, scalar-loaded weights)
    against the oracle's F.conv3d: ragged bricks in every axis, odd depths, a batch, the residual epilogue, a channel-sliced input;
    and BIT-IDENTICAL to the kernel it replaces (PGRF_CONV_NO_C8_BRICK): both run the same fmaf chain per output, in (dz, dy, dx, channel) order."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    g = torch.Generator().manual_seed(B * 1000 + D * 100 + H + W)
    x = torch.randn(B, 8, D, H, W, generator=g)
    w = {'c.weight': torch.randn(1, 8, 3, 3, 3, generator=g) * (2.0 / 216) ** 0.5, 'c.bias': torch.randn(1, generator=g) * 0.1}
    conv = O._conv3(w, 'c', x, 1)
    r = torch.randn(conv.shape, generator=g) if res else None
    ref = torch.nn.functional.leaky_relu(conv, 0.01) + (r if res else 0.0)
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    wide_in = torch.full((B, D, H, W, 16), float('nan'), device='cuda')
    wide_in[..., 4:12] = R._to_ndhwc(x.cuda())
    resv = R._Vol(r.cuda().permute(0, 2, 3, 4, 1).contiguous(), 0, 1) if res else None
    outs = []
    for flag in (None, '1'):
        if flag:
            monkeypatch.setenv('PGRF_CONV_NO_C8_BRICK', flag)
        out = R._conv(pc, R._Vol(wide_in, 4, 8), R._Vol.empty(B, D, H, W, 1, 'cuda'), slope=0.01, residual=resv)
        outs.append(out.buf.permute(0, 4, 1, 2, 3).cpu())
        assert max_abs(outs[-1], ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(outs[0], outs[1])


def test_narrow_layer_kernel_rejects_what_it_cannot_do():
    from panogrf_amd._lib import lib
    assert lib.pgrf_conv3d_small_packed_floats(8, 16, 4) == ((6 * 9 * 16 + 31) // 32) * 1024 + 32
    import ctypes
    buf = (ctypes.c_float * 16)()
    assert lib.pgrf_pack_conv3d_small(ctypes.cast(buf, ctypes.c_void_p), None, 16, 16, 4, ctypes.cast(buf, ctypes.c_void_p)) != 0
    assert b'pgrf_pack_conv3d_small' in lib.pgrf_last_error()


@pytest.mark.parametrize('cin,cout,nt,split', [(32, 64, 4, 0), (64, 64, 4, 0), (64, 32, 2, 0), (96, 64, 1, 0), (64, 64, 1, 1),
                                               (128, 128, 4, 0)])
def test_interior_wave_pipeline_vs_oracle(cin, cout, nt, split, monkeypatch):
    """Waves whose tiles lie strictly inside the volume run the in-place fused pipeline (no padding logic, operand
    refills between the MFMA groups); the faces of the same volume run the masked loop.  W = 128 makes a wave's
    NT*32 voxels whole or partial rows, so both kinds of wave occur; tile shape and split-K are forced through the
    tuning knobs so that every instance of the kernel is covered.  Checked against F.conv3d of the oracle."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    monkeypatch.setenv('PGRF_CONV_NT', str(nt))
    monkeypatch.setenv('PGRF_CONV_SPLIT', str(split))
    monkeypatch.setenv('PANOGRF_CONV_SMALL16_WIDE', '0')          # the general kernel
    g = torch.Generator().manual_seed(cin * 7 + cout * 3 + nt)
    B, D, H, W = 1, 5, 6, 128
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5,
         'c.bias': torch.randn(cout, generator=g) * 0.1}
    ref = torch.nn.functional.leaky_relu(O._conv3(w, 'c', x, 1), 0.01)
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    out = R._conv(pc, R._Vol(R._to_ndhwc(x.cuda()), 0, cin), R._Vol.empty(B, D, H, W, cout, 'cuda'), stride=1)
    got = out.buf.permute(0, 4, 1, 2, 3).cpu()
    assert max_abs(got, ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    # the interior planes/rows (fused pipeline) separately from the faces
    assert max_abs(got[:, :, 1:-1, 1:-1], ref[:, :, 1:-1, 1:-1]) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('cin,cout,B,D,H,W,bias,slope', [(32, 64, 1, 4, 4, 64, True, 0.01), (64, 64, 2, 4, 6, 128, True, 0.01),
                                                         (128, 64, 1, 2, 4, 64, False, 1.0), (64, 128, 1, 6, 2, 64, True, 0.2),
                                                         (32, 64, 1, 8, 8, 128, True, 0.01), (96, 192, 1, 2, 2, 64, True, 0.01),
                                                         (32, 64, 1, 4, 4, 32, True, 0.01), (64, 128, 2, 6, 8, 96, True, 0.01),
                                                         (128, 64, 1, 2, 12, 32, False, 0.2), (32, 64, 1, 4, 4, 16, True, 0.01)])
def test_winograd_conv_vs_oracle(cin, cout, B, D, H, W, bias, slope, monkeypatch):
    """pgrf_conv3d_wino (Winograd F(2x2x2, 3x3x3), csrc/conv3d_wino.hip) against F.conv3d of the oracle: volume faces
    (zero padding along D and H), the circular W seam, interior rows (D = H = 8), several channel blocks and cout groups,
    more rows than one workgroup pass, a channel-sliced input and output.  The same tolerance as the direct kernel: the
    transforms only add ~1e-6 relative round-off."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    from panogrf_amd._lib import lib
    monkeypatch.setenv('PANOGRF_CONV_WINOGRAD', 'force')         # also for layers with too few rows to be preferred
    monkeypatch.setenv('PGRF_WINO_WGS', '3')                     # few workgroups: every one loops over several rows
    assert lib.pgrf_conv3d_wino16_supported(cin, cout, D, H, W) == 1
    g = torch.Generator().manual_seed(cin * 31 + cout * 7 + D + H + W)
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5}
    if bias:
        w['c.bias'] = torch.randn(cout, generator=g) * 0.1
    ref = torch.nn.functional.leaky_relu(O._conv3(w, 'c', x, 1), slope)
    pc = R._PackedConv(w['c.weight'], w.get('c.bias'), 'cuda')
    wide_in = torch.full((B, D, H, W, cin + 8), 3.0, device='cuda')
    wide_in[..., 4:4 + cin] = R._to_ndhwc(x.cuda())
    wide_out = R._Vol.empty(B, D, H, W, cout + 8, 'cuda')
    wide_out.buf.fill_(7.0)
    out = R._conv(pc, R._Vol(wide_in, 4, cin), wide_out.slice(4, cout), stride=1, slope=slope)
    got = out.buf[..., 4:4 + cout].permute(0, 4, 1, 2, 3).cpu()
    assert max_abs(got, ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert bool((out.buf[..., :4] == 7.0).all()) and bool((out.buf[..., 4 + cout:] == 7.0).all())
    # and it really was the Winograd kernel: the direct one gives (slightly) different round-off
    monkeypatch.setenv('PANOGRF_CONV_WINOGRAD', '0')
    direct = R._conv(pc, R._Vol(wide_in, 4, cin), R._Vol.empty(B, D, H, W, cout, 'cuda'), stride=1, slope=slope)
    d = max_abs(direct.buf.permute(0, 4, 1, 2, 3).cpu(), got)
    assert 0.0 < d < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('cin,cout,B,D,H,W,fused', [(32, 64, 1, 4, 4, 64, True), (64, 128, 2, 6, 8, 96, True), (64, 64, 1, 8, 8, 128, True),
                                                    (32, 64, 1, 4, 4, 64, False)])
def test_winograd_conv_with_pooled_output(cin, cout, B, D, H, W, fused, monkeypatch):
    """pgrf_conv3d_wino16_pool: the conv's output AND nn.AvgPool3d(2) of it from the same launch (UNet2's encoder levels,
    models/common_blocks.py:210-218 -- an F(2^3, 3^3) output tile is one pooling window), into a channel slice of a wider pooled buffer;
    and the same call where the Winograd kernel does not take the layer (conv + pgrf_avgpool3d_2)."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    monkeypatch.setenv('PANOGRF_CONV_WINOGRAD', 'force' if fused else '0')
    monkeypatch.setenv('PGRF_WINO16_WGS', '3')
    g = torch.Generator().manual_seed(cin + cout + D * H + W)
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5, 'c.bias': torch.randn(cout, generator=g) * 0.1}
    ref = torch.nn.functional.leaky_relu(O._conv3(w, 'c', x, 1), 0.01)
    ref_pool = torch.nn.functional.avg_pool3d(ref, 2)
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    xin = R._Vol(R._to_ndhwc(x.cuda()), 0, cin)
    assert R._conv_wino16_applies(pc, xin, 1, None) == fused
    out = R._Vol.empty(B, D, H, W, cout, 'cuda')
    wide_pool = R._Vol.empty(B, D // 2, H // 2, W // 2, cout + 8, 'cuda')
    wide_pool.buf.fill_(7.0)
    R._conv(pc, xin, out, pool_out=wide_pool.slice(4, cout))
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert max_abs(out.buf.permute(0, 4, 1, 2, 3).cpu(), ref) < tol
    assert max_abs(wide_pool.buf[..., 4:4 + cout].permute(0, 4, 1, 2, 3).cpu(), ref_pool) < tol
    assert bool((wide_pool.buf[..., :4] == 7.0).all()) and bool((wide_pool.buf[..., 4 + cout:] == 7.0).all())
    # without a pooled output the conv writes the same values
    out2 = R._conv(pc, xin, R._Vol.empty(B, D, H, W, cout, 'cuda'))
    assert torch.equal(out2.buf, out.buf)


@pytest.mark.parametrize('cin,cout,B,D,H,W', [(64, 128, 1, 4, 4, 32), (512, 128, 1, 16, 16, 32), (256, 512, 1, 8, 16, 32), (96, 128, 2, 2, 4, 64),
                                              (128, 128, 1, 6, 2, 32), (256, 512, 1, 8, 8, 16), (64, 128, 2, 4, 6, 16)])
def test_winograd_split_k_vs_oracle(cin, cout, B, D, H, W, monkeypatch):
    """pgrf_conv3d_wino16_splitk: layers with few rows (the low-resolution levels of UNet2) run every row as several work items over shares
    of the 32-channel input blocks, partial sums reduced with bias + LeakyReLU behind.  Same tolerance as the unsplit kernel; the host rule
    (pgrf_conv3d_wino16_ksplit) must pick a split for these shapes, and every split up to the block count must give the same result."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    from panogrf_amd._lib import lib
    g = torch.Generator().manual_seed(cin + 3 * cout + D + H)
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5, 'c.bias': torch.randn(cout, generator=g) * 0.1}
    ref = torch.nn.functional.leaky_relu(O._conv3(w, 'c', x, 1), 0.01)
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    xin = R._Vol(R._to_ndhwc(x.cuda()), 0, cin)
    ks = lib.pgrf_conv3d_wino16_ksplit(cin, cout, B, D, H, W)
    assert 2 <= ks <= cin // 32 and R._wino16_ksplit(pc, xin, 1, None) == ks
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    out = R._conv(pc, xin, R._Vol.empty(B, D, H, W, cout, 'cuda'))                    # the host's own choice
    assert max_abs(out.buf.permute(0, 4, 1, 2, 3).cpu(), ref) < tol
    for s_ in sorted({2, cin // 32}):                                                  # the extremes, into a channel slice of a wider tensor
        wide = R._Vol.empty(B, D, H, W, cout + 8, 'cuda')
        wide.buf.fill_(7.0)
        scratch = torch.empty(s_ * B * D * H * W * cout, device='cuda')
        lib.check(lib.pgrf_conv3d_wino16_splitk(R._ptr(xin.buf), xin.cs, xin.coff, cin, R._ptr(wide.buf), wide.cs, 4, cout, R._ptr(pc.wino16()), 1,
                                                B, D, H, W, 0.01, s_, R._ptr(scratch), scratch.numel(), R._stream()))
        assert max_abs(wide.buf[..., 4:4 + cout].permute(0, 4, 1, 2, 3).cpu(), ref) < tol
        assert bool((wide.buf[..., :4] == 7.0).all()) and bool((wide.buf[..., 4 + cout:] == 7.0).all())
    assert lib.pgrf_conv3d_wino16_splitk(R._ptr(xin.buf), xin.cs, xin.coff, cin, R._ptr(out.buf), out.cs, 0, cout, R._ptr(pc.wino16()), 1,
                                         B, D, H, W, 0.01, cin // 32 + 1, R._ptr(out.buf), 10 ** 9, None) != 0          # more items than blocks


def test_winograd_conv_rejects_what_it_cannot_do():
    import ctypes
    from panogrf_amd._lib import lib
    sup = lib.pgrf_conv3d_wino_supported
    assert sup(64, 64, 4, 4, 64) == 1
    assert sup(48, 64, 4, 4, 64) == 0 and sup(64, 32, 4, 4, 64) == 0          # Cin % 32, Cout % 64
    assert sup(64, 64, 5, 4, 64) == 0 and sup(64, 64, 4, 3, 64) == 0 and sup(64, 64, 4, 4, 48) == 0    # odd D / H, W % 32
    assert sup(64, 64, 4, 4, 96) == 1 and sup(64, 64, 4, 6, 96) == 0 and sup(64, 64, 4, 6, 128) == 1   # 16 x 2 tile rows need H % 4 == 0
    x = torch.zeros(1, 5, 4, 64, 64, device='cuda')
    p = ctypes.c_void_p(x.data_ptr())
    assert lib.pgrf_conv3d_wino(p, 64, 0, 64, p, 64, 0, 64, p, 0, 1, 5, 4, 64, 0.01, None, 0, 0, None) != 0
    assert b'pgrf_conv3d_wino' in lib.pgrf_last_error()
    x = torch.zeros(1, 4, 4, 64, 64, device='cuda')
    p = ctypes.c_void_p(x.data_ptr())
    assert lib.pgrf_conv3d_wino(p, 64, 0, 64, p, 64, 0, 64, p, 0, 1, 4, 4, 64, 0.01, p, 64, 0, None) != 0     # residual
    assert b'residual' in lib.pgrf_last_error()
    assert lib.pgrf_conv3d_wino_packed_floats(64, 32) == 2 * 64 * 1 * 1024 + 64
    pref = lib.pgrf_conv3d_wino_preferred
    assert pref(64, 64, 1, 64, 128, 256) == 1 and pref(512, 128, 1, 16, 16, 32) == 0 and pref(48, 64, 1, 64, 128, 256) == 0


@pytest.mark.parametrize('B,D,H,W,C', [(1, 4, 6, 8, 64), (2, 2, 2, 66, 64), (1, 6, 4, 34, 96)])
def test_single_channel_conv_over_upsampled_concat_vs_oracle(B, D, H, W, C):
    """UNet2's last decoder step, conv(cat((up(x), skip))) with one output channel (models/common_blocks.py:225-241), as two
    projections onto the 27 tap vectors -- x at LOW resolution, the skip half at full resolution -- and
    pgrf_conv3d_cout1_gather_up, which interpolates the low-resolution planes inside the gather: against the oracle's
    F.interpolate + torch.cat + F.conv3d (volume faces, the W seam, axes of length 1 at low resolution)."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    from panogrf_amd._lib import lib
    g = torch.Generator().manual_seed(B + D * 7 + H * 3 + W + C)
    x = torch.randn(B, C, D // 2, H // 2, W // 2, generator=g)
    skip = torch.randn(B, C, D, H, W, generator=g)
    w = {'c.weight': torch.randn(1, 2 * C, 3, 3, 3, generator=g) * (2.0 / (54 * C)) ** 0.5, 'c.bias': torch.randn(1, generator=g) * 0.1}
    up = torch.nn.functional.interpolate(x, scale_factor=2, mode='trilinear', align_corners=False)
    ref = torch.nn.functional.leaky_relu(O._conv3(w, 'c', torch.cat((up, skip), 1)), 0.01)
    pu = R._PackedConv(w['c.weight'][:, :C], None, 'cuda')
    ps = R._PackedConv(w['c.weight'][:, C:], w['c.bias'], 'cuda')
    xl = R._Vol(R._to_ndhwc(x.cuda()), 0, C)
    sk = R._Vol(R._to_ndhwc(skip.cuda()), 0, C)
    y_lo = torch.empty(B, D // 2, H // 2, W // 2, 32, device='cuda')
    y_hi = torch.empty(B, D, H, W, 32, device='cuda')
    lib.check(lib.pgrf_conv3d_cout1_project(R._ptr(xl.buf), xl.cs, xl.coff, C, R._ptr(pu.dev), R._ptr(y_lo), B, D // 2, H // 2, W // 2, R._stream()))
    lib.check(lib.pgrf_conv3d_cout1_project(R._ptr(sk.buf), sk.cs, sk.coff, C, R._ptr(ps.dev), R._ptr(y_hi), B, D, H, W, R._stream()))
    out = torch.empty(B, D, H, W, 1, device='cuda')
    lib.check(lib.pgrf_conv3d_cout1_gather_up(R._ptr(y_hi), R._ptr(y_lo), R._ptr(ps.dev), C, 1, R._ptr(out), 1, 0, B, D, H, W, 0.01, None, 0, 0,
                                              R._stream()))
    got = out.permute(0, 4, 1, 2, 3).cpu()
    assert max_abs(got, ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert lib.pgrf_conv3d_cout1_gather_up(R._ptr(y_hi), R._ptr(y_lo), R._ptr(ps.dev), C, 1, R._ptr(out), 1, 0, B, D + 1, H, W, 0.01, None, 0, 0, None) != 0


def test_unet2_fused_head_equals_unfused(monkeypatch):
    """UNet2 with its last decoder step through pgrf_conv3d_cout1_gather_up (default) against the up-sample + concat + convolution
    form of the same step: a different order of the same sums."""
    from panogrf_amd.cost_reg import UNet2Reg
    torch.manual_seed(5)
    net = UNet2Reg().eval().cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
    x = torch.randn(2, 32, 8, 16, 64, generator=torch.Generator().manual_seed(6)).cuda()
    outs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('PANOGRF_UNET_HEAD_FUSE', flag)
        with torch.no_grad():
            outs.append(net(x).clone())
    d = max_abs(outs[0], outs[1])
    assert 0.0 < d < 2e-5 * max(1.0, float(outs[1].abs().max()))


def test_unet2_winograd_equals_direct(monkeypatch):
    """The whole UNet2 regulariser with its wide layers on the Winograd kernel vs all layers on the direct kernel (a volume
    wide enough that the first two resolution levels qualify)."""
    from panogrf_amd.cost_reg import UNet2Reg
    torch.manual_seed(3)
    net = UNet2Reg().eval().cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
    x = torch.randn(1, 32, 16, 16, 128, generator=torch.Generator().manual_seed(4)).cuda()
    outs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('PANOGRF_CONV_WINOGRAD', flag)
        with torch.no_grad():
            outs.append(net(x).clone())
    d = max_abs(outs[0], outs[1])
    assert 0.0 < d < 1e-4 * max(1.0, float(outs[1].abs().max()))


@pytest.mark.parametrize('B,D,H,W', [(1, 4, 6, 40), (2, 3, 5, 33), (1, 2, 2, 49), (1, 5, 9, 70), (1, 3, 4, 64), (1, 2, 3, 48)])
def test_upsample_fused_brick_conv_vs_oracle(B, D, H, W, monkeypatch):
    """pgrf_conv3d_upbrick (16 -> 8: the low-res brick in LDS, z / y position classes per brick / wave, interior x weights with
    the corrected operand of the two seam columns) against the oracle's F.interpolate + F.conv3d with the LeakyReLU + residual
    epilogue and channel-sliced buffers: ragged bricks in x and y, several bricks per axis, first / interior / last planes and
    rows, the right seam in either tile of its brick (W % 32 = 1, 8, 16, 17, 0), axes of length 2.  And against the class-box
    kernel it replaces (a different evaluation order: close, not equal)."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    cin, cout = 16, 8
    g = torch.Generator().manual_seed(D * 3 + H * 5 + W + B)
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5, 'c.bias': torch.randn(cout, generator=g) * 0.1}
    up = torch.nn.functional.interpolate(x, scale_factor=2, mode='trilinear', align_corners=False)
    conv = O._conv3(w, 'c', up)
    res = torch.randn(conv.shape, generator=g)
    ref = torch.nn.functional.leaky_relu(conv, 0.01) + res
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    wide_in = torch.full((B, D, H, W, cin + 8), 3.0, device='cuda')
    wide_in[..., 4:4 + cin] = R._to_ndhwc(x.cuda())
    resv = R._Vol(res.cuda().permute(0, 2, 3, 4, 1).contiguous(), 0, cout)
    outs = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('PANOGRF_CONV_UPBRICK', flag)
        wide_out = R._Vol.empty(B, 2 * D, 2 * H, 2 * W, cout + 8, 'cuda')
        wide_out.buf.fill_(7.0)
        o = R._upconv(pc, R._Vol(wide_in, 4, cin), wide_out.slice(4, cout), lambda: R._Vol.empty(B, 2 * D, 2 * H, 2 * W, cin, 'cuda'), residual=resv)
        outs[flag] = o.buf[..., 4:4 + cout].permute(0, 4, 1, 2, 3).cpu()
        assert bool((o.buf[..., :4] == 7.0).all()) and bool((o.buf[..., 4 + cout:] == 7.0).all())
        assert max_abs(outs[flag], ref) < 2e-5 * max(1.0, float(ref.abs().max())), flag
    assert 0.0 < max_abs(outs['1'], outs['0']) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert ('upb', D, H, W) in pc._small and ('up', D, H, W) in pc._small          # both kernels really ran


def test_upsample_fused_brick_conv_rejects_what_it_cannot_do():
    from panogrf_amd._lib import lib
    import ctypes
    assert lib.pgrf_conv3d_upbrick_packed_floats(8, 16) == 9 * 4 * 27 * 256 + 16
    assert lib.pgrf_conv3d_upbrick_packed_floats(16, 16) == 0 and lib.pgrf_conv3d_upbrick_packed_floats(8, 32) == 0
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.pgrf_pack_conv3d_upbrick(p, None, 8, 16, 4, 4, 32, p) != 0 and b'pgrf_pack_conv3d_upbrick' in lib.pgrf_last_error()
    assert lib.pgrf_conv3d_upbrick(p, 16, 0, 16, p, 8, 0, 8, p, 1, 1, 4, 4, 32, 0.01, None, 0, 0, None) != 0 and b'W >= 33' in lib.pgrf_last_error()


@pytest.mark.parametrize('cin,cout,B,D,H,W', [(16, 8, 1, 4, 6, 40), (32, 16, 2, 3, 5, 33), (16, 8, 1, 2, 2, 2), (16, 16, 1, 6, 4, 64),
                                              (32, 8, 1, 5, 3, 7)])
def test_upsample_fused_conv_vs_oracle(cin, cout, B, D, H, W, monkeypatch):
    """pgrf_conv3d_upfused (trilinear x2 folded into the 3x3x3 conv, 27 position classes) against the oracle's
    F.interpolate + F.conv3d, with the LeakyReLU + residual epilogue and channel-sliced buffers; ragged class boxes, several
    workgroups per box, the W seam and axes of length 2.  And against the unfused two-launch path."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_reg as R
    g = torch.Generator().manual_seed(cin * 17 + cout + D * 3 + H * 5 + W)
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5, 'c.bias': torch.randn(cout, generator=g) * 0.1}
    up = torch.nn.functional.interpolate(x, scale_factor=2, mode='trilinear', align_corners=False)
    conv = O._conv3(w, 'c', up)
    res = torch.randn(conv.shape, generator=g)
    ref = torch.nn.functional.leaky_relu(conv, 0.01) + res
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    wide_in = torch.full((B, D, H, W, cin + 8), 3.0, device='cuda')
    wide_in[..., 4:4 + cin] = R._to_ndhwc(x.cuda())
    resv = R._Vol(res.cuda().permute(0, 2, 3, 4, 1).contiguous(), 0, cout)
    outs = {}
    monkeypatch.setenv('PANOGRF_CONV_UPBRICK', '0')               # this test is about the class-box kernel
    for flag in ('force', '0'):
        monkeypatch.setenv('PANOGRF_CONV_UPFUSE', flag)
        wide_out = R._Vol.empty(B, 2 * D, 2 * H, 2 * W, cout + 8, 'cuda')
        wide_out.buf.fill_(7.0)
        o = R._upconv(pc, R._Vol(wide_in, 4, cin), wide_out.slice(4, cout), lambda: R._Vol.empty(B, 2 * D, 2 * H, 2 * W, cin, 'cuda'), residual=resv)
        outs[flag] = o.buf[..., 4:4 + cout].permute(0, 4, 1, 2, 3).cpu()
        assert bool((o.buf[..., :4] == 7.0).all()) and bool((o.buf[..., 4 + cout:] == 7.0).all())
        assert max_abs(outs[flag], ref) < 2e-5 * max(1.0, float(ref.abs().max())), flag
    assert 0.0 < max_abs(outs['force'], outs['0']) < 2e-5 * max(1.0, float(ref.abs().max()))        # two different evaluations
