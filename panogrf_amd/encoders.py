"""Per-frame 2-D encoders of the renderer, kept on stock PyTorch-ROCm (MIOpen) as SURVEY.md section 8
f-1 prescribes: they run once per frame, not per ray.  Module/parameter names mirror
network/ops.py:235-455 (ResUNetLight) and network/vis_encoder.py:6-32 (DefaultVisEncoder) so reference
checkpoints load unchanged; the code is written against those interfaces, not copied.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ErpPad(nn.Module):
    """zeros top/bottom, circular left/right (models/common_blocks.py:258-292)."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0:
            return x
        x = F.pad(x, (0, 0, self.p, self.p))
        return F.pad(x, (self.p, self.p, 0, 0), mode='circular')


def erp_conv(cin, cout, k, stride=1, bias=False):
    return nn.Sequential(ErpPad(k // 2), nn.Conv2d(cin, cout, k, stride, 0, bias=bias))


def _inorm(c):
    return nn.InstanceNorm2d(c, track_running_stats=False, affine=True)


class ResBlock2d(nn.Module):
    """BasicBlock (network/ops.py:127-196) with InstanceNorm and ERP padding."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = erp_conv(cin, cout, 3, stride)
        self.bn1 = _inorm(cout)
        self.conv2 = erp_conv(cout, cout, 3)
        self.bn2 = _inorm(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), _inorm(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        s = x if self.downsample is None else self.downsample(x)
        return F.relu(y + s)


class ConvNormElu(nn.Module):
    """`conv` of network/ops.py:198-224: conv + InstanceNorm(affine) + ELU."""

    def __init__(self, cin, cout, k, stride, wrap):
        super().__init__()
        self.conv = erp_conv(cin, cout, k, stride, bias=True) if wrap else nn.Conv2d(cin, cout, k, stride, k // 2)
        self.bn = _inorm(cout)

    def forward(self, x):
        return F.elu(self.bn(self.conv(x)))


class UpConv(nn.Module):
    """`upconv` (network/ops.py:226-233): bilinear x2 (align_corners=True) + zero-padded conv block."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = ConvNormElu(cin, cout, 3, 1, wrap=False)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, align_corners=True, mode='bilinear'))


def _stage(cin, cout, n, stride):
    return nn.Sequential(*[ResBlock2d(cin if i == 0 else cout, cout, stride if i == 0 else 1) for i in range(n)])


class ResUNetLight(nn.Module):
    """image_encoder: [n,3,H,W] -> [n,32,H/4,W/4].  Built as NeuralRayBaseRenderer does
    (network/renderer.py:111-112): layers=[1,2,6,4], inplanes=16, out_dim=32, wrap padding."""

    def __init__(self, cfg=None, in_dim=3, layers=(1, 2, 6, 4), out_dim=32, inplanes=16, use_wrap_padding=True,
                 autoencoder=False):
        super().__init__()
        cfg = cfg or {}
        if autoencoder or not use_wrap_padding or cfg.get('handle_distort') or cfg.get('handle_distort_input_all'):
            raise NotImplementedError('only the shipped eval configuration of ResUNetLight is implemented')
        self.conv1 = erp_conv(in_dim, inplanes, 7, 2)
        self.bn1 = _inorm(inplanes)
        self.layer1 = _stage(inplanes, 32, layers[0], 2)
        self.layer2 = _stage(32, 64, layers[1], 2)
        self.layer3 = _stage(64, 128, layers[2], 2)
        self.upconv3 = UpConv(128, 64)
        self.iconv3 = ConvNormElu(128, 64, 3, 1, wrap=True)
        self.upconv2 = UpConv(64, 32)
        self.iconv2 = ConvNormElu(64, 32, 3, 1, wrap=True)
        self.out_conv = nn.Conv2d(32, out_dim, 1, 1)

    @staticmethod
    def _skip(skip, x):
        """network/ops.py:359-371: the encoder feature is padded to the decoder feature's size and
        concatenated AFTER it: cat([decoder, skip])."""
        dy, dx = x.shape[2] - skip.shape[2], x.shape[3] - skip.shape[3]
        if dy or dx:
            skip = F.pad(skip, (0, 0, dy // 2, dy - dy // 2))
            skip = F.pad(skip, (dx // 2, dx - dx // 2, 0, 0), mode='circular')
        return torch.cat([x, skip], 1)

    def forward(self, x):
        x0 = F.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        y = self.iconv3(self._skip(x2, self.upconv3(x3)))
        y = self.iconv2(self._skip(x1, self.upconv2(y)))
        return self.out_conv(y)


class PreActRes(nn.Module):
    """ResidualBlock (network/ops.py:61-118), norm + wrap variant, dim_in == dim_out."""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Sequential(_inorm(c), nn.ReLU(True), ErpPad(1), nn.Conv2d(c, c, 3, 1, 0, bias=False),
                                  _inorm(c), nn.ReLU(True), ErpPad(1), nn.Conv2d(c, c, 3, 1, 0, bias=False))

    def forward(self, x):
        return self.conv(x) + x


class DefaultVisEncoder(nn.Module):
    """vis_encoder: fuses init-net ray_feats with image features (network/vis_encoder.py:6-32)."""

    def __init__(self, cfg=None):
        super().__init__()
        self.out_conv = nn.Sequential(erp_conv(64, 32, 3), PreActRes(32), PreActRes(32), nn.Conv2d(32, 32, 1, 1, bias=False))

    def forward(self, ray_feats, imgs_feats):
        if imgs_feats.shape[2:4] != ray_feats.shape[2:4]:
            imgs_feats = F.interpolate(imgs_feats, (ray_feats.shape[2], ray_feats.shape[3]), mode='bilinear')
        return self.out_conv(torch.cat([imgs_feats, ray_feats], 1))
