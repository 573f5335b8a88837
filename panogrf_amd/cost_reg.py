"""Host mirror of the reference's two 3-D cost regularisers on top of the HIP conv primitives.

`CostRegNet` (models/cost_reg.py:99-175) and `UNet2Reg` (the UNet2 of Conv3DBlockv2 blocks that
models/test_models.py:84-146 builds by default) keep the reference's parameter names, so a reference
state_dict loads unchanged; the torch modules inside are parameter HOLDERS only — forward() never
calls them, it dispatches every conv / pool / upsample to libpanogrf_hip.so (no fallback).
Interface: forward(x[B,C,D,H,W]) -> [B,1,D,H,W], like `self.unet3d(cost_volume)` at
network/omni_mvsnet/pipeline3_model.py:855.  `forward_ndhwc` takes the channel-last volume that
pgrf_cost_volume_erp writes, skipping the permute of pipeline3_model.py:847.
"""
import ctypes as C

import os

import torch
import torch.nn as nn

from ._lib import lib, PgrfError
from .engine import _ptr, _stream, _require_gpu, pack_map


class _PackedConv:
    def __init__(self, weight, bias, device):
        cout, cin = weight.shape[:2]
        w = weight.detach().to('cpu', torch.float32).contiguous()
        b = None if bias is None else bias.detach().to('cpu', torch.float32).contiguous()
        n = lib.pgrf_conv3d_packed_floats(cout, cin)
        host = torch.empty(n, dtype=torch.float32)
        lib.check(lib.pgrf_pack_conv3d(_ptr(w), _ptr(b), cout, cin, _ptr(host)))
        self.dev = host.to(device)
        self.cout, self.cin, self.has_bias = cout, cin, b is not None
        self._w, self._b, self._device, self._small, self._wino, self._wino16 = w, b, device, {}, None, None

    def wino(self):
        """Packed block of the Winograd kernel (pgrf_conv3d_wino): filters transformed on the host, built on first use."""
        if self._wino is None:
            n = lib.pgrf_conv3d_wino_packed_floats(self.cout, self.cin)
            host = torch.empty(n, dtype=torch.float32)
            lib.check(lib.pgrf_pack_conv3d_wino(_ptr(self._w), _ptr(self._b), self.cout, self.cin, _ptr(host)))
            self._wino = host.to(self._device)
        return self._wino

    def wino16(self):
        """Packed block of the second-generation Winograd kernel (pgrf_conv3d_wino16), built on first use."""
        if self._wino16 is None:
            n = lib.pgrf_conv3d_wino16_packed_floats(self.cout, self.cin)
            host = torch.empty(n, dtype=torch.float32)
            lib.check(lib.pgrf_pack_conv3d_wino16(_ptr(self._w), _ptr(self._b), self.cout, self.cin, _ptr(host)))
            self._wino16 = host.to(self._device)
        return self._wino16

    def small(self, nshift):
        """Packed block of the narrow-layer kernel (pgrf_conv3d_small), built on first use per depth-shift count."""
        if nshift not in self._small:
            n = lib.pgrf_conv3d_small_packed_floats(self.cout, self.cin, nshift)
            host = torch.empty(n, dtype=torch.float32)
            lib.check(lib.pgrf_pack_conv3d_small(_ptr(self._w), _ptr(self._b), self.cout, self.cin, nshift, _ptr(host)))
            self._small[nshift] = host.to(self._device)
        return self._small[nshift]


    def small16(self, nshift):
        """Packed block of the 16-row narrow-layer kernel (pgrf_conv3d_small16), built on first use per depth-shift count."""
        key = ('s16', nshift)
        if key not in self._small:
            n = lib.pgrf_conv3d_small16_packed_floats(self.cout, self.cin, nshift)
            host = torch.empty(n, dtype=torch.float32)
            lib.check(lib.pgrf_pack_conv3d_small16(_ptr(self._w), _ptr(self._b), self.cout, self.cin, nshift, _ptr(host)))
            self._small[key] = host.to(self._device)
        return self._small[key]

    def upfused(self, D, H, W):
        """Packed block of the up-sampling-fused kernel (pgrf_conv3d_upfused) for a LOW-resolution input of D x H x W: 27
        position classes of interpolation-folded kernels, built on first use per size."""
        key = ('up', D, H, W)
        if key not in self._small:
            n = lib.pgrf_conv3d_upfused_packed_floats(self.cout, self.cin)
            host = torch.empty(n, dtype=torch.float32)
            lib.check(lib.pgrf_pack_conv3d_upfused(_ptr(self._w), _ptr(self._b), self.cout, self.cin, D, H, W, _ptr(host)))
            self._small[key] = host.to(self._device)
        return self._small[key]


    def upbrick(self, D, H, W):
        """Packed block of pgrf_conv3d_upbrick (16-row tiles over the low-res brick in LDS), per low-res size."""
        key = ('upb', D, H, W)
        if key not in self._small:
            n = lib.pgrf_conv3d_upbrick_packed_floats(self.cout, self.cin)
            host = torch.empty(n, dtype=torch.float32)
            lib.check(lib.pgrf_pack_conv3d_upbrick(_ptr(self._w), _ptr(self._b), self.cout, self.cin, D, H, W, _ptr(host)))
            self._small[key] = host.to(self._device)
        return self._small[key]


def _upconv(pc, x, out, up_scratch, slope=0.01, residual=None):
    """conv(interpolate(x, scale_factor=2, mode='trilinear')) + LeakyReLU (+ residual) -> `out` (high resolution): CostRegNet's
    decoder step (models/cost_reg.py:58-94).  Fused into ONE launch on the low-resolution tensor where the kernel applies
    (Cin 16 | 32, Cout 8 | 16; PANOGRF_CONV_UPFUSE=0 disables); otherwise up-sample into `up_scratch()` and convolve."""
    want = os.environ.get('PANOGRF_CONV_UPFUSE', '1')
    # (Cout = 16 fills four row tiles by itself: the fused form measured 155 us against 110 + 11 us for conv9 at 64x128x256, so only
    #  the 8-channel layers take it by default; 'force' = wherever the kernel applies)
    fits = (min(x.D, x.H, x.W) >= 2 and x.H * x.W * x.cs < (1 << 23) and x.B * x.D * x.H * x.W * x.cs < (1 << 29))
    if want != '0' and fits and pc.cin == 16 and pc.cout == 8 and x.W >= 33 and os.environ.get('PANOGRF_CONV_UPBRICK', '1') != '0':
        # 16-row tiles over the low-res brick in LDS (conv11 of CostRegNet)
        lib.check(lib.pgrf_conv3d_upbrick(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout,
                                          _ptr(pc.upbrick(x.D, x.H, x.W)), int(pc.has_bias), x.B, x.D, x.H, x.W, float(slope),
                                          _ptr(residual.buf) if residual is not None else None,
                                          residual.cs if residual is not None else 0,
                                          residual.coff if residual is not None else 0, _stream()))
        return out
    if want != '0' and fits and pc.cin in (16, 32) and (pc.cout == 8 or (want == 'force' and pc.cout == 16)):
        lib.check(lib.pgrf_conv3d_upfused(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout,
                                          _ptr(pc.upfused(x.D, x.H, x.W)), int(pc.has_bias), x.B, x.D, x.H, x.W, float(slope),
                                          _ptr(residual.buf) if residual is not None else None,
                                          residual.cs if residual is not None else 0,
                                          residual.coff if residual is not None else 0, _stream()))
        return out
    return _conv(pc, _up(x, up_scratch()), out, slope=slope, residual=residual)


class _Vol:
    """channel-last volume view: buffer [B,D,H,W,cs] + channel slice [coff, coff+C)"""

    def __init__(self, buf, coff, C):
        self.buf, self.coff, self.C = buf, coff, C
        self.B, self.D, self.H, self.W, self.cs = buf.shape

    @staticmethod
    def empty(B, D, H, W, C, device):
        return _Vol(torch.empty(B, D, H, W, C, dtype=torch.float32, device=device), 0, C)

    def slice(self, coff, C):
        return _Vol(self.buf, self.coff + coff, C)


def _small_shift(pc, x, stride):
    """Depth-shift count for the narrow-layer kernel, or 0 to use the general one.  PANOGRF_CONV_SMALL=0 disables."""
    if os.environ.get('PANOGRF_CONV_SMALL', '1') == '0':
        return 0
    if pc.cin not in (8, 16, 32) or pc.cout % 4 or pc.cout > 32 or pc.cout < 4:
        return 0
    Do = (x.D - 1) // stride + 1
    if stride == 1 and pc.cout <= 8 and Do % 4 == 0:
        return 4
    if stride == 1 and pc.cout <= 16 and Do % 2 == 0:
        return 2
    return 1 if pc.cin < 32 else 0


def _small16_shift(pc, x, stride):
    """Depth-shift count for the 16-row narrow-layer kernels (pgrf_conv3d_small16, v_mfma_f32_16x16x4_f32 tiles), or 0:
    8 output channels x 2 shifts from 16 / 32 input channels (LDS-brick kernel, 75 % useful products against the 32-row
    kernel's 50 %: conv0 of CostRegNet 573 -> 341 us) or 16 channels unshifted (100 % against 67 %; stride 1 from 16 / 32
    channels on the LDS-brick kernel, everything else from registers).  PANOGRF_CONV_SMALL16=0 disables."""
    if os.environ.get('PANOGRF_CONV_SMALL16', '1') == '0' or (pc.cin != 8 and pc.cin % 16):
        return 0
    if pc.cout == 8:
        return 2 if pc.cin in (16, 32) and stride == 1 and x.D % 2 == 0 else 0
    if pc.cout == 16 and pc.cin in (8, 16, 32):
        return 1
    # wider layers in groups of 16 output channels (LDS-brick kernel, stride 1): the 1/4- and 1/8-resolution layers of
    # CostRegNet, where the general kernel's waves mostly wait for their own loads.  PANOGRF_CONV_SMALL16_WIDE = max voxels.
    wide = int(os.environ.get('PANOGRF_CONV_SMALL16_WIDE', '65536'))
    if (pc.cout % 16 == 0 and pc.cout <= int(os.environ.get('PANOGRF_CONV_SMALL16_COUT', '64')) and pc.cin % 16 == 0 and pc.cin <= 1024 and stride == 1
            and x.B * x.D * x.H * x.W <= wide):
        wino = os.environ.get('PANOGRF_CONV_WINOGRAD', '1')          # layers the Winograd kernel takes stay there
        if wino == 'force' or (wino != '0' and (lib.pgrf_conv3d_wino_preferred(pc.cin, pc.cout, x.B, x.D, x.H, x.W) or
                                                 lib.pgrf_conv3d_wino16_preferred(pc.cin, pc.cout, x.B, x.D, x.H, x.W))):
            return 0
        return 1
    return 0


def _conv_wino16_applies(pc, x, stride, residual):
    """the dispatch rule of _conv below for the second-generation Winograd kernel"""
    if pc.cout == 1 and pc.cin >= 64 and pc.cin % 4 == 0 and stride == 1:
        return False
    if _small16_shift(pc, x, stride) or _small_shift(pc, x, stride):
        return False
    force = os.environ.get('PANOGRF_CONV_WINOGRAD') == 'force'
    return bool(stride == 1 and residual is None and os.environ.get('PANOGRF_CONV_WINOGRAD', '1') != '0' and os.environ.get('PANOGRF_WINO', '16') != '32' and
                (lib.pgrf_conv3d_wino16_preferred(pc.cin, pc.cout, x.B, x.D, x.H, x.W) or
                 (force and lib.pgrf_conv3d_wino16_supported(pc.cin, pc.cout, x.D, x.H, x.W))))


def _wino16_ksplit(pc, x, stride, residual):
    """split of the input-channel blocks for layers with too few rows to fill the chip (pgrf_conv3d_wino16_splitk), 0 = none"""
    if stride != 1 or residual is not None or os.environ.get('PANOGRF_CONV_WINOGRAD', '1') == '0' or os.environ.get('PANOGRF_WINO', '16') == '32':
        return 0
    if os.environ.get('PANOGRF_WINO_KSPLIT', '1') == '0' or (pc.cout == 1 and pc.cin >= 64):
        return 0
    if _small16_shift(pc, x, stride) or _small_shift(pc, x, stride):
        return 0
    return int(lib.pgrf_conv3d_wino16_ksplit(pc.cin, pc.cout, x.B, x.D, x.H, x.W))


def _conv(pc, x, out, stride=1, slope=0.01, residual=None, pool_out=None):
    """pool_out: AvgPool3d(2) of the result as well (UNet2's encoder levels) -- from the Winograd kernel's epilogue where that kernel
    takes the layer, by pgrf_avgpool3d_2 behind any other."""
    if pool_out is not None:
        if (_conv_wino16_applies(pc, x, stride, residual) and os.environ.get('PANOGRF_POOL_FUSE', '1') != '0' and
                not _wino16_ksplit(pc, x, stride, residual)):
            lib.check(lib.pgrf_conv3d_wino16_pool(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout,
                                                  _ptr(pc.wino16()), int(pc.has_bias), x.B, x.D, x.H, x.W, float(slope), None, 0, 0,
                                                  _ptr(pool_out.buf), pool_out.cs, pool_out.coff, _stream()))
            return out
        _conv(pc, x, out, stride, slope, residual)
        _pool(out, pool_out)
        return out
    if pc.cout == 1 and pc.cin >= 64 and pc.cin % 4 == 0 and stride == 1:
        # wide-input single-channel conv: pointwise MFMA projection onto the 27 tap vectors + 27-tap gather
        y = torch.empty(x.B, x.D, x.H, x.W, 32, dtype=torch.float32, device=x.buf.device)
        lib.check(lib.pgrf_conv3d_cout1_project(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(pc.dev), _ptr(y), x.B, x.D, x.H, x.W, _stream()))
        lib.check(lib.pgrf_conv3d_cout1_gather(_ptr(y), _ptr(pc.dev), pc.cin, int(pc.has_bias), _ptr(out.buf), out.cs, out.coff,
                                               x.B, x.D, x.H, x.W, float(slope),
                                               _ptr(residual.buf) if residual is not None else None,
                                               residual.cs if residual is not None else 0,
                                               residual.coff if residual is not None else 0, _stream()))
        return out
    ns = _small16_shift(pc, x, stride)
    if ns:
        lib.check(lib.pgrf_conv3d_small16(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout,
                                          _ptr(pc.small16(ns)), int(pc.has_bias), ns, x.B, x.D, x.H, x.W, stride, float(slope),
                                          _ptr(residual.buf) if residual is not None else None,
                                          residual.cs if residual is not None else 0,
                                          residual.coff if residual is not None else 0, _stream()))
        return out
    ns = _small_shift(pc, x, stride)
    if ns:
        lib.check(lib.pgrf_conv3d_small(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout,
                                        _ptr(pc.small(ns)), int(pc.has_bias), ns, x.B, x.D, x.H, x.W, stride, float(slope),
                                        _ptr(residual.buf) if residual is not None else None,
                                        residual.cs if residual is not None else 0,
                                        residual.coff if residual is not None else 0, _stream()))
        return out
    ks = _wino16_ksplit(pc, x, stride, residual)
    if ks >= 2:
        # few rows (the low-resolution levels): every row as `ks` work items over shares of the input-channel blocks, partial sums reduced behind
        scratch = torch.empty(ks * x.B * x.D * x.H * x.W * pc.cout, dtype=torch.float32, device=x.buf.device)
        lib.check(lib.pgrf_conv3d_wino16_splitk(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout,
                                                _ptr(pc.wino16()), int(pc.has_bias), x.B, x.D, x.H, x.W, float(slope), ks,
                                                _ptr(scratch), scratch.numel(), _stream()))
        return out
    force = os.environ.get('PANOGRF_CONV_WINOGRAD') == 'force'
    if (stride == 1 and residual is None and os.environ.get('PANOGRF_CONV_WINOGRAD', '1') != '0' and os.environ.get('PANOGRF_WINO', '16') != '32' and
            (lib.pgrf_conv3d_wino16_preferred(pc.cin, pc.cout, x.B, x.D, x.H, x.W) or
             (force and lib.pgrf_conv3d_wino16_supported(pc.cin, pc.cout, x.D, x.H, x.W)))):
        # wide stride-1 layers: Winograd F(2x2x2, 3x3x3), 3.4x fewer MFMAs; two independent workgroups per CU (csrc/conv3d_wino16.hip)
        lib.check(lib.pgrf_conv3d_wino16(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout,
                                         _ptr(pc.wino16()), int(pc.has_bias), x.B, x.D, x.H, x.W, float(slope), None, 0, 0, _stream()))
        return out
    if (stride == 1 and residual is None and os.environ.get('PANOGRF_CONV_WINOGRAD', '1') != '0' and
            (lib.pgrf_conv3d_wino_preferred(pc.cin, pc.cout, x.B, x.D, x.H, x.W) or
             (os.environ.get('PANOGRF_CONV_WINOGRAD') == 'force' and lib.pgrf_conv3d_wino_supported(pc.cin, pc.cout, x.D, x.H, x.W)))):
        # wide stride-1 layers: Winograd F(2x2x2, 3x3x3), 3.4x fewer MFMAs (csrc/conv3d_wino.hip)
        lib.check(lib.pgrf_conv3d_wino(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout,
                                       _ptr(pc.wino()), int(pc.has_bias), x.B, x.D, x.H, x.W, float(slope), None, 0, 0, _stream()))
        return out
    lib.check(lib.pgrf_conv3d_wrap(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout,
                                   _ptr(pc.dev), int(pc.has_bias), x.B, x.D, x.H, x.W, stride, float(slope),
                                   _ptr(residual.buf) if residual is not None else None,
                                   residual.cs if residual is not None else 0,
                                   residual.coff if residual is not None else 0, _stream()))
    return out


def _pool(x, out):
    lib.check(lib.pgrf_avgpool3d_2(_ptr(x.buf), x.cs, x.coff, _ptr(out.buf), out.cs, out.coff, x.B, x.D, x.H, x.W, x.C, _stream()))
    return out


def _up(x, out):
    lib.check(lib.pgrf_upsample3d_2(_ptr(x.buf), x.cs, x.coff, _ptr(out.buf), out.cs, out.coff, x.B, x.D, x.H, x.W, x.C, _stream()))
    return out


def _to_ndhwc(x):
    B, Cc, D, H, W = x.shape
    return pack_map(x.reshape(B, Cc, D * H, W)).view(B, D, H, W, -1)


class _RegBase(nn.Module):
    def __init__(self):
        super().__init__()
        self._packed = None
        self._packed_key = None

    def _key(self):
        return tuple((k, v._version, v.data_ptr()) for k, v in self.state_dict().items())

    def _ensure_packed(self, device):
        key = (self._key(), str(device))
        if self._packed is None or self._packed_key != key:
            self._packed = self._pack(device)
            self._packed_key = key
        return self._packed

    def forward(self, x):
        _require_gpu()
        out = self.forward_ndhwc(_to_ndhwc(x.float()))
        B, D, H, W, _ = out.shape
        return out.view(B, 1, D, H, W)       # C == 1: channel-last and channel-first coincide


class _ConvBnHolder(nn.Module):
    def __init__(self, cin, cout, bn=True):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, 3, bias=False)
        if bn:
            self.bn = nn.BatchNorm3d(cout)


class CostRegNet(_RegBase):
    """models/cost_reg.py:99-175 (norm_act=None branch: BatchNorm3d eval + LeakyReLU(0.01))."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        if cfg.get('inplace_abn', False):
            raise PgrfError('inplace_abn variant is not implemented (no shipped config enables it)')
        if not cfg.get('use_wrap_padding', True):
            raise PgrfError('only use_wrap_padding=True is implemented')
        chans = [(in_channels, 8), (8, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)]
        for i, (ci, co) in enumerate(chans):
            setattr(self, f'conv{i}', _ConvBnHolder(ci, co))
        self.conv7, self.conv9, self.conv11 = _ConvBnHolder(64, 32), _ConvBnHolder(32, 16), _ConvBnHolder(16, 8)
        self.prob = _ConvBnHolder(8, 1, bn=False)

    def _pack(self, device):
        pk = {}
        for name in ['conv0', 'conv1', 'conv2', 'conv3', 'conv4', 'conv5', 'conv6', 'conv7', 'conv9', 'conv11']:
            m = getattr(self, name)
            scale = m.bn.weight.detach() / torch.sqrt(m.bn.running_var.detach() + m.bn.eps)
            w = m.conv.weight.detach() * scale.view(-1, 1, 1, 1, 1)           # eval-mode BN folded into the conv
            b = m.bn.bias.detach() - m.bn.running_mean.detach() * scale
            pk[name] = _PackedConv(w, b, device)
        pk['prob'] = _PackedConv(self.prob.conv.weight, None, device)
        return pk

    def forward_ndhwc(self, vol):
        B, D, H, W, Cc = vol.shape
        if D % 8 or H % 8 or W % 8:
            raise PgrfError('CostRegNet needs D,H,W divisible by 8')
        dev = vol.device
        pk = self._ensure_packed(dev)
        x = _Vol(vol.contiguous(), 0, Cc)
        E = lambda s, c: _Vol.empty(B, D // s, H // s, W // s, c, dev)
        c0 = _conv(pk['conv0'], x, E(1, 8))
        c2 = _conv(pk['conv2'], _conv(pk['conv1'], c0, E(2, 16), stride=2), E(2, 16))
        c4 = _conv(pk['conv4'], _conv(pk['conv3'], c2, E(4, 32), stride=2), E(4, 32))
        t = _conv(pk['conv6'], _conv(pk['conv5'], c4, E(8, 64), stride=2), E(8, 64))
        t = _upconv(pk['conv7'], t, E(4, 32), lambda: E(4, 64), residual=c4)   # conv4 + conv7(up(x))
        t = _upconv(pk['conv9'], t, E(2, 16), lambda: E(2, 32), residual=c2)
        t = _upconv(pk['conv11'], t, E(1, 8), lambda: E(1, 16), residual=c0)
        return _conv(pk['prob'], t, E(1, 1), slope=1.0).buf


class _BlockHolder(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, cout, 3)
        self.conv2 = nn.Conv3d(cout, cout, 3)


class UNet2Reg(_RegBase):
    """The default `unet3d`: UNet2 (models/common_blocks.py:187-242) over Conv3DBlockv2 blocks
    (:366-455, no norm, LeakyReLU 0.01) wired as in models/test_models.py:84-146 with size=4."""

    def __init__(self, in_channels=32, cnn3d_num_layer=3, size=4):
        super().__init__()
        n = cnn3d_num_layer
        enc = [(2 ** (i + size + 1), 2 ** (i + size + 2)) for i in range(n)]
        enc.append((2 ** (n + size + 1), 2 ** (n + size + 2)))
        if enc[0][0] != in_channels:
            raise PgrfError(f'UNet2Reg expects {enc[0][0]} input channels')
        dec = [(2 ** (size + 3), 1)] + [(4 * 2 ** (i + size + 1), 2 ** (i + size + 1)) for i in range(1, n)]
        self.encoders = nn.ModuleList([_BlockHolder(a, b) for a, b in enc])
        self.decoders = nn.ModuleList([_BlockHolder(a, b) for a, b in dec])
        self.n = n

    def _pack(self, device):
        pk = {}
        for grp in ['encoders', 'decoders']:
            for i, blk in enumerate(getattr(self, grp)):
                pk[f'{grp}.{i}.1'] = _PackedConv(blk.conv1.weight, blk.conv1.bias, device)
                pk[f'{grp}.{i}.2'] = _PackedConv(blk.conv2.weight, blk.conv2.bias, device)
        # the last decoder step is a Cout = 1 convolution over cat((up(x), skip)): its two channel halves as convolutions of their own
        # (forward_ndhwc projects the up-sampled half at LOW resolution, pgrf_conv3d_cout1_gather_up)
        w0 = self.decoders[0].conv1.weight
        if self.n >= 2 and w0.shape[0] == 1 and w0.shape[1] % 2 == 0 and w0.shape[1] // 2 >= 64:
            c = w0.shape[1] // 2
            pk['decoders.0.1.up'] = _PackedConv(w0[:, :c], None, device)
            pk['decoders.0.1.skip'] = _PackedConv(w0[:, c:], self.decoders[0].conv1.bias, device)
        return pk

    def forward_ndhwc(self, vol):
        B, D, H, W, Cc = vol.shape
        n = self.n
        if D % (1 << n) or H % (1 << n) or W % (1 << n):
            raise PgrfError(f'UNet2Reg needs D,H,W divisible by {1 << n}')
        dev = vol.device
        pk = self._ensure_packed(dev)
        E = lambda s, c: _Vol.empty(B, D >> s, H >> s, W >> s, c, dev)
        x = _Vol(vol.contiguous(), 0, Cc)
        cats = {}
        for i in range(n + 1):
            c1, c2 = pk[f'encoders.{i}.1'], pk[f'encoders.{i}.2']
            t = _conv(c1, x, E(i, c1.cout))
            pooled = E(i + 1, c2.cout) if i < n else None       # (the pooled tensor comes out of the conv's own epilogue where it can)
            if i < n - 1:
                # unpooled output is the skip of decoder i: write it straight into the second half of the concat buffer
                cat = E(i, 2 * c2.cout)
                cats[i] = cat
                u = _conv(c2, t, cat.slice(c2.cout, c2.cout), pool_out=pooled)
            else:
                u = _conv(c2, t, E(i, c2.cout), pool_out=pooled)
            x = pooled if i < n else u
        d = pk[f'decoders.{n - 1}.1'], pk[f'decoders.{n - 1}.2']
        x = _up(x, E(n - 1, x.C))
        x = _conv(d[1], _conv(d[0], x, E(n - 1, d[0].cout)), E(n - 1, d[1].cout))
        for i in range(n - 2, -1, -1):
            cat = cats[i]
            d = pk[f'decoders.{i}.1'], pk[f'decoders.{i}.2']
            if i == 0 and 'decoders.0.1.up' in pk and x.C * 2 == cat.cs and os.environ.get('PANOGRF_UNET_HEAD_FUSE', '1') != '0':
                # Cout = 1 over cat((up(x), skip)): project x at LOW resolution and the skip half at full resolution onto the 27 tap
                # vectors, interpolate inside the gather -- up(x) is never written (UNet2 at 128x256x64: see profiles/)
                pu, ps = pk['decoders.0.1.up'], pk['decoders.0.1.skip']
                skip = cat.slice(x.C, x.C)
                y_lo = torch.empty(x.B, x.D, x.H, x.W, 32, dtype=torch.float32, device=dev)
                y_hi = torch.empty(skip.B, skip.D, skip.H, skip.W, 32, dtype=torch.float32, device=dev)
                lib.check(lib.pgrf_conv3d_cout1_project(_ptr(x.buf), x.cs, x.coff, pu.cin, _ptr(pu.dev), _ptr(y_lo), x.B, x.D, x.H, x.W, _stream()))
                lib.check(lib.pgrf_conv3d_cout1_project(_ptr(skip.buf), skip.cs, skip.coff, ps.cin, _ptr(ps.dev), _ptr(y_hi), skip.B, skip.D,
                                                        skip.H, skip.W, _stream()))
                t = E(0, 1)
                lib.check(lib.pgrf_conv3d_cout1_gather_up(_ptr(y_hi), _ptr(y_lo), _ptr(ps.dev), ps.cin, int(ps.has_bias), _ptr(t.buf), t.cs,
                                                          t.coff, skip.B, skip.D, skip.H, skip.W, 0.01, None, 0, 0, _stream()))
                x = _conv(d[1], t, E(0, d[1].cout))
                continue
            _up(x, cat.slice(0, x.C))                         # torch.cat((upsampled, skip), dim=1)
            x = _conv(d[1], _conv(d[0], _Vol(cat.buf, 0, cat.cs), E(i, d[0].cout)), E(i, d[1].cout))
        return x.buf


def build_unet3d(args, in_channels=32):
    """The `unet3d` branch of models/test_models.py:95-146."""
    if args.get('use_new_reg3dnet', False):
        # (models/test_models.py:95-100: 64 input channels without group_wise -- the reference's default there never matches its
        #  32-channel cost volume; the caller passes the channel count its volume really has)
        return CostRegNet(args, in_channels=in_channels)
    return UNet2Reg(in_channels=in_channels, cnn3d_num_layer=args.get('cnn3d_num_layer', 3))
