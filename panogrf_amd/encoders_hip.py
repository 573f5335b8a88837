"""The two per-frame 2-D encoders of the renderer on the HIP library.

`HipEncoders(image_encoder, vis_encoder)` takes the torch modules of `panogrf_amd.encoders` as PARAMETER HOLDERS
(reference parameter names, so checkpoints load unchanged) and runs their layer sequence — ResUNetLight
(network/ops.py:235-455: stem 7x7/2, three BasicBlock stages with InstanceNorm, two bilinear-x2 `upconv` + `iconv`
decoder levels, 1x1 head) and DefaultVisEncoder (network/vis_encoder.py:6-32) — through the C ABI:
`pgrf_conv2d_direct` (stem), `pgrf_conv2d` (every 3x3 / 1x1 conv, ERP wrap or zero padding, on the implicit-GEMM MFMA
kernel), `pgrf_instnorm` (statistics + normalise + activation + residual in two launches) and `pgrf_resize_bilinear`.
Maps are channel-last [N,H,W,C]; the skip concatenations are channel slices of one buffer, the ERP padding lives in the
conv kernel, so there are no pad / cat / permute launches: ~110 launches per frame instead of ~250 on MIOpen + torch
elementwise ops.  No fallback inside: an unsupported shape raises; the torch path stays selectable from the renderer
(`cfg['hip_encoders']=False`).
"""
import torch

from ._lib import lib, PgrfError
from .engine import _ptr, _stream, _require_gpu, pack_map


class _Map:
    """channel-last map view: buffer [N,H,W,cs] + channel slice [coff, coff+C)"""

    def __init__(self, buf, coff=0, C=None):
        self.buf, self.coff = buf, coff
        self.N, self.H, self.W, self.cs = buf.shape
        self.C = self.cs - coff if C is None else C

    @staticmethod
    def empty(N, H, W, C, device):
        return _Map(torch.empty(N, H, W, C, dtype=torch.float32, device=device))

    def slice(self, coff, C):
        return _Map(self.buf, self.coff + coff, C)


class _Conv:
    def __init__(self, conv, device):
        w = conv.weight.detach().to('cpu', torch.float32).contiguous()
        b = None if conv.bias is None else conv.bias.detach().to('cpu', torch.float32).contiguous()
        self.cout, self.cin, k, k2 = w.shape
        if k != k2 or k not in (1, 3):
            raise PgrfError(f'conv kernel {k}x{k2}: only 1x1 and 3x3 go through pgrf_conv2d')
        self.k, self.stride, self.has_bias = k, int(conv.stride[0]), b is not None
        host = torch.empty(lib.pgrf_conv2d_packed_floats(self.cout, self.cin), dtype=torch.float32)
        lib.check(lib.pgrf_pack_conv2d(_ptr(w), _ptr(b), self.cout, self.cin, k, _ptr(host)))
        self.dev = host.to(device)


def _plain(conv_or_seq):
    """erp_conv = Sequential(ErpPad, Conv2d): the conv is the last child; plain Conv2d otherwise."""
    return conv_or_seq[-1] if isinstance(conv_or_seq, torch.nn.Sequential) else conv_or_seq


class HipEncoders:
    def __init__(self, image_encoder, vis_encoder, device):
        _require_gpu()
        self.ie, self.ve, self.device = image_encoder, vis_encoder, torch.device(device)
        self._packed, self._key = None, None
        self._partial = torch.empty(2 * 32 * 256 * 2, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ weights
    def _state_key(self):
        return tuple((k, v._version, v.data_ptr()) for m in (self.ie, self.ve) for k, v in m.state_dict().items())

    def _pack(self):
        key = self._state_key()
        if self._packed is not None and self._key == key:
            return self._packed
        d, ie, ve = self.device, self.ie, self.ve
        P = {'stem': _plain(ie.conv1).weight.detach().to(d, torch.float32).contiguous()}
        for name, stage in (('layer1', ie.layer1), ('layer2', ie.layer2), ('layer3', ie.layer3)):
            for i, blk in enumerate(stage):
                P[f'{name}.{i}.conv1'] = _Conv(_plain(blk.conv1), d)
                P[f'{name}.{i}.conv2'] = _Conv(_plain(blk.conv2), d)
                if blk.downsample is not None:
                    P[f'{name}.{i}.ds'] = _Conv(blk.downsample[0], d)
        for name in ('upconv3', 'upconv2'):
            P[name] = _Conv(_plain(getattr(ie, name).conv.conv), d)
        for name in ('iconv3', 'iconv2'):
            P[name] = _Conv(_plain(getattr(ie, name).conv), d)
        P['out'] = _Conv(ie.out_conv, d)
        oc = ve.out_conv
        P['v.in'] = _Conv(_plain(oc[0]), d)
        for r in (1, 2):
            P[f'v.res{r}.a'] = _Conv(oc[r].conv[3], d)
            P[f'v.res{r}.b'] = _Conv(oc[r].conv[7], d)
        P['v.out'] = _Conv(oc[3], d)
        self._packed, self._key = P, key
        return P

    # ------------------------------------------------------------------ primitives
    def _conv(self, pc, x, out, wrap=True, residual=None):
        if x.C != pc.cin or out.C != pc.cout:
            raise PgrfError(f'conv2d: map has {x.C}->{out.C} channels, weights {pc.cin}->{pc.cout}')
        lib.check(lib.pgrf_conv2d(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout, _ptr(pc.dev),
                                  int(pc.has_bias), int(wrap), x.N, x.H, x.W, pc.stride,
                                  _ptr(residual.buf) if residual is not None else None,
                                  residual.cs if residual is not None else 0, residual.coff if residual is not None else 0,
                                  _stream()))
        return out

    def _norm(self, x, norm, act, out=None, residual=None):
        """InstanceNorm2d(affine) + activation ('none' | 'relu' | 'elu'), residual added before the activation."""
        out = x if out is None else out
        need = x.N * lib.pgrf_instnorm_chunks(x.H * x.W) * x.C * 2
        if self._partial.numel() < need:
            self._partial = torch.empty(need, dtype=torch.float32, device=self.device)
        lib.check(lib.pgrf_instnorm(_ptr(x.buf), x.cs, x.coff, x.C, x.N, x.H * x.W, _ptr(norm.weight), _ptr(norm.bias),
                                    float(norm.eps), {'none': 0, 'relu': 1, 'elu': 2}[act],
                                    _ptr(residual.buf) if residual is not None else None,
                                    residual.cs if residual is not None else 0, residual.coff if residual is not None else 0,
                                    _ptr(out.buf), out.cs, out.coff, _ptr(self._partial), _stream()))
        return out

    def _resize(self, x, out, align):
        lib.check(lib.pgrf_resize_bilinear(_ptr(x.buf), x.cs, x.coff, x.C, x.N, x.H, x.W, _ptr(out.buf), out.cs, out.coff,
                                           out.H, out.W, int(align), _stream()))
        return out

    def _block(self, P, name, blk, x, out):
        """BasicBlock (network/ops.py:127-196): relu(bn2(conv2(relu(bn1(conv1 x)))) + shortcut(x)) -> `out` slice."""
        c1, c2 = P[name + '.conv1'], P[name + '.conv2']
        Ho, Wo = (x.H - 1) // c1.stride + 1, (x.W - 1) // c1.stride + 1
        y = self._conv(c1, x, _Map.empty(x.N, Ho, Wo, c1.cout, self.device))
        self._norm(y, blk.bn1, 'relu')
        y2 = self._conv(c2, y, _Map.empty(x.N, Ho, Wo, c2.cout, self.device))
        s = x
        if blk.downsample is not None:
            s = self._conv(P[name + '.ds'], x, _Map.empty(x.N, Ho, Wo, c2.cout, self.device))
            self._norm(s, blk.downsample[1], 'none')
        return self._norm(y2, blk.bn2, 'relu', out=out, residual=s)

    # ------------------------------------------------------------------ forward
    def image_encoder(self, imgs):
        """imgs [N,3,H,W] (NCHW, CUDA) -> img_feats channel-last [N,H/4,W/4,32]  (network/ops.py:332-372)."""
        ie, d = self.ie, self.device
        P = self._pack()
        imgs = imgs.detach().to(d, torch.float32).contiguous()
        N, Cin, H, W = imgs.shape
        if H % 16 or W % 16:
            raise PgrfError(f'HIP encoders need H, W multiples of 16 (got {H}x{W}); use cfg["hip_encoders"]=False')
        x0 = _Map.empty(N, H // 2, W // 2, 16, d)
        lib.check(lib.pgrf_conv2d_direct(_ptr(imgs), N, Cin, H, W, _ptr(P['stem']), 16, 7, 2, 1, _ptr(x0.buf), x0.cs, 0, _stream()))
        self._norm(x0, ie.bn1, 'relu')
        cat2 = _Map.empty(N, H // 4, W // 4, 64, d)            # [upconv2 out (32) | x1 (32)]: input of iconv2
        cat3 = _Map.empty(N, H // 8, W // 8, 128, d)           # [upconv3 out (64) | x2 (64)]: input of iconv3
        x = x0
        for name, stage, final in (('layer1', ie.layer1, cat2.slice(32, 32)), ('layer2', ie.layer2, cat3.slice(64, 64)),
                                   ('layer3', ie.layer3, None)):
            for i, blk in enumerate(stage):
                last = i == len(stage) - 1
                c1 = P[f'{name}.{i}.conv1']
                Ho, Wo = (x.H - 1) // c1.stride + 1, (x.W - 1) // c1.stride + 1
                out = final if (last and final is not None) else _Map.empty(N, Ho, Wo, c1.cout, d)
                x = self._block(P, f'{name}.{i}', blk, x, out)
        # decoder level 3: bilinear x2 (align_corners=True) + zero-padded conv block, then the wrap-padded fusion conv
        up = self._resize(x, _Map.empty(N, H // 8, W // 8, 128, d), True)
        self._conv(P['upconv3'], up, cat3.slice(0, 64), wrap=False)
        self._norm(cat3.slice(0, 64), ie.upconv3.conv.bn, 'elu')
        y = self._conv(P['iconv3'], cat3, _Map.empty(N, H // 8, W // 8, 64, d))
        self._norm(y, ie.iconv3.bn, 'elu')
        up = self._resize(y, _Map.empty(N, H // 4, W // 4, 64, d), True)
        self._conv(P['upconv2'], up, cat2.slice(0, 32), wrap=False)
        self._norm(cat2.slice(0, 32), ie.upconv2.conv.bn, 'elu')
        y = self._conv(P['iconv2'], cat2, _Map.empty(N, H // 4, W // 4, 32, d))
        self._norm(y, ie.iconv2.bn, 'elu')
        return self._conv(P['out'], y, _Map.empty(N, H // 4, W // 4, 32, d)).buf

    def vis_encoder(self, ray_feats, img_feats_nhwc):
        """ray_feats [N,32,hr,wr] (NCHW) + image features channel-last -> fused ray features channel-last
        [N,hr,wr,32]  (network/vis_encoder.py:24-32)."""
        ve, d = self.ve, self.device
        P = self._pack()
        N, C, hr, wr = ray_feats.shape
        cat = _Map.empty(N, hr, wr, 64, d)
        f = _Map(img_feats_nhwc)
        if (f.H, f.W) != (hr, wr):
            self._resize(f, cat.slice(0, 32), False)
        else:
            cat.buf[..., :32] = img_feats_nhwc
        cat.buf[..., 32:] = pack_map(ray_feats)
        oc = ve.out_conv
        v = self._conv(P['v.in'], cat, _Map.empty(N, hr, wr, 32, d))
        for r in (1, 2):
            seq = oc[r].conv
            t = self._norm(v, seq[0], 'relu', out=_Map.empty(N, hr, wr, 32, d))
            t2 = self._conv(P[f'v.res{r}.a'], t, _Map.empty(N, hr, wr, 32, d))
            self._norm(t2, seq[4], 'relu')
            v = self._conv(P[f'v.res{r}.b'], t2, t, residual=v)           # conv + x (PreAct residual block)
        return self._conv(P['v.out'], v, _Map.empty(N, hr, wr, 32, d)).buf

    def __call__(self, imgs, ray_feats):
        """-> (img_feats, ray_feats) as NCHW-shaped VIEWS of channel-last memory (what the reference interface expects
        in ref_imgs_info; panogrf_amd.engine.Frame takes them without a re-layout)."""
        f = self.image_encoder(imgs)
        r = self.vis_encoder(ray_feats.detach().to(self.device, torch.float32), f)
        return f.permute(0, 3, 1, 2), r.permute(0, 3, 1, 2)
