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


def fold_bn(conv, bn):
    """conv (+ bias) followed by an eval-mode BatchNorm2d -> equivalent (weight, bias) of one conv."""
    w = conv.weight.detach().to('cpu', torch.float32)
    scale = bn.weight.detach().cpu().float() / torch.sqrt(bn.running_var.detach().cpu().float() + bn.eps)
    b0 = conv.bias.detach().cpu().float() if conv.bias is not None else torch.zeros(w.shape[0])
    return (w * scale.view(-1, 1, 1, 1)).contiguous(), ((b0 - bn.running_mean.detach().cpu().float()) * scale + bn.bias.detach().cpu().float()).contiguous()


class _Conv:
    def __init__(self, conv, device, bn=None, pad_cin=0):
        """conv: nn.Conv2d (1x1 or 3x3); bn: an eval-mode BatchNorm2d folded into it; pad_cin: zero input channels
        appended so that a 1-channel input can sit in a 4-channel map."""
        if bn is not None:
            w, b = fold_bn(conv, bn)
        else:
            w = conv.weight.detach().to('cpu', torch.float32).contiguous()
            b = None if conv.bias is None else conv.bias.detach().to('cpu', torch.float32).contiguous()
        if pad_cin > w.shape[1]:
            w = torch.cat([w, torch.zeros(w.shape[0], pad_cin - w.shape[1], w.shape[2], w.shape[3])], 1).contiguous()
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
        self._partial = torch.empty(2 * 32 * 256 * 4, dtype=torch.float32, device=self.device)
        self._nrm = {}

    # ------------------------------------------------------------------ weights
    def _state_key(self):
        return tuple((k, v._version, v.data_ptr()) for m in (self.ie, self.ve) if m is not None for k, v in m.state_dict().items())

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
        if ve is not None:
            P.update(self.pack_head('v', ve.out_conv))
        self._packed, self._key = P, key
        return P

    def pack_head(self, tag, seq):
        """conv3x3 -> n x PreActRes -> conv1x1 (DefaultVisEncoder.out_conv, the init net's depth_conv / out_conv)."""
        d, P = self.device, {}
        cin = _plain(seq[0]).in_channels
        P[f'{tag}.in'] = _Conv(_plain(seq[0]), d, pad_cin=(cin + 3) // 4 * 4)
        for r in range(1, len(seq) - 1):
            P[f'{tag}.res{r}.a'] = _Conv(seq[r].conv[3], d)
            P[f'{tag}.res{r}.b'] = _Conv(seq[r].conv[7], d)
        P[f'{tag}.out'] = _Conv(seq[len(seq) - 1], d)
        return P

    def run_head(self, P, tag, seq, x):
        """x: _Map with the head's input channels -> channel-last [N,h,w,32] tensor."""
        d = self.device
        v = self._conv(P[f'{tag}.in'], x, _Map.empty(x.N, x.H, x.W, P[f'{tag}.in'].cout, d))
        for r in range(1, len(seq) - 1):
            blk = seq[r].conv
            t = self._norm(v, blk[0], 'relu', out=_Map.empty(x.N, x.H, x.W, v.C, d))
            t2 = self._conv(P[f'{tag}.res{r}.a'], t, _Map.empty(x.N, x.H, x.W, v.C, d))
            self._norm(t2, blk[4], 'relu')
            v = self._conv(P[f'{tag}.res{r}.b'], t2, t, residual=v)           # conv + x (PreAct residual block)
        return self._conv(P[f'{tag}.out'], v, _Map.empty(x.N, x.H, x.W, P[f'{tag}.out'].cout, d)).buf

    # ------------------------------------------------------------------ primitives
    _ACT = {'none': 0, 'relu': 1, 'lrelu': 2, 'elu': 3}

    def _conv(self, pc, x, out, wrap=True, residual=None, act='none', residual_first=False):
        """3x3 / 1x1 conv (+ bias) -> activation; residual added after the activation, or before it (residual_first)."""
        if x.C != pc.cin or out.C != pc.cout:
            raise PgrfError(f'conv2d: map has {x.C}->{out.C} channels, weights {pc.cin}->{pc.cout}')
        lib.check(lib.pgrf_conv2d(_ptr(x.buf), x.cs, x.coff, pc.cin, _ptr(out.buf), out.cs, out.coff, pc.cout, _ptr(pc.dev),
                                  int(pc.has_bias), int(wrap), x.N, x.H, x.W, pc.stride, self._ACT[act],
                                  _ptr(residual.buf) if residual is not None else None,
                                  residual.cs if residual is not None else 0, residual.coff if residual is not None else 0,
                                  int(residual_first), _stream()))
        return out

    def _norm_wb(self, norm):
        """gamma / beta of a norm module as fp32, contiguous tensors ON self.device (the kernel dereferences raw pointers:
        a module still on the CPU, on another GPU or in half precision must never reach it as it is).  Parameters that
        already qualify are used in place; the copy of any other is refreshed when its version counter moves."""
        w, b = norm.weight, norm.bias
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        cache = self.__dict__.setdefault('_nrm', {})
        ent = cache.get(id(norm))
        if ent is None or ent[0] != key:
            f = lambda t: t.detach().to(self.device, torch.float32).contiguous()
            ent = cache[id(norm)] = (key, f(w), f(b))
        return ent[1], ent[2]

    def _norm(self, x, norm, act, out=None, residual=None):
        """InstanceNorm2d(affine) + activation ('none' | 'relu' | 'elu'), residual added before the activation."""
        out = x if out is None else out
        gamma, beta = self._norm_wb(norm)
        need = x.N * lib.pgrf_instnorm_chunks(x.H * x.W) * x.C * 4
        if self._partial.numel() < need:
            self._partial = torch.empty(need, dtype=torch.float32, device=self.device)
        lib.check(lib.pgrf_instnorm(_ptr(x.buf), x.cs, x.coff, x.C, x.N, x.H * x.W, _ptr(gamma), _ptr(beta),
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
        c0 = P['stem'].shape[0]                                # stem width: 16 (image_encoder) or 32 (init net's res_net)
        x0 = _Map.empty(N, H // 2, W // 2, c0, d)
        lib.check(lib.pgrf_conv2d_direct(_ptr(imgs), N, Cin, H, W, _ptr(P['stem']), None, c0, 7, 2, 1, 0, _ptr(x0.buf), x0.cs, 0,
                                         _stream()))
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
        return self.run_head(P, 'v', ve.out_conv, cat)

    def __call__(self, imgs, ray_feats):
        """-> (img_feats, ray_feats) as NCHW-shaped VIEWS of channel-last memory (what the reference interface expects
        in ref_imgs_info; panogrf_amd.engine.Frame takes them without a re-layout)."""
        f = self.image_encoder(imgs)
        r = self.vis_encoder(ray_feats.detach().to(self.device, torch.float32), f)
        return f.permute(0, 3, 1, 2), r.permute(0, 3, 1, 2)
