"""The 2-D networks of the MVS stage on the HIP library: `HipInitNet(init_net)` runs
panogrf_amd.mvs.CostVolumeInitNet.forward with

  * the Equi feature net (ERP ResNet-18 with its eval-mode BatchNorms folded into the convolutions, wrap-padded max-pool,
    three decoder levels) through pgrf_conv2d_direct / pgrf_maxpool3s2_wrap / pgrf_conv2d / pgrf_resize_bilinear,
  * hypotheses, spherical cost volume and 3-D regulariser (already on HIP),
  * decoders2 (bilinear x2 + wrap-padded conv pairs with LeakyReLU), the init net's ResUNetLight (`res_net`, the layer
    sequence of panogrf_amd.encoders_hip) and its two conv3x3 / PreActRes / conv1x1 heads,

all on channel-last maps (the feature net writes [B,2,h,w,32], exactly the cost-volume kernel's input; U-Net concats
are channel slices).  The torch modules are parameter holders.  What stays on PyTorch: the mono-depth module (user
supplied, out of scope), three small bilinear resizes of NCHW inputs and the inverse-depth normalisation of the MVS
depth.  `decoders1` (the auxiliary low-resolution depth the reference returns as 'rectified_depth_d1') is not consumed
by the init net and is not evaluated here.  Measured on MI355X, 2 reference views, depth 256x512: see DESIGN.md.
"""
import torch
import torch.nn.functional as F

from ._lib import lib, PgrfError
from . import cost_volume as CV
from .engine import _ptr, _stream, pack_map
from .encoders_hip import HipEncoders, _Map, _Conv, fold_bn


class HipInitNet:
    def __init__(self, init_net, device):
        self.net, self.dn, self.device = init_net, init_net.depth_net, torch.device(device)
        self.T = HipEncoders(init_net.res_net, None, device)           # conv / norm / resize launchers + the res_net sequence
        self._packed, self._key = None, None

    def _state_key(self):
        return tuple((k, v._version, v.data_ptr()) for k, v in self.net.state_dict().items())

    def _pack(self):
        key = self._state_key()
        if self._packed is not None and self._key == key:
            return self._packed
        d, P = self.device, {}
        e = self.dn.unet.equi_encoder
        w, b = fold_bn(e.conv1, e.bn1)
        P['stem.w'], P['stem.b'] = w.to(d), b.to(d)
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(e, f'layer{li}')):
                P[f'l{li}.{bi}.c1'] = _Conv(blk.conv1, d, bn=blk.bn1)
                P[f'l{li}.{bi}.c2'] = _Conv(blk.conv2, d, bn=blk.bn2)
                if blk.downsample is not None:
                    P[f'l{li}.{bi}.ds'] = _Conv(blk.downsample[0], d, bn=blk.downsample[1])
        for i, blk in enumerate(self.dn.unet.equi_decoder):
            P[f'dec{i}'] = _Conv(blk.conv.conv, d)
        for i, blk in enumerate(self.dn.decoders2):
            P[f'd2.{i}.a'] = _Conv(blk.conv1, d)
            P[f'd2.{i}.b'] = _Conv(blk.conv2, d)
        P.update(self.T.pack_head('dc', self.net.depth_conv))
        P.update(self.T.pack_head('oc', self.net.out_conv))
        self._packed, self._key = P, key
        return P

    # ------------------------------------------------------------------ Equi feature net (models/my_equi.py:100-141)
    def equi(self, imgs):
        """imgs [n,3,H,W] NCHW -> channel-last [n,H/4,W/4,32]."""
        P, T, d = self._pack(), self.T, self.device
        imgs = imgs.detach().to(d, torch.float32).contiguous()
        n, cin, H, W = imgs.shape
        if H % 32 or W % 32:
            raise PgrfError(f'HIP Equi net needs H, W multiples of 32 (got {H}x{W})')
        x0 = _Map.empty(n, H // 2, W // 2, 64, d)
        lib.check(lib.pgrf_conv2d_direct(_ptr(imgs), n, cin, H, W, _ptr(P['stem.w']), _ptr(P['stem.b']), 64, 7, 2, 1, 1, _ptr(x0.buf), 64,
                                         0, _stream()))
        x = _Map.empty(n, H // 4, W // 4, 64, d)
        lib.check(lib.pgrf_maxpool3s2_wrap(_ptr(x0.buf), 64, 0, 64, n, H // 2, W // 2, _ptr(x.buf), 64, 0, _stream()))
        catA = _Map.empty(n, H // 16, W // 16, 512, d)        # [up(upconv_5) 256 | f3 256]
        catB = _Map.empty(n, H // 8, W // 8, 256, d)          # [up(upconv_4) 128 | f2 128]
        catC = _Map.empty(n, H // 4, W // 4, 128, d)          # [up(upconv_3)  64 | f1  64]
        finals = {1: catC.slice(64, 64), 2: catB.slice(128, 128), 3: catA.slice(256, 256), 4: None}
        e = self.dn.unet.equi_encoder
        for li in range(1, 5):
            blocks = getattr(e, f'layer{li}')
            for bi, blk in enumerate(blocks):
                c1, c2 = P[f'l{li}.{bi}.c1'], P[f'l{li}.{bi}.c2']
                Ho, Wo = (x.H - 1) // c1.stride + 1, (x.W - 1) // c1.stride + 1
                y = T._conv(c1, x, _Map.empty(n, Ho, Wo, c1.cout, d), act='relu')
                s = x
                if blk.downsample is not None:
                    s = T._conv(P[f'l{li}.{bi}.ds'], x, _Map.empty(n, Ho, Wo, c2.cout, d))
                last = bi == len(blocks) - 1
                out = finals[li] if (last and finals[li] is not None) else _Map.empty(n, Ho, Wo, c2.cout, d)
                x = T._conv(c2, y, out, act='relu', residual=s, residual_first=True)
        dec = lambda i, xin, out: T._conv(P[f'dec{i}'], xin, out, act='elu')
        y = dec(0, x, _Map.empty(n, H // 32, W // 32, 256, d))
        T._resize(y, catA.slice(0, 256), False)
        y = dec(2, dec(1, catA, _Map.empty(n, H // 16, W // 16, 256, d)), _Map.empty(n, H // 16, W // 16, 128, d))
        T._resize(y, catB.slice(0, 128), False)
        y = dec(4, dec(3, catB, _Map.empty(n, H // 8, W // 8, 128, d)), _Map.empty(n, H // 8, W // 8, 64, d))
        T._resize(y, catC.slice(0, 64), False)
        return dec(6, dec(5, catC, _Map.empty(n, H // 4, W // 4, 64, d)), _Map.empty(n, H // 4, W // 4, 32, d)).buf

    # ------------------------------------------------------------------ decoders2 (pipeline3_model.py:883-932)
    def decode(self, cat96):
        """cat96: _Map [B,h,w,96] = [cost_reg (64 depth planes as channels) | mono features (32)] -> depth [B,4h,4w,1]."""
        P, T, d = self._pack(), self.T, self.device
        x = cat96
        for i, blk in enumerate(self.dn.decoders2):
            if blk.upscale:
                x = T._resize(x, _Map.empty(x.N, 2 * x.H, 2 * x.W, x.C, d), False)
            act = 'lrelu' if blk.act else 'none'
            a = P[f'd2.{i}.a']
            x = T._conv(a, x, _Map.empty(x.N, x.H, x.W, a.cout, d), act=act)
            x = T._conv(P[f'd2.{i}.b'], x, _Map.empty(x.N, x.H, x.W, a.cout, d), act=act)
        return x.buf

    # ------------------------------------------------------------------ CostVolumeInitNet.forward (init_net.py:578-650)
    def __call__(self, ref, src, is_train=False, mono=None):
        net, dn, c, d, T = self.net, self.dn, self.net.cfg, self.device, self.T
        P = self._pack()
        dh, dw = c['depth_height'], c['depth_width']
        h, w = dh // 4, dw // 4
        ref_s = F.interpolate(ref['imgs'].float(), (dh, dw), mode='bilinear')
        src_s = F.interpolate(src['imgs'].float(), (dh, dw), mode='bilinear')
        # res_net (the init net's ResUNetLight on the reference images) depends on nothing else in the stage and, like the
        # feature net, is a chain of ~30 small launches that cannot fill the chip: it runs on a side stream next to the
        # feature net / cost volume and is joined in front of out_conv (both streams end up in the same hipGraph capture).
        main = torch.cuda.current_stream(d)
        side = self.__dict__.get('_side')
        if side is None:
            side = self.__dict__['_side'] = torch.cuda.Stream(device=d)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            res_feats = T.image_encoder(ref_s)
            ref_s.record_stream(side)
        B = ref_s.shape[0]
        if int(c.get('seq_len', 3)) > 3:
            # construct_input_data_mv (init_net.py:183-231): per reference view [ref, src[nn_ids]]; the reference is view 0.
            # nn_ids as a device tensor keeps the gather free of host copies (hipGraph capture)
            nn_ids = torch.as_tensor(ref['nn_ids']).to(d).long().view(B, -1)
            gather = lambda a, b: torch.cat([a.unsqueeze(1), b[nn_ids]], 1)
            views = gather(ref_s, src_s)                                          # [B,S,3,dh,dw]
            rots = gather(ref['rots'].float(), src['rots'].float())
            trans = gather(ref['trans'].float(), src['trans'].float())
            S = views.shape[1]
            ref_mu, x_d3 = dn._mono(views, (0,), mono, h, w)
            dv = dn._hypotheses(ref_mu.contiguous(), c['mvs_min_depth'], c['mvs_max_depth'])
            feats = self.equi(views.reshape(B * S, 3, dh, dw)).view(B, S, h, w, 32)
            vol = CV.calculate_cost_volume_erp_multiview(dn.args, feats, None, trans, rots, depth_volume=dv, cost_type='abs_diff',
                                                         curr_idx=0)
        else:
            pair = torch.stack([src_s, ref_s], 1)                                # [B,2,3,dh,dw]: 0 = source, 1 = reference
            rots = torch.stack([src['rots'], ref['rots']], 1).float()
            trans = torch.stack([src['trans'], ref['trans']], 1).float()
            ref_mu, x_d3 = dn._mono(pair, (1, 0), mono, h, w)
            dv = dn._hypotheses(ref_mu.contiguous(), c['mvs_min_depth'], c['mvs_max_depth'])
            feats = self.equi(pair.reshape(B * 2, 3, dh, dw)).view(B, 2, h, w, 32)
            vol = CV.calculate_cost_volume_erp(dn.args, feats, None, trans, rots, depth_volume=dv, cost_type='abs_diff')
        reg = dn._regularise(vol)                                                # [B,D,h,w,1]
        cat96 = _Map.empty(B, h, w, 96, d)
        cat96.buf[..., :64] = reg[..., 0].permute(0, 2, 3, 1)                    # depth planes become channels
        cat96.buf[..., 64:] = pack_map(x_d3)
        depth = torch.clamp_min(self.decode(cat96), 0.0)                          # [B,dh,dw,1]
        mvs_depth = depth.view(B, 1, dh, dw)                                      # one channel: NHWC == NCHW
        # ray features: res_net(resized reference image) ++ depth_conv(normalised inverse depth at 1/4) -> out_conv
        cat64 = _Map.empty(B, h, w, 64, d)
        main.wait_stream(side)
        cat64.buf[..., :32] = res_feats
        near_inv, far_inv = -1.0 / c['min_depth'], -1.0 / c['max_depth']
        dd = (-1.0 / torch.clamp(mvs_depth, min=1e-5) - near_inv) / (far_inv - near_inv)
        dd = F.interpolate(torch.clamp(dd, min=0, max=1.0), scale_factor=0.25, mode='bilinear', align_corners=False)
        d4 = torch.zeros(B, h, w, 4, device=d)
        d4[..., 0] = dd[:, 0]
        cat64.buf[..., 32:] = T.run_head(P, 'dc', net.depth_conv, _Map(d4))
        ray = T.run_head(P, 'oc', net.out_conv, cat64)
        return {'mvs_depth': mvs_depth, 'ray_feats': ray.permute(0, 3, 1, 2)}
