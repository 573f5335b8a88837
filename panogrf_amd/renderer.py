"""Drop-in counterpart of the reference's renderer modules (network/renderer.py) for inference.

`name2network['neuray_gen'|'neuray_base'](cfg)` builds an nn.Module with the reference's sub-module and
parameter names (`image_encoder`, `vis_encoder`, `dist_decoder`, `agg_net.agg_impl.*`, `agg_net.prob_embed`,
`fine_dist_decoder`, `fine_agg_net`, `sph_fitter.regs`), so `load_state_dict(ckpt['network_state_dict'])`
works unchanged, and `renderer(data)` / `render(que_imgs_info, ref_imgs_info, is_train)` return the same
dict keys and shapes (network/renderer.py:635-686, 777-786).  The per-frame 2-D encoders run on stock
PyTorch-ROCm; everything per ray goes to the HIP library.  The Linear layers below are parameter HOLDERS:
their forward is never called.  Training mode (`'eval' not in data`): the reference's random sampling and
extra outputs; under autograd the per-ray passes are torch.autograd Functions whose backward is `pgrf_render_pass_backward`
(panogrf_amd.engine), the 2-D encoders train on torch.  `name2network['neuray_ft']` is the per-scene fine-tuning module
(network/renderer.py:788-1200) on the same fast path; the reference's losses are in panogrf_amd/loss.py.
"""
import itertools
import math

import numpy as np

import torch
import torch.nn as nn

from ._lib import PgrfError
from . import engine as E
from .encoders import ResUNetLight, DefaultVisEncoder


def _mlp(dims, acts):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        layers.append(acts[i]())
    return nn.Sequential(*layers)


def _interpolate_feats(feats, points):
    """interpolate_feats(feats [b,f,h,w], points [b,n,2] in pixels, align_corners=True) -> [b,n,f]: bilinear,
    zeros padding (network/ops.py:32-52).  Per-frame ground-truth lookup only, not on the per-ray path."""
    h, w = feats.shape[-2:]
    grid = torch.stack([points[..., 0] / (w - 1) * 2 - 1, points[..., 1] / (h - 1) * 2 - 1], -1).unsqueeze(1)
    out = torch.nn.functional.grid_sample(feats, grid, mode='bilinear', padding_mode='zeros', align_corners=True)
    return out.squeeze(2).permute(0, 2, 1)


class _AddBias(nn.Module):            # index placeholder for network/ops.py:119-125 (no parameters)
    def __init__(self, val=0.05):
        super().__init__()
        self.val = val


class MixtureLogisticsDistDecoder(nn.Module):
    """Parameter holder; layout of network/dist_decoder.py:53-97."""
    default_cfg = {'feats_dim': 32, 'bias_val': 0.05, 'use_vis': True}

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        if self.cfg['feats_dim'] != 32 or abs(self.cfg['bias_val'] - 0.05) > 1e-12:
            raise PgrfError('dist decoder: only feats_dim=32, bias_val=0.05 are implemented')
        self.mean_decoder = _mlp([32, 32, 32, 2], [nn.ELU, nn.ELU, nn.Softplus])
        self.var_decoder = _mlp([32, 32, 32, 2], [nn.ELU, nn.ELU, nn.Softplus])
        self.var_decoder.add_module('6', _AddBias(0.05))
        self.aw_decoder = _mlp([32, 32, 32, 1], [nn.ELU, nn.ELU, nn.Sigmoid])
        if self.cfg['use_vis']:
            self.vis_decoder = _mlp([32, 32, 32, 1], [nn.ELU, nn.ELU, nn.Sigmoid])


class _RayAttention(nn.Module):
    def __init__(self):
        super().__init__()
        self.w_qs = nn.Linear(16, 16, bias=False)
        self.w_ks = nn.Linear(16, 16, bias=False)
        self.w_vs = nn.Linear(16, 16, bias=False)
        self.fc = nn.Linear(16, 16, bias=False)
        self.layer_norm = nn.LayerNorm(16, eps=1e-6)


class IBRNetWithNeuRay(nn.Module):
    """Parameter holder; layer shapes and init of network/ibrnet.py:239-300."""

    def __init__(self, neuray_in_dim=32, in_feat_ch=32, n_samples=64):
        super().__init__()
        if neuray_in_dim != 32 or in_feat_ch != 32:
            raise PgrfError('aggregation net: only neuray_dim=32, in_feat_ch=32 are implemented')
        self.n_samples = n_samples
        self.ray_dir_fc = _mlp([4, 16, 35], [nn.ELU, nn.ELU])
        self.base_fc = _mlp([207, 64, 32], [nn.ELU, nn.ELU])
        self.vis_fc = _mlp([32, 32, 33], [nn.ELU, nn.ELU])
        self.vis_fc2 = _mlp([32, 32, 1], [nn.ELU, nn.Sigmoid])
        self.geometry_fc = _mlp([65, 64, 16], [nn.ELU, nn.ELU])
        self.ray_attention = _RayAttention()
        self.out_geometry_fc = _mlp([16, 16, 1], [nn.ELU, nn.ReLU])
        self.rgb_fc = nn.Sequential(nn.Linear(37, 16), nn.ELU(), nn.Linear(16, 8), nn.ELU(), nn.Linear(8, 1))
        self.neuray_fc = nn.Sequential(nn.Linear(32, 8), nn.ELU(), nn.Linear(8, 1))
        for seq in [self.base_fc, self.vis_fc2, self.vis_fc, self.geometry_fc, self.rgb_fc, self.neuray_fc]:
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.kaiming_normal_(m.weight.data)
                    nn.init.zeros_(m.bias.data)


class DefaultAggregationNet(nn.Module):
    """Parameter holder; network/aggregate_net.py:16-39."""
    default_cfg = {'sample_num': 64, 'neuray_dim': 32, 'use_img_feats': False}

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        if self.cfg.get('level', 0) in [-1] or self.cfg.get('wo_geometry') or self.cfg.get('wo_appearance'):
            raise PgrfError('aggregation net: level=-1 / wo_geometry / wo_appearance are not implemented')
        self.agg_impl = IBRNetWithNeuRay(self.cfg['neuray_dim'], 32, self.cfg['sample_num'])
        self.prob_embed = nn.Sequential(nn.Linear(2 + 32, 32), nn.ReLU(), nn.Linear(32, 32))


class _SphHolder(nn.Module):
    """Only the `regs` buffer of SphericalHarmonicsSolver(3) appears in checkpoints (network/sph_solver.py:5-13);
    direct rendering itself is disabled in the reference."""

    def __init__(self):
        super().__init__()
        regs = [0.0] + [0.001] * 3 + [0.005] * 5 + [0.05] * 7
        self.register_buffer('regs', torch.tensor(regs, dtype=torch.float32))


def _snapshot(named):
    """[(name, value)] -> the same with each tensor's version counter: what `_unchanged` compares the next frame's inputs to.
    Holds the tensors themselves (no copies): an in-place write bumps the counter and invalidates the snapshot.  Returns
    None when an input is an inference tensor (torch.inference_mode: no version counter, so nothing can be kept)."""
    if any(torch.is_tensor(v) and v.is_inference() for _, v in named):
        return None
    return [(k, v, v._version if torch.is_tensor(v) else None) for k, v in named]


def _unchanged(snap, named, compare_contents=False):
    """True when every input is the snapshot's: the same tensor object at the same version counter (no device work).  With
    `compare_contents` a NEW tensor object of equal contents also counts (one device compare and a host sync per such
    tensor: cfg['hip_reuse_source_maps']='equal'); by default a new object means new data."""
    named = list(named)
    if snap is None or len(snap) != len(named):
        return False
    for (k0, v0, ver0), (k, v) in zip(snap, named):
        if k0 != k or torch.is_tensor(v0) != torch.is_tensor(v):
            return False
        if not torch.is_tensor(v):
            if v0 is not v and not np.array_equal(np.asarray(v0), np.asarray(v)):
                return False
            continue
        if v.is_inference() or v0._version != ver0:
            return False
        if v is not v0:
            if not compare_contents:
                return False
            if v.shape != v0.shape or v.dtype != v0.dtype or v.device != v0.device or not torch.equal(v, v0):
                return False
    return True


class NeuralRayBaseRenderer(nn.Module):
    base_cfg = {
        'vis_encoder_type': 'default', 'vis_encoder_cfg': {}, 'dist_decoder_type': 'mixture_logistics',
        'dist_decoder_cfg': {}, 'autoencoder': False, 'agg_net_type': 'default', 'agg_net_cfg': {},
        'use_hierarchical_sampling': False, 'fine_agg_net_cfg': {}, 'fine_dist_decoder_cfg': {},
        'fine_depth_sample_num': 64, 'fine_depth_use_all': False, 'ray_batch_num': 2048, 'depth_sample_num': 64,
        'alpha_value_ground_state': -15, 'use_dr_prediction': False, 'use_nr_color_for_dr': False,
        'use_self_hit_prob': False, 'use_ray_mask': True, 'ray_mask_view_num': 1, 'ray_mask_point_num': 8,
        'patchsize': (128, 128), 'render_depth': False,
    }

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.base_cfg, **cfg}
        c = self.cfg
        c['agg_net_cfg'], c['fine_agg_net_cfg'] = dict(c['agg_net_cfg']), dict(c['fine_agg_net_cfg'])
        if 'sample_num' in cfg:
            c['agg_net_cfg']['sample_num'] = c['sample_num']
            c['fine_agg_net_cfg']['sample_num'] = c['sample_num']
        for key in ('level', 'wo_geometry', 'wo_appearance'):
            if key in cfg:
                c['agg_net_cfg'][key] = cfg[key]
                c['fine_agg_net_cfg'][key] = cfg[key]
        if c.get('dataset_name', 'm3d') not in E.DATASETS:
            raise PgrfError(f"dataset_name {c.get('dataset_name')!r}: spherical convention not implemented "
                            f"(have {sorted(E.DATASETS)}; network/spt_utils.py)")
        for key in ('debug', 'diner_depth_guided_sampling', 'autoencoder'):
            if c.get(key):
                raise PgrfError(f"cfg['{key}'] is not implemented on the HIP path")
        if c.get('render_uncert') and not c['render_depth']:
            raise PgrfError("cfg['render_uncert'] needs cfg['render_depth'] (the reference reads outputs['render_depth'], renderer.py:312)")
        if c.get('local_feature_type') == 'ERP+TP':
            raise PgrfError('ERP+TP image encoder is out of scope')
        self.vis_encoder = DefaultVisEncoder(c)
        self.dist_decoder = MixtureLogisticsDistDecoder(c['dist_decoder_cfg'])
        self.image_encoder = ResUNetLight(c, 3, [1, 2, 6, 4], 32, inplanes=16, use_wrap_padding=c.get('use_wrap_padding', True))
        self.agg_net = DefaultAggregationNet(c['agg_net_cfg'])
        if c['use_hierarchical_sampling'] and not c.get('one_mlp', False):
            self.fine_dist_decoder = MixtureLogisticsDistDecoder(c['fine_dist_decoder_cfg'])
            self.fine_agg_net = DefaultAggregationNet(c['fine_agg_net_cfg'])
        self.sph_fitter = _SphHolder()
        self._rr = None
        self._rr_key = None

    # ------------------------------------------------------------------ HIP engine plumbing
    def _per_ray_state(self):
        keep = ('dist_decoder.', 'agg_net.', 'fine_dist_decoder.', 'fine_agg_net.')
        return {k: v for k, v in self.state_dict().items() if k.startswith(keep)}

    def _engine(self, device):
        sd = self._per_ray_state()
        key = (tuple((k, v._version, v.data_ptr()) for k, v in sd.items()), str(device))
        if self._rr is None or self._rr_key != key:
            ecfg = dict(self.cfg)
            ecfg['dist_decoder_cfg'] = {'use_vis': self.dist_decoder.cfg['use_vis']}
            if hasattr(self, 'fine_dist_decoder'):
                ecfg['fine_dist_decoder_cfg'] = {'use_vis': self.fine_dist_decoder.cfg['use_vis']}
            # the reference adds pos_encoding [1,n_samples,16] to [rn,dn,16]: dn must equal sample_num (ibrnet.py:358)
            ns = self.agg_net.agg_impl.n_samples
            dn_f = (ecfg['depth_sample_num'] + ecfg['fine_depth_sample_num']) if ecfg['fine_depth_use_all'] else ecfg['fine_depth_sample_num']
            if ecfg['depth_sample_num'] != ns or (ecfg['use_hierarchical_sampling'] and dn_f != ns):
                raise PgrfError(f'samples per pass must equal agg_net sample_num={ns} (positional encoding)')
            self._rr = E.RayRenderer(ecfg, sd, device=device)
            self._rr_key = key
        return self._rr

    def _check_train(self, is_train):
        """Training mode: the reference's random importance samples and hit_prob_* outputs; with autograd
        enabled the per-ray passes run as torch.autograd Functions whose backward is pgrf_render_pass_backward
        (panogrf_amd.engine.RenderPassFunction) and the 2-D encoders stay on torch.  What has no backward (the split-f16 mode) is refused."""
        if is_train and torch.is_grad_enabled():
            if self.cfg.get('hip_precision', 'f32') != 'f32':
                raise NotImplementedError("training runs in exact fp32 (cfg['hip_precision']='f32')")

    # ------------------------------------------------------------------ reference interface
    def render_impl(self, que_imgs_info, ref_imgs_info, is_train, is_perspec=False):
        """One ray batch (network/renderer.py:567-633).  `ref_imgs_info` must carry 'img_feats' and
        (vis-encoded) 'ray_feats' like after lines 641-642 of the reference."""
        self._check_train(is_train)
        imgs = ref_imgs_info['imgs']
        dev = imgs.device
        rr = self._engine(dev)
        if is_train and torch.is_grad_enabled():
            out = self._render_rays_autograd(que_imgs_info, ref_imgs_info, is_perspec)
        else:
            frame = ref_imgs_info.get('_pgrf_frame')
            if frame is None:
                frame = self._make_frame(que_imgs_info, ref_imgs_info, is_perspec)
            out = rr.render_rays(frame, que_imgs_info['coords'][0], is_train=bool(is_train))
        c = self.cfg
        res = {}
        rn = que_imgs_info['coords'].shape[1]
        for suffix in ([''] + (['_fine'] if c['use_hierarchical_sampling'] else [])):
            dn = out['density_nr' + suffix].shape[1]
            for k in ('pixel_colors_nr', 'hit_prob_nr', 'colors_nr', 'density_nr'):
                res[k + suffix] = out[k + suffix].unsqueeze(0)
            gt_key = 'imgs' if 'imgs' in que_imgs_info else ('cube_imgs' if 'cube_imgs' in que_imgs_info else None)
            if gt_key is not None:      # ground-truth colours of the query pixels (renderer.py:277-288)
                res['pixel_colors_gt' + suffix] = _interpolate_feats(que_imgs_info[gt_key], que_imgs_info['coords'])
                if c.get('use_polar_weighted_loss') and suffix == '' and gt_key == 'imgs':       # renderer.py:283-285
                    res['polar_weights'] = _interpolate_feats(que_imgs_info['polar_weights'], que_imgs_info['coords'])
            if c['use_ray_mask']:
                ok = (imgs.shape[0] >= c['ray_mask_view_num']) and (dn > c['ray_mask_point_num'])
                res['ray_mask' + suffix] = torch.full((1, rn), bool(ok), dtype=torch.bool, device=dev)
            if c['render_depth']:
                res['render_depth' + suffix] = out['render_depth' + suffix].unsqueeze(0)
            if c.get('render_uncert'):       # sum_i w_i (z_i - depth)^2 + 1e-5 (renderer.py:305-312)
                res['render_uncert' + suffix] = out['render_uncert' + suffix].unsqueeze(0)
        return res

    def _render_rays_autograd(self, que, ref, is_perspec):
        """render_impl's per-ray part under autograd (network/renderer.py:567-633 as the trainer runs it): deterministic
        coarse depths (:607 passes random_sample=False), coarse pass, importance sampling from the DETACHED hit
        probabilities with torch.rand draws (:437, render_ops.py:449), fine pass.  Gradients reach the module's own
        parameters and ref['img_feats'] / ref['ray_feats'] (the torch encoders' outputs)."""
        c = self.cfg
        imgs = ref['imgs']
        dev = imgs.device
        coords = que['coords'][0].detach().to(dev, torch.float32).contiguous()
        rn, dn = coords.shape[0], int(c['depth_sample_num'])
        params = dict(self.named_parameters())
        kw = dict(dataset_name=c.get('dataset_name', 'm3d'))
        que_c2w = None
        if is_perspec:
            if que['poses'].shape[0] != 1:
                raise PgrfError('one query view per call (qn = 1)')
            kw.update(que_pose=que['poses'], que_K=que['Ks'])
        else:
            if que['c2w'].shape[0] != 1:
                raise PgrfError('que_imgs_info c2w.shape[0]=1')
            que_c2w = que['c2w']
        use_disp = bool(c.get('use_disp', True))
        depth = E.sample_coarse_depth(c['min_depth'], c['max_depth'], rn, dn, use_disp=use_disp, device=dev)
        common = (imgs, ref['img_feats'], ref['ray_feats'], ref['w2c'], ref['depth_range'], que_c2w, que['depth_range'], params)
        out = E.render_pass_autograd(*common, 'dist_decoder', 'agg_net', self.dist_decoder.cfg['use_vis'], coords, depth, **kw)
        out['depth'] = depth
        if c['use_hierarchical_sampling']:
            fdn = int(c['fine_depth_sample_num'])
            u = torch.rand([1, rn, fdn])[0].to(dev)
            fd = E.sample_fine_depth(depth, out['hit_prob_nr'].detach(), E.host_floats(que['depth_range'], 2), fdn, use_disp=use_disp,
                                     use_all=bool(c['fine_depth_use_all']), u=u)
            if c.get('one_mlp', False):
                pre, vis_f = ('dist_decoder', 'agg_net'), self.dist_decoder.cfg['use_vis']
            else:
                pre, vis_f = ('fine_dist_decoder', 'fine_agg_net'), self.fine_dist_decoder.cfg['use_vis']
            fine = E.render_pass_autograd(*common, pre[0], pre[1], vis_f, coords, fd, **kw)
            fine['depth'] = fd
            if c.get('render_c2f_all'):      # one composite over the sorted union of both passes' samples (renderer.py:484-518)
                fine.update(E.merge_c2f_autograd(depth, out['colors_nr'], out['density_nr'], fd, fine['colors_nr'], fine['density_nr']))
                # the merged composite's depths (constants of the graph): what render_uncert below pairs with the merged hit
                # probabilities (renderer.py:492,515); equal depths give equal terms, so the order among ties does not matter
                fine['depth'] = torch.sort(torch.cat([depth, fd], -1), dim=-1, stable=True)[0]
            for k, v in fine.items():
                out[k + '_fine'] = v
        if c.get('render_uncert'):        # renderer.py:305-312, in torch so that it stays differentiable
            for sfx in [''] + (['_fine'] if c['use_hierarchical_sampling'] else []):
                out['render_uncert' + sfx] = ((out['depth' + sfx] - out['render_depth' + sfx].unsqueeze(-1)).pow(2) * out['hit_prob_nr' + sfx]).sum(-1) + 1e-5
        return out

    def _torch_encoders(self, imgs, ray_feats):
        feats = self.image_encoder(imgs)
        return feats, self.vis_encoder(ray_feats, feats)

    def _encoder_fn(self, imgs):
        """The per-frame encoders run on the HIP library (panogrf_amd/encoders_hip.py) when the input is
        on the GPU, autograd is off and the frame size allows it (H, W multiples of 16); cfg['hip_encoders']=False, or
        any other case, keeps the same modules on PyTorch-ROCm / MIOpen."""
        h, w = imgs.shape[-2:]
        if not (imgs.is_cuda and self.cfg.get('hip_encoders', True)) or torch.is_grad_enabled() or h % 16 or w % 16:
            return self._torch_encoders
        he = self.__dict__.get('_hip_enc')
        if he is None or he.device != imgs.device:
            from .encoders_hip import HipEncoders
            he = self.__dict__['_hip_enc'] = HipEncoders(self.image_encoder, self.vis_encoder, imgs.device)
        return he

    def _graphed(self, tag, fn, inputs, modules=()):
        """Run fn(*inputs) (CUDA tensors in, tuple of CUDA tensors out, static shapes, no host sync inside) as a replayed
        hipGraph: captured once per (tag, input shapes), replayed on copies of the inputs; outputs are cloned because the
        next replay overwrites the graph's buffers.  `modules`: the nn.Modules fn reads weights from -- the HIP paths
        keep packed copies of them, which a replay would not refresh, so any in-place update or re-assignment of a
        parameter / buffer (load_state_dict, an optimiser step) drops the capture and takes a new one."""
        key = (tag,) + tuple((tuple(t.shape), t.dtype, str(t.device)) for t in inputs)
        version = hash(tuple((v._version, v.data_ptr()) for m in modules for v in itertools.chain(m.parameters(), m.buffers())))
        graphs = self.__dict__.setdefault('_graphs', {})
        ent = graphs.get(key)
        if ent is not None and ent[3] != version:
            ent = None
            del graphs[key]
        if ent is None:
            dev = inputs[0].device
            statics = [t.clone() for t in inputs]
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):                      # warm-up outside capture (weight packing, MIOpen find, allocations)
                    fn(*statics)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                outs = fn(*statics)
            ent = graphs[key] = (g, statics, tuple(outs), version)
        g, statics, outs, _ = ent
        for s_, t in zip(statics, inputs):
            s_.copy_(t)
        g.replay()
        return tuple(o.clone() for o in outs)

    def _enc_shard(self, imgs):
        """Multi-GPU: the image encoder runs once per SOURCE view, each independent of the others, so under a
        process group the views are dealt to the ranks -- to the ranks the MVS stage leaves idle when the node has at least
        2 * rfn of them (panogrf_amd.dist.encoder_offset), so that the two stages run side by side -- and the feature maps
        (4 MB per 512x1024 view) are all-gathered; the small visibility encoder stays replicated.  -> (mine, offset) or None.
        cfg['hip_shard_encoders']=False keeps every rank on all views."""
        import torch.distributed as tdist
        # (never in training mode, like the MVS stage's view shards: a data-parallel job whose ranks each hold their own batch and
        #  run a training-mode forward under no_grad must not mix each other's feature maps, nor hang in an all-gather only some
        #  ranks reach)
        if not (self.cfg.get('hip_shard_encoders', True) and imgs.is_cuda and not torch.is_grad_enabled() and
                not self.__dict__.get('_pgrf_is_train', False) and
                tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1 and imgs.shape[0] > 1):
            return None
        from . import dist as D
        rank, world, n = tdist.get_rank(), tdist.get_world_size(), imgs.shape[0]
        off = D.encoder_offset(world, n)
        return D.view_shard(rank, world, n, off), off

    def _encode_sharded(self, imgs, ray_feats, shard):
        """Image encoder on this rank's views only (started early on a side stream when the MVS stage runs in front), one
        all-gather of the feature maps, visibility encoder on every rank."""
        from . import dist as D
        mine, off = shard
        dev = imgs.device
        n, _, H, W = imgs.shape
        enc = self._encoder_fn(imgs)
        hip = enc != self._torch_encoders
        early = self.__dict__.pop('_early_feats', None)
        if early is not None and early[0] is imgs:
            local, side = early[1], early[2]
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
                if local.numel():
                    local.record_stream(torch.cuda.current_stream(dev))
        else:
            local = self._encode_local_views(imgs, mine)
        feats = D.allgather_views(local.contiguous(), n, offset=off)
        use_graph = self.cfg.get('hip_graph_encoders', True)
        vis = (lambda rf, f: (enc.vis_encoder(rf, f.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2),)) if hip else \
              (lambda rf, f: (self.vis_encoder(rf, f),))
        if use_graph:
            return feats, self._graphed('enc_vis_hip' if hip else 'enc_vis_torch', vis, (ray_feats.contiguous(), feats), modules=(self.vis_encoder,))[0]
        return feats, vis(ray_feats.contiguous(), feats)[0]

    def _encode_local_views(self, imgs, mine):
        """image_encoder(imgs[mine]) -> [len(mine),32,H/4,W/4] (contiguous NCHW; an empty tensor for a rank without views)."""
        dev = imgs.device
        n, _, H, W = imgs.shape
        if not mine:
            return torch.zeros(0, 32, H // 4, W // 4, device=dev)
        enc = self._encoder_fn(imgs)
        hip = enc != self._torch_encoders
        sub = imgs[torch.as_tensor(mine, device=dev)].contiguous()
        img_fn = (lambda x: (enc.image_encoder(x).permute(0, 3, 1, 2).contiguous(),)) if hip else (lambda x: (self.image_encoder(x),))
        if self.cfg.get('hip_graph_encoders', True):
            return self._graphed('enc_img_hip_l' if hip else 'enc_img_torch_l', img_fn, (sub,), modules=(self.image_encoder,))[0]
        return img_fn(sub)[0]

    def _encode(self, imgs, ray_feats):
        """The two per-frame 2-D encoders (network/renderer.py:639-642).  On a GPU their ~70 small kernels are
        launch-bound (3.4 ms eager for 2x512x1024), so the launch sequence is captured once per input shape into a
        hipGraph and replayed (1.9 ms); cfg['hip_graph_encoders']=False keeps eager launches."""
        shard = self._enc_shard(imgs)
        if shard is not None:
            return self._encode_sharded(imgs, ray_feats, shard)
        enc = self._encoder_fn(imgs)
        if not (imgs.is_cuda and self.cfg.get('hip_graph_encoders', True)) or torch.is_grad_enabled():
            return enc(imgs, ray_feats)
        early = self.__dict__.pop('_early_feats', None)
        if early is not None and early[0] is imgs:
            # the image encoder was started on a side stream in front of the MVS stage (_encode_images_early): join it and
            # run the visibility encoder alone
            feats, side = early[1], early[2]
            main = torch.cuda.current_stream(imgs.device)
            main.wait_stream(side)
            feats.record_stream(main)
            hip = enc != self._torch_encoders
            vis = (lambda rf, f: (enc.vis_encoder(rf, f.permute(0, 2, 3, 1)).permute(0, 3, 1, 2),)) if hip else \
                  (lambda rf, f: (self.vis_encoder(rf, f),))
            return feats, self._graphed('enc_vis_hip' if hip else 'enc_vis_torch', vis, (ray_feats.contiguous(), feats), modules=(self.vis_encoder,))[0]
        return self._graphed('enc_torch' if enc == self._torch_encoders else 'enc_hip', enc, (imgs.contiguous(), ray_feats.contiguous()),
                             modules=(self.image_encoder, self.vis_encoder))

    @staticmethod
    def _weights_version(modules):
        return hash(tuple((v._version, v.data_ptr()) for m in modules for v in itertools.chain(m.parameters(), m.buffers())))

    def _reuse_enabled(self, imgs, is_train):
        """Source-map reuse (new; the reference recomputes per frame): a trajectory renders many query poses against the
        SAME source views, and everything in front of
        the per-ray kernel -- MVS stage, 2-D encoders: 8.5 of 84 ms at the shipped configuration -- depends on the source
        views and the weights only.  With cfg['hip_reuse_source_maps'] (default on; eval, GPU, no autograd) those results
        are kept and returned again while the inputs are unchanged (`_unchanged`) and the weights' version counters are.
        True (default) recognises the SAME tensor objects at the same version counters only -- no device work, no host
        sync: the caller keeps its ref / src infos across the frames of a trajectory (render.py of THIS repository does; the
        frame timings published with "source-map reuse" use this mode).  The reference's render.py:248-278 rebuilds the infos
        from disk for every frame: such a caller gets new tensor objects each frame, which this mode treats as new data (the
        stage is recomputed, as in the reference) -- 'equal' additionally accepts new tensors of equal contents, at one
        torch.equal = one sync per tensor and frame; False recomputes always.  Off under
        torch.inference_mode(): inference tensors have no version counter to watch."""
        return (bool(self.cfg.get('hip_reuse_source_maps', True)) and imgs.is_cuda and not is_train
                and not torch.is_grad_enabled() and not torch.is_inference_mode_enabled())

    def _reuse_by_contents(self):
        return self.cfg.get('hip_reuse_source_maps', True) == 'equal'

    def _encode_reusing(self, imgs, ray_feats, is_train):
        if not self._reuse_enabled(imgs, is_train):
            return self._encode(imgs, ray_feats)
        named = [('imgs', imgs), ('ray_feats', ray_feats)]
        wver = self._weights_version((self.image_encoder, self.vis_encoder))
        c = self.__dict__.get('_enc_cache')
        if (c is not None and c['wver'] == wver and all(o._version == ver for o, ver in c['outs']) and
                _unchanged(c['snap'], named, self._reuse_by_contents())):
            self.__dict__.pop('_early_feats', None)
            return c['outs'][0][0], c['outs'][1][0]
        outs = self._encode(imgs, ray_feats)
        snap = _snapshot(named)
        self.__dict__['_enc_cache'] = None if snap is None else dict(snap=snap, wver=wver, outs=[(o, o._version) for o in outs])
        return outs

    def _encode_images_early(self, imgs):
        """The image encoder depends on the source images only, the MVS stage in front of render() on nothing the encoder
        produces, and both are chains of small launches: start the encoder's hipGraph on a side stream so that it runs next
        to the MVS stage; _encode() joins it.  (GPU, no autograd, graphs enabled; otherwise a no-op.)"""
        if not (imgs.is_cuda and self.cfg.get('hip_graph_encoders', True) and self.cfg.get('hip_overlap_encoders', True)) or torch.is_grad_enabled():
            return
        shard = self._enc_shard(imgs)
        if shard is not None:       # multi-GPU: this rank's share of the views only (none on the ranks busy with the MVS stage)
            dev = imgs.device
            if not shard[0]:
                self.__dict__['_early_feats'] = (imgs, self._encode_local_views(imgs, []), None)
                return
            side = self.__dict__.get('_enc_stream')
            if side is None or side.device != dev:
                side = self.__dict__['_enc_stream'] = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                local = self._encode_local_views(imgs, shard[0])
            self.__dict__['_early_feats'] = (imgs, local, side)
            return
        enc = self._encoder_fn(imgs)
        hip = enc != self._torch_encoders
        dev = imgs.device
        side = self.__dict__.get('_enc_stream')
        if side is None or side.device != dev:
            side = self.__dict__['_enc_stream'] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        img_fn = (lambda x: (enc.image_encoder(x).permute(0, 3, 1, 2),)) if hip else (lambda x: (self.image_encoder(x),))
        with torch.cuda.stream(side):
            feats = self._graphed('enc_img_hip' if hip else 'enc_img_torch', img_fn, (imgs.contiguous(),), modules=(self.image_encoder,))[0]
        self.__dict__['_early_feats'] = (imgs, feats, side)

    def _make_frame(self, que, ref, is_perspec):
        """ERP query: que['c2w'] (render_ops.py:76-106).  Pinhole query (is_perspec, render_cubes.py): que['poses'] and
        que['Ks'] (render_ops.py:35-74)."""
        kw = dict(device=ref['imgs'].device, dataset_name=self.cfg.get('dataset_name', 'm3d'))
        if is_perspec:
            if que['poses'].shape[0] != 1:
                raise PgrfError('one query view per call (qn = 1)')
            return E.Frame(ref['imgs'], ref['img_feats'], ref['ray_feats'], ref['w2c'], ref['depth_range'], None,
                           que['depth_range'], que_pose=que['poses'], que_K=que['Ks'], **kw)
        if que['c2w'].shape[0] != 1:
            raise PgrfError('que_imgs_info c2w.shape[0]=1')
        return E.Frame(ref['imgs'], ref['img_feats'], ref['ray_feats'], ref['w2c'], ref['depth_range'], que['c2w'],
                       que['depth_range'], **kw)       # whole [1,...] tensors: the first view's values lead (host_floats)

    def render(self, que_imgs_info, ref_imgs_info, is_train, is_perspec=False):
        """Whole query view (network/renderer.py:635-686).  Mutates ref_imgs_info['img_feats'/'ray_feats'] and
        que_imgs_info['coords'] like the reference does."""
        self._check_train(is_train)
        self.__dict__['_pgrf_is_train'] = bool(is_train)
        ref_img_feats, ray_feats = self._encode_reusing(ref_imgs_info['imgs'], ref_imgs_info['ray_feats'], is_train)
        ref_imgs_info['img_feats'] = ref_img_feats
        ref_imgs_info['ray_feats'] = ray_feats
        ref = dict(ref_imgs_info)
        if not (is_train and torch.is_grad_enabled()):
            ref['_pgrf_frame'] = self._make_frame(que_imgs_info, ref, is_perspec)
        coords = que_imgs_info['coords']
        # Multi-GPU (new, the reference is single-GPU): under an initialised process group every rank renders a
        # contiguous chunk of the rays (whole row tiles for a full panorama) and ONE all-gather per output reassembles them.
        # cfg['hip_shard_rays']=False keeps every rank on all rays; cfg['hip_gather']='all' also gathers the per-sample
        # tensors (colors_nr, density_nr, hit_prob_nr: hundreds of MB per frame), the default 'image' returns per-ray keys only.
        shard = self._ray_shard_world()
        n_all = coords.shape[1]
        if shard is not None and n_all < shard[1]:
            shard = None        # fewer rays than ranks: a rank without rays would produce no keys to gather -> every rank renders all
        if shard is not None:
            from . import dist as D
            r0, r1 = D.ray_shard(shard[0], shard[1], n_all)
            coords = coords[:, r0:r1]
        rb = self.cfg['ray_batch_num']
        # rays are independent: the reference's batch loop (renderer.py:656-683) only bounds memory, so by default
        # batches are merged up to `hip_ray_batch_num` rays per launch (results are batching-invariant)
        rb = max(rb, int(self.cfg.get('hip_ray_batch_num', 1 << 20)))
        parts = {}
        for s in range(0, coords.shape[1], rb):
            que_imgs_info['coords'] = coords[:, s:s + rb]
            info = self.render_impl(que_imgs_info, ref, is_train, is_perspec)
            for k, v in info.items():
                if k.startswith('hit_prob') and not is_train:        # kept in training mode only (renderer.py:675)
                    continue
                parts.setdefault(k, []).append(v)
        out = {k: torch.cat(v, 1) if len(v) > 1 else v[0] for k, v in parts.items()}
        if shard is not None:
            per_ray = ('pixel_colors', 'render_depth', 'render_uncert', 'ray_mask')
            keep = out if self.cfg.get('hip_gather', 'image') == 'all' else {k: v for k, v in out.items() if k.startswith(per_ray)}
            out = {k: D.allgather_rays(v, n_all, 1) for k, v in keep.items()}
        return out

    def _ray_shard_world(self):
        """(rank, world) when rays are to be sharded over the default process group, else None."""
        import torch.distributed as tdist
        if not (self.cfg.get('hip_shard_rays', True) and tdist.is_available() and tdist.is_initialized()) or torch.is_grad_enabled():
            return None          # (under autograd every rank renders its own batch: data parallelism is the trainer's, not ours)
        world = tdist.get_world_size()
        return (tdist.get_rank(), world) if world > 1 else None

    def forward(self, data, is_perspec=False):
        ref = dict(data['ref_imgs_info'])
        que = dict(data['que_imgs_info'])
        return self.render(que, ref, 'eval' not in data, is_perspec)      # is_train = 'eval' not in data (renderer.py:780)


class NeuralRayGenRenderer(NeuralRayBaseRenderer):
    """Generalisable renderer (network/renderer.py:688-786).  With cfg['init_net_type'] == 'cost_volume' (the shipped
    yaml) the init net is panogrf_amd.mvs.CostVolumeInitNet — the mono-guided 360-degree MVS stage of
    network/init_net.py + network/omni_mvsnet/pipeline3_model.py with the reference's parameter names, hypotheses / cost
    volume / 3-D regulariser on the HIP library; its mono-depth net (UniFuse, out of scope)
    is passed as `d_net=`.  Any other module with the reference's call signature
    `init_net(ref_imgs_info, src_imgs_info, is_train) -> {'ray_feats','mvs_depth',...}` can be given as `init_net=`."""
    default_cfg = {'init_net_type': 'depth', 'init_net_cfg': {}, 'use_depth_loss': False, 'depth_loss_coords_num': 8192}

    def __init__(self, cfg, init_net=None, d_net=None):
        super().__init__({**self.default_cfg, **cfg})
        if init_net is not None:
            self.init_net = init_net
        elif self.cfg.get('init_net_type') == 'cost_volume' and 'depth_height' in self.cfg:
            # name2init_net[cfg['init_net_type']](cfg) of the reference (network/renderer.py:700); the mono-depth net is
            # not mirrored: pass it as d_net= (see panogrf_amd/mvs.py)
            from .mvs import CostVolumeInitNet
            self.init_net = CostVolumeInitNet(self.cfg, d_net)

    def render_call(self, que_imgs_info, ref_imgs_info, is_train, src_imgs_info=None, is_perspec=False):
        if not hasattr(self, 'init_net'):
            raise PgrfError('NeuralRayGenRenderer needs an init_net module (see class docstring); '
                            'or put ray_feats into ref_imgs_info and call render()')
        self.__dict__['_pgrf_is_train'] = bool(is_train)
        ret = self._init_net_reusing(ref_imgs_info, src_imgs_info, is_train)
        ref_imgs_info['ray_feats'] = ret['ray_feats']
        if 'mvs_depth' in ret:
            ref_imgs_info['mvs_depth'] = ret['mvs_depth']
        if 'mvs_uncert' in ret:                  # cfg['uncert_tune'] (network/renderer.py:710-711)
            ref_imgs_info['mvs_uncert'] = ret['mvs_uncert']
        return self.render(que_imgs_info, ref_imgs_info, is_train, is_perspec)

    def _init_net_reusing(self, ref, src, is_train):
        """MVS stage, or its kept result when the source views and the weights are those of the previous frame (see
        `_reuse_enabled`).  Entries this module writes itself (ray_feats, img_feats, mvs_depth) are not part of the key."""
        reuse = self._reuse_enabled(ref['imgs'], is_train)
        if reuse:
            mine = ('ray_feats', 'img_feats', 'mvs_depth', 'mvs_uncert')
            named = [('ref.' + k, ref[k]) for k in sorted(ref) if k not in mine and not k.startswith('_pgrf')]
            named += [('src.' + k, src[k]) for k in sorted(src or {})]
            wver = self._weights_version((self.init_net,)) if isinstance(self.init_net, nn.Module) else None
            c = self.__dict__.get('_init_cache')
            if (c is not None and wver is not None and c['wver'] == wver and all(o._version == ver for o, ver in c['outs']) and
                    _unchanged(c['snap'], named, self._reuse_by_contents())):
                return dict(c['ret'])
        self._encode_images_early(ref['imgs'])
        ret = self._run_init_net_sharded(ref, src, is_train)
        snap = _snapshot(named) if (reuse and wver is not None) else None
        if snap is not None:
            self.__dict__['_init_cache'] = dict(snap=snap, wver=wver, ret=dict(ret),
                                                outs=[(o, o._version) for o in ret.values() if torch.is_tensor(o)])
        return ret

    def _run_init_net_sharded(self, ref, src, is_train):
        """Multi-GPU: the MVS stage builds one cost volume per reference view, independent of each other (init_net.py:164-171),
        so under a process group the views are dealt round-robin to the ranks (panogrf_amd.dist.view_shard) and the
        per-view results all-gathered (ray_feats + mvs_depth: < 1 MB per view).  cfg['hip_shard_views']=False disables."""
        from .mvs import CostVolumeInitNet
        from . import dist as D
        import torch.distributed as tdist
        net = self.init_net
        B = ref['imgs'].shape[0]
        if not (self.cfg.get('hip_shard_views', True) and tdist.is_available() and tdist.is_initialized() and
                tdist.get_world_size() > 1 and B > 1 and isinstance(net, CostVolumeInitNet) and src is not None and not is_train and
                not net.cfg.get('uncert_tune')):          # (the sharded / captured forms below return ray_feats and mvs_depth only)
            return self._run_init_net(ref, src, is_train)
        rank, world = tdist.get_rank(), tdist.get_world_size()
        mine = D.view_shard(rank, world, B)
        dev = ref['imgs'].device
        dh, dw = net.cfg['depth_height'], net.cfg['depth_width']
        if mine:
            idx = torch.as_tensor(mine, device=dev)
            ref_l = {k: ref[k][idx] for k in ('imgs', 'rots', 'trans')}
            if int(net.cfg.get('seq_len', 3)) > 3:          # multi-view: [ref_i, src[nn_ids_i]] needs all source views
                nn = torch.as_tensor(ref['nn_ids']).long().view(B, -1)
                ref_l['nn_ids'] = nn[torch.as_tensor(mine)]
                src_l = src
            else:                                            # two-view: source i pairs with reference i
                src_l = {k: src[k][idx] for k in ('imgs', 'rots', 'trans')}
            out = self._run_init_net(ref_l, src_l, is_train)
            rf, dp = out['ray_feats'].float().contiguous(), out['mvs_depth'].float().contiguous()
        else:
            rf = torch.zeros(0, 32, dh // 4, dw // 4, device=dev)
            dp = torch.zeros(0, 1, dh, dw, device=dev)
        return {'ray_feats': D.allgather_views(rf, B), 'mvs_depth': D.allgather_views(dp, B)}

    def _run_init_net(self, ref, src, is_train):
        """The MVS stage is ~150 small launches (2-D CNNs on MIOpen + the HIP cost-volume / regulariser kernels) with static
        shapes: like the encoders it is captured into a hipGraph once per input shape and replayed
        (cfg['hip_graph_init_net']=False keeps eager launches; panogrf_amd.mvs.CostVolumeInitNet only, two-view or multi-view: a
        user-supplied init net, or a mono net that syncs with the host, cannot be captured)."""
        from .mvs import CostVolumeInitNet
        net = self.init_net
        mv = int(net.cfg.get('seq_len', 3)) > 3 if isinstance(net, CostVolumeInitNet) else False
        ok = (isinstance(net, CostVolumeInitNet) and src is not None and (not mv or 'nn_ids' in ref) and
              ref['imgs'].is_cuda and not torch.is_grad_enabled() and self.cfg.get('hip_graph_init_net', True) and
              not net.cfg.get('hip_check_nan') and not net.cfg.get('uncert_tune'))
        if not ok:
            return net(ref, src, is_train)
        run = net
        if self.cfg.get('hip_init_net', True) and net.cfg['depth_height'] % 32 == 0 and net.cfg['depth_width'] % 32 == 0:
            # the stage's 2-D networks on the HIP library as well (panogrf_amd/mvs_hip.py)
            run = self.__dict__.get('_hip_init')
            if run is None or run.device != ref['imgs'].device or run.net is not net:
                from .mvs_hip import HipInitNet
                run = self.__dict__['_hip_init'] = HipInitNet(net, ref['imgs'].device)

        def fn(ri, rr, rt, si, sr, st, nn_ids=None):
            r = {'imgs': ri, 'rots': rr, 'trans': rt}
            if nn_ids is not None:
                r['nn_ids'] = nn_ids
            o = run(r, {'imgs': si, 'rots': sr, 'trans': st}, False)
            return o['ray_feats'], o['mvs_depth']
        f = lambda t: t.detach().float().contiguous()
        inputs = (f(ref['imgs']), f(ref['rots']), f(ref['trans']), f(src['imgs']), f(src['rots']), f(src['trans']))
        if mv:      # neighbour ids as a device tensor: the [ref, src[nn_ids]] gather then has no host copy inside the capture
            inputs += (torch.as_tensor(ref['nn_ids']).long().to(ref['imgs'].device).view(ref['imgs'].shape[0], -1),)
        ray_feats, depth = self._graphed('init_net_hip' if run is not net else 'init_net', fn, inputs, modules=(net,))
        return {'ray_feats': ray_feats, 'mvs_depth': depth}

    def forward(self, data, is_perspec=False):          # renderer(data, is_perspec=True): render_cubes.py:274
        ref = dict(data['ref_imgs_info'])
        que = dict(data['que_imgs_info'])
        src = dict(data['src_imgs_info']) if 'src_imgs_info' in data else None
        is_train = 'eval' not in data
        if 'ray_feats' in ref and not hasattr(self, 'init_net'):
            return self.render(que, ref, is_train, is_perspec)
        return self.render_call(que, ref, is_train, src, is_perspec)


def imgs_info_slice(imgs_info, indices):
    """utils/imgs_info.py:273-277"""
    return {k: v[indices] for k, v in imgs_info.items()}


def get_coords_mask(que_mask, train_ray_num, foreground_ratio):
    """Training pixels of a query view (utils/base_utils.py:585-604): `train_ray_num * foreground_ratio` random pixels inside the
    mask, the rest from everything left; draws from np.random like the reference."""
    min_pos_num = int(train_ray_num * foreground_ratio)
    y0, x0 = np.nonzero(que_mask)
    y1, x1 = np.nonzero(~que_mask)
    xy0 = np.stack([x0, y0], 1).astype(np.float32)
    xy1 = np.stack([x1, y1], 1).astype(np.float32)
    idx = np.arange(xy0.shape[0])
    np.random.shuffle(idx)
    xy0 = xy0[idx]
    coords0 = xy0[:min_pos_num]
    if min_pos_num < train_ray_num:
        xy1 = np.concatenate([xy1, xy0[min_pos_num:]], 0)
        idx = np.arange(xy1.shape[0])
        np.random.shuffle(idx)
        return np.concatenate([coords0, xy1[idx[:(train_ray_num - min_pos_num)]]], 0)
    return coords0


class NeuralRayFtRenderer(NeuralRayBaseRenderer):
    """Per-scene fine-tuning (network/renderer.py:788-1200): the source panoramas' raw visibility features `ray_feats` are
    PARAMETERS (one [1, dim, h/8, w/8] tensor per source view, initialised by the generalisable model's init net or from noise) and
    are optimised together with the encoders and the per-ray networks; `renderer(data)` without 'eval' is `train_step()` (a random
    source view becomes the query, `train_ray_num` of its pixels are rendered from the others), with 'eval' `validate_step(index)`.

    The reference reads its scene from a Habitat / Replica database inside the constructor (:826-871; datasets are out of scope, SURVEY.md
    section 2): here the scene is GIVEN -- `ref_imgs_info` (the source views in `ref_ids` order) and optionally `val_imgs_info`, dicts
    as built by panogrf_amd.imgs_info.build_imgs_info (`imgs, w2c, c2w, depth_range, masks`, ...; numpy or torch) -- and the
    generalisable model as an object (`gen_renderer=`, a NeuralRayGenRenderer with its init net) instead of a checkpoint path."""
    default_cfg = {
        'gen_cfg': None, 'use_validation': True, 'validate_initialization': False, 'init_src_view_num': 1, 'train_include_self': False,
        'ft_include_self': False, 'include_self_prob': 0.01, 'train_ray_num': 512, 'foreground_ratio': 1.0, 'ray_feats_res': [64, 128],
        'ray_feats_dim': 32, 'depth_guided_ray_sampling': False, 'has_depth': True,
    }

    def __init__(self, cfg, ref_imgs_info=None, val_imgs_info=None, gen_renderer=None):
        cfg = {**self.default_cfg, **cfg}
        cfg['ray_feats_res'] = [cfg['height'] // 8, cfg['width'] // 8]                       # :822
        super().__init__(cfg)
        if ref_imgs_info is None:
            raise PgrfError('NeuralRayFtRenderer: the scene is passed in as ref_imgs_info= (see class docstring); the reference\'s '
                            'Habitat / Replica readers are out of scope')
        for key in ('depth_guided_ray_sampling', 'use_self_hit_prob'):
            if self.cfg.get(key):
                raise PgrfError(f"cfg['{key}'] is not implemented on the HIP path")
        from .imgs_info import imgs_info_to_torch
        self.ref_imgs_info = imgs_info_to_torch(dict(ref_imgs_info))
        n_ref = self.ref_imgs_info['imgs'].shape[0]
        self.ref_ids = np.arange(n_ref)
        if self.cfg['use_validation'] and val_imgs_info is not None:
            self.val_imgs_info = imgs_info_to_torch(dict(val_imgs_info))
            self.val_num = self.val_imgs_info['imgs'].shape[0]
        else:
            self.val_imgs_info, self.val_num = None, 0
        self._initialization(gen_renderer)

    # ---- initialisation (network/renderer.py:918-1028)
    def _init_by_cost_volume(self, ref_i, init_net):
        ref = imgs_info_slice(self.ref_imgs_info, torch.tensor([ref_i]))
        src_index = [i for i in range(len(self.ref_ids)) if i != ref_i]
        src = imgs_info_slice(self.ref_imgs_info, torch.tensor(src_index))
        dev = next(init_net.parameters()).device
        with torch.no_grad():
            ret = init_net({k: v.to(dev) for k, v in ref.items()}, {k: v.to(dev) for k, v in src.items()}, False)
        out = [ret['ray_feats'].detach().cpu(), ret['mvs_depth'].detach().cpu()]
        if self.cfg.get('render_uncert'):
            out.append(ret['mvs_uncert'].detach().cpu())
        return out

    def _initialization(self, gen_renderer):
        self.ray_feats = nn.ParameterList()
        self.mvs_depths, self.mvs_uncert = [], []
        if gen_renderer is not None:
            gen_renderer = gen_renderer.eval()
            for ref_i in range(len(self.ref_ids)):
                got = self._init_by_cost_volume(ref_i, gen_renderer.init_net)
                self.ray_feats.append(nn.Parameter(got[0]))
                self.mvs_depths.append(got[1])
                if self.cfg.get('render_uncert'):
                    self.mvs_uncert.append(got[2])
            for name in ('vis_encoder', 'dist_decoder', 'agg_net', 'sph_fitter', 'image_encoder'):
                getattr(self, name).load_state_dict(getattr(gen_renderer, name).state_dict())
            if self.cfg['use_hierarchical_sampling'] and not self.cfg.get('one_mlp'):
                self.fine_dist_decoder.load_state_dict(gen_renderer.fine_dist_decoder.state_dict())
                self.fine_agg_net.load_state_dict(gen_renderer.fine_agg_net.state_dict())
        else:                                   # 'init from scratch !' (:1021-1027)
            fh, fw = self.cfg['ray_feats_res']
            for _ in range(len(self.ref_ids)):
                self.ray_feats.append(nn.Parameter(torch.randn(1, self.cfg['ray_feats_dim'], fh, fw)))

    # ---- batches (network/renderer.py:1029-1107)
    def slice_imgs_info(self, ref_idx, val_idx, is_train):
        dev = self.ray_feats[0].device
        ref = {k: v.to(dev) for k, v in imgs_info_slice(self.ref_imgs_info, torch.as_tensor(np.asarray(ref_idx)).long()).items()}
        ref['ray_feats'] = torch.cat([self.ray_feats[i] for i in ref_idx], 0)
        if self.mvs_depths:
            ref['mvs_depth'] = torch.cat([self.mvs_depths[i] for i in ref_idx], 0).to(dev)
            if self.cfg.get('render_uncert'):
                ref['mvs_uncert'] = torch.cat([self.mvs_uncert[i] for i in ref_idx], 0).to(dev)
        if is_train:
            que = imgs_info_slice(self.ref_imgs_info, torch.tensor([val_idx]))
            if self.cfg.get('my_debug'):
                _, _, hn, wn = que['imgs'].shape
                coords = np.stack(np.meshgrid(np.arange(wn), np.arange(hn)), -1).reshape([1, -1, 2]).astype(np.float32)
            else:
                mask = que['masks'][0, 0].cpu().numpy() > 0
                coords = get_coords_mask(mask, self.cfg['train_ray_num'], self.cfg['foreground_ratio']).reshape([1, -1, 2])
        else:
            if self.val_imgs_info is None:
                raise PgrfError('NeuralRayFtRenderer: no validation views (val_imgs_info=)')
            que = imgs_info_slice(self.val_imgs_info, torch.tensor([val_idx]))
            _, _, hn, wn = que['imgs'].shape
            coords = np.stack(np.meshgrid(np.arange(wn), np.arange(hn)), -1).reshape([1, -1, 2]).astype(np.float32)
        que = {k: v.to(dev) for k, v in que.items()}
        que['coords'] = torch.from_numpy(coords).to(dev)
        if is_train and self.mvs_depths:
            que['mvs_depth'] = self.mvs_depths[val_idx]
            if self.cfg.get('render_uncert'):
                que['mvs_uncert'] = self.mvs_uncert[val_idx]
        return ref, que

    def validate_step(self, val_idx, step=0):
        ref_idx = list(range(len(self.ref_ids)))
        ref, que = self.slice_imgs_info(ref_idx, val_idx, False)
        with torch.no_grad():
            out = self.render(dict(que), dict(ref), False)
        ref.pop('ray_feats')
        out.update({'ref_imgs_info': ref, 'que_imgs_info': que})
        return out

    def train_step(self):
        n = len(self.ref_ids)
        que_i = np.random.randint(0, n)
        others = [i for i in range(n) if i != que_i]
        if self.cfg['ft_include_self'] and not (np.random.random() > self.cfg['include_self_prob']):
            ref_idx = list(range(n))
            np.random.shuffle(ref_idx)
        else:
            ref_idx = others
        ref, que = self.slice_imgs_info(ref_idx, que_i, True)
        out = self.render(dict(que), dict(ref), True)
        out.update({'que_imgs_info': que, 'ref_imgs_info': ref, 'que_id': que_i, 'all_info': self.ref_imgs_info})
        return out

    def render_pose(self, render_imgs_info, is_perspec=False):
        """Arbitrary query poses from all source views (network/renderer.py:1162-1187; render_cube_pose = is_perspec=True)."""
        dev = self.ray_feats[0].device
        ref = {k: v.to(dev) for k, v in self.ref_imgs_info.items()}
        ref['ray_feats'] = torch.cat(list(self.ray_feats), 0)
        with torch.no_grad():
            return self.render(render_imgs_info, ref, False, is_perspec=is_perspec)

    def render_cube_pose(self, render_imgs_info):
        return self.render_pose(render_imgs_info, is_perspec=True)

    def forward(self, data, is_perspec=False):
        if 'eval' not in data:
            return self.train_step()
        return self.validate_step(data['index'], data.get('step', 0))


name2network = {
    'neuray_gen': NeuralRayGenRenderer,
    'neuray_ft': NeuralRayFtRenderer,
    'neuray_base': NeuralRayBaseRenderer,
}
