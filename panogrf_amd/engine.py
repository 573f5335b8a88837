"""Host-side driver of the HIP per-ray renderer: owns no math, only buffers and calls.

torch is used for device memory and streams only; every tensor handed to the library is a
caller-owned, contiguous float32 CUDA(HIP) tensor passed as a raw pointer.
"""
import ctypes as C
import os
import threading

import torch

from ._lib import DATASETS, lib, PgrfError, PgrfFrame, PgrfPassOut, PgrfPassDebug, PgrfRenderCfg

FEAT_CH = 32


def _require_gpu():
    if not torch.cuda.is_available():
        raise PgrfError("panogrf_amd needs a ROCm GPU (gfx950); there is no CPU fallback")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t, device=None):
    t = t.detach()
    if device is not None:
        t = t.to(device)
    return t.to(torch.float32).contiguous()


def host_floats(t, n):
    """First n values of a small tensor as Python floats for a by-value kernel argument.  A tensor that
    panogrf_amd.imgs_info.imgs_info_to_torch moved to the GPU carries its host values (`_pgrf_host`), so no device->host copy
    and no host sync happens; any other GPU tensor is read back."""
    mirror = getattr(t, '_pgrf_host', None)
    if mirror is not None and not t.is_inference() and mirror[1] == t._version and mirror[0].numel() >= n:
        return [float(x) for x in mirror[0][:n]]
    return [float(x) for x in _f32c(t, 'cpu').reshape(-1)[:n]]


def raw_names():
    return [lib.pgrf_raw_name(i).decode() for i in range(lib.pgrf_raw_count())]


def pack_pass_weights(state_dict, dd_prefix, agg_prefix, use_vis, max_dn=128):
    """Pack one pass (dist decoder `dd_prefix`, aggregation net `agg_prefix`) of a reference
    state_dict into the MFMA-ordered blob (host tensor).  Key names: SURVEY.md Appendix A."""
    names = raw_names()
    keep = []          # keep the contiguous host copies alive during the call
    arr = (C.c_void_p * len(names))()
    present = []
    for i, n in enumerate(names):
        if n.startswith('dist_decoder.'):
            key = dd_prefix + n[len('dist_decoder'):]
        else:
            key = agg_prefix + n[len('agg_net'):]
        if key not in state_dict:
            if '.vis_decoder.' in key and not use_vis:
                arr[i] = None
                continue
            raise PgrfError(f"state_dict lacks '{key}'")
        present.append((i, state_dict[key].detach()))
    if present and all(t.is_cuda for _, t in present):
        # GPU-resident parameters (training: the weights change every optimizer step): packed ON the device, stream ordered, no
        # host round trip -- the host packer cost a device->host copy (a sync), 1-3 ms of CPU work and an upload per pass
        dev = present[0][1].device
        for i, t in present:
            t = _f32c(t, dev)
            keep.append(t)
            arr[i] = t.data_ptr()
        blob = torch.empty(lib.pgrf_blob_floats(), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            lib.check(lib.pgrf_pack_weights_device(arr, len(names), int(bool(use_vis)), int(max_dn), _ptr(blob), _stream()))
        return blob
    for i, t in present:
        t = t.to('cpu', torch.float32).contiguous()
        keep.append(t)
        arr[i] = t.data_ptr()
    blob = torch.empty(lib.pgrf_blob_floats(), dtype=torch.float32)
    lib.check(lib.pgrf_pack_weights(arr, len(names), int(bool(use_vis)), int(max_dn), C.c_void_p(blob.data_ptr())))
    return blob


def pack_map(t):
    """[n,C,h,w] CUDA tensor -> channel-last [n,h,w,Cpad] (Cpad=4 for rgb)."""
    _require_gpu()
    t = _f32c(t)
    n, c, h, w = t.shape
    cpad = (c + 3) // 4 * 4
    out = torch.empty(n, h, w, cpad, dtype=torch.float32, device=t.device)
    lib.check(lib.pgrf_pack_map(_ptr(t), n, c, h, w, _ptr(out), _stream()))
    return out


def _channel_last(t, dev):
    """[n,C,h,w] -> channel-last [n,h,w,C].  A tensor that already is an NCHW-shaped view of channel-last memory (what the
    HIP encoders return, or torch.channels_last) is taken as it is; anything else goes through pgrf_pack_map."""
    t = t.detach()
    if t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.dim() == 4 and t.shape[1] % 4 == 0:
        v = t.permute(0, 2, 3, 1)
        if v.is_contiguous():
            return v
    return pack_map(_f32c(t, dev))


class Frame:
    """Source-view data of one frame in the layout the kernels read (PgrfFrame)."""

    def __init__(self, imgs, img_feats, ray_feats, w2c, ref_depth_range, que_c2w, que_depth_range, device='cuda',
                 dataset_name='m3d', que_pose=None, que_K=None):
        """ERP query: que_c2w [3,4] (que_imgs_info['c2w']).  Pinhole query (is_perspec=True, render_cubes.py): pass
        que_pose [3,4] world->camera and que_K [3,3] instead (que_imgs_info['poses'], ['Ks']); coords are then pinhole
        pixels and que_c2w is derived exactly as network/render_ops.py:45-46 does (rot = R^T, trans = -rot @ t)."""
        _require_gpu()
        if dataset_name not in DATASETS:
            raise PgrfError(f'dataset_name {dataset_name!r}: spherical convention not implemented (have {sorted(DATASETS)})')
        self.dataset = DATASETS[dataset_name]
        self.perspective = que_K is not None
        self.que_kinv = [0.0] * 9
        if self.perspective:
            pose = torch.tensor(host_floats(que_pose, 12)).reshape(3, 4)
            rot = pose[:, :3].t().contiguous()
            trans = -rot @ pose[:, 3:]
            que_c2w = torch.cat([rot, trans], 1)
            self.que_kinv = [float(x) for x in torch.inverse(torch.tensor(host_floats(que_K, 9)).reshape(3, 3)).reshape(-1)]
        dev = torch.device(device)
        if img_feats.shape[1] != FEAT_CH or ray_feats.shape[1] != FEAT_CH or imgs.shape[1] != 3:
            raise PgrfError(f"expected imgs [rfn,3,H,W], img_feats/ray_feats [rfn,{FEAT_CH},h,w]")
        self.rfn, _, self.H, self.W = imgs.shape
        self.imgs = pack_map(_f32c(imgs, dev))
        self.img_feats = _channel_last(img_feats, dev)
        self.ray_feats = _channel_last(ray_feats, dev)
        self.w2c = _f32c(w2c, dev).reshape(self.rfn, 12)
        self.ref_dr = _f32c(ref_depth_range, dev).reshape(self.rfn, 2)
        self.que_c2w = host_floats(que_c2w, 12)
        self.que_dr = host_floats(que_depth_range, 2)
        self.device = dev

    def c_struct(self):
        f = PgrfFrame()
        f.rfn, f.H, f.W = self.rfn, self.H, self.W
        f.hf, f.wf = self.img_feats.shape[1:3]
        f.hr, f.wr = self.ray_feats.shape[1:3]
        f.imgs, f.img_feats, f.ray_feats = self.imgs.data_ptr(), self.img_feats.data_ptr(), self.ray_feats.data_ptr()
        f.w2c, f.ref_depth_range = self.w2c.data_ptr(), self.ref_dr.data_ptr()
        f.que_c2w = (C.c_float * 12)(*self.que_c2w)
        f.que_depth_range = (C.c_float * 2)(*self.que_dr)
        f.dataset, f.perspective = self.dataset, int(self.perspective)
        f.que_kinv = (C.c_float * 9)(*self.que_kinv)
        return f


_DEBUG_SHAPES = {
    'px': lambda v, r, d: (v, r, d, 2), 'depth_v': lambda v, r, d: (v, r, d), 'ddiff': lambda v, r, d: (v, r, d, 4),
    'vis': lambda v, r, d: (v, r, d), 'hit': lambda v, r, d: (v, r, d), 'ray_feats': lambda v, r, d: (v, r, d, 32),
    'rgb': lambda v, r, d: (v, r, d, 3), 'img_feats': lambda v, r, d: (v, r, d, 32),
    'neuray': lambda v, r, d: (r, d, v, 32), 'base_x': lambda v, r, d: (r, d, v, 32),
    'geo': lambda v, r, d: (r, d, 16), 'attn': lambda v, r, d: (r, d, 16),
}


def sample_coarse_depth(min_depth, max_depth, rn, dn, use_disp=True, device='cuda', jitter=None):
    """sample_depth (render_ops.py:292-339).  jitter [rn,dn-2] uniform draws = training mode (random_sample=True)."""
    _require_gpu()
    out = torch.empty(rn, dn, dtype=torch.float32, device=device)
    if jitter is not None:
        jitter = _f32c(jitter, device)
        if tuple(jitter.shape) != (rn, dn - 2):
            raise PgrfError(f'jitter must be [rn, dn-2] = [{rn}, {dn - 2}], got {tuple(jitter.shape)}')
    lib.check(lib.pgrf_sample_coarse_depth_train(float(min_depth), float(max_depth), int(use_disp), rn, dn, _ptr(jitter),
                                                 _ptr(out), _stream()))
    return out


def sample_fine_depth(depth, hit_prob, que_depth_range, fdn, use_disp=True, use_all=False, u=None):
    """sample_fine_depth + sort (render_ops.py:413-473, renderer.py:469-472).  u [rn,fdn] uniform draws = training mode."""
    _require_gpu()
    depth, hit_prob = _f32c(depth), _f32c(hit_prob)
    rn, dn = depth.shape
    out = torch.empty(rn, (dn + fdn) if use_all else fdn, dtype=torch.float32, device=depth.device)
    if u is not None:
        u = _f32c(u, depth.device)
        if tuple(u.shape) != (rn, fdn):
            raise PgrfError(f'u must be [rn, fdn] = [{rn}, {fdn}], got {tuple(u.shape)}')
    lib.check(lib.pgrf_sample_fine_depth_train(_ptr(depth), _ptr(hit_prob), float(que_depth_range[0]), float(que_depth_range[1]),
                                               int(use_disp), int(use_all), rn, dn, fdn, _ptr(u), _ptr(out), _stream()))
    return out


def _alloc_out(rn, dn, dev, nan_flag=None, uncert=False):
    t = dict(pixel_colors_nr=torch.empty(rn, 3, device=dev), render_depth=torch.empty(rn, device=dev),
             hit_prob_nr=torch.empty(rn, dn, device=dev), colors_nr=torch.empty(rn, dn, 3, device=dev),
             density_nr=torch.empty(rn, dn, device=dev))
    if uncert:
        t['render_uncert'] = torch.empty(rn, device=dev)
    o = PgrfPassOut(t['pixel_colors_nr'].data_ptr(), t['render_depth'].data_ptr(), t['hit_prob_nr'].data_ptr(),
                    t['colors_nr'].data_ptr(), t['density_nr'].data_ptr(),
                    nan_flag.data_ptr() if nan_flag is not None else None,
                    t['render_uncert'].data_ptr() if uncert else None)
    return t, o


PRECISIONS = {'f32': 0, 'f16x3': 1}


def render_pass(frame, blob_dev, use_vis, coords, depth, debug=False, precision='f32', uncert=False):
    """One network pass; returns dict of [rn,...] CUDA tensors (reference key names without the
    leading qn axis).  `debug=True` also returns the intermediate tensors under 'dbg'; `uncert=True` adds
    'render_uncert' (cfg render_uncert, renderer.py:305-312)."""
    _require_gpu()
    coords, depth = _f32c(coords, frame.device), _f32c(depth, frame.device)
    rn, dn = depth.shape
    t, o = _alloc_out(rn, dn, frame.device, uncert=uncert)
    dbg_t, dbg_s = None, None
    if debug:
        dbg_t = {k: torch.zeros(*f(frame.rfn, rn, dn), device=frame.device) for k, f in _DEBUG_SHAPES.items()}
        dbg_s = PgrfPassDebug(*[dbg_t[n].data_ptr() for n, _ in PgrfPassDebug._fields_])
    fs = frame.c_struct()
    nws = lib.pgrf_render_pass_workspace_bytes(C.byref(fs))
    ws = torch.empty(nws, dtype=torch.uint8, device=frame.device)
    flags = (1 if use_vis else 0) | (2 if PRECISIONS[precision] == 1 else 0)
    lib.check(lib.pgrf_render_pass(C.byref(fs), _ptr(blob_dev), flags, _ptr(coords), _ptr(depth), rn, dn,
                                   C.byref(o), C.byref(dbg_s) if debug else None, _ptr(ws), C.c_size_t(nws), _stream()))
    if debug:
        t['dbg'] = dbg_t
    return t


def make_cfg(cfg):
    """reference cfg dict -> PgrfRenderCfg (keys: renderer.py:36-64)."""
    c = PgrfRenderCfg()
    c.min_depth, c.max_depth = float(cfg['min_depth']), float(cfg['max_depth'])
    c.use_disp = int(cfg.get('use_disp', True))
    c.dn = int(cfg.get('depth_sample_num', 64))
    c.fdn = int(cfg.get('fine_depth_sample_num', 64))
    c.hierarchical = int(cfg.get('use_hierarchical_sampling', False))
    c.fine_use_all = int(cfg.get('fine_depth_use_all', False))
    c.c2f_all = int(cfg.get('render_c2f_all', False))
    c.use_vis_coarse = int(cfg.get('dist_decoder_cfg', {}).get('use_vis', True))
    c.use_vis_fine = int(cfg.get('fine_dist_decoder_cfg', {}).get('use_vis', True))
    c.precision = PRECISIONS[cfg.get('hip_precision', 'f32')]
    return c


class RayRenderer:
    """Coarse(+fine) render of arbitrary ray sets against one Frame: the HIP counterpart of
    NeuralRayBaseRenderer.render_impl (network/renderer.py:567-633)."""

    def __init__(self, cfg, state_dict, device='cuda'):
        _require_gpu()
        self.cfg = dict(cfg)
        self.ccfg = make_cfg(cfg)
        dn_max = max(self.ccfg.dn, self.ccfg.fdn, self.ccfg.dn + self.ccfg.fdn if self.ccfg.fine_use_all else 0)
        if dn_max > 128:
            raise PgrfError("at most 128 samples per pass are supported")
        self.blob_c = pack_pass_weights(state_dict, 'dist_decoder', 'agg_net', self.ccfg.use_vis_coarse).to(device)
        self.blob_f = None
        if self.ccfg.hierarchical:
            if cfg.get('one_mlp', False):
                self.blob_f = self.blob_c
                self.ccfg.use_vis_fine = self.ccfg.use_vis_coarse
            else:
                self.blob_f = pack_pass_weights(state_dict, 'fine_dist_decoder', 'fine_agg_net',
                                                self.ccfg.use_vis_fine).to(device)
        self.device = torch.device(device)
        self._ws = None

    def _workspace(self, fs, rn):
        need = lib.pgrf_render_workspace_bytes(C.byref(fs), C.byref(self.ccfg), rn)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    def render_rays(self, frame, coords, want=('pixel', 'depth', 'density', 'colors'), is_train=False, jitter=None):
        """coords [rn,2] -> dict with the reference's key names (no qn axis).  The caller drops hit_prob_* in eval like
        the reference does (renderer.py:675).  is_train: as in the reference's render_impl, the COARSE depths stay
        deterministic (renderer.py:607 passes random_sample=False even when training) and the importance sampler takes
        uniform draws (renderer.py:437-438 -> render_ops.py:449), made here exactly where the reference makes them
        (torch.rand([1, rn, fdn]) on the CPU generator), so a seeded run reproduces its sample positions.  `jitter`
        ([rn, dn-2] uniform draws) additionally moves the coarse samples like sample_depth(random_sample=True) would.
        Forward only."""
        coords = _f32c(coords, self.device)
        rn = coords.shape[0]
        c = self.ccfg
        jit = None if jitter is None else _f32c(jitter, self.device)
        u_fine = None
        if is_train and rn > 0 and c.hierarchical:
            u_fine = torch.rand([1, rn, c.fdn])[0].to(self.device).contiguous()
        fdn_eff = c.dn + c.fdn if c.fine_use_all else c.fdn
        dn_f_out = c.dn + fdn_eff if c.c2f_all else fdn_eff
        # cfg['hip_check_nan']: the reference drops into ipdb when a NaN shows up (renderer.py:264-265,459-462); here the
        # kernels raise a device flag and the host turns it into an exception (costs one device->host sync per call)
        flag = torch.zeros(1, dtype=torch.int32, device=self.device) if self.cfg.get('hip_check_nan') else None
        unc = bool(self.cfg.get('render_uncert', False))
        tc, oc = _alloc_out(rn, c.dn, self.device, flag, unc)
        out = dict(tc)
        of = None
        fd = None
        if c.hierarchical:
            tf, of = _alloc_out(rn, dn_f_out, self.device, flag, unc)
            fd = torch.empty(rn, fdn_eff, device=self.device)
            for k, v in tf.items():
                out[k + '_fine'] = v
            out['depth_fine'] = fd
        fs = frame.c_struct()
        ws, nbytes = self._workspace(fs, rn)
        cd = torch.empty(rn, c.dn, device=self.device) if (is_train or jit is not None) else None
        if cd is not None:
            out['depth'] = cd
        lib.check(lib.pgrf_render_rays_train(C.byref(fs), _ptr(self.blob_c), _ptr(self.blob_f), C.byref(c), _ptr(coords), rn,
                                             _ptr(jit), _ptr(u_fine), C.byref(oc), C.byref(of) if of is not None else None,
                                             _ptr(cd), _ptr(fd), _ptr(ws), C.c_size_t(nbytes), _stream()))
        if flag is not None and int(flag.item()):
            raise PgrfError('NaN in the rendered samples (density / colour / hit probability)')
        return out


# ------------------------------------------------------------------------------------------------ training (backward)
def raw_weight_tensors(state_dict, dd_prefix, agg_prefix, use_vis, device):
    """The pass's weights as fp32 contiguous DEVICE tensors in pgrf_raw_name order (None for an absent vis_decoder)."""
    out = []
    for n in raw_names():
        key = (dd_prefix + n[len('dist_decoder'):]) if n.startswith('dist_decoder.') else (agg_prefix + n[len('agg_net'):])
        if key not in state_dict:
            if '.vis_decoder.' in key and not use_vis:
                out.append(None)
                continue
            raise PgrfError(f"state_dict lacks '{key}'")
        out.append(state_dict[key].detach().to(device, torch.float32).contiguous())
    return out


def render_pass_staged_forward(frame, raw, use_vis, coords, depth):
    """Phase 1 of pgrf_render_pass_backward_phase: the staged forward of one pass with every activation kept.  -> (outputs dict, workspace
    tensor) -- hand the workspace to render_pass_backward(..., workspace=ws) for the reverse sweep (same frame / raw / coords / depth)."""
    _require_gpu()
    dev = frame.device
    coords, depth = _f32c(coords, dev), _f32c(depth, dev)
    rn, dn = depth.shape
    n = len(raw)
    raw_arr = (C.c_void_p * n)()
    for i, t in enumerate(raw):
        raw_arr[i] = None if t is None else t.data_ptr()
    fs = frame.c_struct()
    fwd_t, fwd_s = _alloc_out(rn, dn, dev)
    nws = lib.pgrf_render_pass_backward_workspace_bytes(frame.rfn, rn, dn)
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
    if rn > 0:
        lib.check(lib.pgrf_render_pass_backward_phase(C.byref(fs), raw_arr, n, int(bool(use_vis)), _ptr(coords), _ptr(depth), rn, dn,
                                                      None, None, None, None, C.byref(fwd_s), _ptr(ws), C.c_size_t(nws), _stream(), 1))
    return fwd_t, ws


def render_pass_backward(frame, raw, use_vis, coords, depth, grads, want_maps=True, want_forward=False, workspace=None):
    """Backward of one network pass (pgrf_render_pass_backward).  raw: list from raw_weight_tensors; grads: dict with any of
    'pixel_colors_nr' [rn,3], 'render_depth' [rn], 'hit_prob_nr' [rn,dn], 'colors_nr' [rn,dn,3], 'density_nr' [rn,dn].
    -> (list of weight gradients in raw order (None where raw is None), g_img_feats [rfn,32,hf,wf], g_ray_feats [rfn,32,hr,wr]
    (NCHW, or None), forward outputs of the staged recompute when want_forward)."""
    from ._lib import PgrfPassGrad
    _require_gpu()
    dev = frame.device
    coords, depth = _f32c(coords, dev), _f32c(depth, dev)
    rn, dn = depth.shape
    g = {k: (_f32c(v, dev) if v is not None else None) for k, v in grads.items()}
    pg = PgrfPassGrad(*[(g[k].data_ptr() if g.get(k) is not None else None)
                        for k in ('pixel_colors_nr', 'render_depth', 'hit_prob_nr', 'colors_nr', 'density_nr')])
    n = len(raw)
    raw_arr, g_arr = (C.c_void_p * n)(), (C.c_void_p * n)()
    g_raw = []
    # one zeroed buffer for all weight gradients (62-68 torch.zeros_like launches per pass otherwise), 64-float aligned views
    offs, total = [], 0
    for t in raw:
        offs.append(total)
        total += 0 if t is None else (t.numel() + 63) // 64 * 64
    flat = torch.zeros(max(total, 1), dtype=torch.float32, device=dev)
    for i, t in enumerate(raw):
        raw_arr[i] = None if t is None else t.data_ptr()
        gt = None if t is None else flat[offs[i]:offs[i] + t.numel()].view(t.shape)
        g_raw.append(gt)
        g_arr[i] = None if gt is None else gt.data_ptr()
    fs = frame.c_struct()
    g_if = torch.zeros_like(frame.img_feats) if want_maps else None         # channel-last, like the frame's maps
    g_rf = torch.zeros_like(frame.ray_feats) if want_maps else None
    fwd_t, fwd_s = None, None
    if want_forward:
        fwd_t, fwd_s = _alloc_out(rn, dn, dev)
    nws = lib.pgrf_render_pass_backward_workspace_bytes(frame.rfn, rn, dn)
    if workspace is not None:
        # the reverse sweep alone, over the activations render_pass_staged_forward left in `workspace`
        if workspace.numel() < nws:
            raise PgrfError('render_pass_backward: workspace smaller than the pass needs')
        lib.check(lib.pgrf_render_pass_backward_phase(C.byref(fs), raw_arr, n, int(bool(use_vis)), _ptr(coords), _ptr(depth), rn, dn,
                                                      C.byref(pg), g_arr, _ptr(g_if), _ptr(g_rf), None, _ptr(workspace), C.c_size_t(nws), _stream(), 2))
    else:
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        lib.check(lib.pgrf_render_pass_backward(C.byref(fs), raw_arr, n, int(bool(use_vis)), _ptr(coords), _ptr(depth), rn, dn,
                                                C.byref(pg), g_arr, _ptr(g_if), _ptr(g_rf), C.byref(fwd_s) if want_forward else None,
                                                _ptr(ws), C.c_size_t(nws), _stream()))
    to_nchw = lambda t: None if t is None else t.permute(0, 3, 1, 2)
    return g_raw, to_nchw(g_if), to_nchw(g_rf), fwd_t


class RenderPassFunction(torch.autograd.Function):
    """One network pass under torch.autograd: forward = the fused HIP kernel (pgrf_render_pass), backward =
    pgrf_render_pass_backward (rematerialising).  Differentiable inputs: img_feats, ray_feats (NCHW, the 2-D encoders'
    outputs) and the pass's weights in pgrf_raw_name order; everything else (source images, poses, sample depths) is a
    constant, as in the reference's training graph (the importance sampler's inputs are detached, renderer.py:437)."""

    # The channel-last Frame (shared by the coarse and the fine pass of a step) and the packed weight blob are reused while
    # they are built from the same tensors (same storage address, shape and version counter; the cache holds the tensors, so
    # their addresses cannot be recycled under it): an optimiser step or a new encoder output rebuilds them, a second pass over
    # the same tensors does not (was: ~70 device->host copies and two map re-layouts per pass).
    _frames, _blobs = [], []

    _lock = threading.Lock()

    @staticmethod
    def _lookup(cache, ts, extra):
        with RenderPassFunction._lock:
            entries = list(cache)
        for ent_ts, ent_ver, ent_extra, val in entries:
            if ent_extra == extra and len(ent_ts) == len(ts) and all((a is b) or (a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.dtype == b.dtype) for a, b in zip(ent_ts, ts)) and \
                    ent_ver == [None if t is None else t._version for t in ts]:
                return val
        return None

    @staticmethod
    def _store(cache, ts, extra, val, keep):
        # detached aliases as keys: same storage address and version counter (the lookup still matches, the address still cannot be
        # recycled under the cache), but no grad_fn -- the encoders' autograd graph is not kept alive by the cache
        with RenderPassFunction._lock:
            del cache[:max(0, len(cache) - keep + 1)]
            cache.append(([None if t is None else t.detach() for t in ts], [None if t is None else t._version for t in ts], extra, val))

    @staticmethod
    def clear_caches():
        """Drop the cached channel-last Frame and packed weight blobs (after training / validation, or to release their memory)."""
        with RenderPassFunction._lock:
            del RenderPassFunction._frames[:], RenderPassFunction._blobs[:]

    @staticmethod
    def forward(ctx, meta, coords, depth, img_feats, ray_feats, *raw):
        imgs, w2c, ref_dr, que_c2w, que_dr, use_vis, precision, frame_kw = meta[:8]
        grad_mode = meta[8] if len(meta) > 8 else True
        F_ = RenderPassFunction
        kw_t = [v for _, v in sorted(frame_kw.items()) if torch.is_tensor(v)]
        f_ts = [imgs, img_feats, ray_feats, w2c, ref_dr, que_c2w if torch.is_tensor(que_c2w) else None, que_dr] + kw_t
        f_extra = repr(sorted((k, v) for k, v in frame_kw.items() if not torch.is_tensor(v))) + repr(None if torch.is_tensor(que_c2w) else que_c2w)
        frame = F_._lookup(F_._frames, f_ts, f_extra)
        if frame is None:
            frame = Frame(imgs, img_feats.detach(), ray_feats.detach(), w2c, ref_dr, que_c2w, que_dr, device=imgs.device, **frame_kw)
            F_._store(F_._frames, f_ts, f_extra, frame, keep=1)
        # Training keeps a pass's activations between forward and backward (PANOGRF_TRAIN_KEEP_ACTS=0: the fused kernel forward + a
        # rematerialising backward, ~9 KB less memory per (ray, sample, view) held): the staged forward's outputs ARE the pass's outputs
        # (equal to the fused kernel's to 2e-5, tests/test_gpu_train_backward.py), so the fused launch and the second forward are saved.
        rn = coords.shape[0]
        # (only when a backward can follow: needs_input_grad mirrors the inputs' requires_grad flags whatever the grad mode, and grad
        # mode is always off inside Function.forward, so the CALLER's mode rides in `meta` -- a caller under torch.no_grad() or with
        # frozen inputs would otherwise pay the staged forward and its workspace for nothing)
        keep = grad_mode and any(ctx.needs_input_grad[3:]) and os.environ.get('PANOGRF_TRAIN_KEEP_ACTS', '1') != '0' and precision in (None, 'f32') and rn > 0 and \
            lib.pgrf_render_pass_backward_workspace_bytes(frame.rfn, rn, depth.shape[1]) <= int(float(os.environ.get('PANOGRF_TRAIN_KEEP_ACTS_MAX_GB', '16')) * 2 ** 30)
        ctx.ws = None
        if keep:
            raw_now = [(t.detach().to(frame.device, torch.float32).contiguous() if t is not None else None) for t in raw]
            out, ctx.ws = render_pass_staged_forward(frame, raw_now, use_vis, coords, depth)
        else:
            blob = F_._lookup(F_._blobs, list(raw), bool(use_vis))
            if blob is None:
                names = raw_names()
                blob = pack_pass_weights({n: t.detach() for n, t in zip(names, raw) if t is not None}, 'dist_decoder', 'agg_net', use_vis).to(frame.device)
                F_._store(F_._blobs, list(raw), bool(use_vis), blob, keep=4)
            out = render_pass(frame, blob, use_vis, coords, depth, precision=precision)
        ctx.frame, ctx.use_vis = frame, use_vis
        ctx.coords, ctx.depth = coords.detach(), depth.detach()
        # the weights go through save_for_backward: an in-place update between forward and backward (an optimiser step) is then
        # caught by autograd's version check instead of silently rematerialising the forward with the NEW weights
        ctx.present = [t is not None for t in raw]
        ctx.save_for_backward(*[t for t in raw if t is not None])
        ctx.need_maps = img_feats.requires_grad or ray_feats.requires_grad
        return out['pixel_colors_nr'], out['render_depth'], out['hit_prob_nr'], out['colors_nr'], out['density_nr']

    @staticmethod
    def backward(ctx, g_pixel, g_depth, g_hit, g_colors, g_density):
        grads = {'pixel_colors_nr': g_pixel, 'render_depth': g_depth, 'hit_prob_nr': g_hit, 'colors_nr': g_colors, 'density_nr': g_density}
        saved = iter(ctx.saved_tensors)
        raw = [(next(saved).detach().to(ctx.frame.device, torch.float32).contiguous() if p else None) for p in ctx.present]
        g_raw, g_if, g_rf, _ = render_pass_backward(ctx.frame, raw, ctx.use_vis, ctx.coords, ctx.depth, grads, want_maps=ctx.need_maps,
                                                    workspace=ctx.ws)
        ctx.ws = None
        return (None, None, None, g_if, g_rf) + tuple(g_raw)


class MergeC2fFunction(torch.autograd.Function):
    """`render_c2f_all` under torch.autograd (network/renderer.py:484-518): forward = pgrf_merge_c2f (stable sort of the concatenated
    coarse and fine samples + one composite over the union), backward = pgrf_merge_c2f_backward (the sort is a permutation: a
    gather).  Differentiable inputs: the two passes' per-sample colours and densities; depths are constants."""

    @staticmethod
    def forward(ctx, depth_c, colors_c, density_c, depth_f, colors_f, density_f):
        dev = depth_c.device
        args = [_f32c(t, dev) for t in (depth_c, colors_c, density_c, depth_f, colors_f, density_f)]
        rn, dn_c = args[0].shape
        dn_f = args[3].shape[1]
        t, o = _alloc_out(rn, dn_c + dn_f, dev)
        lib.check(lib.pgrf_merge_c2f(_ptr(args[0]), _ptr(args[1]), _ptr(args[2]), dn_c, _ptr(args[3]), _ptr(args[4]), _ptr(args[5]), dn_f, rn,
                                     C.byref(o), _stream()))
        ctx.save_for_backward(*args)
        return t['pixel_colors_nr'], t['render_depth'], t['hit_prob_nr'], t['colors_nr'], t['density_nr']

    @staticmethod
    def backward(ctx, g_pixel, g_depth, g_hit, g_colors, g_density):
        from ._lib import PgrfPassGrad
        zc, cc, dc, zf, cf, df = ctx.saved_tensors
        dev = zc.device
        rn, dn_c = zc.shape
        dn_f = zf.shape[1]
        gs = [None if g is None else _f32c(g, dev) for g in (g_pixel, g_depth, g_hit, g_colors, g_density)]
        pg = PgrfPassGrad(*[(None if g is None else g.data_ptr()) for g in gs])
        gcc, gdc, gcf, gdf = torch.empty_like(cc), torch.empty_like(dc), torch.empty_like(cf), torch.empty_like(df)
        lib.check(lib.pgrf_merge_c2f_backward(_ptr(zc), _ptr(cc), _ptr(dc), dn_c, _ptr(zf), _ptr(cf), _ptr(df), dn_f, rn, C.byref(pg),
                                              _ptr(gcc), _ptr(gdc), _ptr(gcf), _ptr(gdf), _stream()))
        return None, gcc, gdc, None, gcf, gdf


def merge_c2f_autograd(depth_c, colors_c, density_c, depth_f, colors_f, density_f):
    """-> dict with the merged composite's outputs (the keys the reference overwrites in the fine pass's dict, renderer.py:499-506)"""
    px, rd, hit, col, den = MergeC2fFunction.apply(depth_c, colors_c, density_c, depth_f, colors_f, density_f)
    return {'pixel_colors_nr': px, 'render_depth': rd, 'hit_prob_nr': hit, 'colors_nr': col, 'density_nr': den}


def render_pass_autograd(imgs, img_feats, ray_feats, w2c, ref_depth_range, que_c2w, que_depth_range, named_params, dd_prefix,
                         agg_prefix, use_vis, coords, depth, precision='f32', **frame_kw):
    """named_params: name -> Parameter (the module's own, so the gradients land on them).  -> dict like render_pass()."""
    names = raw_names()
    raw = []
    for n in names:
        key = (dd_prefix + n[len('dist_decoder'):]) if n.startswith('dist_decoder.') else (agg_prefix + n[len('agg_net'):])
        if key not in named_params:
            if '.vis_decoder.' in key and not use_vis:
                raw.append(None)
                continue
            raise PgrfError(f"module lacks parameter '{key}'")
        raw.append(named_params[key])
    meta = (imgs.detach(), w2c.detach(), ref_depth_range.detach(), que_c2w, que_depth_range, bool(use_vis), precision, frame_kw,
            torch.is_grad_enabled())
    px, rd, hit, col, den = RenderPassFunction.apply(meta, coords, depth, img_feats, ray_feats, *raw)
    return {'pixel_colors_nr': px, 'render_depth': rd, 'hit_prob_nr': hit, 'colors_nr': col, 'density_nr': den}
