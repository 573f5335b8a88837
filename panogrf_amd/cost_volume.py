"""Host mirror of the reference's spherical cost-volume entry points, dispatching to the HIP
library.  Function names / argument meaning follow models/spherical_cost_volume.py:231 and
models/spherical_cost_volume_mv.py:219 so the call sites in
network/omni_mvsnet/pipeline3_model.py:823-832,1160-1202 read unchanged."""
import ctypes as C

import numpy as np
import torch

from ._lib import DATASETS, lib, PgrfError
from .engine import _f32c, _ptr, _stream, _require_gpu

_COST = {'abs_diff': 0, 'dot': 1, 'none': 2}


def depth_sampling(n_samples=5, sampling_range=3):
    """k_list of the mono-guided hypotheses (host-side constants; pipeline3_model.py:537-545)."""
    from scipy.special import erf
    from scipy.stats import norm
    p_total = erf(sampling_range / np.sqrt(2))
    idx = np.arange(0, n_samples + 1)
    p = (1 - p_total) / 2 + (idx / n_samples) * p_total
    k = norm.ppf(p)
    return list((k[1:] + k[:-1]) / 2)


_UNIFORM = {}


def _uniform_planes(lo, hi, n, device):
    """torch.linspace(lo, hi, n) on the device, made once per (range, count, device): a host->device copy per call would
    also forbid capturing the MVS stage into a hipGraph."""
    key = (lo, hi, n, str(device))
    if key not in _UNIFORM:
        _UNIFORM[key] = torch.linspace(lo, hi, n).to(device)
    return _UNIFORM[key]


def build_depth_volume(ref_mu, k_list, fixed_sigma, min_depth, max_depth, n_total=64):
    """ref_mu [B,1,h,w] (CUDA) -> sorted hypothesis volume [B,n_total,h,w]
    (pipeline3_model.py:723-733,774-815 with the shipped flags)."""
    _require_gpu()
    ref_mu = _f32c(ref_mu)
    B, one, h, w = ref_mu.shape
    if one != 1:
        raise PgrfError('ref_mu must be [B,1,h,w]')
    n_g = len(k_list)
    n_u = n_total - n_g
    offs = (C.c_float * max(n_g, 1))(*[float(k * fixed_sigma) for k in k_list])
    uni = _uniform_planes(float(min_depth), float(max_depth), n_u, ref_mu.device) if n_u > 0 else None
    out = torch.empty(B, n_total, h, w, dtype=torch.float32, device=ref_mu.device)
    lib.check(lib.pgrf_depth_hypotheses(_ptr(ref_mu), B, h * w, offs, n_g, float(min_depth), float(max_depth),
                                        _ptr(uni), n_u, _ptr(out), _stream()))
    return out


def _dataset(args):
    name = args.get('dataset_name', 'm3d')
    if name not in DATASETS:
        raise PgrfError(f'dataset_name {name!r}: spherical convention not implemented (have {sorted(DATASETS)})')
    return DATASETS[name]


def _run(images, depth_volume, trans, rots, ref_idx, src_idx, div, cost_type, dataset):
    _require_gpu()
    if cost_type not in _COST:
        raise ValueError('Unknown cost type')
    images, depth_volume = _f32c(images), _f32c(depth_volume)
    rots, trans = _f32c(rots).to(images.device), _f32c(trans).to(images.device)
    B, S, h, w, Cc = images.shape
    D = depth_volume.shape[1]
    out = torch.empty(B, D, h, w, Cc, dtype=torch.float32, device=images.device)
    idx = (C.c_int * len(src_idx))(*src_idx)
    lib.check(lib.pgrf_cost_volume_erp(_ptr(images), _ptr(depth_volume), _ptr(rots), _ptr(trans), B, S, h, w, Cc, D,
                                       int(ref_idx), idx, len(src_idx), float(div), _COST[cost_type], int(dataset),
                                       _ptr(out), _stream()))
    return out


def _volume_from(args, images, depths, depth_volume):
    B, S, h, w, _ = images.shape
    if args.get('contain_dnet', False):
        if depth_volume is None:
            raise PgrfError('contain_dnet needs depth_volume [B,D,h,w]')
        return depth_volume
    d = torch.as_tensor(depths, dtype=torch.float32, device=images.device).reshape(1, -1, 1, 1)
    return d.expand(B, d.shape[1], h, w).contiguous()


def calculate_cost_volume_erp(args, images, depths, trans, rots, depth_volume=None, cost_type='abs_diff',
                              ref_gmms=None, nghbr_gmms=None, thres=None, direction='up'):
    """images [B,2,H,W,C] (index 0 = source, 1 = reference) -> [B,D,H,W,C]."""
    dv = _volume_from(args, images, depths, depth_volume)
    return _run(images, dv, trans, rots, 1, [0], 1.0, cost_type, _dataset(args))


def calculate_cost_volume_erp_multiview(args, images, depths, trans, rots, depth_volume=None, cost_type='abs_diff',
                                        ref_gmms=None, nghbr_gmms=None, thres=None, direction='up', curr_idx=0):
    """images [B,S,H,W,C]; the reference view is `curr_idx`; like the reference, the LAST view of the
    stack and the reference itself are skipped and the rest are averaged with 1/(S-2)."""
    S = images.shape[1]
    assert S > 2
    src = [v for v in range(S - 1) if v != curr_idx]
    dv = _volume_from(args, images, depths, depth_volume)
    return _run(images, dv, trans, rots, curr_idx, src, float(S - 2), cost_type, _dataset(args))
