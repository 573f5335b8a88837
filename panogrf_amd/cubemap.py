"""ERP -> cube-map conversion on the GPU: the counterpart of the reference's `Equirec2Cube`
(UniFuse-Unidirectional-Fusion/UniFuse/datasets/util.py:7-111) and of `FullPipeline.e2c_process` /
`normalize_input` (network/omni_mvsnet/pipeline3_model.py:226-244, 266-282), which build the cube-map input of the
mono-depth net per view and frame with a device -> host copy, scipy.ndimage.map_coordinates and a host -> device copy.
Here the sampling (and optionally the ImageNet normalisation) is one launch of `pgrf_equirec_to_cube` on device memory;
only the converter's constant coordinate table is built on the host, once per size.
"""
import math

import numpy as np
import torch

from ._lib import lib, PgrfError
from .engine import _ptr, _stream


def _tables(equ_h, equ_w, face_w):
    """(coor_y, coor_x, cos_map), each [face_w, 6*face_w] float32: where each cube-map pixel looks in the ERP image
    (util.py:27-70) and its ray-distance -> z-depth factor (util.py:22-24).  Faces [F R B L U D]; per face the two in-plane
    axes are the pixel grid (a: left->right, b: top->bottom = +0.5 -> -0.5) and the third is +-0.5."""
    lin = np.linspace(-0.5, 0.5, num=face_w, dtype=np.float32)
    a = np.broadcast_to(lin[None, :], (face_w, face_w))
    b = np.broadcast_to(-lin[:, None], (face_w, face_w))
    p, m = np.full((face_w, face_w), 0.5, np.float32), np.full((face_w, face_w), -0.5, np.float32)
    #            x            y   z
    layout = {'F': (a,          b, p),
              'R': (p,          b, a[:, ::-1]),
              'B': (a[:, ::-1], b, m),
              'L': (m,          b, a),
              'U': (a,          p, b[::-1]),
              'D': (a,          m, b)}
    x, y, z = (np.concatenate([layout[f][k] for f in 'FRBLUD'], 1).astype(np.float32) for k in range(3))
    lon = np.arctan2(x, z)
    lat = np.arctan2(y, np.sqrt(x ** 2 + z ** 2))
    coor_x = (lon / (2 * np.pi) + 0.5) * equ_w - 0.5
    coor_y = (-lat / np.pi + 0.5) * equ_h - 0.5
    cos = 1 / np.sqrt((2 * a) ** 2 + (2 * b) ** 2 + 1)
    return (np.ascontiguousarray(coor_y, np.float32), np.ascontiguousarray(coor_x, np.float32),
            np.ascontiguousarray(np.concatenate(6 * [cos], 1), np.float32))


class Equirec2Cube:
    """`Equirec2Cube(equ_h, equ_w, face_w)`; `run(equ_img [.., H, W, C], equ_dep=None)` like the reference, but on CUDA
    tensors (any leading batch dims) and returning CUDA tensors [.., face_w, 6*face_w, C].  `to_cube(x, channels_first,
    mean, std)` is the fused form used by the MVS stage.  Inputs must already have the converter's size (the reference
    falls back to cv2.resize otherwise)."""

    def __init__(self, equ_h, equ_w, face_w, device='cuda'):
        self.equ_h, self.equ_w, self.face_w = int(equ_h), int(equ_w), int(face_w)
        self.device = torch.device(device)
        cy, cx, cos = _tables(self.equ_h, self.equ_w, self.face_w)
        self.coor_y, self.coor_x, self.cosmaps = (torch.from_numpy(t).to(self.device) for t in (cy, cx, cos))

    def to_cube(self, x, channels_first=False, order=1, scale=False, mean=None, std=None):
        """x [N,H,W,C] (or [N,C,H,W] with channels_first) float32 on the GPU -> same layout, spatial size face_w x 6*face_w."""
        if not x.is_cuda:
            raise PgrfError('Equirec2Cube runs on the GPU (HIP library); got a CPU tensor')
        x = x.detach().float()
        if x.dim() != 4:
            raise PgrfError('to_cube expects a 4-D tensor')
        if channels_first:
            N, C, H, W = x.shape
            sn, sc, sh, sw = x.stride()
        else:
            N, H, W, C = x.shape
            sn, sh, sw, sc = x.stride()
        if (H, W) != (self.equ_h, self.equ_w):
            raise PgrfError(f'input is {H}x{W}, the converter was built for {self.equ_h}x{self.equ_w} (resize first)')
        fw = self.face_w
        out = torch.empty((N, C, fw, 6 * fw) if channels_first else (N, fw, 6 * fw, C), dtype=torch.float32, device=x.device)
        if channels_first:
            on, oc, oh, ow = out.stride()
        else:
            on, oh, ow, oc = out.stride()
        mean_t = std_t = None
        if mean is not None:
            mean_t = torch.as_tensor(mean, dtype=torch.float32, device=x.device).reshape(-1).contiguous()
            std_t = torch.as_tensor(std, dtype=torch.float32, device=x.device).reshape(-1).contiguous()
            if mean_t.numel() != C or std_t.numel() != C:
                raise PgrfError('mean / std need one value per channel')
        lib.check(lib.pgrf_equirec_to_cube(_ptr(x), N, H, W, C, sn, sh, sw, sc, _ptr(self.coor_y), _ptr(self.coor_x), fw, int(order),
                                           _ptr(self.cosmaps) if scale else None, _ptr(mean_t) if mean_t is not None else None,
                                           _ptr(std_t) if std_t is not None else None, _ptr(out), on, oh, ow, oc, _stream()))
        return out

    def run(self, equ_img, equ_dep=None):
        lead = equ_img.shape[:-3]
        cube = self.to_cube(equ_img.reshape((-1,) + tuple(equ_img.shape[-3:])), order=1)
        cube = cube.reshape(tuple(lead) + tuple(cube.shape[1:]))
        if equ_dep is None:
            return cube
        dep = self.to_cube(equ_dep.reshape((-1,) + tuple(equ_dep.shape[-3:])), order=0, scale=True)
        return cube, dep.reshape(tuple(lead) + tuple(dep.shape[1:]))


def e2c_process(panos_small, e2c):
    """FullPipeline.e2c_process (pipeline3_model.py:266-282): [B,S,H,W,3] -> [B,S,face_w,6*face_w,3], on the device."""
    return e2c.run(panos_small)
