"""The dict-of-tensors interface between the caller and the renderer (keys and shapes of
utils/imgs_info.py:105-181 in the reference), rebuilt for synthetic / in-memory panoramas.  Host plumbing only."""
import math

import numpy as np
import torch


def build_render_imgs_info(que_pose, que_shape, que_depth_range):
    """Query-view info from a world->camera pose [3,4] (utils/imgs_info.py:158-181): row-major pixel
    coords [1,h*w,2] as (x,y), w2c / c2w / rots / trans / depth_range / shape."""
    h, w = int(que_shape[0]), int(que_shape[1])
    que_pose = np.asarray(que_pose, np.float64)
    coords = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(1, -1, 2).astype(np.float32)
    c2w = np.linalg.inv(np.concatenate([que_pose, np.array([[0, 0, 0, 1.0]])], 0))[:3, :4]
    return {'w2c': que_pose.astype(np.float32)[None], 'c2w': c2w.astype(np.float32)[None],
            'rots': que_pose[:3, :3].astype(np.float32)[None], 'trans': que_pose[:3, 3].astype(np.float32)[None],
            'coords': coords, 'depth_range': np.asarray(que_depth_range, np.float32)[None, :], 'shape': (h, w)}


def build_render_cube_imgs_info(que_pose, que_K, que_shape, que_depth_range):
    """Pinhole (cube-face) query info for renderer(data, is_perspec=True) (utils/imgs_info.py:253-264)."""
    h, w = int(que_shape[0]), int(que_shape[1])
    coords = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(1, -1, 2).astype(np.float32)
    return {'poses': np.asarray(que_pose, np.float32)[None], 'Ks': np.asarray(que_K, np.float32)[None], 'coords': coords,
            'depth_range': np.asarray(que_depth_range, np.float32)[None, :], 'shape': (h, w)}


def build_imgs_info(imgs, w2c, depth_range):
    """Source-view info from in-memory panoramas (the reference reads them from its M3D database,
    utils/imgs_info.py:105-156): imgs [n,3,H,W] in [0,1], w2c [n,3,4], depth_range [n,2]."""
    imgs = np.asarray(imgs, np.float32)
    w2c = np.asarray(w2c, np.float64)
    n, _, H, W = imgs.shape
    c2w = np.stack([np.linalg.inv(np.concatenate([p, np.array([[0, 0, 0, 1.0]])], 0))[:3, :4] for p in w2c])
    sin_phi = np.sin((np.arange(H, dtype=np.float32) + 0.5) * math.pi / H)
    return {'imgs': imgs, 'rots': w2c[:, :3, :3].astype(np.float32), 'trans': w2c[:, :3, 3].astype(np.float32),
            'c2w': c2w.astype(np.float32), 'w2c': w2c.astype(np.float32),
            'depth_range': np.asarray(depth_range, np.float32), 'masks': np.ones((n, 1, H, W), np.float32),
            'polar_weights': np.broadcast_to(sin_phi.reshape(1, 1, H, 1), (n, 1, H, W)).copy()}


def imgs_info_to_torch(info, device=None):
    out = {}
    for k, v in info.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        if torch.is_tensor(v) and device is not None:
            host = v
            v = v.to(device)
            if v is not host and host.numel() <= 64 and not v.is_inference():
                # poses, intrinsics and depth ranges become BY-VALUE kernel arguments (PgrfFrame.que_c2w, ...): keep the host
                # values next to the device copy so that panogrf_amd.engine.host_floats does not read them back (a
                # device->host copy = one host sync per frame); valid while the device tensor's version counter stands
                v._pgrf_host = (host.detach().to(torch.float32).reshape(-1).clone(), v._version)
        out[k] = v
    return out
