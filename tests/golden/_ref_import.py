"""Import harness for the PanoGRF reference (ONLY usable in the build container
where /root/reference exists).  Used by tests/golden/make_fixtures.py to
generate golden vectors; nothing under tests/ run with `-m gpu` imports this.

The harness registers stub modules for third-party packages the reference
imports at module scope but never executes on the inference hot path
(SURVEY.md Appendix B), and replaces the `cuda:0`-hard-coded position
encoding builder by a device-neutral one computing the same table.
"""
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = os.environ.get("PANOGRF_REFERENCE", "/root/reference")


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        m = MagicMock(name=f"{self.__name__}.{k}")
        setattr(self, k, m)
        return m


def _stub(name):
    m = _Stub(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m


_STUBS = [
    "kornia", "easydict", "skimage", "skimage.io", "skimage.metrics", "cv2", "inplace_abn",
    "torchvision", "torchvision.transforms", "lmdb", "h5py", "plyfile", "transforms3d",
    "transforms3d.axangles", "transforms3d.euler", "imageio", "tensorboardX", "lpips",
    "habitat", "habitat_sim", "data_readers.habitat_data_neuray_ft",
    "data_readers.habitat_data_neuray_ft_lmdb",
]


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "network"))


def setup():
    """Make `network.*`, `models.*`, `helpers.*` of the reference importable."""
    if getattr(setup, "_done", False):
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, os.path.join(REF_ROOT, "UniFuse-Unidirectional-Fusion", "UniFuse"))
    for n in _STUBS:
        if n not in sys.modules:
            _stub(n)
    import numpy as np
    import torch
    import network.ibrnet as ib

    def posenc(self, d_hid, n_samples):
        pos = np.arange(n_samples, dtype=np.float64)[:, None]
        j = np.arange(d_hid)[None, :]
        tab = pos / np.power(10000, 2 * (j // 2) / d_hid)
        tab[:, 0::2] = np.sin(tab[:, 0::2])
        tab[:, 1::2] = np.cos(tab[:, 1::2])
        return torch.from_numpy(tab).float().unsqueeze(0)

    ib.IBRNetWithNeuRay.posenc = posenc
    setup._done = True
