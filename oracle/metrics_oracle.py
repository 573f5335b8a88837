"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): plain-numpy restatement, in double, of the reference's image metrics.
Pinned by tests/golden/metrics_fixtures.npz (outputs of the reference's own functions, tests/golden/make_metrics_fixtures.py).

  to_uint8        utils/base_utils.py:496-499  (color_map_backward)
  psnr_uint8      network/metrics.py:17-32     (compute_psnr)
  ws_psnr         network/metrics.py:124-171   (WSPSNR.ws_psnr)"""
import numpy as np


def to_uint8(rgb):
    out = np.empty(rgb.shape, np.uint8)
    flat, o = rgb.reshape(-1), out.reshape(-1)
    for i in range(flat.size):                       # scale in the input's precision, clip, truncate toward zero
        v = flat[i] * rgb.dtype.type(255)
        v = min(max(v, 0), 255)
        o[i] = int(v)
    return out


def psnr_uint8(gt, pr):
    g = gt.reshape(-1, 3).astype(np.float64)
    p = pr.reshape(-1, 3).astype(np.float64)
    mse = ((g - p) ** 2).mean(0).mean()
    return 10.0 * np.log10(255.0 * 255.0 / mse)


def ws_psnr(pred, true, max_val=1.0):
    """[B,H,W,C] -> [B]"""
    B, H, W, C = pred.shape
    w = np.sin((np.arange(H) + 0.5) * (np.pi / H)).reshape(1, H, 1, 1)
    se = (pred.astype(np.float64) - true.astype(np.float64)) ** 2
    wmse = (w * se).sum((1, 2, 3)) / (w.sum() * W * C)
    return 10.0 * np.log10(max_val * max_val / wmse)
