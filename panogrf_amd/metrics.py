"""Image-quality metrics of the reference's evaluation (`PSNR` row): the numbers parity is reported in.

  color_map_backward / color_map_forward   utils/base_utils.py:492-499   float [0,1] <-> uint8
  compute_psnr                              network/metrics.py:17-32      10 log10(255^2 / MSE) on uint8 images
  psnr_float                                10 log10(max^2 / MSE) on float images (the "float variant" of the same row)
  WSPSNR.get_weights / calculate_wsmse / ws_psnr   network/metrics.py:124-171   rows weighted by sin((i + 1/2) pi / H)

Host-side measurement code (numpy / torch on whatever device the images are on): no kernel of its own."""
import numpy as np
import torch


def color_map_forward(rgb):
    return rgb.astype(np.float32) / 255


def color_map_backward(rgb):
    """float image -> uint8: scale, clip, TRUNCATE (numpy astype), as the reference does before every PSNR / jpg."""
    rgb = rgb * 255
    return np.clip(rgb, a_min=0, a_max=255).astype(np.uint8)


def compute_psnr(img_gt, img_pr, use_vis_scores=False, vis_scores=None, vis_scores_thresh=1.5):
    """PSNR of two uint8 (or 0..255 float) images, mean over pixels then over the three channels.  The reference's
    `use_vis_scores` branch computes a masked MSE and then overwrites it with the unmasked one (metrics.py:18-28): the
    arguments are accepted and, as there, have no effect on the result."""
    img_gt = np.asarray(img_gt).reshape([-1, 3]).astype(np.float32)
    img_pr = np.asarray(img_pr).reshape([-1, 3]).astype(np.float32)
    mse = np.mean(np.mean((img_gt - img_pr) ** 2, 0))
    return 10 * np.log10(255 * 255 / mse)


def psnr_float(pred, ref, max_val=1.0):
    """10 log10(max^2 / MSE) on float images / ray colours of any shape (torch or numpy), accumulated in double."""
    a = torch.as_tensor(pred).double()
    b = torch.as_tensor(ref).double().to(a.device)
    mse = float(((a - b) ** 2).mean())
    return 10 * np.log10(max_val * max_val / max(mse, 1e-30))


def erp_row_weights(height):
    """Solid-angle weight of every ERP row: a pixel of row i (centre at polar angle (i + 1/2) pi / H) covers an area of the
    sphere proportional to the sine of that angle.  float64 [H]."""
    return np.sin((np.arange(height, dtype=np.float64) + 0.5) * (np.pi / height))


class WSPSNR:
    """Weighted-to-spherical PSNR on ERP images: the squared error of every pixel counts with its row's solid angle
    (interface of the reference's evaluator, network/metrics.py:118-171: get_weights / calculate_wsmse / ws_psnr)."""

    def __init__(self):
        self._rows = {}

    def get_weights(self, height=1080, width=1920):
        """[height, width] weight map (every column is erp_row_weights(height))."""
        if height not in self._rows:
            self._rows[height] = erp_row_weights(height)
        return np.repeat(self._rows[height][:, None], width, axis=1)

    def calculate_wsmse(self, reconstructed, reference):
        """[B,H,W,C] image batches -> [B]: sum_y w_y * err^2 / (W * C * sum_y w_y), evaluated per row first."""
        H, W, C = reconstructed.shape[1:]
        w = torch.as_tensor(self.get_weights(H, 1)[:, 0], dtype=reconstructed.dtype, device=reconstructed.device)
        row_err = ((reconstructed - reference) ** 2).sum(dim=(2, 3))          # [B,H]
        return (row_err * w).sum(1) / (w.sum() * (W * C))

    def ws_psnr(self, y_pred, y_true, max_val=1.0):
        return 10 * torch.log10(max_val * max_val / self.calculate_wsmse(y_pred, y_true))

    @staticmethod
    def ws_psnr_rays(pred, ref, rows, height, max_val=1.0):
        """The same metric on a SUBSET of an ERP image's pixels ([n,3] colours, their row indices): what bench.py can
        evaluate against the CPU oracle, which renders a sample of the frame's rays only."""
        w = torch.as_tensor(erp_row_weights(height))[torch.as_tensor(rows).long().cpu()].view(-1, 1)
        a = torch.as_tensor(pred).double().cpu()
        b = torch.as_tensor(ref).double().cpu()
        wmse = float((w * (a - b) ** 2).sum() / (w.sum() * a.shape[1]))
        return 10 * np.log10(max_val * max_val / max(wmse, 1e-30))
