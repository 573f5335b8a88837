"""CPU restatement of the reference's training losses (TEST INFRASTRUCTURE: imported by tests/ only; the product's own losses are
panogrf_amd/loss.py).  Plain functions over tensors, each citing the reference lines it follows; pinned by
tests/golden/loss_fixtures.npz, which tests/golden/make_loss_fixtures.py generates by calling the REFERENCE's RenderLoss / DepthLoss /
ConsistencyLoss (loss values and gradients)."""
import torch
import torch.nn.functional as F


def render_loss_term(rgb_pr, rgb_gt, ray_mask=None, polar_weights=None):
    """network/loss.py:103-145 (`compute_loss`, the branch without the black-pixel mask): rgb [1,rn,3], ray_mask [1,rn] or None
    (cfg use_ray_mask false), polar_weights [1,rn,1] or None (cfg use_polar_weighted_loss false) -> [1]"""
    se = (rgb_pr - rgb_gt) ** 2
    if polar_weights is not None:
        per_ray = (se * polar_weights).sum(-1)                                        # :126
        if ray_mask is not None:
            return (per_ray * ray_mask).sum(1) / ((ray_mask * polar_weights.squeeze(-1)).sum(1) + 1e-7)   # :135
        return per_ray.sum(1) / (polar_weights.squeeze(-1).sum(1) + 1e-7)               # :137
    per_ray = se.sum(-1)                                                              # :128
    if ray_mask is not None:
        return (per_ray * ray_mask).sum(1) / (ray_mask.sum(1) + 1e-7)                   # :143
    return per_ray.mean(1)                                                            # :145


def render_loss(data_pr, cfg):
    """network/loss.py:66-165 for cfg without one_mlp / woblack: {'loss_rgb_nr', ['loss_rgb_nr_fine']}"""
    mask = data_pr['ray_mask'] if cfg.get('use_ray_mask', True) else None
    pw = data_pr['polar_weights'] if cfg.get('use_polar_weighted_loss', False) else None
    out = {}
    if not cfg.get('fix_coarse', False):
        out['loss_rgb_nr'] = render_loss_term(data_pr['pixel_colors_nr'], data_pr['pixel_colors_gt'], mask, pw)     # :155
    if cfg.get('use_hierarchical_sampling', False) and cfg.get('use_nr_fine_loss', False):
        out['loss_rgb_nr_fine'] = render_loss_term(data_pr['pixel_colors_nr_fine'], data_pr['pixel_colors_gt'], mask, pw)   # :164-165
    return out


def consistency_loss(hit_prob_nr, hit_prob_self):
    """network/loss.py:36-44: cross entropy against the detached rendered probabilities, mean over samples then rays"""
    p0 = hit_prob_nr.detach()
    ce = -p0 * torch.log(hit_prob_self + 1e-5) - (1 - p0) * torch.log(1 - hit_prob_self + 1e-5)
    return ce.mean(-1).mean(1)


def depth_loss(depth_mean, depth_coords, true_depth, depth_range, loss_type='l2', beta=0.05):
    """network/loss.py:188-224: depth_mean [rfn,pn] (normalised inverse depth), depth_coords [rfn,pn,2] pixels, true_depth
    [rfn,1,h,w], depth_range [rfn,2] -> [rfn]"""
    rfn, _, h, w = true_depth.shape
    x = depth_coords[..., 0] / (w - 1) * 2 - 1                                          # network/ops.py:47-50
    y = depth_coords[..., 1] / (h - 1) * 2 - 1
    gt = F.grid_sample(true_depth, torch.stack([x, y], -1).unsqueeze(1), mode='bilinear', padding_mode='border', align_corners=True)[:, 0, 0]
    near, far = -1 / depth_range[:, 0:1], -1 / depth_range[:, 1:2]
    gt = ((-1 / gt.clamp(min=1e-5)) - near) / (far - near)                              # :199-204
    gt = gt.clamp(0, 1.0)
    if loss_type == 'l2':
        per = (gt - depth_mean) ** 2
    else:
        d = (gt - depth_mean).abs()
        per = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)                 # nn.SmoothL1Loss(reduction='none', beta)
    return per.mean(1)


def total_loss(results):
    """train/trainer.py:382-388"""
    t = 0
    for k, v in results.items():
        if 'loss' in k:
            t = t + v.mean()
    return t
