"""The reference's training losses for the per-ray path, under the reference's own names (network/loss.py): `name2loss['render']`,
`['depth']`, `['consist']` are callables `loss(data_pr, data_gt, step) -> {'loss_*': tensor}` over the dict `renderer(data)` returns in
training mode, and `total_loss` is the trainer's reduction (train/trainer.py:377-388: the sum of the means of every entry whose key
contains 'loss').  Plain torch ops on the device tensors the HIP passes produced: the gradients enter the fast path through
`pgrf_render_pass_backward` (panogrf_amd.engine.RenderPassFunction)."""
import torch
import torch.nn.functional as F


class Loss:
    def __init__(self, keys):
        self.keys = keys           # (the reference's multi-GPU DummyLoss lists them; kept for interface parity)

    def __call__(self, data_pr, data_gt, step, **kwargs):
        return {}


def _interpolate_feats(feats, points, h=None, w=None, padding_mode='zeros', align_corners=False):
    """network/ops.py:32-52"""
    _, _, ch, cw = feats.shape
    if h is None and w is None:
        h, w = ch, cw
    x = points[:, :, 0] / (w - 1) * 2 - 1
    y = points[:, :, 1] / (h - 1) * 2 - 1
    grid = torch.stack([x, y], -1).unsqueeze(1)
    return F.grid_sample(feats, grid, mode='bilinear', padding_mode=padding_mode, align_corners=align_corners).squeeze(2).permute(0, 2, 1)


class ConsistencyLoss(Loss):
    """network/loss.py:23-53: cross entropy between the rendered hit probabilities (detached) and the query view's own."""
    default_cfg = {'use_ray_mask': False, 'use_dr_loss': False, 'use_dr_fine_loss': False, 'use_nr_fine_loss': False}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}
        super().__init__(['loss_prob', 'loss_prob_fine'])

    def __call__(self, data_pr, data_gt, step, **kwargs):
        if 'hit_prob_self' not in data_pr:
            return {}

        def ce(p0, p1):
            p0 = p0.detach()
            c = -p0 * torch.log(p1 + 1e-5) - (1 - p0) * torch.log(1 - p1 + 1e-5)
            return torch.mean(torch.mean(c, -1), 1)
        out = {'loss_prob': ce(data_pr['hit_prob_nr'], data_pr['hit_prob_self'])}
        if 'hit_prob_nr_fine' in data_pr:
            out['loss_prob_fine'] = ce(data_pr['hit_prob_nr_fine'], data_pr['hit_prob_self_fine'])
        return out


class RenderLoss(Loss):
    """network/loss.py:55-165: squared colour error per ray, summed over the channels, averaged over the (masked) rays; with
    `use_polar_weighted_loss` weighted by sin(latitude) and normalised by the weights' sum."""
    default_cfg = {'use_ray_mask': True, 'use_dr_loss': False, 'use_dr_fine_loss': False, 'use_nr_fine_loss': False,
                   'fix_all': False, 'fix_coarse': False, 'use_polar_weighted_loss': False, 'use_hierarchical_sampling': False}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}
        super().__init__(['loss_rgb'])

    def __call__(self, data_pr, data_gt, step, **kwargs):
        c = self.cfg
        if c['fix_all']:
            return {'loss_depth': torch.zeros([1], dtype=torch.float32, device=data_pr['pixel_colors_nr'].device)}
        one_mlp = bool(c.get('one_mlp'))
        if one_mlp:
            rgb_gt, rgb_nr, ray_mask = data_pr['pixel_colors_gt_fine'], None, data_pr['ray_mask_fine'].float()
        else:
            rgb_gt, rgb_nr, ray_mask = data_pr['pixel_colors_gt'], data_pr['pixel_colors_nr'], data_pr['ray_mask']
        polar = bool(c['use_polar_weighted_loss'])
        weights_gt = data_pr['polar_weights'] if polar else None          # 1,rn,1

        def photometric(rgb_pr, rgb_gt):
            if polar:
                loss = torch.sum((rgb_pr - rgb_gt) ** 2 * weights_gt, -1)
                if c['use_ray_mask']:
                    return torch.sum(loss * ray_mask, dim=1) / (torch.sum(ray_mask * weights_gt.squeeze(-1), dim=1) + 1e-7)
                return torch.sum(loss, dim=1) / (torch.sum(weights_gt.squeeze(-1), dim=1) + 1e-7)
            loss = torch.sum((rgb_pr - rgb_gt) ** 2, -1)
            if c['use_ray_mask']:
                return torch.sum(loss * ray_mask, 1) / (torch.sum(ray_mask, 1) + 1e-7)
            return torch.mean(loss, 1)

        terms = []          # (loss key, predicted-colour key) in the reference's order
        if not c['fix_coarse'] and not one_mlp:
            terms.append(('loss_rgb_nr', 'pixel_colors_nr'))
            if c['use_dr_loss']:
                terms.append(('loss_rgb_dr', 'pixel_colors_dr'))
        if c['use_hierarchical_sampling'] or c.get('diner_depth_guided_sampling'):
            if c['use_dr_fine_loss']:
                terms.append(('loss_rgb_dr_fine', 'pixel_colors_dr_fine'))
            if c['use_nr_fine_loss']:
                terms.append(('loss_rgb_nr_fine', 'pixel_colors_nr_fine'))
        return {name: photometric(rgb_nr if src == 'pixel_colors_nr' else data_pr[src], rgb_gt) for name, src in terms}


class DepthLoss(Loss):
    """network/loss.py:167-230: the depth the visibility decoder predicts at `depth_coords` of the source views against their true
    depth maps, both in normalised inverse depth; zero when the batch carries no `true_depth`."""
    default_cfg = {'depth_correct_thresh': 0.02, 'depth_loss_type': 'l2', 'depth_loss_l1_beta': 0.05}

    def __init__(self, cfg):
        super().__init__(['loss_depth'])
        self.cfg = {**self.default_cfg, **cfg}

    def __call__(self, data_pr, data_gt, step, **kwargs):
        if 'true_depth' not in data_gt['ref_imgs_info']:
            key = 'pixel_colors_nr_fine' if self.cfg.get('diner_depth_guided_sampling') else 'pixel_colors_nr'
            return {'loss_depth': torch.zeros([1], dtype=torch.float32, device=data_pr[key].device)}
        coords, depth_pr = data_pr['depth_coords'], data_pr['depth_mean']            # rfn,pn,2 / rfn,pn
        true_depth = data_gt['ref_imgs_info']['true_depth']                          # [rfn, 1, h, w]
        h, w = true_depth.shape[-2:]
        depth_gt = _interpolate_feats(true_depth, coords, h, w, padding_mode='border', align_corners=True)[..., 0]
        depth_range = data_gt['ref_imgs_info']['depth_range']
        near, far = -1 / depth_range[:, 0:1], -1 / depth_range[:, 1:2]

        inv = -1 / torch.clamp(depth_gt, min=1e-5)                                    # normalised inverse depth in [0, 1]
        depth_gt = torch.clamp((inv - near) / (far - near), min=0, max=1.0)

        def depth_term(pr):
            if self.cfg['depth_loss_type'] == 'l2':
                loss = (depth_gt - pr) ** 2
            elif self.cfg['depth_loss_type'] == 'smooth_l1':
                loss = F.smooth_l1_loss(depth_gt, pr, reduction='none', beta=self.cfg['depth_loss_l1_beta'])
            else:
                raise ValueError(f"depth_loss_type {self.cfg['depth_loss_type']!r}")
            return torch.mean(loss, 1)
        out = {'loss_depth': depth_term(depth_pr)}
        if 'depth_mean_fine' in data_pr:
            out['loss_depth_fine'] = depth_term(data_pr['depth_mean_fine'])
        return out


name2loss = {'render': RenderLoss, 'depth': DepthLoss, 'consist': ConsistencyLoss}


def total_loss(loss_results):
    """train/trainer.py:382-388: sum of torch.mean(v) over every entry whose key contains 'loss'."""
    loss = 0
    for k, v in loss_results.items():
        if 'loss' in k:
            loss = loss + torch.mean(v)
    return loss
