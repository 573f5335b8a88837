"""CPU oracle for the 360-degree spherical cost volume and its 3-D regularisers
(SURVEY.md section 8 rows a-11 .. a-14).

TEST INFRASTRUCTURE: see oracle/__init__.py.  Plain PyTorch fp32 restatement; file:line
citations are into /root/reference.  Pinned by tests/golden/cv_*.npz captured from the
reference's own calculate_cost_volume_erp(_multiview), CostRegNet and UNet2.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- hypotheses (a-14)
def gaussian_k_list(n_samples=5, sampling_range=3.0):
    """Mid-points of equal-probability Gaussian quantile bins inside +-sampling_range sigma.
    pipeline3_model.py:537-545."""
    from scipy.special import erf
    from scipy.stats import norm
    p_total = erf(sampling_range / np.sqrt(2))
    idx = np.arange(0, n_samples + 1)
    p = (1 - p_total) / 2 + (idx / n_samples) * p_total
    k = norm.ppf(p)
    return list((k[1:] + k[:-1]) / 2)


def depth_hypotheses(ref_mu, k_list, fixed_sigma, min_depth, max_depth, n_total=64):
    """ref_mu [B,1,h,w] -> sorted hypothesis volume [B,n_total,h,w]: len(k_list) mono-guided
    planes + uniform linspace planes.  pipeline3_model.py:723-733, 774-815 (shipped flags:
    use_depth_sampling, !revise_range, !wo_hdh, !mono_uncertainty)."""
    B, _, h, w = ref_mu.shape
    guided = [torch.clamp(ref_mu + k * fixed_sigma, min=min_depth, max=max_depth) for k in k_list]
    n_uni = n_total - len(k_list)
    uni = torch.linspace(min_depth, max_depth, n_uni).reshape(1, n_uni, 1, 1).repeat(B, 1, h, w)
    vol = torch.cat(guided + [uni], dim=1)
    return torch.sort(vol, dim=1)[0]


# --------------------------------------------------------------------------- cost volume (a-11)
def _centre_dirs(h, w, dataset='m3d'):
    """Unit directions at the pixel positions the dataset's convention uses, [h,w,3].
    spherical_cost_volume.py:268-297, my_torch_helpers.py:34-56."""
    phi = torch.arange(0, h, dtype=torch.float32)
    theta = torch.arange(0, w, dtype=torch.float32)
    if dataset == 'm3d':
        phi = (phi + 0.5) * (np.pi / h)
        theta = (theta + 0.5) * (2 * np.pi / w) - np.pi / 2
    elif dataset == 'replica_test':
        theta = (2 * np.pi / w) * (theta + 0.5) - np.pi
        phi = -(phi + 0.5) * np.pi / h + np.pi * 0.5
    elif dataset == 'residential':
        theta = np.pi * (2 * theta / (w - 1) - 1.5)
        phi = np.pi * (0.5 - phi / (h - 1))
    elif dataset == 'CoffeeArea':
        theta = (-2 * np.pi / (w - 1)) * theta + 2 * np.pi
        phi = (np.pi / (h - 1)) * phi
    else:
        raise ValueError(f'unknown dataset convention {dataset!r}')
    phi, theta = torch.meshgrid(phi, theta, indexing='ij')
    if dataset == 'm3d':
        s = torch.sin(phi)
        return torch.stack((s * torch.cos(theta), torch.cos(phi), s * torch.sin(theta)), -1)
    if dataset == 'replica_test':
        return torch.stack((torch.sin(theta) * torch.cos(phi), -torch.sin(phi), torch.cos(theta) * torch.cos(phi)), -1)
    if dataset == 'residential':
        return torch.stack((torch.cos(theta) * torch.cos(phi), torch.sin(phi), torch.sin(theta) * torch.cos(phi)), -1)
    return torch.stack((torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi)), -1)


def _cart_to_sph(xyz, lin=np.deg2rad(10), dataset='m3d'):
    """theta, phi per dataset; m3d linearises acos within 10 degrees of the poles.  my_torch_helpers.py:62-120
    (replica_test takes -asin(z / r): reproduced as written there)."""
    cos_deg = np.cos(lin)
    radius = torch.norm(xyz, dim=-1)
    x, y, z = torch.unbind(xyz, dim=-1)
    if dataset == 'm3d':
        theta = torch.atan2(z, x)
        yor = y / radius
        valid = yor.abs() < cos_deg
        zero = torch.zeros((), dtype=xyz.dtype)
        phi = torch.where(valid, torch.acos(torch.where(valid, yor, zero)),
                          torch.where(y >= 0, lin * (1 - yor) / (1 - cos_deg),
                                      np.pi - lin * (yor + 1) / (-cos_deg + 1)))
    elif dataset == 'replica_test':
        theta = torch.atan2(x, z)
        phi = -torch.asin(z / radius)
    elif dataset == 'residential':
        theta = -torch.atan2(-z, x)
        phi = torch.asin(y / radius)
        theta = torch.where(torch.logical_and(theta.gt(np.pi * 0.5), theta.le(2 * np.pi)), theta - 2 * np.pi, theta)
    elif dataset == 'CoffeeArea':
        theta = torch.atan2(y, x)
        phi = torch.acos(z / radius)
        theta = torch.where(theta.lt(0), theta + 2 * np.pi, theta)
    else:
        raise ValueError(f'unknown dataset convention {dataset!r}')
    return theta, phi


def _warp_plane(depth, dirs, rot_ref, t_ref, rot_src, t_src, src_cf, dataset='m3d'):
    """One hypothesis plane: back-project the reference pixels to `depth`, move them into the
    source camera, sample the source features.  depth [B,h,w]; src_cf [B,C,h,w] -> [B,C,h,w].
    spherical_cost_volume.py:135-198."""
    B, h, w = depth.shape
    m = (depth.unsqueeze(-1) * dirs).view(B, -1, 3, 1)
    wpt = torch.matmul(torch.inverse(rot_ref.unsqueeze(1)), m - t_ref.view(B, 1, 3, 1))
    c = torch.matmul(rot_src.unsqueeze(1), wpt) + t_src.view(B, 1, 3, 1)
    theta, phi = _cart_to_sph(c.view(B, h, w, 3), dataset=dataset)
    if dataset == 'm3d':
        u = torch.fmod(theta + np.pi / 2 + 2 * np.pi, 2 * np.pi)
        v = phi
    elif dataset == 'replica_test':
        u = torch.fmod(theta + np.pi + 2 * np.pi, 2 * np.pi)
        v = -phi + 0.5 * np.pi
    elif dataset == 'residential':
        u = torch.fmod(theta + 3 / 4.0 * 2 * np.pi, 2 * np.pi)
        v = 0.5 * np.pi - phi
    else:
        u = 2 * np.pi - theta
        v = phi
    u = u / np.pi - 1
    v = 2 * v / np.pi - 1
    return F.grid_sample(src_cf, torch.stack((u, v), -1), mode='bilinear', align_corners=True)


def cost_volume_erp(feats, depth_volume, rots, trans, cost_type='abs_diff', dataset='m3d'):
    """2-view volume.  feats [B,2,h,w,C] (0 = source, 1 = reference), depth_volume [B,D,h,w],
    rots [B,2,3,3], trans [B,2,3] -> [B,D,h,w,C].  spherical_cost_volume.py:231-341."""
    B, _, h, w, C = feats.shape
    dirs = _centre_dirs(h, w, dataset).unsqueeze(0).expand(B, h, w, 3)
    src_cf = feats[:, 0].permute(0, 3, 1, 2)
    ref_cf = feats[:, 1].permute(0, 3, 1, 2)
    planes = []
    for d in range(depth_volume.shape[1]):
        warped = _warp_plane(depth_volume[:, d], dirs, rots[:, 1], trans[:, 1], rots[:, 0], trans[:, 0], src_cf, dataset)
        planes.append(_cost(warped, ref_cf, cost_type))
    return torch.stack(planes, 1).permute(0, 1, 3, 4, 2)


def _cost(warped, ref_cf, cost_type):
    if cost_type == 'abs_diff':
        return torch.abs(warped - ref_cf)
    if cost_type == 'dot':
        return warped * ref_cf
    if cost_type == 'none':
        return warped
    raise ValueError('Unknown cost type')


def cost_volume_erp_multiview(feats, depth_volume, rots, trans, curr_idx=0, cost_type='abs_diff', dataset='m3d'):
    """Multi-view volume: mean over the source views, where view `curr_idx` (the reference) and the
    LAST view of the stack are skipped (range(seq_len-1)).  spherical_cost_volume_mv.py:219-347."""
    B, S, h, w, C = feats.shape
    assert S > 2
    dirs = _centre_dirs(h, w, dataset).unsqueeze(0).expand(B, h, w, 3)
    ref_cf = feats[:, curr_idx].permute(0, 3, 1, 2)
    total = 0
    for v in range(S - 1):
        if v == curr_idx:
            continue
        src_cf = feats[:, v].permute(0, 3, 1, 2)
        planes = []
        for d in range(depth_volume.shape[1]):
            warped = _warp_plane(depth_volume[:, d], dirs, rots[:, curr_idx], trans[:, curr_idx], rots[:, v],
                                 trans[:, v], src_cf, dataset)
            planes.append(_cost(warped, ref_cf, cost_type))
        total = total + torch.stack(planes, 1).permute(0, 1, 3, 4, 2) / (S - 2)
    return total


# --------------------------------------------------------------------------- 3-D regularisers (a-12, a-13)
def wrap_pad3d(x, p=1):
    """zeros along D and H, circular along W. common_blocks.py:458-503."""
    x = F.pad(x, (0, 0, p, p, p, p))
    return torch.cat((x[..., -p:], x, x[..., :p]), dim=-1)


def _conv3(w, name, x, stride=1):
    return F.conv3d(wrap_pad3d(x), w[name + '.weight'], w.get(name + '.bias'), stride=stride)


def _bn_lrelu(w, name, x, eps=1e-5):
    x = F.batch_norm(x, w[name + '.running_mean'], w[name + '.running_var'], w[name + '.weight'],
                     w[name + '.bias'], False, 0.0, eps)
    return F.leaky_relu(x, 0.01)


def cost_reg_net(w, x, pre=''):
    """MVSNet-style 3-D U-Net, eval mode (BatchNorm running stats).  x [B,Cin,D,h,w] -> [B,1,D,h,w].
    cost_reg.py:15-175."""
    def cbr(name, t, stride=1):
        return _bn_lrelu(w, pre + name + '.bn', _conv3(w, pre + name + '.conv', t, stride))

    def up(name, t):
        t = F.interpolate(t, scale_factor=2, mode='trilinear', align_corners=False)
        return _bn_lrelu(w, pre + name + '.bn', _conv3(w, pre + name + '.conv', t))

    c0 = cbr('conv0', x)
    c2 = cbr('conv2', cbr('conv1', c0, 2))
    c4 = cbr('conv4', cbr('conv3', c2, 2))
    t = cbr('conv6', cbr('conv5', c4, 2))
    t = c4 + up('conv7', t)
    t = c2 + up('conv9', t)
    t = c0 + up('conv11', t)
    return _conv3(w, pre + 'prob.conv', t)


def _block(w, name, x):
    """Conv3DBlockv2 without norm: (pad, conv3^3 + bias, LeakyReLU) x2. common_blocks.py:366-455."""
    x = F.leaky_relu(_conv3(w, name + '.conv1', x), 0.01)
    return F.leaky_relu(_conv3(w, name + '.conv2', x), 0.01)


def unet2(w, x, pre='', n_layers=3):
    """Default regulariser: encoders E0..E3 (avg-pool after E0..E2), decoders D2, D1, D0 with
    trilinear x2 upsampling; only E1's and E0's unpooled outputs are concatenated.
    common_blocks.py:187-242, test_models.py:84-146."""
    skips = []
    for i in range(n_layers + 1):
        x = _block(w, f'{pre}encoders.{i}', x)
        skips.append(x)
        if i < n_layers:
            x = F.avg_pool3d(x, 2)
    x = F.interpolate(x, scale_factor=2, mode='trilinear', align_corners=False)
    x = _block(w, f'{pre}decoders.{n_layers - 1}', x)
    for i in range(n_layers - 2, -1, -1):
        x = F.interpolate(x, scale_factor=2, mode='trilinear', align_corners=False)
        x = torch.cat((x, skips[i]), dim=1)
        x = _block(w, f'{pre}decoders.{i}', x)
    return x
