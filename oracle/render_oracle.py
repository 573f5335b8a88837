"""CPU oracle for the per-ray spherical volume renderer (SURVEY.md section 8 rows a-1 .. a-10, a-15).

TEST INFRASTRUCTURE: see oracle/__init__.py.  Plain PyTorch fp32, written from the math of
the reference; file:line citations are into /root/reference.  Pinned by
tests/golden/render_*.npz (captured from the reference's own NeuralRayBaseRenderer).

Conventions: rfn = source views, rn = rays, dn = samples per ray.  `weights` is a flat
dict keyed with the reference's state_dict names (e.g. 'agg_net.agg_impl.base_fc.0.weight').
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- geometry (a-1, a-2)
def erp_ray_directions(H, W, dataset='m3d'):
    """Unit camera-frame direction of every ERP pixel, [H,W,3], in the dataset's spherical convention.
    network/ray_utils.py:4-16, network/spt_utils.py:21-35 (pixel grid), :45-83 (pixel -> angles per dataset),
    :106-124 (angles -> cartesian per dataset)."""
    x = torch.linspace(0, W - 1, W).view(1, W).expand(H, W)
    y = torch.linspace(0, H - 1, H).view(H, 1).expand(H, W)
    if dataset == 'm3d':
        theta = x / (W - 1) * 2 * math.pi - 0.5 * math.pi
        phi = y / (H - 1) * math.pi
        s = torch.sin(phi)
        d = torch.stack([s * torch.cos(theta), torch.cos(phi), s * torch.sin(theta)], -1)
    elif dataset == 'replica_test':
        theta = x * 2 * math.pi / (W - 1) - math.pi
        phi = -y * math.pi / (H - 1) + math.pi * 0.5
        d = torch.stack([torch.sin(theta) * torch.cos(phi), -torch.sin(phi), torch.cos(theta) * torch.cos(phi)], -1)
    elif dataset == 'residential':
        theta = math.pi * (2 * x / (W - 1) - 1.5)
        phi = math.pi * (0.5 - y / (H - 1))
        d = torch.stack([torch.cos(theta) * torch.cos(phi), torch.sin(phi), torch.sin(theta) * torch.cos(phi)], -1)
    elif dataset == 'CoffeeArea':
        theta = (-2 * math.pi / (W - 1)) * x + 2 * math.pi
        phi = (math.pi / (H - 1)) * y
        d = torch.stack([torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi)], -1)
    else:
        raise ValueError(f'unknown dataset convention {dataset!r}')
    return d / torch.norm(d, p=2, dim=-1, keepdim=True)


def sample_coarse_depth(min_depth, max_depth, rn, dn, use_disp=True, device='cpu', jitter=None):
    """Stratified depths, [rn,dn]. network/render_ops.py:292-339.  Eval: deterministic; training: `jitter` [rn,dn-2]
    uniform draws move the interior samples by (r - 0.5) * 0.999 bins (:310-311, :329-330)."""
    near = torch.tensor(float(min_depth), device=device)
    far = torch.tensor(float(max_depth), device=device)
    val = torch.arange(1, dn - 1, dtype=torch.float32, device=device).view(1, dn - 2).expand(rn, dn - 2)
    if jitter is not None:
        val = val + (jitter.to(device) - 0.5) * 0.999
    zero = torch.zeros(rn, 1, device=device)
    if use_disp:
        interval = (1 / far - 1 / near) / (dn - 1)
        ticks = torch.cat([zero, interval * val, (1 / far - 1 / near).view(1, 1).expand(rn, 1)], -1)
        depth = 1 / (1 / near + ticks)
    else:
        interval = (far - near) / (dn - 1)
        ticks = torch.cat([zero, interval * val, (far - near).view(1, 1).expand(rn, 1)], -1)
        depth = near + ticks
    return depth.contiguous()


def normalized_inverse_depth(depth, near, far):
    """(-1/z - (-1/near)) / ((-1/far) - (-1/near)); render_ops.py:115-121, dist_decoder.py:16-23."""
    n, f = -1 / near, -1 / far
    return (-1 / depth - n) / (f - n)


def inv_dists(depth, que_depth_range):
    """Consecutive differences of normalised inverse depth, last = 1e6. render_ops.py:110-122."""
    near, far = que_depth_range[0], que_depth_range[1]
    di = normalized_inverse_depth(depth, near, far)
    d = di[..., 1:] - di[..., :-1]
    return torch.cat([d, torch.full_like(di[..., :1], 1e6)], -1)


def ray_points(coords, c2w, depth, H, W, dataset='m3d'):
    """World sample points and reversed unit view directions of an ERP query.
    coords [rn,2] (x,y), c2w [3,4], depth [rn,dn] -> pts [rn,dn,3], que_dir [rn,3].
    render_ops.py:76-106, ray_utils.py:53-71."""
    dirs = erp_ray_directions(H, W, dataset).to(coords.device)
    d_cam = dirs[coords[:, 1].long(), coords[:, 0].long()]            # [rn,3]
    d = torch.matmul(c2w[:3, :3].view(1, 3, 3), d_cam.unsqueeze(-1)).squeeze(-1)
    o = c2w[:, 3].view(1, 3)
    pts = o.unsqueeze(1) + d.unsqueeze(1) * depth.unsqueeze(-1)
    que_dir = -d / torch.norm(d, dim=1, keepdim=True)
    return pts, que_dir


def ray_points_perspective(coords, pose, K, depth):
    """Pinhole query (is_perspec=True): coords [rn,2] pinhole pixels, pose [3,4] world->camera, K [3,3].
    Directions are NOT normalised (depth is the camera-space z).  render_ops.py:35-74."""
    rot = pose[:, :3].t()
    trans = -rot @ pose[:, 3:]                                          # [3,1] camera centre
    rn = coords.shape[0]
    hc = torch.cat([coords, torch.ones(rn, 1, dtype=torch.float32, device=coords.device)], 1)
    cam = torch.inverse(K).view(1, 3, 3) @ hc.unsqueeze(-1)              # [rn,3,1]
    cam = rot.view(1, 3, 3) @ cam + trans.view(1, 3, 1)
    centers = trans.view(1, 3).expand(rn, 3)
    d = cam.squeeze(-1) - centers
    pts = centers.unsqueeze(1) + d.unsqueeze(1) * depth.unsqueeze(-1)
    que_dir = -d / torch.norm(d, dim=1, keepdim=True)
    return pts, que_dir


# --------------------------------------------------------------------------- projection (a-3, a-4)
def project_to_views(pts, w2c, H, W, dataset='m3d'):
    """pts [pn,3], w2c [rfn,3,4] -> px [rfn,pn,2] (x,y in full-image pixels), depth [rfn,pn].
    render_ops.py:158-191, ray_utils.py:18-22, spt_utils.py:128-156 (cartesian -> angles per dataset),
    :172-192 (angles -> pixel per dataset)."""
    R, t = w2c[:, :, :3], w2c[:, :, 3]
    pc = torch.einsum('vij,pj->vpi', R, pts) + t[:, None, :]
    r = torch.linalg.norm(pc, dim=-1)
    xc, yc, zc = pc[..., 0], pc[..., 1], pc[..., 2]
    if dataset == 'm3d':
        theta = torch.atan2(zc, xc)
        phi = torch.acos(yc / (r + 1e-5))
        theta = torch.remainder(theta + 0.5 * math.pi, 2 * math.pi)
        x = theta / (2 * math.pi) * (W - 1)
        y = phi / math.pi * (H - 1)
    elif dataset == 'replica_test':
        theta = torch.atan2(xc, zc)
        phi = -torch.asin(yc / r)
        x = ((W - 1) / (2 * math.pi)) * (theta + math.pi)
        y = (H - 1) / math.pi * (-phi + 0.5 * math.pi)
    elif dataset == 'residential':
        theta = -torch.atan2(-zc, xc)
        phi = torch.asin(yc / r)
        mask = torch.logical_and(theta.gt(math.pi * 0.5), theta.le(2 * math.pi))
        theta = torch.where(mask, theta - 2 * math.pi, theta)
        x = ((1 / (2.0 * math.pi)) * theta + (3 / 4.0)) * (W - 1)
        y = (0.5 - phi / math.pi) * (H - 1)
    elif dataset == 'CoffeeArea':
        theta = torch.atan2(yc, xc)
        phi = torch.acos(zc / r)
        theta = torch.where(theta.lt(0), theta + 2 * math.pi, theta)
        x = (W - 1) * (1 - theta / (2.0 * math.pi))
        y = phi * (H - 1) / math.pi
    else:
        raise ValueError(f'unknown dataset convention {dataset!r}')
    return torch.stack([x, y], -1), r


def project_directions(pts, w2c):
    """Unit direction from sample point towards each source camera centre, [rfn,pn,3].
    render_ops.py:195-209."""
    cam = -torch.matmul(w2c[:, :, :3].transpose(1, 2), w2c[:, :, 3:])      # [rfn,3,1]
    d = pts.unsqueeze(0) - cam.transpose(1, 2)
    return -d / torch.clamp_min(torch.norm(d, dim=2, keepdim=True), 1e-5)


def dir_diff(prj_dir, que_dir):
    """[prj_dir - que_dir, <prj_dir,que_dir>] -> [...,4]. aggregate_net.py:8-14."""
    return torch.cat([prj_dir - que_dir, torch.sum(prj_dir * que_dir, -1, keepdim=True)], -1)


# --------------------------------------------------------------------------- gather (a-5)
def gather_map(fmap, px, H, W):
    """Bilinear gather of fmap [rfn,C,fh,fw] at px [rfn,pn,2] given in FULL-image pixels (H,W).
    border padding, align_corners iff the map is at full resolution.
    render_ops.py:126-143, ops.py:32-52."""
    fh, fw = fmap.shape[-2:]
    ac = (fh == H and fw == W)
    xn = px[..., 0] / (W - 1) * 2 - 1
    yn = px[..., 1] / (H - 1) * 2 - 1
    grid = torch.stack([xn, yn], -1).unsqueeze(1)
    out = F.grid_sample(fmap, grid, mode='bilinear', padding_mode='border', align_corners=ac)
    return out.squeeze(2).permute(0, 2, 1)                                  # [rfn,pn,C]


def gather_map_explicit(fmap, px, H, W):
    """Same as gather_map with the grid_sample arithmetic spelled out (documents what the HIP
    kernel implements; checked equal to gather_map in tests)."""
    rfn, C, fh, fw = fmap.shape
    ac = (fh == H and fw == W)
    xn = px[..., 0] / (W - 1) * 2 - 1
    yn = px[..., 1] / (H - 1) * 2 - 1
    if ac:
        ix = (xn + 1) / 2 * (fw - 1)
        iy = (yn + 1) / 2 * (fh - 1)
    else:
        ix = ((xn + 1) * fw - 1) / 2
        iy = ((yn + 1) * fh - 1) / 2
    ix = ix.clamp(0, fw - 1)
    iy = iy.clamp(0, fh - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    wnw = (x1 - ix) * (y1 - iy)
    wne = (ix - x0) * (y1 - iy)
    wsw = (x1 - ix) * (iy - y0)
    wse = (ix - x0) * (iy - y0)
    out = torch.zeros(rfn, px.shape[1], C)
    for v in range(rfn):
        m = fmap[v].permute(1, 2, 0)                                        # [fh,fw,C]

        def tap(yy, xx, ww):
            ok = (xx >= 0) & (xx <= fw - 1) & (yy >= 0) & (yy <= fh - 1)
            xi = xx.clamp(0, fw - 1).long()
            yi = yy.clamp(0, fh - 1).long()
            return m[yi, xi] * (ww * ok.float()).unsqueeze(-1)
        out[v] = tap(y0[v], x0[v], wnw[v]) + tap(y0[v], x1[v], wne[v]) + \
            tap(y1[v], x0[v], wsw[v]) + tap(y1[v], x1[v], wse[v])
    return out


# --------------------------------------------------------------------------- MLP helpers
def _lin(w, name, x):
    return F.linear(x, w[name + '.weight'], w.get(name + '.bias'))


def _mlp3(w, pre, x, last):
    x = F.elu(_lin(w, pre + '.0', x))
    x = F.elu(_lin(w, pre + '.2', x))
    return last(_lin(w, pre + '.4', x))


# --------------------------------------------------------------------------- dist decoder (a-6)
def dist_decoder(w, pre, feats, use_vis=False, bias_val=0.05):
    """Mixture-of-2-logistics parameters from gathered ray_feats [...,32].
    dist_decoder.py:53-107."""
    mean = _mlp3(w, pre + '.mean_decoder', feats, F.softplus)
    var = _mlp3(w, pre + '.var_decoder', feats, F.softplus) + bias_val
    aw = _mlp3(w, pre + '.aw_decoder', feats, torch.sigmoid)
    vis = _mlp3(w, pre + '.vis_decoder', feats, torch.sigmoid) if use_vis else None
    return mean, var, vis, aw


def compute_prob(depth_v, que_dists, mean, var, vis, aw, ref_depth_range):
    """Visibility / hit probability of every sample seen from every source view.
    depth_v [rfn,rn,dn] (distance of the point from the source camera), que_dists [rn,dn],
    mean,var [rfn,rn,dn,2], aw [rfn,rn,dn,1], ref_depth_range [rfn,2].
    dist_decoder.py:6-51 (is_ref branch), :109-140."""
    near_r = ref_depth_range[:, 0].view(-1, 1, 1)
    far_r = ref_depth_range[:, 1].view(-1, 1, 1)
    z = normalized_inverse_depth(torch.clamp(depth_v, min=1e-5), near_r, far_r)
    half = (que_dists / 2).unsqueeze(0)
    ext = torch.cat([half[..., 0:1], half], -1)
    near = (z - ext[..., :-1]).unsqueeze(-1)
    far = (z + ext[..., 1:]).unsqueeze(-1)
    mix = torch.cat([aw, 1 - aw], -1)
    cdf0 = 0.5 + 0.5 * torch.tanh((near - mean) * var)
    cdf1 = 0.5 + 0.5 * torch.tanh((far - mean) * var)
    if vis is not None:
        cdf0, cdf1 = cdf0 * vis, cdf1 * vis
    visibility = torch.sum((1 - cdf0) * mix, -1)
    hit_prob = torch.sum((cdf1 - cdf0) * mix, -1)
    return visibility, hit_prob


# --------------------------------------------------------------------------- aggregation (a-7, a-8)
def pos_encoding(n_samples, d_hid=16):
    """Sinusoid table [n_samples,16]. ibrnet.py:305-313."""
    pos = np.arange(n_samples, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    tab = pos / np.power(10000, 2 * (j // 2) / d_hid)
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return torch.from_numpy(tab).float()


def _wmean_var(x, wgt):
    """Weighted mean / variance over the view axis (dim 2). ibrnet.py:112-116."""
    mean = torch.sum(x * wgt, dim=2, keepdim=True)
    var = torch.sum(wgt * (x - mean) ** 2, dim=2, keepdim=True)
    return mean, var


def ray_attention(w, pre, g, masked=False):
    """4-head self attention over the samples of each ray + residual + LayerNorm.
    g [rn,dn,16]. ibrnet.py:52-102, :7-27 (temperature 2).
    The reference passes mask=(num_valid_obs > 1) (ibrnet.py:358-360) and every projection counts as valid, so the
    mask is all ones for rfn >= 2 and all ZERO for a single source view: `masked_fill(mask == 0, -1e9)` (ibrnet.py:20)
    then sets every logit to -1e9 and the softmax is uniform -> `masked=True` gives the mean of V over the ray."""
    rn, dn, _ = g.shape
    q = F.linear(g, w[pre + '.w_qs.weight']).view(rn, dn, 4, 4).transpose(1, 2)
    k = F.linear(g, w[pre + '.w_ks.weight']).view(rn, dn, 4, 4).transpose(1, 2)
    v = F.linear(g, w[pre + '.w_vs.weight']).view(rn, dn, 4, 4).transpose(1, 2)
    attn = torch.matmul(q / 2.0, k.transpose(2, 3))
    if masked:
        attn = torch.full_like(attn, -1e9)
    attn = F.softmax(attn, dim=-1)
    o = torch.matmul(attn, v).transpose(1, 2).reshape(rn, dn, 16)
    o = F.linear(o, w[pre + '.fc.weight']) + g
    return F.layer_norm(o, (16,), w[pre + '.layer_norm.weight'], w[pre + '.layer_norm.bias'], 1e-6)


def aggregate(w, pre, rgb, img_feats, ray_feats, hit_prob, vis, ddiff, want_debug=False):
    """IBRNet-with-NeuRay aggregation.  All inputs view-major:
    rgb [rfn,rn,dn,3], img_feats [rfn,rn,dn,32], ray_feats [rfn,rn,dn,32],
    hit_prob, vis [rfn,rn,dn], ddiff [rfn,rn,dn,4].  Returns density [rn,dn], colors [rn,dn,3].
    aggregate_net.py:41-89, ibrnet.py:315-373."""
    ag = pre + '.agg_impl'
    rfn, rn, dn, _ = rgb.shape
    dbg = {}
    pe_in = torch.cat([ray_feats, ((hit_prob - 0.5) * 2).unsqueeze(-1), ((vis - 0.5) * 2).unsqueeze(-1)], -1)
    neuray = _lin(w, pre + '.prob_embed.2', F.relu(_lin(w, pre + '.prob_embed.0', pe_in)))
    # to [rn,dn,rfn,*]
    neuray = neuray.permute(1, 2, 0, 3)
    rgb_feat = torch.cat([rgb, img_feats], -1).permute(1, 2, 0, 3)
    rdiff = ddiff.permute(1, 2, 0, 3)
    rgb_in = rgb_feat[..., :3]

    dfeat = F.elu(_lin(w, ag + '.ray_dir_fc.2', F.elu(_lin(w, ag + '.ray_dir_fc.0', rdiff))))
    rgb_feat = rgb_feat + dfeat
    wgt = torch.full((rn, dn, rfn, 1), 1.0, device=rgb.device) / (rfn + 1e-8)
    nf = _lin(w, ag + '.neuray_fc.2', F.elu(_lin(w, ag + '.neuray_fc.0', neuray)))
    wgt0 = torch.sigmoid(nf) * wgt
    mean0, var0 = _wmean_var(rgb_feat, wgt0)
    mean1, var1 = _wmean_var(rgb_feat, wgt)
    glob = torch.cat([mean0, var0, mean1, var1], -1).expand(-1, -1, rfn, -1)
    x = torch.cat([glob, rgb_feat, neuray], -1)
    x = F.elu(_lin(w, ag + '.base_fc.2', F.elu(_lin(w, ag + '.base_fc.0', x))))
    dbg['base_x'] = x
    xv = F.elu(_lin(w, ag + '.vis_fc.2', F.elu(_lin(w, ag + '.vis_fc.0', x * wgt))))
    x_res, v1 = xv[..., :32], torch.sigmoid(xv[..., 32:33])
    x = x + x_res
    v2 = torch.sigmoid(_lin(w, ag + '.vis_fc2.2', F.elu(_lin(w, ag + '.vis_fc2.0', x * v1))))
    wgt2 = v2 / (torch.sum(v2, dim=2, keepdim=True) + 1e-8)
    mean, var = _wmean_var(x, wgt2)
    g = torch.cat([mean.squeeze(2), var.squeeze(2), wgt2.mean(dim=2)], -1)
    g = F.elu(_lin(w, ag + '.geometry_fc.2', F.elu(_lin(w, ag + '.geometry_fc.0', g))))
    dbg['geo'] = g
    g = g + pos_encoding(dn).unsqueeze(0).to(g.device)
    g = ray_attention(w, ag + '.ray_attention', g, masked=(rfn <= 1))
    dbg['attn'] = g
    sigma = F.relu(_lin(w, ag + '.out_geometry_fc.2', F.elu(_lin(w, ag + '.out_geometry_fc.0', g))))
    xr = torch.cat([x, v2, rdiff], -1)
    xr = F.elu(_lin(w, ag + '.rgb_fc.0', xr))
    xr = F.elu(_lin(w, ag + '.rgb_fc.2', xr))
    xr = _lin(w, ag + '.rgb_fc.4', xr)
    blend = F.softmax(xr, dim=2)
    colors = torch.sum(rgb_in * blend, dim=2)
    if want_debug:
        dbg['neuray'] = neuray
        return sigma.squeeze(-1), colors, dbg
    return sigma.squeeze(-1), colors


# --------------------------------------------------------------------------- composite (a-9)
def composite(density, colors, depth):
    """alpha = 1-exp(-relu(sigma)); T_i = prod_{j<i}(1-alpha_j+1e-10); w = alpha*T.
    renderer.py:210-219,291-304; render_ops.py:145-153."""
    alpha = 1.0 - torch.exp(-torch.relu(density))
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1. - alpha + 1e-10], -1), -1)[..., :-1]
    hit = alpha * T
    pixel = torch.sum(hit.unsqueeze(-1) * colors, -2)
    rdepth = torch.sum(hit * depth, -1)
    return hit, pixel, rdepth


# --------------------------------------------------------------------------- importance sampling (a-10)
def sample_fine_depth(depth, hit_prob, que_depth_range, fdn, use_disp=True, u=None):
    """Inverse-CDF resampling in normalised inverse depth, then sort.  Eval: bin-centre positions; training: `u`
    [rn,fdn] uniform draws (render_ops.py:449).
    depth, hit_prob [rn,dn] -> [rn,fdn].  render_ops.py:413-473, renderer.py:469-472."""
    near, far = que_depth_range[0], que_depth_range[1]
    if use_disp:
        d = normalized_inverse_depth(depth, near, far)
    else:
        d = depth
    centre = torch.cat([d[..., :1], (d[..., 1:] + d[..., :-1]) / 2, d[..., -1:]], -1)   # dn+1
    hp = hit_prob + 1e-5
    pdf = hp / torch.sum(hp, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)         # dn+1
    interval = 1 / fdn
    if u is None:
        u = (0.5 * interval + torch.arange(fdn) * interval).to(cdf.device).expand(*cdf.shape[:-1], fdn).contiguous()
    else:
        u = u.to(cdf.device).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(centre, -1, below), torch.gather(centre, -1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    fd = b0 + (u - c0) / denom * (b1 - b0)
    if use_disp:
        n, f = -1 / near, -1 / far
        fd = -1 / (fd * (f - n) + n)
    return torch.sort(fd, -1)[0]


# --------------------------------------------------------------------------- one pass / whole render
def render_pass(weights, dd_pre, agg_pre, frame, coords, depth, cfg, want_debug=False):
    """One network pass (coarse or fine) over rays `coords` [rn,2] at sample depths [rn,dn].
    frame: dict(imgs [rfn,3,H,W], img_feats [rfn,32,hf,wf], ray_feats [rfn,32,hr,wr], w2c [rfn,3,4],
                ref_depth_range [rfn,2], que_c2w [3,4], que_depth_range [2]; optional dataset_name,
                and for a pinhole query que_pose [3,4] + que_K [3,3]).
    renderer.py:223-304 (render_by_depth, non-debug branch)."""
    H, W = frame['imgs'].shape[-2:]
    rfn = frame['imgs'].shape[0]
    rn, dn = depth.shape
    qd = inv_dists(depth, frame['que_depth_range'])
    ds = frame.get('dataset_name', 'm3d')
    if frame.get('que_K') is not None:        # pinhole query: is_perspec=True
        pts, que_dir = ray_points_perspective(coords, frame['que_pose'], frame['que_K'], depth)
    else:
        pts, que_dir = ray_points(coords, frame['que_c2w'], depth, H, W, ds)
    flat = pts.reshape(rn * dn, 3)
    px, dv = project_to_views(flat, frame['w2c'], H, W, ds)
    pdir = project_directions(flat, frame['w2c'])
    rf = gather_map(frame['ray_feats'], px, H, W)
    rgb = gather_map(frame['imgs'], px, H, W)
    imf = gather_map(frame['img_feats'], px, H, W)
    use_vis = bool(cfg.get('use_vis', False))
    mean, var, vis_p, aw = dist_decoder(weights, dd_pre, rf, use_vis)
    v5 = lambda t: None if t is None else t.reshape(rfn, rn, dn, -1)
    vis, hit = compute_prob(dv.reshape(rfn, rn, dn), qd, v5(mean), v5(var), v5(vis_p), v5(aw),
                            frame['ref_depth_range'])
    qdir_pts = que_dir.unsqueeze(1).expand(rn, dn, 3).reshape(1, rn * dn, 3)
    dd = dir_diff(pdir, qdir_pts).reshape(rfn, rn, dn, 4)
    out = aggregate(weights, agg_pre, rgb.reshape(rfn, rn, dn, 3), imf.reshape(rfn, rn, dn, 32),
                    rf.reshape(rfn, rn, dn, 32), hit, vis, dd, want_debug)
    density, colors = out[0], out[1]
    hitq, pixel, rdepth = composite(density, colors, depth)
    res = {'pixel_colors_nr': pixel, 'hit_prob_nr': hitq, 'colors_nr': colors, 'density_nr': density,
           'render_depth': rdepth}
    if cfg.get('render_uncert'):
        res['render_uncert'] = depth_uncertainty(depth, rdepth, hitq)
    if want_debug:
        res['dbg'] = dict(px=px.reshape(rfn, rn, dn, 2), depth_v=dv.reshape(rfn, rn, dn), ddiff=dd,
                          ray_feats=rf.reshape(rfn, rn, dn, 32), rgb=rgb.reshape(rfn, rn, dn, 3),
                          img_feats=imf.reshape(rfn, rn, dn, 32), vis=vis, hit=hit, que_dists=qd,
                          **out[2])
    return res


def depth_uncertainty(depth, rdepth, hit):
    """cfg render_uncert: variance of the sample depths under the hit probabilities, + 1e-5.  renderer.py:305-312."""
    return ((depth - rdepth.unsqueeze(-1)).pow(2) * hit).sum(-1) + 1e-5


def merge_c2f_all(coarse_depth, coarse, fine_depth, fine, uncert=False):
    """`render_c2f_all`: one composite over the union of coarse and fine samples.
    renderer.py:484-518."""
    z = torch.cat([coarse_depth, fine_depth], -1)
    col = torch.cat([coarse['colors_nr'], fine['colors_nr']], -2)
    den = torch.cat([coarse['density_nr'], fine['density_nr']], -1)
    z, idx = z.sort(-1)
    col = torch.gather(col, -2, idx.unsqueeze(-1).expand_as(col))
    den = torch.gather(den, -1, idx)
    hit, pixel, rdepth = composite(den, col, z)
    res = {'pixel_colors_nr': pixel, 'hit_prob_nr': hit, 'colors_nr': col, 'density_nr': den,
           'render_depth': rdepth}
    if uncert:
        res['render_uncert'] = depth_uncertainty(z, rdepth, hit)
    return res


def render_rays(weights, frame, coords, cfg, want_debug=False, jitter=None, u=None):
    """Coarse (+ fine) render of rays `coords` [rn,2]; keys follow renderer.py:567-633 with the
    leading qn=1 axis dropped.  jitter / u: the uniform draws of a training-mode forward (is_train=True)."""
    rn = coords.shape[0]
    dn = cfg['depth_sample_num']
    depth = sample_coarse_depth(cfg['min_depth'], cfg['max_depth'], rn, dn, cfg.get('use_disp', True), coords.device, jitter)
    unc = bool(cfg.get('render_uncert', False))
    ccfg = {'use_vis': cfg.get('dist_decoder_cfg', {}).get('use_vis', True), 'render_uncert': unc}
    out = render_pass(weights, 'dist_decoder', 'agg_net', frame, coords, depth, ccfg, want_debug)
    out['depth'] = depth
    if cfg.get('use_hierarchical_sampling', False):
        fdn = cfg['fine_depth_sample_num']
        fd = sample_fine_depth(depth, out['hit_prob_nr'], frame['que_depth_range'], fdn, cfg.get('use_disp', True), u)
        if cfg.get('fine_depth_use_all', False):
            fd = torch.sort(torch.cat([depth, fd], -1), -1)[0]
        fcfg = {'use_vis': cfg.get('fine_dist_decoder_cfg', {}).get('use_vis', True), 'render_uncert': unc}
        # quirk kept: the fine pass evaluates compute_prob through the COARSE decoder object,
        # i.e. with the coarse decoder's use_vis flag (renderer.py:128)
        fcfg_prob = dict(fcfg)
        if cfg.get('one_mlp', False):
            fine = render_pass(weights, 'dist_decoder', 'agg_net', frame, coords, fd, ccfg, want_debug)
        else:
            fine = render_pass(weights, 'fine_dist_decoder', 'fine_agg_net', frame, coords, fd, fcfg_prob, want_debug)
        if cfg.get('render_c2f_all', False):
            fine.update(merge_c2f_all(depth, out, fd, fine, unc))
        fine['depth'] = fd
        for k, v in fine.items():
            out[k + '_fine'] = v
    return out


def render_frame(weights, frame, cfg, ray_batch_num=None):
    """Whole panorama in ray batches; the result must not depend on the batch size.
    renderer.py:635-686 (without the per-frame 2-D encoders)."""
    H, W = frame['imgs'].shape[-2:]
    dev = frame['imgs'].device
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev),
                            indexing='ij')
    coords = torch.stack([xs, ys], -1).reshape(-1, 2)
    rb = ray_batch_num or cfg.get('ray_batch_num', 2048)
    outs = {}
    for s in range(0, coords.shape[0], rb):
        o = render_rays(weights, frame, coords[s:s + rb], cfg)
        for k, v in o.items():
            if k.startswith('hit_prob') or k.startswith('dbg'):
                continue
            outs.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in outs.items()}
