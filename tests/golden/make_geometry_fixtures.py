"""Golden vectors for the geometry of the other spherical conventions (cfg dataset_name = replica_test, residential,
CoffeeArea; network/spt_utils.py) and of pinhole queries (is_perspec=True; network/render_ops.py:35-74), produced by
the REFERENCE's own functions.  Build container only (needs /root/reference).  Re-run:
    python tests/golden/make_geometry_fixtures.py
(add --cv for the cost-volume conventions -> cv_conventions.npz)
Writes tests/golden/geometry_conventions.npz: inputs (coords, poses, depths) and the reference's sample points, view
directions, projected pixels and projected depths."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402
from make_fixtures import rot_xyz  # noqa: E402


def main():
    torch.set_num_threads(1)
    R.setup()
    from network import render_ops as ro
    from network.spt_utils import Utils
    H, W, rn, dn, rfn = 16, 32, 40, 6, 3
    g = torch.Generator().manual_seed(77)
    out = {}
    coords = torch.stack([torch.randint(0, W, (rn,), generator=g), torch.randint(0, H, (rn,), generator=g)], -1).float()
    # include the image corners (poles / seam)
    coords[:4] = torch.tensor([[0.0, 0.0], [W - 1.0, 0.0], [0.0, H - 1.0], [W - 1.0, H - 1.0]])
    depth = 0.5 + 6.0 * torch.rand(rn, dn, generator=g)
    Rq = torch.from_numpy(rot_xyz(0.4, -1.1, 0.3))
    tq = torch.tensor([0.03, -0.02, 0.05])
    c2w = torch.cat([Rq.t(), (-Rq.t() @ tq)[:, None]], 1)
    w2c = []
    for v in range(rfn):
        Rm = torch.from_numpy(rot_xyz(*((torch.rand(3, generator=g) - 0.5) * 3.0).tolist()))
        t = (torch.rand(3, generator=g) - 0.5) * 1.5
        w2c.append(torch.cat([Rm, t[:, None]], 1))
    w2c = torch.stack(w2c)
    out.update(coords=coords, depth=depth, que_c2w=c2w, w2c=w2c, H=np.int64(H), W=np.int64(W))
    for ds in ['m3d', 'replica_test', 'residential', 'CoffeeArea']:
        cfg = dict(height=H, width=W, dataset_name=ds, batch_size=1)
        try:
            spt = Utils(cfg)
        except TypeError:
            spt = Utils(H, W, ds, cfg) if False else None
        if spt is None:
            raise RuntimeError('Utils signature changed')
        que = dict(coords=coords[None], c2w=c2w[None])
        pts, qdir = ro.depth2points_spherical(que, depth[None], spt)
        px, dv = ro.project_points_coords(pts.reshape(-1, 3), w2c, spt)
        out[f'{ds}.pts'] = pts[0]
        out[f'{ds}.que_dir'] = qdir[0, :, 0]
        out[f'{ds}.px'] = px
        out[f'{ds}.depth_v'] = dv
    # pinhole query (a cube face: 90 degree field of view)
    Hp = Wp = 24
    K = torch.tensor([[Wp / 2.0, 0.0, Wp / 2.0], [0.0, Hp / 2.0, Hp / 2.0], [0.0, 0.0, 1.0]])
    pose = torch.cat([Rq, tq[:, None]], 1)
    pc = torch.stack([torch.randint(0, Wp, (rn,), generator=g), torch.randint(0, Hp, (rn,), generator=g)], -1).float()
    que = dict(coords=pc[None], poses=pose[None], Ks=K[None])
    pts, qdir = ro.depth2points_perspec(que, depth[None])
    out.update({'persp.coords': pc, 'persp.pose': pose, 'persp.K': K, 'persp.pts': pts[0], 'persp.que_dir': qdir[0, :, 0]})
    np.savez_compressed(os.path.join(HERE, 'geometry_conventions.npz'),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print('wrote geometry_conventions.npz', {k: tuple(np.asarray(v).shape) for k, v in out.items()})


if __name__ == '__main__' and '--cv' not in sys.argv:
    main()


def cost_volume_conventions():
    """2-view and multi-view spherical cost volumes of the REFERENCE in every dataset convention
    (models/spherical_cost_volume.py:231-341, spherical_cost_volume_mv.py) -> cv_conventions.npz"""
    torch.set_num_threads(1)
    R.setup()
    from models.spherical_cost_volume import calculate_cost_volume_erp
    from models.spherical_cost_volume_mv import calculate_cost_volume_erp_multiview
    g = torch.Generator().manual_seed(123)
    B, h, w, C, D, S = 1, 8, 16, 4, 5, 4
    feats = torch.randn(B, S, h, w, C, generator=g)
    rots = torch.stack([torch.from_numpy(rot_xyz(*((torch.rand(3, generator=g) - 0.5) * 1.0).tolist())) for _ in range(S)])[None]
    trans = (torch.rand(B, S, 3, generator=g) - 0.5) * 0.8
    dv = (0.6 + 7.0 * torch.rand(B, D, h, w, generator=g))
    out = dict(feats=feats, rots=rots, trans=trans, depth_volume=dv)
    for ds in ['m3d', 'replica_test', 'residential', 'CoffeeArea']:
        args = dict(dataset_name=ds, contain_dnet=True, mono_uncertainty=False)
        two = calculate_cost_volume_erp(args, feats[:, :2], None, trans[:, :2], rots[:, :2], depth_volume=dv)
        mv = calculate_cost_volume_erp_multiview(args, feats, None, trans, rots, depth_volume=dv, curr_idx=1)
        out[f'{ds}.two'] = two
        out[f'{ds}.mv'] = mv
    np.savez_compressed(os.path.join(HERE, 'cv_conventions.npz'),
                        **{k: v.detach().numpy() for k, v in out.items()})
    print('wrote cv_conventions.npz', {k: tuple(v.shape) for k, v in out.items()})


if __name__ == '__main__' and '--cv' in sys.argv:
    cost_volume_conventions()
