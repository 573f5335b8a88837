"""Randomised spherical cost volumes (views, channels, datasets, cost types) against the oracle (script)."""
import sys
import torch
sys.path.insert(0, '.')
from oracle import cost_volume_oracle as O
from panogrf_amd import cost_volume as CV

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g))


def rot():
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.float()


worst = 0.0
for it in range(n):
    ds = ['m3d', 'replica_test', 'residential', 'CoffeeArea'][ri(0, 4)]
    ct = ['abs_diff', 'dot', 'none'][ri(0, 3)]
    C = [4, 8, 16, 32, 64][ri(0, 5)]
    B, h, w, D = ri(1, 3), ri(2, 12), ri(2, 24), ri(1, 7)
    S = ri(2, 6)
    feats = torch.randn(B, S, h, w, C, generator=g)
    rots = torch.stack([torch.stack([rot() for _ in range(S)]) for _ in range(B)])
    trans = (torch.rand(B, S, 3, generator=g) - 0.5) * 1.0
    dv = 0.5 + 8 * torch.rand(B, D, h, w, generator=g)
    args = dict(dataset_name=ds, contain_dnet=True)
    if S == 2:
        ref = O.cost_volume_erp(feats, dv, rots, trans, ct, ds)
        got = CV.calculate_cost_volume_erp(args, feats.cuda(), None, trans.cuda(), rots.cuda(), depth_volume=dv.cuda(), cost_type=ct)
    else:
        ci = ri(0, S)
        ref = O.cost_volume_erp_multiview(feats, dv, rots, trans, ci, ct, ds)
        got = CV.calculate_cost_volume_erp_multiview(args, feats.cuda(), None, trans.cuda(), rots.cuda(), depth_volume=dv.cuda(), cost_type=ct, curr_idx=ci)
    e = float((got.cpu() - ref).abs().max())
    worst = max(worst, e)
    print(f'{it:3d} {ds:12s} {ct:8s} C={C:2d} B={B} S={S} {h}x{w} D={D}  err {e:.2e}' + ('' if e < 5e-4 else '  <-- CHECK'))
print('worst', worst)
