"""Timing of the cost-volume side on the GPU box (not part of the bench contract).
python tools/time_cost_volume.py [h w]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panogrf_amd import cost_volume as CV
from panogrf_amd import cost_reg as R

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
only = sys.argv[3] if len(sys.argv) > 3 else None     # 'costregnet' | 'unet2': time just that regulariser
B, D, C = int(os.environ.get('PGRF_TCV_B', '1')), 64, 32      # PGRF_TCV_B=2: the two reference views of the MVS stage in one batch
g = torch.Generator().manual_seed(0)
feats = torch.randn(B, 2, h, w, C, generator=g).cuda()
ref_mu = (1 + 7 * torch.rand(B, 1, h, w, generator=g)).cuda()
eye = torch.eye(3).view(1, 1, 3, 3).repeat(B, 2, 1, 1).cuda()
trans = torch.tensor([[[0.5, 0, 0], [-0.5, 0, 0]]]).repeat(B, 1, 1).cuda()
args = dict(dataset_name='m3d', contain_dnet=True)
k = CV.depth_sampling(5, 3)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


dv = CV.build_depth_volume(ref_mu, k, 0.5, 0.1, 10, 64)
t = timeit(lambda: CV.build_depth_volume(ref_mu, k, 0.5, 0.1, 10, 64))
print(f'hypotheses {h}x{w}: {t*1e3:.1f} us')
cv = CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv)
t = timeit(lambda: CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv))
nbytes = B * D * h * w * (C + 1) * 4 + B * 2 * h * w * C * 4
print(f'cost volume {h}x{w}xD{D}: {t*1e3:.1f} us  -> {nbytes/t/1e6:.1f} GB/s algorithmic ({nbytes/1e6:.1f} MB)')
for name, net, gf in [('CostRegNet', R.CostRegNet(dict(use_wrap_padding=True), 32), 16.2e9 * (h * w) / (64 * 128)),
                      ('UNet2', R.UNet2Reg(), 438.5e9 * (h * w) / (64 * 128))]:
    if only and name.lower() != only:
        continue
    net = net.cuda().eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
        out = net.forward_ndhwc(cv)
        t = timeit(lambda: net.forward_ndhwc(cv), n=20)         # (n = 3 until round 3: that included ~15 % of start-up effects)
    print(f'{name} {h}x{w}: {t:.2f} ms -> {gf/t/1e9:.1f} TFLOP/s algorithmic, out finite={bool(torch.isfinite(out).all())}')
