"""Polluted-cache proxy of the cost-volume stage: build + CostRegNet / UNet2 with plain vs streamed stores of the volume."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panogrf_amd import cost_volume as CV
from panogrf_amd import cost_reg as CR
dev = 'cuda'
big = torch.zeros(256 * 1024 * 1024, device=dev)
KEYS = ('PGRF_CV_PIPE', 'PGRF_CV_TILE_W', 'PGRF_CV_DCH', 'PGRF_CV_ORDER', 'PGRF_CV_STREAM_STORES')
def setenv(kw):
    for kk in KEYS: os.environ.pop(kk, None)
    for kk, v in kw.items(): os.environ[kk] = str(v)
k = CV.depth_sampling(5, 3); args = dict(dataset_name='m3d', contain_dnet=True)
for regname in ('costregnet', 'unet2'):
    net = (CR.UNet2Reg() if regname == 'unet2' else CR.CostRegNet(dict(use_wrap_padding=True), 32)).to(dev).eval()
    for (V, ch, cw) in [(1, 128, 256), (2, 128, 256), (1, 64, 128), (2, 64, 128)]:
        g = torch.Generator().manual_seed(7)
        feats = torch.randn(V, 2, ch, cw, 32, generator=g).to(dev)
        ref_mu = (1.0 + 7.0 * torch.rand(V, 1, ch, cw, generator=g)).to(dev)
        rots = torch.eye(3).view(1, 1, 3, 3).repeat(V, 2, 1, 1).to(dev)
        trans = torch.tensor([[[0.5, 0.0, 0.0], [-0.5, 0.0, 0.0]]]).repeat(V, 1, 1).to(dev)
        def measure(kw, n=6):
            setenv(kw)
            evs = []
            with torch.no_grad():
                for i in range(n + 2):
                    big.add_(1.0)
                    dv = CV.build_depth_volume(ref_mu, k, 0.5, 0.1, 10.0, 64)
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    e[0].record()
                    vol = CV.calculate_cost_volume_erp(args, feats, None, trans, rots, depth_volume=dv)
                    e[1].record()
                    out = net.forward_ndhwc(vol)
                    e[2].record()
                    if i >= 2: evs.append(e)
            torch.cuda.synchronize()
            b = sorted(a[0].elapsed_time(a[1]) * 1e3 / V for a in evs); t = sorted(a[0].elapsed_time(a[2]) * 1e3 / V for a in evs)
            return b[len(b) // 2], t[len(t) // 2]
        for name, kw in [('r05 kernel, r06 stores', dict(PGRF_CV_PIPE=0, PGRF_CV_TILE_W=64)), ('shipped', {}), ('shipped, plain stores', dict(PGRF_CV_STREAM_STORES=0)),
                         ('shipped, streamed stores', dict(PGRF_CV_STREAM_STORES=1))]:
            r = [measure(kw) for _ in range(2)]
            print(f'{regname:10s} {ch}x{cw} B={V} {name:26s} build ' + ' '.join(f'{x[0]:6.1f}' for x in r) + ' us   build + regulariser ' + ' '.join(f'{x[1]:7.1f}' for x in r) + ' us per view', flush=True)
