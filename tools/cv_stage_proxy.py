"""In-context proxy of bench.py's cost-volume stage: per step [cache pollution like a render pass: 1 GB read-modify-write] ->
hypotheses -> (event) cost volume (event) -> CostRegNet.  One process, environment knobs flipped per configuration."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panogrf_amd import cost_volume as CV
from panogrf_amd import cost_reg as CR
dev = 'cuda'
V, ch, cw = int(os.environ.get('PROXY_B', '2')), int(os.environ.get('PROXY_H', '128')), int(os.environ.get('PROXY_W', '256'))
g = torch.Generator().manual_seed(7)
feats = torch.randn(V, 2, ch, cw, 32, generator=g).to(dev)
ref_mu = (1.0 + 7.0 * torch.rand(V, 1, ch, cw, generator=g)).to(dev)
rots = torch.eye(3).view(1, 1, 3, 3).repeat(V, 2, 1, 1).to(dev)
trans = torch.tensor([[[0.5, 0.0, 0.0], [-0.5, 0.0, 0.0]]]).repeat(V, 1, 1).to(dev)
k = CV.depth_sampling(5, 3); args = dict(dataset_name='m3d', contain_dnet=True)
net = CR.CostRegNet(dict(use_wrap_padding=True), 32).to(dev).eval()
big = torch.zeros(256 * 1024 * 1024, device=dev)
KEYS = ('PGRF_CV_PIPE', 'PGRF_CV_TILE_W', 'PGRF_CV_DCH', 'PGRF_CV_ORDER', 'PGRF_CV_STREAM_STORES')
def setenv(kw):
    for kk in KEYS: os.environ.pop(kk, None)
    for kk, v in kw.items(): os.environ[kk] = str(v)
def measure(kw, n=6):
    setenv(kw)
    evs = []
    with torch.no_grad():
        for i in range(n + 2):
            big.add_(1.0)
            dv = CV.build_depth_volume(ref_mu, k, 0.5, 0.1, 10.0, 64)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            vol = CV.calculate_cost_volume_erp(args, feats, None, trans, rots, depth_volume=dv)
            e1.record()
            out = net.forward_ndhwc(vol)
            if i >= 2: evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 / V for a, b in evs)
    return t[len(t) // 2], t[0]
cfgs = [('old', dict(PGRF_CV_PIPE=0, PGRF_CV_TILE_W=64)), ('old_tw16', dict(PGRF_CV_PIPE=0, PGRF_CV_TILE_W=16))]
for pipe, tw, dch, order in itertools.product((1, 2, 4), (64, 16), (2, 4, 8, 16, 32), (0, 1)):
    cfgs.append((f'p{pipe}_tw{tw}_d{dch}_o{order}', dict(PGRF_CV_PIPE=pipe, PGRF_CV_TILE_W=tw, PGRF_CV_DCH=dch, PGRF_CV_ORDER=order)))
res = {}
for rnd in range(2):
    for name, kw in cfgs:
        res.setdefault(name, []).append(measure(kw)[0])
nbytes = 64 * ch * cw * 33 * 4 + 2 * ch * cw * 32 * 4
for name, v in sorted(res.items(), key=lambda q: min(q[1])):
    print(f'{ch}x{cw} B={V} {name:22s} ' + ' '.join(f'{x:6.1f}' for x in v) + f'  us per view -> {nbytes / min(v) / 1e6 / 8 * 100:5.1f} % of 8 TB/s')
