"""cost-volume build with COLD caches (a 2 GB read-modify-write between launches, as a frame's other kernels leave them) against a
plain fill of the same buffer: what the HBM write path gives a kernel that writes 268 MB once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panogrf_amd import cost_volume as CV
D, C = 64, 32
k = CV.depth_sampling(5, 3)
args = dict(dataset_name='m3d', contain_dnet=True)
big = torch.zeros(512 * 1024 * 1024, device='cuda')      # 2 GB
def one(fn, cold):
    torch.cuda.synchronize()
    # the GPU is busy with the cache-trashing kernel (~0.6 ms) while the host enqueues the events and the launch behind it: no host time
    # inside the timed interval (warm: a 64 MB kernel keeps the queue busy without evicting the 256 MB last-level cache)
    if cold: big.add_(1.0)
    else: big[:16 * 1024 * 1024].add_(1.0); big[:16 * 1024 * 1024].add_(1.0); big[:16 * 1024 * 1024].add_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
def setenv(**kw):
    for kk in ('PGRF_CV_PIPE', 'PGRF_CV_TILE_W', 'PGRF_CV_DCH', 'PGRF_CV_ORDER', 'PGRF_CV_STREAM_STORES'): os.environ.pop(kk, None)
    for kk, v in kw.items(): os.environ[kk] = str(v)
for (h, w, B) in [(128, 256, 1), (128, 256, 2)]:
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(B, 2, h, w, C, generator=g).cuda()
    ref_mu = (1 + 7 * torch.rand(B, 1, h, w, generator=g)).cuda()
    eye = torch.eye(3).view(1, 1, 3, 3).repeat(B, 2, 1, 1).cuda()
    trans = torch.tensor([[[0.5, 0, 0], [-0.5, 0, 0]]]).repeat(B, 1, 1).cuda()
    dv = CV.build_depth_volume(ref_mu, k, 0.5, 0.1, 10, 64)
    out = torch.empty(B, D, h, w, C, device='cuda')
    nbytes = B * D * h * w * (C + 1) * 4 + B * 2 * h * w * C * 4
    variants = {'fill_': (lambda: out.fill_(1.0), {}),
                'copy_ (read 268 MB + write 268 MB per view)': (None, {}),
                'r05 kernel (single stage, strips)': (lambda: CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv), dict(PGRF_CV_PIPE=0, PGRF_CV_TILE_W=64)),
                'single stage, 16-wide patches': (lambda: CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv), dict(PGRF_CV_PIPE=0, PGRF_CV_TILE_W=16)),
                'pipelined (shipped)': (lambda: CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv), {}),
                'pipelined, streamed stores': (lambda: CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv), dict(PGRF_CV_STREAM_STORES=1)),
                'pipelined, plain stores': (lambda: CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv), dict(PGRF_CV_STREAM_STORES=0)),
                'pipelined dch=8': (lambda: CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv), dict(PGRF_CV_DCH=8)),
                'pipelined P=1': (lambda: CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv), dict(PGRF_CV_PIPE=1)),
                'pipelined order=1': (lambda: CV.calculate_cost_volume_erp(args, feats, None, trans, eye, depth_volume=dv), dict(PGRF_CV_ORDER=1)),
                }
    src = torch.empty_like(out)
    variants['copy_ (read 268 MB + write 268 MB per view)'] = (lambda: out.copy_(src), {})
    for cold in (True, False):
        res = {n: [] for n in variants}
        for rep in range(8):
            for n, (fn, env) in variants.items():
                setenv(**env)
                res[n].append(one(fn, cold))
        for n, v in res.items():
            v = sorted(v[1:]); med = v[len(v) // 2]
            print(f'{h}x{w} B={B} {"cold" if cold else "warm"}  {n:48s}: median {med / B:6.1f} us per view (min {v[0] / B:6.1f})  -> {nbytes / med / 1e6:6.2f} TB/s algorithmic', flush=True)
