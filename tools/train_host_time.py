"""Host (CPU enqueue) time of the training step of bench.py --train, by segment: how far the host runs ahead of the GPU.
   python tools/train_host_time.py   (on a GPU box)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from panogrf_amd import engine as E
dev = torch.device('cuda:0')
H, W, V, dn, rays = 512, 1024, 2, 64, 2048
case = bench.synthetic_case(H, W, V, dn, seed=0, fine=True)
f = {k: v.to(dev) for k, v in case['frame'].items()}
params = {k: torch.nn.Parameter(v.to(dev).clone()) for k, v in case['weights'].items()}
img_feats = f['img_feats'].clone().requires_grad_(True); ray_feats = f['ray_feats'].clone().requires_grad_(True)
opt = torch.optim.Adam(list(params.values()), lr=1e-4, fused=True)
target_img = f['imgs'][0].permute(1, 2, 0).reshape(-1, 3)
g = torch.Generator().manual_seed(100)
N = 45
idxs = [torch.randperm(H * W, generator=g)[:rays].to(dev) for _ in range(N)]
us = [torch.rand(rays, dn, generator=g).to(dev) for _ in range(N)]
common = (f['imgs'], img_feats, ray_feats, f['w2c'], f['ref_depth_range'], case['frame']['que_c2w'], case['frame']['que_depth_range'], params)
acc = {}
def seg(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
for i in range(N):
    if i == 5: acc.clear()
    t = time.perf_counter()
    idx = idxs[i]; coords = torch.stack([(idx % W).float(), (idx // W).float()], -1); tgt = target_img[idx]
    depth = E.sample_coarse_depth(0.5, 15.0, rays, dn, device=dev); t = seg('prep', t)
    oc = E.render_pass_autograd(*common, 'dist_decoder', 'agg_net', False, coords, depth); t = seg('fwd coarse', t)
    fd = E.sample_fine_depth(depth, oc['hit_prob_nr'].detach(), [0.5, 15.0], dn, u=us[i]); t = seg('sampler', t)
    of = E.render_pass_autograd(*common, 'fine_dist_decoder', 'fine_agg_net', False, coords, fd); t = seg('fwd fine', t)
    loss = ((of['pixel_colors_nr'] - tgt) ** 2).mean() + ((oc['pixel_colors_nr'] - tgt) ** 2).mean(); t = seg('loss', t)
    opt.zero_grad(set_to_none=True); img_feats.grad = ray_feats.grad = None; t = seg('zero_grad', t)
    loss.backward(); t = seg('backward', t)
    opt.step(); t = seg('adam', t)
torch.cuda.synchronize()
for k, v in acc.items(): print('%-12s %.3f ms / step (CPU enqueue)' % (k, v * 1e3 / (N - 5)))
print('total %.3f' % (sum(acc.values()) * 1e3 / (N - 5)))
