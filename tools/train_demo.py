"""A few optimisation steps through the drop-in module on the HIP forward / backward kernels: synthetic
source panoramas, a fixed target image, Adam on ALL parameters (per-ray networks on the HIP backward, 2-D encoders on torch
autograd), `train_ray_num` random rays per step as the reference's trainer samples them.  Prints the loss curve and the time
per step; not a reproduction of the reference's training (no dataset here), a demonstration that the gradient path is whole.

  python tools/train_demo.py [steps] [rays]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_case, synthetic_weights
from panogrf_amd.renderer import name2network

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H, W, V, dn = 256, 512, 2, 64
case = synthetic_case(H, W, V, dn, seed=0, fine=True)
cfg = dict(case['cfg'], network='neuray_base', use_wrap_padding=True, render_depth=True, sample_num=dn)
torch.manual_seed(0)
net = name2network['neuray_base'](cfg)
net.load_state_dict(synthetic_weights(0), strict=False)
net = net.cuda().train()
f = case['frame']
ref0 = {'imgs': f['imgs'].cuda(), 'w2c': f['w2c'].cuda(), 'depth_range': f['ref_depth_range'].cuda(), 'ray_feats': f['ray_feats'].cuda()}
target_img = f['imgs'][0].permute(1, 2, 0).reshape(-1, 3).cuda()          # a fixed image to fit
opt = torch.optim.Adam(net.parameters(), lr=2e-4)
g = torch.Generator().manual_seed(5)
t_f = t_b = 0.0
for it in range(steps):
    idx = torch.randperm(H * W, generator=g)[:rays]
    coords = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    que = {'c2w': f['que_c2w'][None].cuda(), 'coords': coords[None].cuda(), 'depth_range': f['que_depth_range'][None].cuda()}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = net({'que_imgs_info': que, 'ref_imgs_info': dict(ref0)})
    tgt = target_img[idx.cuda()]
    loss = ((out['pixel_colors_nr_fine'][0] - tgt) ** 2).mean() + ((out['pixel_colors_nr'][0] - tgt) ** 2).mean()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if it >= 3:
        t_f += t1 - t0; t_b += t2 - t1
    if it % 5 == 0 or it == steps - 1:
        print(f'step {it:3d}  loss {float(loss.detach()):.5f}')
n = max(steps - 3, 1)
print(f'{rays} rays x {dn}+{dn} samples x {V} views per step: forward {t_f / n * 1e3:.1f} ms (encoders on torch + 2 fused passes), '
      f'backward + Adam {t_b / n * 1e3:.1f} ms (2 rematerialising HIP backward passes + torch autograd of the encoders)')
