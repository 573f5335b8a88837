import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import render_oracle as O
from bench import synthetic_case
from panogrf_amd import engine as E
case = synthetic_case(32, 64, 2, 64, seed=6, fine=True)
case['cfg']['fine_depth_use_all'] = True
f = case['frame']
fr = E.Frame(f['imgs'], f['img_feats'], f['ray_feats'], f['w2c'], f['ref_depth_range'], f['que_c2w'], f['que_depth_range'])
rr = E.RayRenderer(case['cfg'], case['weights'])
idx = torch.arange(0, 32 * 64, 11)
coords = torch.stack([(idx % 64).float(), (idx // 64).float()], -1)
out = rr.render_rays(fr, coords)
ref = O.render_rays(case['weights'], case['frame'], coords, case['cfg'])
d = (out['depth_fine'].cpu() - ref['depth_fine']).abs()
print('max', float(d.max()), 'n>2e-3', int((d > 2e-3).sum()), 'n>1e-4', int((d > 1e-4).sum()), 'of', d.numel())
i = int(d.argmax()); r, c = i // d.shape[1], i % d.shape[1]
print('ray', r, 'col', c, out['depth_fine'][r, max(c-2,0):c+3].cpu(), ref['depth_fine'][r, max(c-2,0):c+3])
hp = (out['hit_prob_nr'].cpu() - ref['hit_prob_nr']).abs()
print('coarse hit_prob max err', float(hp.max()), 'density err', float((out['density_nr'].cpu() - ref['density_nr']).abs().max()))
print('coarse hit_prob row', ref['hit_prob_nr'][r].sum(), ref['hit_prob_nr'][r].max())
