import os, sys, torch
sys.path.insert(0, '.')
from panogrf_amd import engine as E
dn = 64
for rn in (2048, 16384, 32768, 65536, 131072, 262144):
    g = torch.Generator().manual_seed(1)
    depth = torch.sort(torch.rand(rn, dn, generator=g) * 10 + 0.5, -1)[0].cuda()
    hit = torch.rand(rn, dn, generator=g).cuda()
    res = {}
    for knob in ('wave', 'lane'):
        os.environ['PGRF_FINE_DEPTH'] = knob
        for _ in range(3):
            E.sample_fine_depth(depth, hit, (0.5, 15.0), dn)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            E.sample_fine_depth(depth, hit, (0.5, 15.0), dn)
        e1.record(); torch.cuda.synchronize()
        res[knob] = e0.elapsed_time(e1) / 10 * 1e3
    print(rn, 'wave %.1f us  lane %.1f us' % (res['wave'], res['lane']))
