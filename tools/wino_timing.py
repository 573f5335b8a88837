"""Per-phase cycle breakdown of the Winograd kernel's MFMA wave 0 (needs a -DPGRF_WINO_TIMING build of conv3d_wino.hip,
selected with PANOGRF_HIP_LIB, PGRF_ALLOW_ABLATION_LIB=1): python tools/wino_timing.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panogrf_amd import cost_reg as R
from panogrf_amd._lib import lib
from panogrf_amd.engine import _ptr, _stream
g = torch.Generator().manual_seed(0)
GEN = int(sys.argv[1]) if len(sys.argv) > 1 else 16      # kernel generation: 16 (two workgroups per CU) or 32
names = ['row setup/R init', 'wait B1 (loaders: T free)', 'wait B2 (T ready)', 'MFMA phase', 'wait E1', 'partials -> LDS', 'wait E2', 'epilogue']
for (cin, cout, D, H, W) in [(32, 64, 64, 128, 256), (64, 64, 64, 128, 256), (128, 128, 32, 64, 128), (256, 256, 16, 32, 64)]:
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    pc = R._PackedConv(w, torch.zeros(cout), 'cuda')
    x = torch.randn(1, D, H, W, cin, generator=g).cuda()
    out = torch.empty(1, D, H, W, cout, device='cuda')
    dbg = torch.zeros(8, dtype=torch.int64, device='cuda')
    for rep in range(2):
        dbg.zero_()
        if GEN == 16:
            lib.check(lib.pgrf_conv3d_wino16(_ptr(x), cin, 0, cin, _ptr(out), cout, 0, cout, _ptr(pc.wino16()), 1, 1, D, H, W, 0.01, _ptr(dbg), 0, 0, _stream()))
        else:
            lib.check(lib.pgrf_conv3d_wino(_ptr(x), cin, 0, cin, _ptr(out), cout, 0, cout, _ptr(pc.wino()), 1, 1, D, H, W, 0.01, _ptr(dbg), 0, 0, _stream()))
        torch.cuda.synchronize()
    t = dbg.cpu().double()
    nwg = min(512, D // 2 * (H // 2) * (W // 32) * (cout // 64)) if GEN == 16 else min(256, D // 2 * (H // 2) * (W // 64) * (cout // 64))
    print(f'cin={cin} cout={cout} {D}x{H}x{W}: total {t.sum()/nwg/100e6*1e3:.0f} us at 100 MHz counter per WG; shares: ' +
          ', '.join(f'{n} {100*v/t.sum():.1f}%' for n, v in zip(names, t.tolist())))
