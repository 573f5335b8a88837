"""Winograd vs direct 3-D convolution on the GPU: max error on small random cases, then timing of the UNet2 layer shapes.
python tools/check_wino.py [time]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panogrf_amd import cost_reg as R

dev = 'cuda'
g = torch.Generator().manual_seed(0)


def run(cin, cout, B, D, H, W, wino, bias=True, reps=1):
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1 if bias else None
    pc = R._PackedConv(w, b, dev)
    x = R._Vol(torch.randn(B, D, H, W, cin, generator=g).to(dev), 0, cin)
    out = R._Vol.empty(B, D, H, W, cout, dev)
    os.environ['PANOGRF_CONV_WINOGRAD'] = '1' if wino else '0'
    R._conv(pc, x, out, 1, 0.01)
    torch.cuda.synchronize()
    if reps > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            R._conv(pc, x, out, 1, 0.01)
        e1.record(); torch.cuda.synchronize()
        return out.buf, e0.elapsed_time(e1) / reps
    return out.buf, 0.0


if len(sys.argv) < 2:
    for (cin, cout, B, D, H, W) in [(32, 32, 1, 2, 2, 64), (32, 64, 1, 4, 4, 64), (64, 64, 2, 4, 6, 128), (128, 64, 1, 2, 4, 64),
                                    (64, 128, 1, 6, 2, 64), (32, 96, 1, 2, 2, 64)]:
        st = g.get_state()
        a, _ = run(cin, cout, B, D, H, W, True)
        g.set_state(st)
        b, _ = run(cin, cout, B, D, H, W, False)
        err = float((a - b).abs().max()); ref = float(b.abs().max())
        print(f'cin={cin:3d} cout={cout:3d} {B}x{D}x{H}x{W}: max|wino-direct| = {err:.3e}  (max |out| {ref:.3f})  finite={bool(torch.isfinite(a).all())}')
else:
    for (cin, cout, D, H, W) in [(32, 64, 64, 128, 256), (64, 64, 64, 128, 256), (64, 128, 32, 64, 128), (128, 128, 32, 64, 128),
                                 (256, 64, 32, 64, 128), (64, 64, 32, 64, 128), (128, 256, 16, 32, 64), (256, 256, 16, 32, 64)]:
        gf = 27 * cin * cout * 2 * D * H * W / 1e9
        os.environ['PANOGRF_WINO'] = '16'
        _, tw = run(cin, cout, 1, D, H, W, True, reps=5)
        os.environ['PANOGRF_WINO'] = '32'
        _, t32 = run(cin, cout, 1, D, H, W, True, reps=5)
        os.environ['PANOGRF_WINO'] = '16'
        _, td = run(cin, cout, 1, D, H, W, False, reps=5)
        # roof of the Winograd-domain products: direct FLOPs / 3.375 at the f32 MFMA peak (157.3 TFLOP/s)
        print(f'cin={cin:3d} cout={cout:3d} {D}x{H}x{W}: winograd {tw*1e3:8.1f} us ({gf/tw:6.1f} TF/s alg, {gf / 3.375 / tw / 157.3 * 100:4.1f} % of the Winograd-MFMA roof)   '
              f'first generation (32-tile rows) {t32*1e3:8.1f} us   direct {td*1e3:8.1f} us ({gf/td:6.1f} TF/s)   x{td/tw:.2f}')
