"""CostRegNet forward repeated: every output must equal the first bit for bit (no races in the LDS-brick kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panogrf_amd import cost_reg as R
for (h, w) in ((128, 256), (40, 72)):
    cv = torch.randn(1, 64 if h == 128 else 24, h, w, 32, device='cuda')
    net = R.CostRegNet(dict(use_wrap_padding=True), 32).cuda().eval()
    with torch.no_grad():
        first = net.forward_ndhwc(cv).clone()
        bad = 0
        for i in range(30):
            if not torch.equal(net.forward_ndhwc(cv), first):
                bad += 1
    print(f'{h}x{w}: {bad} of 30 forwards differ from the first; finite={bool(torch.isfinite(first).all())}')
