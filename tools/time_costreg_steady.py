"""Is a CostRegNet forward host-bound?  wall time of enqueueing n forwards vs GPU time of the same n."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panogrf_amd import cost_reg as R
h, w = 128, 256
cv = torch.randn(1, 64, h, w, 32, device='cuda')
net = R.CostRegNet(dict(use_wrap_padding=True), 32).cuda().eval()
with torch.no_grad():
    for p in net.parameters():
        p.mul_(0.5)
    net.forward_ndhwc(cv); torch.cuda.synchronize()
    for n in (3, 20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); e0.record()
        for _ in range(n):
            net.forward_ndhwc(cv)
        e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f'n={n}: enqueue {1e3 * (t1 - t0) / n:.3f} ms/forward, gpu events {e0.elapsed_time(e1) / n:.3f} ms/forward, wall to sync {1e3 * (t2 - t0) / n:.3f}')
