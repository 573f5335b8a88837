"""Is the Python-orchestrated regulariser launch-bound?  Eager vs hipGraph replay of the same launch sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panogrf_amd import cost_reg as R

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for h, w in ((64, 128), (128, 256)):
    vol = torch.randn(1, 64, h, w, 32, device='cuda')
    for name, net in (('CostRegNet', R.CostRegNet(dict(use_wrap_padding=True), 32)), ('UNet2', R.UNet2Reg())):
        net = net.cuda().eval()
        with torch.no_grad():
            t_eager = timeit(lambda: net.forward_ndhwc(vol))
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                net.forward_ndhwc(vol)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = net.forward_ndhwc(vol)
            t_graph = timeit(g.replay)
        print(f'{name} {h}x{w}: eager {t_eager:.3f} ms, graph replay {t_graph:.3f} ms')
