"""Time the per-frame 2-D encoders (torch ops on ROCm) next to the HIP render pass: the Amdahl remainder of render()."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panogrf_amd import encoders as EN

dev = torch.device('cuda', 0)
torch.manual_seed(0)
H, W, V = 512, 1024, 2
enc = EN.ResUNetLight().to(dev).eval()
vis = EN.DefaultVisEncoder().to(dev).eval()
imgs = torch.rand(V, 3, H, W, device=dev)
ray_feats = torch.randn(V, 32, H // 8, W // 8, device=dev)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

with torch.no_grad():
    feats = enc(imgs)
    print('ResUNetLight out', tuple(feats.shape))
    print(f'ResUNetLight {V}x{H}x{W}: {timeit(lambda: enc(imgs)):.2f} ms')
    out = vis(ray_feats, feats)
    print('DefaultVisEncoder out', tuple(out.shape))
    print(f'DefaultVisEncoder: {timeit(lambda: vis(ray_feats, feats)):.2f} ms')

# hipGraph replay of the same launch sequence (are the encoders launch-bound?)
with torch.no_grad():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            feats = enc(imgs); out = vis(ray_feats, feats)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        feats_g = enc(imgs)
        out_g = vis(ray_feats, feats_g)
    print(f'encoders, hipGraph replay: {timeit(g.replay):.2f} ms (eager {timeit(lambda: vis(ray_feats, enc(imgs))):.2f} ms)')

# the same layers on the HIP library (panogrf_amd/encoders_hip.py)
from panogrf_amd.encoders_hip import HipEncoders
with torch.no_grad():
    he = HipEncoders(enc, vis, dev)
    f_h, r_h = he(imgs, ray_feats)
    f_t = enc(imgs); r_t = vis(ray_feats, f_t)
    print('HIP vs torch: max rel err feats %.2e  ray_feats %.2e' % (float((f_h - f_t).abs().max() / f_t.abs().max()),
                                                                  float((r_h - r_t).abs().max() / r_t.abs().max())))
    print(f'encoders on HIP, eager launches: {timeit(lambda: he(imgs, ray_feats)):.2f} ms')
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            he(imgs, ray_feats)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        o = he(imgs, ray_feats)
    t_hip_graph = timeit(g2.replay)
    print(f'encoders on HIP, hipGraph replay: {t_hip_graph:.2f} ms')

# roofline of the stage (round 4): algorithmic FLOPs of both encoders from torch's FlopCounterMode on the same modules (CPU), and the
# compulsory bytes (input images + every layer's output written once + read once), against the time of the HIP graph replay
try:
    from torch.utils.flop_counter import FlopCounterMode
    enc_c, vis_c = EN.ResUNetLight().eval(), EN.DefaultVisEncoder().eval()
    enc_c.load_state_dict(enc.state_dict()); vis_c.load_state_dict(vis.state_dict())
    sizes = []
    hooks = [m.register_forward_hook(lambda mod, i, o: sizes.append(o.numel() * 4) if torch.is_tensor(o) else None)
             for m in list(enc_c.modules()) + list(vis_c.modules()) if isinstance(m, (torch.nn.Conv2d, torch.nn.InstanceNorm2d))]
    with torch.no_grad(), FlopCounterMode(display=False) as fc:
        f_c = enc_c(imgs.cpu())
        vis_c(ray_feats.cpu(), f_c)
    flops = fc.get_total_flops()
    for h in hooks:
        h.remove()
    bytes_ = imgs.numel() * 4 + 2 * sum(sizes)
    t_ms = t_hip_graph          # (measured above, with the GPU warm: after the CPU flop count the clocks have dropped)
    print(f'encoders roofline: {flops / 1e9:.1f} GFLOP, ~{bytes_ / 1e6:.0f} MB of layer outputs written + read once; HIP graph {t_ms:.2f} ms = '
          f'{flops / t_ms / 1e9:.1f} TFLOP/s = {flops / t_ms / 1e9 / 157.3 * 100:.1f} % of the f32 MFMA peak, {bytes_ / t_ms / 1e9 * 1e3 / 1e3:.2f} TB/s = '
          f'{bytes_ / t_ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s: bound by neither -- ~110 dependent launches of 5-30 us (ms per launch {t_ms / 111:.4f})')
except Exception as ex:       # the roofline line must never take the timing down
    print('encoders roofline: n/a', repr(ex)[:200])
