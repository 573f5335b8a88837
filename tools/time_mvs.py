"""Timing of the MVS stage (init net) on the GPU box: python tools/time_mvs.py [depth_h depth_w]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import mvs_fixtures as MF
from panogrf_amd.mvs import CostVolumeInitNet

dh, dw = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
H, W = 512, 1024
cfg = dict(MF.MVS_ARGS, device='cuda', height=dh, width=dw, mono_height=dh, mono_width=dw, depth_height=dh, depth_width=dw,
           use_wrap_padding=True, seq_len=3)
cfg.update(min_depth=0.5, max_depth=15.0)
torch.manual_seed(0)
net = CostVolumeInitNet(cfg, d_net=MF.StubMono()).cuda().eval()
g = torch.Generator().manual_seed(1)
ref = {'imgs': torch.rand(2, 3, H, W, generator=g).cuda(), 'rots': torch.eye(3).repeat(2, 1, 1).cuda(),
       'trans': torch.tensor([[0.5, 0, 0], [-0.5, 0, 0]]).cuda()}
src = {'imgs': ref['imgs'].flip(0).contiguous(), 'rots': ref['rots'], 'trans': ref['trans'].flip(0).contiguous()}


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    out = net(ref, src, False)
    print('ray_feats', tuple(out['ray_feats'].shape), 'mvs_depth', tuple(out['mvs_depth'].shape), 'finite', bool(torch.isfinite(out['ray_feats']).all()))
    print(f'init net (2 reference views, depth {dh}x{dw}, cost volume {dh//4}x{dw//4}x64): {timeit(lambda: net(ref, src, False)):.2f} ms')
    x0 = torch.rand(2, 3, dh, dw, device='cuda')
    from panogrf_amd.renderer import NeuralRayGenRenderer
    holder = NeuralRayGenRenderer.__new__(NeuralRayGenRenderer)
    torch.nn.Module.__init__(holder)
    holder.cfg = {}; holder.init_net = net
    holder._run_init_net(ref, src, False)
    holder.cfg = {'hip_init_net': False}
    holder._run_init_net(ref, src, False)
    print(f'init net, 2-D nets on MIOpen, replayed hipGraph: {timeit(lambda: holder._run_init_net(ref, src, False)):.2f} ms')
    holder.cfg = {}
    o2 = holder._run_init_net(ref, src, False)
    print('HIP vs torch init net: ray_feats max rel err %.2e' % float((o2['ray_feats'] - out['ray_feats']).abs().max() / out['ray_feats'].abs().max()))
    print(f'init net, 2-D nets on HIP, replayed hipGraph: {timeit(lambda: holder._run_init_net(ref, src, False)):.2f} ms')
    from panogrf_amd.mvs_hip import HipInitNet
    hip = HipInitNet(net, 'cuda:0')
    print(f'  HIP init net, eager launches: {timeit(lambda: hip(ref, src, False)):.2f} ms')
    pair = torch.rand(4, 3, dh, dw, device='cuda')
    print(f'  HIP Equi feature net, 4 images: {timeit(lambda: hip.equi(pair)):.2f} ms')
    print(f'  HIP res_net, 2 images: {timeit(lambda: hip.T.image_encoder(x0)):.2f} ms')
    x0 = torch.rand(2, 3, dh, dw, device='cuda')
    dn = net.depth_net
    panos = torch.rand(2, 2, dh, dw, 3, device='cuda')
    pcf = panos.permute(0, 1, 4, 2, 3)
    print(f'  Equi feature net, 4 images {dh}x{dw}: {timeit(lambda: (dn.unet(pcf[:, 0]), dn.unet(pcf[:, 1]))):.2f} ms')
    x = torch.rand(2, 3, dh, dw, device='cuda')
    print(f'  res_net (ResUNetLight 2,3,6,3) on 2 images: {timeit(lambda: net.res_net(x)):.2f} ms')
    vol = torch.rand(2, 64, dh // 4, dw // 4, 32, device='cuda')
    print(f'  3-D regulariser, 2 volumes: {timeit(lambda: dn.unet3d.forward_ndhwc(vol)):.2f} ms')
    cr = torch.rand(2, 64, dh // 4, dw // 4, device='cuda'); xd = torch.rand(2, 32, dh // 4, dw // 4, device='cuda')
    def dec():
        t = torch.cat([cr, xd], 1)
        for b in dn.decoders2: t = b(t)
        return t
    print(f'  decoders2: {timeit(dec):.2f} ms')
    # roofline of the stage (round 4): the 3-D regulariser is its only MFMA-bound part -- UNet2: 438.5 GFLOP per 64x128x64 volume (direct
    # convolution count, scales with the volume), executed as Winograd products (/ 3.375) -- the 2-D nets around it are launch / latency bound
    t_reg = timeit(lambda: dn.unet3d.forward_ndhwc(vol))
    t_all = timeit(lambda: holder._run_init_net(ref, src, False))
    gf = 438.5 * (dh // 4) * (dw // 4) / (64 * 128) * 2
    print(f'MVS stage roofline: regulariser {t_reg:.2f} of {t_all:.2f} ms = {gf / 3.375 / t_reg / 157.3 * 100:.1f} % of the f32 MFMA roof of its executed '
          f'products ({gf:.0f} GFLOP direct / 3.375); the remaining {t_all - t_reg:.2f} ms are ~150 launches of 2-D layers (feature net, decoders, heads) at '
          f'{(t_all - t_reg) / 150 * 1e3:.0f} us per launch on average: bound by launch latency and dependent small kernels, not by HBM or MFMA')
