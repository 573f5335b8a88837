#!/usr/bin/env python3
"""Throughput of the HIP per-ray renderer on synthetic panoramas (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N=1: plain python; N>1: torchrun, one rank per GPU)

A step = one whole query panorama (H*W rays) rendered against `views` source panoramas already
resident in HBM: NCHW->NHWC re-layout of the source maps, depth sampling, the fused render pass
(and the importance sampler + fine pass with --fine).  With N>1 the panorama is sharded by row tile
across ranks and reassembled with one all-gather (RCCL), i.e. strong scaling of one frame.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import subprocess
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


# ------------------------------------------------------------------------------------------------
# synthetic workload: data only, no reference code involved
def _smooth_pano(g, C, H, W):
    th = torch.linspace(0, 2 * math.pi, W).view(1, 1, W)
    ph = torch.linspace(0, math.pi, H).view(1, H, 1)
    out = torch.zeros(C, H, W)
    for _ in range(8):
        a = torch.randint(1, 5, (C, 1, 1), generator=g).float()
        b = torch.randint(1, 4, (C, 1, 1), generator=g).float()
        p = torch.rand(C, 1, 1, generator=g) * 2 * math.pi
        q = torch.rand(C, 1, 1, generator=g) * 2 * math.pi
        out = out + torch.sin(a * th + p) * torch.cos(b * ph + q)
    return (out - out.amin()) / (out.amax() - out.amin())


_LINEARS = {   # name -> (out, in, kaiming_normal?)   shapes: SURVEY.md Appendix A
    'prob_embed.0': (32, 34, False), 'prob_embed.2': (32, 32, False),
    'agg_impl.ray_dir_fc.0': (16, 4, False), 'agg_impl.ray_dir_fc.2': (35, 16, False),
    'agg_impl.base_fc.0': (64, 207, True), 'agg_impl.base_fc.2': (32, 64, True),
    'agg_impl.vis_fc.0': (32, 32, True), 'agg_impl.vis_fc.2': (33, 32, True),
    'agg_impl.vis_fc2.0': (32, 32, True), 'agg_impl.vis_fc2.2': (1, 32, True),
    'agg_impl.geometry_fc.0': (64, 65, True), 'agg_impl.geometry_fc.2': (16, 64, True),
    'agg_impl.out_geometry_fc.0': (16, 16, False), 'agg_impl.out_geometry_fc.2': (1, 16, False),
    'agg_impl.rgb_fc.0': (16, 37, True), 'agg_impl.rgb_fc.2': (8, 16, True), 'agg_impl.rgb_fc.4': (1, 8, True),
    'agg_impl.neuray_fc.0': (8, 32, True), 'agg_impl.neuray_fc.2': (1, 8, True),
}


def synthetic_weights(seed, use_vis=False):
    """Random-init per-ray networks with the reference's key names/shapes and PyTorch's default init
    statistics, plus a seeded density boost so the rendered image is not black."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def lin(key, o, i, kaiming):
        if kaiming:
            sd[key + '.weight'] = torch.randn(o, i, generator=g) * math.sqrt(2.0 / i)
            sd[key + '.bias'] = torch.zeros(o)
        else:
            b = 1.0 / math.sqrt(i)
            sd[key + '.weight'] = (torch.rand(o, i, generator=g) * 2 - 1) * b
            sd[key + '.bias'] = (torch.rand(o, generator=g) * 2 - 1) * b

    for agg, dd in [('agg_net', 'dist_decoder'), ('fine_agg_net', 'fine_dist_decoder')]:
        for name, (o, i, k) in _LINEARS.items():
            lin(f'{agg}.{name}', o, i, k)
        for nm in ['w_qs', 'w_ks', 'w_vs', 'fc']:
            sd[f'{agg}.agg_impl.ray_attention.{nm}.weight'] = (torch.rand(16, 16, generator=g) * 2 - 1) * 0.25
        sd[f'{agg}.agg_impl.ray_attention.layer_norm.weight'] = torch.ones(16)
        sd[f'{agg}.agg_impl.ray_attention.layer_norm.bias'] = torch.zeros(16)
        for dec, outs in [('mean_decoder', 2), ('var_decoder', 2), ('aw_decoder', 1)] + ([('vis_decoder', 1)] if use_vis else []):
            lin(f'{dd}.{dec}.0', 32, 32, False)
            lin(f'{dd}.{dec}.2', 32, 32, False)
            lin(f'{dd}.{dec}.4', outs, 32, False)
        # density boost
        sd[f'{agg}.agg_impl.out_geometry_fc.2.bias'] += 0.6
        sd[f'{agg}.agg_impl.out_geometry_fc.2.weight'] *= 3.0
        sd[f'{agg}.agg_impl.out_geometry_fc.0.weight'] *= 2.0
        sd[f'{dd}.var_decoder.4.bias'] += 2.0
    return sd


def synthetic_case(H, W, rfn, dn, seed=0, fine=False, use_vis=False):
    g = torch.Generator().manual_seed(1234 + seed)
    imgs = torch.stack([_smooth_pano(g, 3, H, W) for _ in range(rfn)])
    img_feats = torch.randn(rfn, 32, H // 4, W // 4, generator=g)
    ray_feats = torch.randn(rfn, 32, H // 8, W // 8, generator=g)
    w2c = torch.zeros(rfn, 3, 4)
    for v in range(rfn):
        a = 2 * math.pi * v / rfn
        yaw = math.radians(5.0) * (1 if v % 2 else -1)
        c, s = math.cos(yaw), math.sin(yaw)
        w2c[v, :, :3] = torch.tensor([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        w2c[v, :, 3] = torch.tensor([0.5 * math.cos(a), 0.0, 0.5 * math.sin(a)])   # 1.0 m baseline (cfg gen_hr_1.0)
    frame = dict(imgs=imgs, img_feats=img_feats, ray_feats=ray_feats, w2c=w2c,
                 ref_depth_range=torch.tensor([[0.5, 15.0]] * rfn),
                 que_c2w=torch.tensor([[1.0, 0, 0, 0.02], [0, 1.0, 0, 0.01], [0, 0, 1.0, -0.03]]),
                 que_depth_range=torch.tensor([0.5, 15.0]))
    cfg = dict(min_depth=0.5, max_depth=15.0, use_disp=True, depth_sample_num=dn, fine_depth_sample_num=dn,
               use_hierarchical_sampling=bool(fine), dist_decoder_cfg={'use_vis': use_vis},
               fine_dist_decoder_cfg={'use_vis': use_vis}, height=H, width=W)
    return dict(frame=frame, cfg=cfg, weights=synthetic_weights(seed, use_vis))


def flop_per_ray(rfn, dn, passes):
    """Algorithmic matmul FLOP per ray (equals FlopCounterMode on the reference)."""
    return passes * dn * 2 * (rfn * 28464 + 6480 + 32 * dn)


# ------------------------------------------------------------------------------------------------
def end_to_end_frames(H, W, V, dn, fine, dev, frames=5):
    """SURVEY.md 8d asks for the rate 'with / without per-frame encoders and cost volume': the drop-in module's whole
    `renderer(data)` call (MVS stage at depth H/2 x W/2 as shipped -> 2-D encoders -> render passes; random-init networks, a
    constant mono-depth prior standing in for UniFuse) timed next to the headline.  Reported beside `value`, never as it."""
    import numpy as np
    from render import DEFAULT_CFG, MVS_CFG, ConstantMono
    from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, imgs_info_to_torch
    from panogrf_amd.renderer import name2network
    cfg = dict(DEFAULT_CFG, height=H, width=W, depth_sample_num=dn, fine_depth_sample_num=dn, sample_num=dn,
               use_hierarchical_sampling=bool(fine))
    cfg.update(MVS_CFG, depth_height=H // 2, depth_width=W // 2)
    torch.manual_seed(0)
    net = name2network['neuray_gen'](cfg, d_net=ConstantMono(3.0)).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(V, 3, H, W, generator=g).numpy()
    pose = lambda tx: np.concatenate([np.eye(3), np.array([[tx], [0.0], [0.0]])], 1)
    w2c = np.stack([pose(0.5 - v / max(V - 1, 1)) for v in range(V)])
    rng = [[0.5, 15.0]] * V
    ref = imgs_info_to_torch(build_imgs_info(imgs, w2c, rng), dev)
    src = imgs_info_to_torch(build_imgs_info(imgs[::-1].copy(), w2c[::-1].copy(), rng), dev)
    que = imgs_info_to_torch(build_render_imgs_info(pose(0.0), (H, W), [0.5, 15.0]), dev)

    def run(reuse):
        net.cfg['hip_reuse_source_maps'] = reuse
        with torch.no_grad():
            for i in range(2 + frames):
                if i == 2:
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                net({'que_imgs_info': dict(que), 'ref_imgs_info': dict(ref), 'src_imgs_info': dict(src), 'eval': True})
            torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / frames * 1e3
    every, kept = run(False), run(True)
    return {'what': f'renderer(data) of the drop-in module: MVS stage (depth {H // 2}x{W // 2}, UNet2) + 2-D encoders + '
                    f'{"coarse and fine passes" if fine else "one pass"}; constant mono prior',
            'ms_per_frame_everything_per_frame': every, 'rays_per_s_everything_per_frame': H * W / every * 1e3,
            'ms_per_frame_source_maps_reused': kept, 'rays_per_s_source_maps_reused': H * W / kept * 1e3}


def cpu_baseline_config0(budget_s=6.0):
    """BASELINE configs[0] -- the reference's own CPU-runnable case: 256x512 ERP, 2 source views, 32 coarse + 32 fine samples --
    through the CPU oracle on a bounded random sample of the frame's rays."""
    case0 = synthetic_case(256, 512, 2, 32, seed=0, fine=True)
    base, _, _ = cpu_baseline(case0, budget_s=budget_s)
    base['workload'] = '256x512 ERP, 2 source views, 32 + 32 samples (BASELINE configs[0])'
    return base


def cpu_baseline(case, budget_s=15.0):
    """The CPU oracle (a port of the reference algorithm) timed on a bounded sample of the workload."""
    from oracle import render_oracle as O
    H, W = case['frame']['imgs'].shape[-2:]
    # host cores actually usable: the scheduler affinity mask, capped — GPU boxes often expose 256 logical
    # CPUs behind a much smaller cgroup quota, where 256 OpenMP threads are far slower than 16
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    avail = cores
    cores = max(1, min(cores, int(os.environ.get('PGRF_CPU_BASELINE_THREADS', '16'))))
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(7)

    def run(n):
        idx = torch.randperm(H * W, generator=g)[:n]
        coords = torch.stack([(idx % W).float(), (idx // W).float()], -1)
        t0 = time.perf_counter()
        with torch.no_grad():
            out = O.render_rays(case['weights'], case['frame'], coords, case['cfg'])
        return time.perf_counter() - t0, coords, out

    t_probe, _, _ = run(64)
    n = int(max(64, min(65536, 64 * budget_s / max(t_probe, 1e-3))))
    n = n // 64 * 64
    dt, coords, out = run(n)
    torch.set_num_threads(prev_threads)
    return dict(value=n / dt, unit='rays/s', cores=cores, kind='port',
                sample=f'{n} random rays of the same frame through oracle.render_oracle.render_rays '
                       f'({dt:.1f} s, torch CPU fp32; {cores} of the {avail} CPUs the affinity mask shows: GPU boxes expose far more '
                       f'logical CPUs than their cgroup quota runs, and torch with every one of them as a thread is slower than with 16 '
                       f'-- PGRF_CPU_BASELINE_THREADS overrides)'), coords, out


def eager_gpu_baseline(case, dev, ray_batch_num=2048, frames=1):
    """'Reference single-GPU PyTorch rays/s': the oracle (an op-for-op port of the reference's eager PyTorch
    path) moved to the GPU and run the way the reference runs it — the ray_batch_num loop of
    network/renderer.py:656-683.  Reported next to the CPU baseline; never the product path."""
    from oracle import render_oracle as O
    fr = {k: v.to(dev) for k, v in case['frame'].items()}
    w = {k: v.to(dev) for k, v in case['weights'].items()}
    H, W = fr['imgs'].shape[-2:]
    with torch.no_grad():
        O.render_frame(w, {**fr, 'imgs': fr['imgs'][:, :, :32], 'img_feats': fr['img_feats'][:, :, :8], 'ray_feats': fr['ray_feats'][:, :, :4]},
                       case['cfg'], ray_batch_num)                                  # warm-up on a 32-row strip
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            out = O.render_frame(w, fr, case['cfg'], ray_batch_num)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dict(value=H * W * frames / dt, unit='rays/s', ray_batch_num=ray_batch_num,
                kind='port of the reference eager path on torch-ROCm, same GPU', ms_per_frame=dt / frames * 1e3), out


def eager_strip_baseline(case, dev, rows=64, budget_s=20.0):
    """Bounded, DEFAULT leg of the same denominator (the '>= 10x the reference single-GPU PyTorch rays/s' of the north star):
    the eager port timed on a fixed strip of `rows` query rows around the equator of the same frame, through the
    reference's ray_batch_num loop (network/renderer.py:656-683) at 2048 (its default) and 4096 (render.py) rays per
    batch.  Per-batch cost does not depend on which rays a batch holds, so the strip rate is the frame rate."""
    from oracle import render_oracle as O
    fr = {k: v.to(dev) for k, v in case['frame'].items()}
    w = {k: v.to(dev) for k, v in case['weights'].items()}
    H, W = fr['imgs'].shape[-2:]
    rows = min(rows, H)
    r0 = (H - rows) // 2
    ys, xs = torch.meshgrid(torch.arange(r0, r0 + rows, dtype=torch.float32, device=dev),
                            torch.arange(W, dtype=torch.float32, device=dev), indexing='ij')
    coords = torch.stack([xs, ys], -1).reshape(-1, 2)
    res = {'unit': 'rays/s', 'kind': 'port, strip-extrapolated',
           'sample': f'{rows} rows x {W} px around the equator of the same frame ({coords.shape[0]} rays), eager torch-ROCm port of '
                     f'the reference path on this GPU, ray_batch_num 2048 and 4096'}
    t_start = time.perf_counter()
    outs = None
    with torch.no_grad():
        O.render_rays(w, fr, coords[:2048], case['cfg'])            # warm-up (allocator, kernel selection)
        torch.cuda.synchronize()
        for rb in (2048, 4096):
            n_done, parts = 0, []
            t0 = time.perf_counter()
            for s_ in range(0, coords.shape[0], rb):
                parts.append(O.render_rays(w, fr, coords[s_:s_ + rb], case['cfg']))
                n_done += min(rb, coords.shape[0] - s_)
                if time.perf_counter() - t_start > budget_s:         # hard bound on slow hosts
                    break
            torch.cuda.synchronize()
            res[f'value_ray_batch_{rb}'] = n_done / (time.perf_counter() - t0)
            if rb == 2048:
                outs = (coords[:n_done], parts)
    res['value'] = max(res['value_ray_batch_2048'], res['value_ray_batch_4096'])
    return res, outs


def train_bench(args, dev, rank, world, distributed):
    """one TRAINING step of the per-ray networks as the reference's trainer runs it through
    network/renderer.py:567-633 -- coarse pass, importance sampling from the detached weights, fine pass, a photometric loss on
    both images, backward to every per-ray weight and to the two feature maps, Adam.  The 2-D encoders that produce the maps are
    torch modules outside this path (their outputs are leaf tensors here).  Forward = the staged HIP forward with its activations kept, backward = the
    reverse sweep over them (pgrf_render_pass_backward_phase 1 / 2).  Ranks are data-parallel: each its own rays, gradients all-reduced (RCCL)."""
    from panogrf_amd import engine as E
    H, W, V, dn, rays = args.height, args.width, args.views, args.samples, args.train_rays
    case = synthetic_case(H, W, V, dn, seed=0, fine=True)
    f = {k: v.to(dev) for k, v in case['frame'].items()}
    params = {k: torch.nn.Parameter(v.to(dev).clone()) for k, v in case['weights'].items()}
    img_feats = f['img_feats'].clone().requires_grad_(True)
    ray_feats = f['ray_feats'].clone().requires_grad_(True)
    opt = torch.optim.Adam(list(params.values()), lr=1e-4, fused=True)      # (one fused kernel for the ~120 small tensors; the foreach form is 1.4 ms of host time per step)
    use_vis = case['cfg']['dist_decoder_cfg']['use_vis']
    target_img = f['imgs'][0].permute(1, 2, 0).reshape(-1, 3)
    g = torch.Generator().manual_seed(100 + rank)
    n_batches = args.warmup + args.steps
    idxs = [torch.randperm(H * W, generator=g)[:rays].to(dev) for _ in range(n_batches)]
    us = [torch.rand(rays, dn, generator=g).to(dev) for _ in range(n_batches)]
    # the batches (pixel coordinates and their ground-truth colours) are the data loader's product, prepared like the draws above
    batches = [(torch.stack([(idx % W).float(), (idx // W).float()], -1), target_img[idx]) for idx in idxs]
    que_dr = [0.5, 15.0]
    common = (f['imgs'], img_feats, ray_feats, f['w2c'], f['ref_depth_range'], case['frame']['que_c2w'], case['frame']['que_depth_range'], params)
    flat_grads = None

    def step(i):
        coords, tgt = batches[i]
        depth = E.sample_coarse_depth(case['cfg']['min_depth'], case['cfg']['max_depth'], rays, dn, device=dev)
        oc = E.render_pass_autograd(*common, 'dist_decoder', 'agg_net', use_vis, coords, depth)
        fd = E.sample_fine_depth(depth, oc['hit_prob_nr'].detach(), que_dr, dn, u=us[i])
        of = E.render_pass_autograd(*common, 'fine_dist_decoder', 'fine_agg_net', use_vis, coords, fd)
        loss = torch.nn.functional.mse_loss(of['pixel_colors_nr'], tgt) + torch.nn.functional.mse_loss(oc['pixel_colors_nr'], tgt)
        opt.zero_grad(set_to_none=True)
        img_feats.grad = ray_feats.grad = None
        loss.backward()
        if distributed:
            gl = [p.grad for p in params.values() if p.grad is not None]
            flat = torch.cat([t.reshape(-1) for t in gl])
            torch.distributed.all_reduce(flat)
            flat /= world
            off = 0
            for t in gl:
                t.copy_(flat[off:off + t.numel()].view_as(t)); off += t.numel()
        opt.step()
        return loss

    def sync_all():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    losses = []
    for i in range(args.warmup):
        step(i)
    sync_all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(args.steps):
        losses.append(step(args.warmup + i))
    e1.record()
    sync_all()
    dt = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1) / args.steps
    if distributed:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # algorithmic FLOPs of a training step: forward + 2x for the backward (dX and dW products) = 3x the forward of both passes
    flops_step = 3 * flop_per_ray(V, dn, 2) * rays
    achieved = flops_step / (gpu_ms * 1e-3) / 1e12
    res = {'metric': f'trained rays/sec ({dn} + {dn} samples, {V} views, forward + backward + Adam of the per-ray networks)',
           'value': rays * world * args.steps / dt, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
           'data': 'synthetic',
           'config': {'workload': f'training step: {rays} random rays per rank of a {H}x{W} ERP query, {V} source views, {dn} coarse + {dn} fine '
                                  f'samples; gradients to all per-ray weights and to img_feats / ray_feats; Adam; encoders outside',
                      'rays_per_step_per_rank': rays, 'parallelism': f'data-parallel x{world}, gradient all-reduce',
                      # (what the timed step contains since round 5 -- records of rounds <= 4 under this metric name also prepared the
                      #  batch and ran the foreach Adam and a 15-kernel loss inside the step, so their ms_per_step is not comparable)
                      'batch_prep_in_step': False, 'optimizer': 'adam-fused', 'loss': 'mse_loss (coarse + fine)'},
           'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MFMA_TFLOPS,
                        'traffic': None, 'kernel': 'whole step on the stream (2 staged forward passes with their activations kept, 2 reverse sweeps, sampler, loss, Adam)',
                        'kernel_ms': gpu_ms, 'flop_per_ray': 3 * flop_per_ray(V, dn, 2), 'rays_per_launch': rays,
                        'note': 'algorithmic = 3 x forward FLOPs = what the step executes (PANOGRF_TRAIN_KEEP_ACTS=0: fused forward + rematerialising backward, 4 x)'},
           'loss_first_last': [float(losses[0].detach()), float(losses[-1].detach())]}
    under_profiler = 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
    if rank == 0 and world == 1 and not distributed and not (args.no_live_pmc or args.pmc_child or under_profiler):
        # the step is a chain of ~270 launches over materialised [rows, C] activations: beside the MFMA roof of its FLOPs, the HBM side --
        # bytes per step from two child passes of this command under rocprofv3 --pmc, against the 8 TB/s peak
        tb, tsrc = live_hbm_traffic(['--train', '--height', str(H), '--width', str(W), '--views', str(V), '--samples', str(dn),
                                     '--train-rays', str(rays)], kernel=None, steps_total=3)
        if tb is not None:
            ach = tb / (gpu_ms * 1e-3) / 1e9
            res['roofline_hbm'] = {'bound': 'hbm', 'achieved': ach, 'peak': 8000.0, 'unit': 'GB/s', 'frac': ach / 8000.0, 'traffic': tb,
                                   'traffic_source': tsrc, 'kernel_ms': gpu_ms,
                                   'note': 'bytes per step (all kernels) / step time; FETCH_SIZE doubled as for wide coalesced reads (narrower accesses are uncalibrated on gfx950)'}
        else:
            res['roofline_hbm'] = {'error': tsrc}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the eager-autograd port on the same GPU: the oracle's torch ops (the reference's graph) under torch.autograd, same rays
        from oracle import render_oracle as O
        pw = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        fr = {k: v for k, v in f.items()}
        fr['img_feats'] = f['img_feats'].clone().requires_grad_(True)
        fr['ray_feats'] = f['ray_feats'].clone().requires_grad_(True)
        eopt = torch.optim.Adam(list(pw.values()), lr=1e-4, fused=True)
        qdr = case['frame']['que_depth_range'].to(dev)

        def estep(i):
            coords, tgt = batches[i]
            depth = O.sample_coarse_depth(case['cfg']['min_depth'], case['cfg']['max_depth'], rays, dn).to(dev)
            oc = O.render_pass(pw, 'dist_decoder', 'agg_net', fr, coords, depth, {'use_vis': use_vis})
            fd = O.sample_fine_depth(depth, oc['hit_prob_nr'].detach(), qdr, dn, u=us[i])
            fd = torch.sort(fd, -1)[0]
            of = O.render_pass(pw, 'fine_dist_decoder', 'fine_agg_net', fr, coords, fd, {'use_vis': use_vis})
            loss = torch.nn.functional.mse_loss(of['pixel_colors_nr'], tgt) + torch.nn.functional.mse_loss(oc['pixel_colors_nr'], tgt)
            eopt.zero_grad(set_to_none=True)
            loss.backward()
            eopt.step()
            return loss
        try:
            for i in range(2):
                estep(i)
            torch.cuda.synchronize()
            n_e = max(1, min(args.steps, 5))
            t0 = time.perf_counter()
            for i in range(n_e):
                el = estep(args.warmup + i)
            torch.cuda.synchronize()
            et = (time.perf_counter() - t0) / n_e
            res['torch_eager_gpu_baseline'] = {'value': rays / et, 'unit': 'rays/s', 'ms_per_step': et * 1e3,
                                               'kind': 'port: the oracle (reference graph) under torch.autograd + Adam on this GPU, same rays',
                                               'speedup_hip_over_eager': (rays * world * args.steps / dt) / (rays / et)}
        except Exception as ex:      # the baseline must never take the measurement down
            res['torch_eager_gpu_baseline'] = {'error': repr(ex)[:300]}
    return res


def launch_ranks(n, argv):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as a CHILD torchrun (this process has not
    touched the GPU yet and never execs), relay their output and rank 0's JSON line, exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, env=env).returncode


def live_hbm_traffic(argv_shape, timeout_s=150, kernel='render_pass', steps_total=None):
    """HBM-side bytes per render_pass launch of THIS configuration, measured now: two child runs of this script under
    `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters in their own runs, no
    other trace domain beside them -- MI355X_MICROARCH.md, rocprofv3 section), two steps each.  FETCH_SIZE is doubled (gfx950 tallies
    the 128-byte requests of wide coalesced reads at 64 B); both counters are KiB.  Returns (bytes, description) or (None, why).
    kernel=None, steps_total=n: the counters are summed over EVERY kernel of the child process and divided by its n steps
    (warm-up included) -- bytes per step of a multi-kernel step (bench.py --train)."""
    import csv
    import glob
    import shutil
    import tempfile
    rp = shutil.which('rocprofv3')
    if rp is None:
        return None, 'rocprofv3 not on PATH'
    vals = {}
    env = dict(os.environ)
    env['TMPDIR'] = '/tmp'
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = tempfile.mkdtemp(prefix='pgrf_pmc_', dir='/tmp')
        # the profiled program is python itself (no shell, env or launcher hop between rocprofv3 and the process that opens the GPU)
        cmd = [rp, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.abspath(__file__),
               '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-end-to-end', '--pmc-child'] + argv_shape
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                for row in csv.DictReader(open(f)):
                    if (kernel is None or kernel in row.get('Kernel_Name', '')) and row.get('Counter_Name') == counter:
                        tot += float(row['Counter_Value']); n += 1
            if r.returncode != 0 or n == 0:
                return None, f'rocprofv3 --pmc {counter} pass failed (rc {r.returncode}, {n} kernel rows)'
            vals[counter] = (tot / (n if steps_total is None else steps_total), n)
        except subprocess.TimeoutExpired:
            return None, f'rocprofv3 --pmc {counter} pass exceeded {timeout_s} s'
        except Exception as ex:          # measurement garnish: never takes the bench line down
            return None, f'rocprofv3 --pmc {counter} pass: {ex!r}'[:200]
        finally:
            shutil.rmtree(out, ignore_errors=True)
    b = (2.0 * vals['FETCH_SIZE'][0] + vals['WRITE_SIZE'][0]) * 1024.0
    what = (f"averages over {vals['FETCH_SIZE'][1]} / {vals['WRITE_SIZE'][1]} {kernel} launches" if steps_total is None else
            f"sums over all {vals['FETCH_SIZE'][1]} / {vals['WRITE_SIZE'][1]} kernel launches of the child's {steps_total} steps (warm-up and set-up kernels included), per step")
    return b, (f"measured by this run: two child passes of this command under rocprofv3 --kernel-trace --pmc (FETCH_SIZE {vals['FETCH_SIZE'][0]:.4e} KiB x 2 per the "
               f"gfx950 note + WRITE_SIZE {vals['WRITE_SIZE'][0]:.4e} KiB, {what})")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--height', type=int, default=512)
    ap.add_argument('--width', type=int, default=1024)
    ap.add_argument('--views', type=int, default=2)
    ap.add_argument('--samples', type=int, default=64)
    ap.add_argument('--fine', action='store_true', help='coarse + importance-sampled fine pass (BASELINE config 4)')
    ap.add_argument('--with-cost-volume', action='store_true',
                    help='BASELINE config 3: also build + regularise the D=64 spherical cost volume of every source view each step')
    ap.add_argument('--cv-res', type=int, nargs=2, default=None, help='cost-volume h w (default H/4 W/4)')
    ap.add_argument('--reg', choices=['unet2', 'costregnet'], default='unet2')
    ap.add_argument('--precision', choices=['f32', 'f16x3'], default='f32',
                    help="f32 = exact fp32 MFMA (default); f16x3 = opt-in split-f16 MFMA (3 per product, fp32 accumulate)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-end-to-end', action='store_true', help='skip the whole-module frame timing reported beside the headline (--no-cpu-baseline skips it too)')
    ap.add_argument('--eager-baseline', action='store_true', help='also time the eager PyTorch-ROCm port on this GPU over the WHOLE frame (minutes); a bounded strip is timed by default')
    ap.add_argument('--train', action='store_true', help='time a coarse + fine TRAINING step of the per-ray networks (forward, backward, Adam) on --train-rays rays per rank')
    ap.add_argument('--train-rays', type=int, default=2048)
    ap.add_argument('--no-live-pmc', action='store_true',
                    help='do not spawn the two rocprofv3 --pmc passes that measure roofline.traffic of this configuration (the committed counter file is quoted instead)')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)      # (set by the parent on the profiled child runs: never spawns again)
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the process group and run every collective even with ONE rank (exercises the RCCL branch on a 1-GPU box)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above has touched the GPU
        # (torch.cuda.is_available() not called yet), and the ranks are fresh child processes, not an exec of this one.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the HIP renderer has no CPU fallback')
    backend = os.environ.get('PGRF_BENCH_BACKEND', 'nccl')      # 'gloo' = test-only: ranks may then share one GPU
    if backend != 'nccl':
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    force_dist = args.force_dist or os.environ.get('PGRF_FORCE_COLLECTIVES') == '1'
    if force_dist:
        os.environ['PGRF_FORCE_COLLECTIVES'] = '1'      # panogrf_amd.dist: do not short-cut the collectives at world size 1
    distributed = world > 1 or force_dist
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # the RCCL build the collectives run on (torch's "nccl" backend IS RCCL on ROCm), recorded with every distributed line
    rccl_version = None
    if distributed and backend == 'nccl':
        try:
            rccl_version = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001 -- a record field, never a reason to fail the bench
            rccl_version = 'unknown (%s)' % type(e).__name__

    from panogrf_amd import engine as E
    from panogrf_amd import dist as D

    H, W, V, dn = args.height, args.width, args.views, args.samples
    if args.train:
        res = train_bench(args, dev, rank, world, distributed)
        if rank == 0:
            print(json.dumps(res))
        if distributed:
            torch.distributed.destroy_process_group()
        return
    case = synthetic_case(H, W, V, dn, seed=0, fine=args.fine)
    case['cfg']['hip_precision'] = args.precision
    fr_in = {k: v.to(dev) for k, v in case['frame'].items()}
    rr = E.RayRenderer(case['cfg'], case['weights'], device=dev)
    coords = D.tile_coords(rank, world, H, W, device=dev)
    rn = coords.shape[0]
    passes = 2 if args.fine else 1

    kern_ms = []
    cv = None
    if args.with_cost_volume:
        # one cost volume per source view acting as MVS reference (network/init_net.py:164-171: B = rfn pairs)
        from panogrf_amd import cost_volume as CVm
        from panogrf_amd import cost_reg as CR
        ch, cw = args.cv_res if args.cv_res else (H // 4, W // 4)
        g = torch.Generator().manual_seed(7)
        cv = dict(h=ch, w=cw, feats=torch.randn(V, 2, ch, cw, 32, generator=g).to(dev),
                  ref_mu=(1.0 + 7.0 * torch.rand(V, 1, ch, cw, generator=g)).to(dev),
                  rots=torch.eye(3).view(1, 1, 3, 3).repeat(V, 2, 1, 1).to(dev),
                  trans=torch.tensor([[[0.5, 0.0, 0.0], [-0.5, 0.0, 0.0]]]).repeat(V, 1, 1).to(dev),
                  k=CVm.depth_sampling(5, 3), args=dict(dataset_name='m3d', contain_dnet=True),
                  net=(CR.UNet2Reg() if args.reg == 'unet2' else CR.CostRegNet(dict(use_wrap_padding=True), 32)).to(dev).eval(),
                  CV=CVm, ms=[], sub=[])
        with torch.no_grad():
            for prm in cv['net'].parameters():
                prm.mul_(0.5)
            cv['reg_numel'] = cv['net'].forward_ndhwc(torch.zeros(1, 64, ch, cw, 32, device=dev)).numel()

    def step(timed):
        # per-frame: channel-last re-layout of the source maps (inputs stay NCHW in HBM like the reference's)
        fr = E.Frame(fr_in['imgs'], fr_in['img_feats'], fr_in['ray_feats'], fr_in['w2c'], fr_in['ref_depth_range'],
                     case['frame']['que_c2w'], case['frame']['que_depth_range'], device=dev)
        if cv is not None:
            # one cost volume per reference view, independent of each other: sharded by view over the ranks
            # (panogrf_amd/dist.py view_shard), regularised volumes all-gathered in view order
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            regs = []
            mine = list(D.view_shard(rank, world, V))
            if mine:
                # this rank's reference views as ONE batch, like the product stage (panogrf_amd/mvs_hip.py: B = rfn pairs in one call,
                # network/init_net.py:164-171): the regulariser's small low-resolution layers fill the chip twice as well, same values
                # bit for bit (2 views, one GPU: CostRegNet stage 1.80 -> 1.61 ms, UNet2 11.30 -> 11.03 ms)
                sl = slice(mine[0], mine[-1] + 1) if mine == list(range(mine[0], mine[-1] + 1)) else mine      # (round-robin shards: index list)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]      # sub-stage boundaries (roofline sub-blocks)
                dv = cv['CV'].build_depth_volume(cv['ref_mu'][sl], cv['k'], 0.5, 0.1, 10.0, 64)
                ev[0].record()
                vol = cv['CV'].calculate_cost_volume_erp(cv['args'], cv['feats'][sl], None, cv['trans'][sl],
                                                         cv['rots'][sl], depth_volume=dv)
                ev[1].record()
                regs = list(cv['net'].forward_ndhwc(vol).reshape(len(mine), -1))
                ev[2].record()
                if timed and rank == 0:
                    cv['sub'].append(ev)
                    cv['sub_views'] = len(mine)
            local = torch.stack(regs, 0) if regs else torch.zeros(0, cv['reg_numel'], device=dev)
            cv['reg_all'] = D.allgather_views(local, V)
            c1.record()
            if timed and rank == 0:
                cv['ms'].append((c0, c1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = rr.render_rays(fr, coords)
        e1.record()
        key = 'pixel_colors_nr_fine' if args.fine else 'pixel_colors_nr'
        img = D.allgather_rows(out[key], H, W)
        if timed:
            kern_ms.append((e0, e1))
        return img, out

    def sync_all():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img, out = step(True)
    sync_all()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    rays_per_s = H * W * args.steps / dt
    gather_diff = None
    if distributed and rank == 0:
        # outside the timed region: the reassembled image against this rank rendering a sample of ALL pixels by itself
        # (rays are independent and every kernel is deterministic, so the all-gathered frame must match bit for bit)
        gsel = torch.randperm(H * W, generator=torch.Generator().manual_seed(11))[:1024].to(dev)
        allc = torch.stack([(gsel % W).float(), (gsel // W).float()], -1)
        fr_chk = E.Frame(fr_in['imgs'], fr_in['img_feats'], fr_in['ray_feats'], fr_in['w2c'], fr_in['ref_depth_range'],
                         case['frame']['que_c2w'], case['frame']['que_depth_range'], device=dev)
        chk = rr.render_rays(fr_chk, allc)['pixel_colors_nr_fine' if args.fine else 'pixel_colors_nr']
        gather_diff = float((chk - img[gsel]).abs().max())

    # dominant kernel = render_pass_kernel; events bracket render_rays on the launch stream
    # (coarse depth fill + render pass [+ sampler + fine pass]); its share is > 99.9 % render pass.
    k_ms = sum(a.elapsed_time(b) for a, b in kern_ms) / max(len(kern_ms), 1)
    launches = passes
    flops_launch = flop_per_ray(V, dn, 1) * rn
    achieved = flops_launch / (k_ms / launches * 1e-3) / 1e12 if k_ms > 0 else 0.0
    # f32 mode: the dense f32 MFMA peak.  f16x3 mode executes three f16 MFMA products per algorithmic product, so its
    # roof for ALGORITHMIC flops is a third of the dense f16 peak (2.5 PF, MI355X_MICROARCH.md).
    peak_tf = PEAK_F32_MFMA_TFLOPS if args.precision == 'f32' else 2500.0 / 3.0

    # HBM-side traffic of the dominant kernel comes from rocprofv3 PMC passes of this same command (they cannot be
    # collected from inside the process): profiles/<round>_render_pass_pmc.txt, FETCH_SIZE doubled per the gfx950
    # note of MI355X_MICROARCH.md (wide coalesced reads are tallied at half), both counters in KiB.
    traffic, traffic_source = None, None
    live_why = None
    # (never from inside a profiled process: tools/profile_round.sh and the tools/pmc_*.sh scripts run this file under rocprofv3 themselves)
    under_profiler = 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
    if rank == 0 and world == 1 and not distributed and not (args.no_live_pmc or args.pmc_child or args.with_cost_volume or under_profiler):
        shape = ['--height', str(H), '--width', str(W), '--views', str(V), '--samples', str(dn), '--precision', args.precision] + (['--fine'] if args.fine else [])
        traffic, traffic_source = live_hbm_traffic(shape)
        if traffic is None:
            live_why, traffic_source = traffic_source, None
    if traffic is None and (H, W, V, dn, world) == (512, 1024, 2, 64, 1) and not args.fine and args.precision == 'f32':
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_render_pass_pmc.txt')))[-1:]:
            vals = {}
            for ln in open(f):
                t = ln.split()
                if len(t) >= 3 and t[1] in ('FETCH_SIZE', 'WRITE_SIZE'):
                    vals[t[1]] = float(t[2])
            if len(vals) == 2:
                traffic = (2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0
                traffic_source = ('profiles/' + os.path.basename(f) + ': (2 x FETCH_SIZE + WRITE_SIZE) KiB of a rocprofv3 --pmc pass of '
                                  'this command, NOT measured by this run (tools/profile_round.sh regenerates it)'
                                  + (f'; live passes: {live_why}' if live_why else ''))
    res = {
        'metric': f'rendered rays/sec ({H}x{W} ERP, {dn} samples)', 'value': rays_per_s, 'unit': 'rays/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32' if args.precision == 'f32' else 'f16x3-split products, f32 accumulate', 'data': 'synthetic',
        'config': {'workload': f'{H}x{W} ERP query, {V} source views, {dn} samples/ray'
                               + (' coarse + ' + str(dn) + ' fine' if args.fine else ', single pass')
                               + ' (BASELINE configs[1])' * (not args.fine and (H, W, V, dn) == (512, 1024, 2, 64)),
                   'rays_per_step': H * W, 'ray_tiles': world, 'parallelism': f'ray-tile x{world} + all-gather',
                   'gathered_image_max_abs_diff_vs_unsharded': gather_diff,
                   'collectives': (None if not distributed else backend + (', forced at world size 1' if world == 1 else '')),
                   'rccl_version': rccl_version},
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak_tf, 'unit': 'TFLOP/s',
                     'frac': achieved / peak_tf, 'traffic': traffic, 'traffic_source': traffic_source,
                     'kernel': 'render_pass_kernel', 'kernel_ms': k_ms / launches,
                     'flop_per_ray': flop_per_ray(V, dn, 1), 'rays_per_launch': rn},
    }
    if rank == 0 and cv is not None:
        cms = sum(a.elapsed_time(b) for a, b in cv['ms']) / max(len(cv['ms']), 1)
        gf = (438.5e9 if args.reg == 'unet2' else 16.2e9) * (cv['h'] * cv['w']) / (64 * 128) * V
        # sub-stage rooflines, per launch = per reference view: the build is HBM-bound -- algorithmic bytes =
        # D h w (C + 1) 4 (volume written, hypotheses read) + S h w C 4 (features read); the regulariser is f32-MFMA-bound.  UNet2's
        # wide layers run Winograd F(2x2x2, 3x3x3), which executes 1 / 3.375 of the direct convolution's multiplies: its roof
        # fraction is priced on the EXECUTED products (direct FLOPs / 3.375 against the f32 MFMA peak), not on the direct count.
        n_sub = max(len(cv['sub']), 1) * cv.get('sub_views', 1)          # (the events bracket a batch of this rank's views: per view)
        build_ms = sum(e[0].elapsed_time(e[1]) for e in cv['sub']) / n_sub
        reg_ms = sum(e[1].elapsed_time(e[2]) for e in cv['sub']) / n_sub
        cv_bytes = 64 * cv['h'] * cv['w'] * (32 + 1) * 4 + 2 * cv['h'] * cv['w'] * 32 * 4
        gf_view = gf / V
        wino = 3.375 if args.reg == 'unet2' else 1.0
        rl_build = {'bound': 'hbm', 'achieved': cv_bytes / (build_ms * 1e-3) / 1e9 if build_ms > 0 else 0.0, 'peak': 8000.0, 'unit': 'GB/s',
                    'kernel': 'cost_volume_pipe_kernel', 'kernel_ms': build_ms, 'algorithmic_bytes_per_launch': cv_bytes, 'traffic': None}
        rl_build['frac'] = rl_build['achieved'] / rl_build['peak']
        rl_reg = {'bound': 'mfma', 'achieved': gf_view / wino / (reg_ms * 1e-3) / 1e12 if reg_ms > 0 else 0.0, 'peak': PEAK_F32_MFMA_TFLOPS,
                  'unit': 'TFLOP/s', 'kernel': args.reg + ' forward (all its launches)', 'kernel_ms': reg_ms,
                  'direct_conv_flop_per_launch': gf_view, 'executed_flop_divisor': wino,
                  'note': ('achieved = direct-convolution FLOPs / 3.375 (Winograd F(2^3, 3^3) executes 8 / 27 of the multiplies) / time'
                           if wino > 1 else 'achieved = direct-convolution FLOPs / time')}
        rl_reg['frac'] = rl_reg['achieved'] / rl_reg['peak']
        res['cost_volume_stage'] = {'ms_per_step': cms, 'views': V, 'res': [cv['h'], cv['w']], 'D': 64, 'C': 32,
                                    'regulariser': args.reg, 'reg_direct_conv_tflops': gf_view / (reg_ms * 1e-3) / 1e12 if reg_ms > 0 else 0.0,
                                    'roofline': rl_reg, 'roofline_build': rl_build,
                                    'view_shards': min(world, V),
                                    'note': 'hypotheses + spherical cost volume + 3-D regulariser per source view (a rank runs its views as one batch; the sub-block times are per view), views sharded over ranks + all-gather; included in ms_per_step/value'}
        res['config']['workload'] += f' + D=64 cost volume {cv["h"]}x{cv["w"]} + {args.reg} per view (BASELINE configs[2])'
    if rank == 0 and world == 1 and args.eager_baseline:
        eg, eout = eager_gpu_baseline(case, dev, 2048)
        eg4, _ = eager_gpu_baseline(case, dev, 4096)
        key = 'pixel_colors_nr_fine' if args.fine else 'pixel_colors_nr'
        mse = float(((img.double() - eout[key].double()) ** 2).mean())
        eg['psnr_hip_vs_eager_db'] = 10 * math.log10(1.0 / max(mse, 1e-20))
        eg['value_ray_batch_4096'] = eg4['value']
        eg['speedup_hip_over_eager'] = rays_per_s / max(eg['value'], eg4['value'])
        res['torch_eager_gpu_baseline_whole_frame'] = eg
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # driver-run denominator of the north star's '>= 10x the reference single-GPU PyTorch path': bounded strip, <= 20 s
        es, eouts = eager_strip_baseline(case, dev)
        if eouts is not None:
            ec, parts = eouts
            key = 'pixel_colors_nr_fine' if args.fine else 'pixel_colors_nr'
            eimg = torch.cat([p_[key] for p_ in parts], 0)
            eidx = (ec[:eimg.shape[0], 1].long() * W + ec[:eimg.shape[0], 0].long())
            mse = float(((img[eidx].double() - eimg.double()) ** 2).mean())
            es['psnr_hip_vs_eager_db'] = 10 * math.log10(1.0 / max(mse, 1e-20))
        es['speedup_hip_over_eager'] = rays_per_s / es['value']
        res['torch_eager_gpu_baseline'] = es
    if rank == 0 and world == 1 and not (args.no_end_to_end or args.no_cpu_baseline) and V == 2 and args.precision == 'f32' and H % 64 == 0 and W % 64 == 0:
        res['end_to_end'] = end_to_end_frames(H, W, V, dn, args.fine, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, ccoords, cout = cpu_baseline(case)
        res['cpu_baseline'] = base
        if (H, W, V, dn) == (512, 1024, 2, 64) and not args.fine:
            res['cpu_baseline_config0'] = cpu_baseline_config0()
        # PSNR of the HIP image against the oracle on the sampled rays
        idx = (ccoords[:, 1].long() * W + ccoords[:, 0].long()).to(dev)
        key = 'pixel_colors_nr_fine' if args.fine else 'pixel_colors_nr'
        mine, want = img[idx].cpu(), cout[key]
        mse = float(((mine.double() - want.double()) ** 2).mean())
        res['psnr_vs_oracle_db'] = 10 * math.log10(1.0 / max(mse, 1e-20))
        # the reference's own metrics on the same rays (network/metrics.py): uint8 PSNR after
        # color_map_backward, WS-PSNR with the rays' row weights, and |PSNR(HIP, GT) - PSNR(oracle, GT)| against a fixed
        # target image (source view 0 at the query pixels: the synthetic views share no scene, any fixed image does)
        from panogrf_amd import metrics as MT
        rows = ccoords[:, 1].long()
        gt = case['frame']['imgs'][0].permute(1, 2, 0).reshape(-1, 3)[idx.cpu()]
        u8 = lambda t: MT.color_map_backward(t.numpy())
        same8 = bool((u8(mine) == u8(want)).all())
        res['psnr'] = {
            'float_vs_oracle_db': res['psnr_vs_oracle_db'],
            'uint8_vs_oracle_db': None if same8 else float(MT.compute_psnr(u8(want), u8(mine))),
            'uint8_images_identical': same8,
            'ws_vs_oracle_db': MT.WSPSNR.ws_psnr_rays(mine, want, rows, H),
            'hip_vs_gt_db': MT.psnr_float(mine, gt), 'oracle_vs_gt_db': MT.psnr_float(want, gt),
            'abs_delta_vs_gt_db': abs(MT.psnr_float(mine, gt) - MT.psnr_float(want, gt)),
            'rays': int(idx.numel())}
    if rank == 0:
        print(json.dumps(res))
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
