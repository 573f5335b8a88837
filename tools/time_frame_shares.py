"""Per-rank shares of ONE whole `renderer(data)` frame under the 8-GPU split, measured on ONE GPU (the 8-GPU run itself is
the driver's): what each rank executes between the frame's start and the final image all-gather, at the shipped
configuration (512x1024, 2 views, 64 + 64 samples, MVS stage at depth 256x512).

  rank 0, 1   MVS stage (init net) of ONE reference view            -> all-gather ray_feats / mvs_depth
  rank 2, 3   image encoder of ONE source view (side by side)       -> all-gather img_feats (4 MB per view)
  every rank  visibility encoder (replicated, small), then its 64-row tile of the coarse + fine render
              -> all-gather of the final image (786 KiB per rank)

Frame time at 8 ranks = max(MVS share, encoder share) + visibility encoder + slowest row tile + collectives; the collectives
are priced, not measured: bytes / 50 GB/s (one xGMI link direction, conservative) + 30 us latency each.
python tools/time_frame_shares.py [world]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from render import DEFAULT_CFG, MVS_CFG, ConstantMono
from panogrf_amd import dist as D
from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, imgs_info_to_torch
from panogrf_amd.renderer import name2network

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W, V = 512, 1024, 2
cfg = dict(DEFAULT_CFG, height=H, width=W, depth_sample_num=64, fine_depth_sample_num=64, sample_num=64)
cfg.update(MVS_CFG, depth_height=H // 2, depth_width=W // 2, hip_reuse_source_maps=False)
torch.manual_seed(0)
net = name2network['neuray_gen'](cfg, d_net=ConstantMono(3.0)).cuda().eval()
g = torch.Generator().manual_seed(1)
imgs = torch.rand(V, 3, H, W, generator=g).numpy()
pose = lambda tx: np.concatenate([np.eye(3), np.array([[tx], [0.0], [0.0]])], 1)
ref = imgs_info_to_torch(build_imgs_info(imgs, np.stack([pose(0.5), pose(-0.5)]), [[0.5, 15.0]] * V), 'cuda')
src = imgs_info_to_torch(build_imgs_info(imgs[::-1].copy(), np.stack([pose(-0.5), pose(0.5)]), [[0.5, 15.0]] * V), 'cuda')
que = imgs_info_to_torch(build_render_imgs_info(pose(0.0), (H, W), [0.5, 15.0]), 'cuda')


def timeit(fn, n=10, warm=3, reps=3):
    """median over `reps` batches of the mean of n calls (a single batch now and then catches a clock dip: one of eight equal row
    tiles read 9.6 ms against 8.7-8.8 for its neighbours in the first round-5 collection)"""
    for _ in range(warm):
        fn()
    vals = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        vals.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(vals)[len(vals) // 2]


with torch.no_grad():
    whole = timeit(lambda: net({'que_imgs_info': dict(que), 'ref_imgs_info': dict(ref), 'src_imgs_info': dict(src), 'eval': True}))
    one = lambda d, i: {k: (v[i:i + 1] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == V else v) for k, v in d.items()}
    mvs1 = timeit(lambda: net._run_init_net(one(ref, 0), one(src, 0), False))
    mvs2 = timeit(lambda: net._run_init_net(ref, src, False))
    enc1 = timeit(lambda: net._encode_local_views(ref['imgs'], [0]))
    enc2 = timeit(lambda: net._encode_local_views(ref['imgs'], [0, 1]))
    ret = net._run_init_net(ref, src, False)
    feats = net._encode_local_views(ref['imgs'], [0, 1])
    he = net._encoder_fn(ref['imgs'])
    vis_fn = lambda: net._graphed('enc_vis_hip', lambda rf, f: (he.vis_encoder(rf, f.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2),),
                                  (ret['ray_feats'].contiguous(), feats), modules=(net.vis_encoder,))
    vis = timeit(vis_fn)
    # render tiles: the module's render() on this rank's rows, maps given
    rfe = vis_fn()[0]
    refm = dict(ref, img_feats=feats, ray_feats=rfe)
    refm['_pgrf_frame'] = net._make_frame(que, refm, False)
    tiles = []
    for r in range(world):
        r0, r1 = D.ray_shard(r, world, H * W)
        q = dict(que, coords=que['coords'][:, r0:r1])
        tiles.append(timeit(lambda: net.render_impl(q, refm, False), n=5, warm=2))
    full_render = timeit(lambda: net.render_impl(dict(que), refm, False), n=5, warm=2)

link = 50e9
coll = lambda nbytes: nbytes / link * 1e3 + 0.03
g_mvs = coll(V * (32 * (H // 8) * (W // 8) + (H // 2) * (W // 2)) * 4)
g_enc = coll(V * 32 * (H // 4) * (W // 4) * 4)
g_img = coll(H * W * 8 * 4)
front = max(mvs1, enc1) if world >= 2 * V else (mvs1 + enc1 if world >= V else mvs2 + enc2)
frame8 = front + g_mvs + g_enc + vis + max(tiles) + g_img
print(f'whole frame, one GPU, everything recomputed per frame: {whole:.2f} ms')
print(f'MVS stage: 2 views {mvs2:.2f} ms, ONE view (a rank\'s share) {mvs1:.2f} ms')
print(f'image encoder: 2 views {enc2:.2f} ms, ONE view {enc1:.2f} ms;  visibility encoder (replicated) {vis:.2f} ms')
print(f'render (coarse + fine), whole frame {full_render:.2f} ms; {world} row tiles: ' + ' '.join(f'{t:.2f}' for t in tiles) + f'  max {max(tiles):.2f} ms')
print(f'collectives priced at 50 GB/s + 30 us: ray_feats/mvs_depth {g_mvs:.3f} ms, img_feats {g_enc:.3f} ms, image {g_img:.3f} ms')
print(f'projected frame at {world} ranks: max(MVS {mvs1:.2f}, encoder {enc1:.2f}) + {g_mvs + g_enc:.2f} + vis {vis:.2f} + tile {max(tiles):.2f} + {g_img:.2f} '
      f'= {frame8:.2f} ms  ->  {whole / frame8:.2f}x over one GPU ({whole:.2f} ms)')
replicated = mvs1 + enc2 + g_mvs + vis + max(tiles) + g_img
print(f'(round-2 layout, encoders replicated on every rank: {replicated:.2f} ms -> {whole / replicated:.2f}x)')
