"""Worker of tests/test_gpu_bench_multirank.py::test_renderer_module_shards_rays_and_views: launched by torch.distributed.run
with 2 ranks over gloo on ONE GPU; renders through the renderer MODULE with ray / view sharding and compares, on every rank,
with the same call under cfg hip_shard_rays / hip_shard_views / hip_shard_encoders = False.  With 4 ranks the image-encoder
views go to ranks 2 and 3 (the MVS stage keeps 0 and 1 busy) and ranks render a quarter of the rays each."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import mvs_fixtures as MF  # noqa: E402


def main():
    dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
    torch.cuda.set_device(0)
    from panogrf_amd.renderer import name2network
    from panogrf_amd.imgs_info import build_imgs_info, build_render_imgs_info, imgs_info_to_torch
    H, W, dn = 64, 128, 32
    args = dict(MF.MVS_ARGS, height=32, width=64, mono_height=32, mono_width=64, depth_height=32, depth_width=64)
    cfg = dict(args, network='neuray_gen', init_net_type='cost_volume', use_wrap_padding=True, seq_len=3, height=H, width=W,
               use_disp=True, use_hierarchical_sampling=True, render_depth=True, depth_sample_num=dn, fine_depth_sample_num=dn,
               sample_num=dn, dist_decoder_cfg={'use_vis': False}, fine_dist_decoder_cfg={'use_vis': False})
    cfg.update(min_depth=0.5, max_depth=15.0)
    torch.manual_seed(0)                                       # same weights on every rank
    net = name2network['neuray_gen'](cfg, d_net=MF.StubMono()).cuda().eval()
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(2, 3, H, W, generator=g).numpy()
    pose = lambda tx: np.concatenate([np.eye(3), np.array([[tx], [0.0], [0.0]])], 1)
    mk = lambda: (imgs_info_to_torch(build_imgs_info(imgs, np.stack([pose(0.5), pose(-0.5)]), [[0.5, 15.0]] * 2), 'cuda'),
                  imgs_info_to_torch(build_imgs_info(imgs[::-1].copy(), np.stack([pose(-0.5), pose(0.5)]), [[0.5, 15.0]] * 2), 'cuda'),
                  imgs_info_to_torch(build_render_imgs_info(pose(0.0), (H, W), [0.5, 15.0]), 'cuda'))
    outs = {}
    for sharded in (True, False):
        net.cfg['hip_shard_rays'] = net.cfg['hip_shard_views'] = net.cfg['hip_shard_encoders'] = sharded
        ref, src, que = mk()
        with torch.no_grad():
            outs[sharded] = net({'que_imgs_info': que, 'ref_imgs_info': ref, 'src_imgs_info': src, 'eval': True})
    a, b = outs[True], outs[False]
    assert set(a) == {k for k in b if k.startswith(('pixel_colors', 'render_depth', 'ray_mask'))}, (sorted(a), sorted(b))
    for k in a:
        assert a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
        d = float((a[k].float() - b[k].float()).abs().max())
        assert d <= 1e-5 * max(1.0, float(b[k].float().abs().max())), (k, d)
    print(f'rank {dist.get_rank()}: sharded == unsharded on {len(a)} keys', flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
