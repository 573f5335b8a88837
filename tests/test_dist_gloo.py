"""Multi-rank path on CPU: world_size-2 gloo run of the ray-tile sharding + image all-gather
(panogrf_amd/dist.py).  The per-tile render is replaced by a closed-form function of the pixel
coordinate so the check is exact; the HIP render itself is covered by the -m gpu tests."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from panogrf_amd import dist as D
    coords = D.tile_coords(rank, world, H, W)
    shard = torch.stack([coords[:, 0] * 0.5 + coords[:, 1], coords[:, 0] - coords[:, 1], coords[:, 1] * 2.0], -1)
    full = D.allgather_rows(shard, H, W)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    exp = torch.stack([xs * 0.5 + ys, xs - ys, ys * 2.0], -1).reshape(-1, 3)
    ok = torch.equal(full, exp)
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def _run(world, H, W):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, H, W, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_allgather_equal_tiles_world2():
    _run(2, 8, 6)


def test_allgather_ragged_tiles_world2():
    _run(2, 7, 5)


def test_allgather_equal_tiles_world4():
    _run(4, 16, 5)


def _view_worker(rank, world, port, n_views, ret, offset=0):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from panogrf_amd import dist as D
    mine = D.view_shard(rank, world, n_views, offset)
    local = torch.stack([torch.full((3, 4), float(v)) + torch.arange(4.0) for v in mine], 0) if mine else torch.zeros(0, 3, 4)
    full = D.allgather_views(local, n_views, offset=offset)
    exp = torch.stack([torch.full((3, 4), float(v)) + torch.arange(4.0) for v in range(n_views)], 0)
    ret[rank] = torch.equal(full, exp)
    dist.barrier()
    dist.destroy_process_group()


def _run_views(world, n_views, offset=0):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_view_worker, args=(world, _free_port(), n_views, ret, offset), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_view_shard_partition():
    from panogrf_amd import dist as D
    for world in (1, 2, 3, 4, 8):
        for nv in (1, 2, 3, 4):
            got = sorted(v for r in range(world) for v in D.view_shard(r, world, nv))
            assert got == list(range(nv))


def test_allgather_views_world2_even():
    _run_views(2, 4)


def test_allgather_views_world2_ragged_and_idle_rank():
    _run_views(2, 3)
    _run_views(2, 1)


def _ray_worker(rank, world, port, n, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from panogrf_amd import dist as D
    r0, r1 = D.ray_shard(rank, world, n)
    idx = torch.arange(r0, r1, dtype=torch.float32)
    colours = torch.stack([idx, idx * 2, idx + 0.5], -1).unsqueeze(0)        # [1, n_rank, 3] like 'pixel_colors_nr'
    mask = (torch.arange(r0, r1) % 3 == 0).unsqueeze(0)                      # [1, n_rank] bool like 'ray_mask'
    full_c, full_m = D.allgather_rays(colours, n, 1), D.allgather_rays(mask, n, 1)
    all_idx = torch.arange(n, dtype=torch.float32)
    ok = (torch.equal(full_c, torch.stack([all_idx, all_idx * 2, all_idx + 0.5], -1).unsqueeze(0)) and full_m.dtype == torch.bool and
          torch.equal(full_m, (torch.arange(n) % 3 == 0).unsqueeze(0)))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_rays_world2_ragged_and_bool():
    """The renderer module's sharding (renderer.render under a process group): contiguous ray chunks that differ by one ray,
    reassembled in ray order, bool masks included."""
    for n in (10, 11):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_ray_worker, args=(2, _free_port(), n, ret), nprocs=2, join=True)
        assert all(ret[r] for r in range(2)), dict(ret)
    from panogrf_amd import dist as D
    assert [D.ray_shard(g, 3, 10) for g in range(3)] == [(0, 4), (4, 7), (7, 10)]


def test_encoder_views_go_to_the_ranks_the_mvs_stage_leaves_idle():
    """panogrf_amd.dist.encoder_offset / view_shard(offset): with >= 2 * rfn ranks the image-encoder views land on ranks
    rfn..2*rfn-1 (disjoint from the MVS stage's ranks 0..rfn-1), otherwise they share them; always a partition."""
    from panogrf_amd import dist as D
    for world in (2, 3, 4, 8):
        for nv in (2, 4):
            off = D.encoder_offset(world, nv)
            owners = {v: [r for r in range(world) if v in D.view_shard(r, world, nv, off)] for v in range(nv)}
            assert all(len(o) == 1 for o in owners.values()), (world, nv, owners)
            mvs = {v: [r for r in range(world) if v in D.view_shard(r, world, nv)][0] for v in range(nv)}
            if world >= 2 * nv:
                assert off == nv and not set(mvs.values()) & {o[0] for o in owners.values()}, (world, nv)
            else:
                assert off == 0


def test_allgather_views_world4_with_owner_offset():
    _run_views(4, 2, offset=2)       # views owned by ranks 2 and 3; ranks 0 and 1 contribute empty shards


def test_allgather_tiles_and_views_world8():
    """world size 8 (the driver's scaling run): equal and ragged row tiles; 2 and 4 reference views over 8 ranks (6 / 4 idle ranks
    pass empty shards through the view all-gather), with and without the encoder's owner offset."""
    _run(8, 16, 6)
    _run(8, 13, 5)
    _run_views(8, 2)
    _run_views(8, 4)
    _run_views(8, 2, offset=2)
