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
