"""ONE rank on the `nccl` (= RCCL) backend with the collectives forced (PGRF_FORCE_COLLECTIVES=1): the calls the 8-GPU run
makes -- all_gather_into_tensor (equal tiles), all_gather (ragged tiles), barrier, MAX all-reduce -- execute on an MI355X
through RCCL instead of being short-cut at world size 1.  Launched by tests/test_gpu_bench_multirank.py under torchrun."""
import os
import sys

import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ['PGRF_FORCE_COLLECTIVES'] = '1'
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    tdist.init_process_group('nccl', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']), device_id=dev)
    assert tdist.get_backend() == 'nccl' and tdist.get_world_size() == 1
    from panogrf_amd import dist as D
    g = torch.Generator().manual_seed(0)
    H, W = 16, 32
    shard = torch.rand(H * W, 3, generator=g).to(dev)
    full = D.allgather_rows(shard, H, W)
    assert full.data_ptr() != shard.data_ptr() and torch.equal(full, shard)      # a real collective made a new tensor
    views = torch.rand(2, 4, 5, generator=g).to(dev)
    assert torch.equal(D.allgather_views(views, 2), views)
    rays = torch.rand(1, 37, 3, generator=g).to(dev)
    out = D.allgather_rays(rays, 37, 1)
    assert out.data_ptr() != rays.data_ptr() and torch.equal(out, rays)
    mask = torch.rand(1, 37, generator=g).to(dev) > 0.5
    assert torch.equal(D.allgather_rays(mask, 37, 1), mask)
    tdist.barrier()
    t = torch.tensor([1.5], device=dev)
    tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    assert float(t) == 1.5
    torch.cuda.synchronize()
    print('rccl one-rank collectives ok')
    tdist.destroy_process_group()


if __name__ == '__main__':
    main()
