"""Ray-tile sharding of one panorama across the GPUs of a node.

Every ray is independent given the per-frame source maps, so rank g renders the contiguous row
tile [g*H/G, (g+1)*H/G) with NO data-path collective; one all-gather (RCCL over xGMI when the
backend is "nccl") reassembles the final image.  The reference has no multi-GPU inference at all
(train/trainer.py:232-234): this is new, not a translation.
"""
import os

import torch
import torch.distributed as dist


def _world(group=None):
    """World size the collectives act on; 1 = nothing to gather.  With PGRF_FORCE_COLLECTIVES=1 a single-rank process
    group is NOT short-cut (-1 is returned: 'one rank, but run the collective'), so that the RCCL calls themselves
    execute on a one-GPU box (bench.py --force-dist, tests/mp_nccl_one_rank.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    w = dist.get_world_size(group)
    return -1 if (w == 1 and os.environ.get('PGRF_FORCE_COLLECTIVES') == '1') else w


def tile_rows(rank, world, H):
    """Row range [r0, r1) of rank `rank`: equal contiguous tiles, remainder rows to the first ranks."""
    base, rem = divmod(H, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def tile_coords(rank, world, H, W, device='cpu'):
    """(x,y) pixel coords [rows*W,2] of this rank's tile, row-major like utils/imgs_info.py:158-181."""
    r0, r1 = tile_rows(rank, world, H)
    ys, xs = torch.meshgrid(torch.arange(r0, r1, dtype=torch.float32, device=device),
                            torch.arange(W, dtype=torch.float32, device=device), indexing='ij')
    return torch.stack([xs, ys], -1).reshape(-1, 2)


def allgather_rows(shard, H, W, group=None):
    """shard [rows_g*W, C] of every rank -> full [H*W, C] on every rank.  Equal tiles use one
    all_gather_into_tensor (a single ncclAllGather); ragged tiles fall back to padded gather."""
    world = _world(group)
    if world == 1:
        return shard
    world = abs(world)
    C = shard.shape[1]
    if dist.get_backend(group) == 'gloo' and shard.is_cuda:
        # test-only path (several ranks sharing one GPU cannot use RCCL): stage through host memory
        return allgather_rows(shard.cpu(), H, W, group).to(shard.device)
    if H % world == 0:
        full = torch.empty(H * W, C, dtype=shard.dtype, device=shard.device)
        dist.all_gather_into_tensor(full, shard.contiguous(), group=group)
        return full
    rows_max = -(-H // world)
    pad = torch.zeros(rows_max * W, C, dtype=shard.dtype, device=shard.device)
    pad[:shard.shape[0]] = shard
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for g in range(world):
        r0, r1 = tile_rows(g, world, H)
        out.append(parts[g][:(r1 - r0) * W])
    return torch.cat(out, 0)


# ---- MVS stage: one cost volume per reference view (network/init_net.py:164-171 builds B = rfn pairs), each
# independent of the others -> shard BY REFERENCE VIEW, then all-gather the per-view result.

def view_shard(rank, world, n_views, offset=0):
    """Reference views handled by `rank`: view v belongs to rank (v + offset) % world -- round-robin, so 2 or 4 views spread
    over 2 or 4 ranks and the remaining ranks (a node has 8) idle through that stage.  `offset` moves the owners: the
    per-frame image encoder is dealt with offset = n_views, i.e. to the ranks the MVS stage leaves idle (encoder_offset)."""
    return [v for v in range(n_views) if (v + offset) % world == rank]


def encoder_offset(world, n_views):
    """Owner offset of the per-frame image encoder's views: the MVS stage occupies ranks 0..n_views-1 (one cost volume per
    reference view), so with at least 2 * n_views ranks the encoder images go to ranks n_views..2*n_views-1 and both stages
    run side by side; with fewer ranks they share the MVS ranks (offset 0)."""
    return n_views if world >= 2 * n_views else 0


def allgather_views(local, n_views, group=None, offset=0):
    """local [len(view_shard(rank, offset=offset)), ...] of every rank -> [n_views, ...] in view order on every rank.  Ranks
    own different numbers of views (including none), so every rank contributes ceil(n_views/world) padded slots."""
    world = _world(group)
    if world == 1:
        return local
    world = abs(world)
    if dist.get_backend(group) == 'gloo' and local.is_cuda:      # test-only: ranks sharing one GPU
        return allgather_views(local.cpu(), n_views, group, offset).to(local.device)
    slots = -(-n_views // world)
    pad = torch.zeros((slots,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    full = torch.empty((world * slots,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, pad.contiguous(), group=group)
    full = full.view((world, slots) + tuple(local.shape[1:]))
    out = [None] * n_views
    for g in range(world):
        for i, v in enumerate(view_shard(g, world, n_views, offset)):
            out[v] = full[g, i]
    return torch.stack(out, 0)


# ---- generic ray sharding for the renderer module (any coords list, not only whole row tiles)

def ray_shard(rank, world, n):
    """Contiguous chunk [r0, r1) of n rays for `rank`; the first n % world ranks get one ray more."""
    return tile_rows(rank, world, n)


def allgather_rays(local, n, dim=1, group=None):
    """local [..., n_rank, ...] (this rank's chunk of ray_shard along `dim`) -> [..., n, ...] on every rank, in ray order.
    Chunks differ by at most one ray: every rank contributes ceil(n / world) padded slots to ONE all-gather."""
    world = _world(group)
    if world == 1:
        return local
    world = abs(world)
    if dist.get_backend(group) == 'gloo' and local.is_cuda:      # test-only: ranks sharing one GPU
        return allgather_rays(local.cpu(), n, dim, group).to(local.device)
    was_bool = local.dtype == torch.bool
    x = local.to(torch.uint8) if was_bool else local
    x = x.movedim(dim, 0).contiguous()
    slots = -(-n // world)
    pad = torch.zeros((slots,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[:x.shape[0]] = x
    full = torch.empty((world * slots,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(full, pad, group=group)
    parts = []
    for g in range(world):
        r0, r1 = ray_shard(g, world, n)
        parts.append(full[g * slots:g * slots + (r1 - r0)])
    out = torch.cat(parts, 0).movedim(0, dim)
    return out.to(torch.bool) if was_bool else out
