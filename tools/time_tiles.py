"""Per-rank cost of the 8-GPU split of a 512x1024 frame, measured on ONE GPU: the kernel time of each contiguous 64-row tile
(bench.py / dist.tile_coords) and of each interleaved row set g, g+8, ... -- the run's time is the MAX over ranks.
python tools/time_tiles.py [world]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from panogrf_amd import engine as E, dist as D

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W, V, dn = 512, 1024, 2, 64
case = B.synthetic_case(H, W, V, dn)
dev = torch.device('cuda:0')
f = case['frame']
fr = E.Frame(f['imgs'], f['img_feats'], f['ray_feats'], f['w2c'], f['ref_depth_range'], f['que_c2w'], f['que_depth_range'], device=dev)
rr = E.RayRenderer(case['cfg'], case['weights'], device=dev)


def ms(coords, n=5):
    for _ in range(2):
        rr.render_rays(fr, coords)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        rr.render_rays(fr, coords)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
full = torch.stack([xs, ys], -1).to(dev)
print(f'whole frame: {ms(full.reshape(-1, 2)):.2f} ms -> ideal per rank at {world} ranks {ms(full.reshape(-1, 2)) / world:.2f} ms')
contig = lambda g: full[slice(*D.tile_rows(g, world, H))].reshape(-1, 2).contiguous()
ms(contig(0))                                   # clocks / caches warm before the first measured share
tiles = [ms(contig(g)) for g in range(world)]
print('contiguous row tiles :', ' '.join(f'{t:.2f}' for t in tiles), f' max {max(tiles):.2f} ms')
print(f'  (tile 0 measured again, last: {ms(contig(0)):.2f} ms; tile {world - 1}: {ms(contig(world - 1)):.2f} ms)')
inter = [ms(full[g::world].reshape(-1, 2).contiguous()) for g in range(world)]
print('interleaved rows     :', ' '.join(f'{t:.2f}' for t in inter), f' max {max(inter):.2f} ms')
for blk in (8, 16, 32):
    if H % (blk * world):
        continue
    inter8 = [ms(full.reshape(H // blk, blk, W, 2)[g::world].reshape(-1, 2).contiguous()) for g in range(world)]
    print(f'interleaved {blk}-row blocks:', ' '.join(f'{t:.2f}' for t in inter8), f' max {max(inter8):.2f} ms')
