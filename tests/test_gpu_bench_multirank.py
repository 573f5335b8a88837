"""bench.py's N>1 code path (tile sharding, image all-gather, max-over-ranks timing) exercised on ONE GPU
with 2 ranks over gloo (RCCL cannot run two ranks on one device; the 8-GPU run is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, PGRF_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29517', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--height', '64', '--width', '128', '--samples', '32']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['config']['ray_tiles'] == 2 and d['scaling'] == 'strong'
    assert d['roofline']['rays_per_launch'] == 64 * 128 // 2
    assert d['value'] > 0
    assert d['config']['gathered_image_max_abs_diff_vs_unsharded'] == 0.0


def test_bench_four_ranks_gathered_frame_is_the_unsharded_frame():
    """4 ranks, coarse + fine: bench.py itself checks the reassembled frame against unsharded rendering of a pixel sample."""
    env = dict(os.environ, PGRF_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1',
           '--master-port', '29523', os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '1', '--warmup', '1',
           '--height', '64', '--width', '128', '--samples', '32', '--fine']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 4 and d['config']['ray_tiles'] == 4
    assert d['config']['gathered_image_max_abs_diff_vs_unsharded'] == 0.0


def test_bench_two_ranks_cost_volume_sharded_by_view():
    env = dict(os.environ, PGRF_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29519', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--height', '64', '--width', '128', '--samples', '32', '--with-cost-volume', '--cv-res', '16', '32',
           '--reg', 'costregnet']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['cost_volume_stage']['view_shards'] == 2 and d['cost_volume_stage']['ms_per_step'] > 0


def test_renderer_module_shards_rays_and_views():
    """`renderer(data)` under a process group: rays split over the ranks + one all-gather per output, the MVS stage split by
    reference view -- the same result as the unsharded call (tests/mp_render_shard.py, 2 ranks, gloo)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29521', os.path.join(ROOT, 'tests', 'mp_render_shard.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert r.stdout.count('sharded == unsharded') == 2
