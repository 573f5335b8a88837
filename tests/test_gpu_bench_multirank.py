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


@pytest.mark.parametrize('ranks', [2, 4])
def test_renderer_module_shards_rays_and_views(ranks):
    """`renderer(data)` under a process group: rays split over the ranks + one all-gather per output, the MVS stage split by
    reference view, the image encoder by source view on the ranks the MVS stage leaves idle -- the same
    result as the unsharded call (tests/mp_render_shard.py, 2 and 4 ranks, gloo, one GPU)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
           '--master-port', str(29521 + ranks), os.path.join(ROOT, 'tests', 'mp_render_shard.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert r.stdout.count('sharded == unsharded') == ranks


def test_rccl_collectives_execute_with_one_rank():
    """backend = nccl (RCCL) on this one GPU: panogrf_amd.dist's all-gathers, the barrier and the MAX all-reduce really run
    (PGRF_FORCE_COLLECTIVES=1 stops them from being short-cut at world size 1)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29525', os.path.join(ROOT, 'tests', 'mp_nccl_one_rank.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert 'rccl one-rank collectives ok' in r.stdout


def test_bench_distributed_path_on_rccl_with_one_rank():
    """bench.py itself with backend nccl, world size 1 and --force-dist: process group on RCCL, tile all-gather through
    all_gather_into_tensor, barrier + MAX all-reduce around the timed region, gathered-frame check."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29527', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--height', '64', '--width', '128', '--samples', '32', '--force-dist', '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['value'] > 0
    assert d['config']['gathered_image_max_abs_diff_vs_unsharded'] == 0.0
    assert d['config']['collectives'] == 'nccl, forced at world size 1'
    assert isinstance(d['config']['rccl_version'], str) and d['config']['rccl_version'][0].isdigit()      # what the run was on, for the first real 8-GPU record


def test_plain_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` as typed (no torchrun, WORLD_SIZE unset): bench.py starts the ranks itself as child
    processes (gloo here: two ranks share this one GPU) and relays rank 0's JSON line; exit code 0."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['PGRF_BENCH_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--height', '64',
           '--width', '128', '--samples', '32']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['config']['ray_tiles'] == 2
    assert d['config']['gathered_image_max_abs_diff_vs_unsharded'] == 0.0


@pytest.mark.parametrize('extra', [[], ['--fine'], ['--with-cost-volume', '--cv-res', '16', '32', '--reg', 'costregnet'], ['--views', '4']],
                         ids=['single-pass', 'coarse+fine', 'cost-volume', 'four-views'])
def test_plain_bench_gpus_8_on_one_gpu(extra):
    """The driver's 8-GPU invocation, as typed -- `python bench.py --gpus 8 ...` -- on the one GPU of the test box (gloo: eight
    self-launched ranks share it): 8 row tiles of 8 rows each, frame all-gathered and bit-identical to unsharded rendering; with
    the cost-volume stage 2 (or 4) ranks hold a reference view each and 6 (4) ranks pass an empty shard through the view
    all-gather (round 4; the earlier tests stopped at 4 ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['PGRF_BENCH_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--height', '64',
           '--width', '128', '--samples', '32', '--no-cpu-baseline'] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 8 and d['config']['ray_tiles'] == 8 and d['scaling'] == 'strong'
    assert d['roofline']['rays_per_launch'] == 64 * 128 // 8
    assert d['config']['gathered_image_max_abs_diff_vs_unsharded'] == 0.0
    if '--with-cost-volume' in extra:
        assert d['cost_volume_stage']['view_shards'] == 2 and d['cost_volume_stage']['ms_per_step'] > 0


def test_bench_train_two_ranks_data_parallel():
    """`bench.py --train --gpus 2` (gloo, one GPU): each rank its own rays, per-ray weight gradients all-reduced, weak scaling; the
    loss falls over the steps and the line carries the training roofline block."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['PGRF_BENCH_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--train', '--gpus', '2', '--steps', '6', '--warmup', '1', '--height', '64',
           '--width', '128', '--samples', '32', '--train-rays', '256', '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
    assert d['roofline']['bound'] == 'mfma' and d['roofline']['flop_per_ray'] == 3 * 2 * 32 * 2 * (2 * 28464 + 6480 + 32 * 32)
    assert d['loss_first_last'][1] < d['loss_first_last'][0]


def test_renderer_module_shards_rays_and_views_eight_ranks():
    """`renderer(data)` under an 8-rank process group (gloo, one GPU): 8 ray shards, the MVS stage on 2 ranks, the image encoder on
    2 others, 4 ranks idle through both -- the same result as the unsharded call on every rank."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(ROOT, 'tests', 'mp_render_shard.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert r.stdout.count('sharded == unsharded') == 8


def test_cabi_rccl_image_allgather_one_rank():
    """The C-ABI collective entry points (include/panogrf_hip.h: pgrf_comm_unique_id / pgrf_comm_init / pgrf_allgather_image /
    pgrf_comm_destroy) for hosts without torch.distributed: a one-rank communicator on this GPU, the image
    all-gather through ncclAllGather, in a fresh process (RCCL resolved by dlopen)."""
    code = r'''
import ctypes as C, torch
from panogrf_amd._lib import lib
torch.cuda.set_device(0)
uid = (C.c_char * 128)()
lib.check(lib.pgrf_comm_unique_id(C.cast(uid, C.c_void_p)))
comm = C.c_void_p()
lib.check(lib.pgrf_comm_init(C.cast(uid, C.c_void_p), 0, 1, C.byref(comm)))
assert comm.value
shard = torch.rand(64 * 128, 3, device='cuda')
full = torch.zeros_like(shard)
st = torch.cuda.current_stream().cuda_stream
lib.check(lib.pgrf_allgather_image(comm, C.c_void_p(shard.data_ptr()), C.c_void_p(full.data_ptr()), shard.numel(), C.c_void_p(st)))
torch.cuda.synchronize()
assert torch.equal(full, shard)
assert lib.pgrf_allgather_image(None, C.c_void_p(shard.data_ptr()), C.c_void_p(full.data_ptr()), 4, None) != 0
# in place, as SURVEY.md 8(e) words the collective: the rank rendered into ITS rows of the frame buffer (shard == full + rank * count)
frame = torch.rand(64 * 128, 3, device='cuda'); want = frame.clone()
lib.check(lib.pgrf_allgather_image(comm, C.c_void_p(frame.data_ptr()), C.c_void_p(frame.data_ptr()), frame.numel(), C.c_void_p(st)))
torch.cuda.synchronize()
assert torch.equal(frame, want)
ver = C.c_int(0)
lib.check(lib.pgrf_comm_rccl_version(C.byref(ver)))
assert ver.value >= 20000, ver.value
print('rccl version code', ver.value)
lib.check(lib.pgrf_comm_destroy(comm))
print('cabi rccl allgather ok')
'''
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2500:])
    assert 'cabi rccl allgather ok' in r.stdout


def test_bench_line_measures_its_own_hbm_traffic():
    """The plain one-GPU bench line carries `roofline.traffic` measured by the run itself: two child passes of the same command under
    `rocprofv3 --kernel-trace --pmc` (FETCH_SIZE, WRITE_SIZE), per render_pass launch.  A small frame keeps the three processes short;
    the per-sample outputs alone (hit probability, colours, density: 20 bytes per sample + 28 per ray) bound the bytes from below."""
    import shutil
    if shutil.which('rocprofv3') is None:
        pytest.skip('rocprofv3 not installed')
    H, W, dn = 64, 128, 32
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--height', str(H), '--width', str(W),
           '--samples', str(dn), '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    rl = d['roofline']
    assert rl['traffic_source'] and rl['traffic_source'].startswith('measured by this run'), rl['traffic_source']
    assert rl['traffic'] > H * W * dn * 20 and rl['traffic'] < 64e9
    # and the switch that keeps the committed file's figure (or null off the headline configuration)
    r2 = subprocess.run(cmd + ['--no-live-pmc'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    d2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith('{')][-1])
    assert d2['roofline']['traffic'] is None
