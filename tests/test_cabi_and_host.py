"""CPU-side checks: the C-ABI library loads here (no GPU) and exports every symbol the header declares;
host logic (weight packing, tiling, error paths) behaves; no compute kernels are launched."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'panogrf_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(pgrf_[a-z0-9_]+)\s*\(', txt)))


def test_library_loads_and_exports_every_declared_symbol():
    from panogrf_amd._lib import lib, SYMBOLS, library_path
    assert os.path.exists(library_path()), 'build with: make -C panogrf_amd/csrc -j8'
    declared = _header_symbols()
    assert len(declared) >= 18
    dll = C.CDLL(library_path())
    for name in declared:
        assert hasattr(dll, name), f'{name} declared in include/panogrf_hip.h but not exported'
    assert set(declared) == set(SYMBOLS), set(declared) ^ set(SYMBOLS)
    assert lib.pgrf_abi_version() >= 1


def test_argument_validation_returns_errors_not_crashes():
    from panogrf_amd._lib import lib, PgrfError, PgrfFrame, PgrfPassOut
    assert lib.pgrf_sample_coarse_depth(0.5, 15.0, 1, 4, 2, None, None) < 0       # null output
    assert b'null' in lib.pgrf_last_error()
    buf = torch.zeros(16)
    assert lib.pgrf_sample_coarse_depth(0.5, 15.0, 1, 4, 2, C.c_void_p(buf.data_ptr()), None) < 0   # dn <= 2
    assert lib.pgrf_sample_fine_depth(C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), 0.5, 15.0, 1, 0, 1, 500, 8,
                                      C.c_void_p(buf.data_ptr()), None) == -3          # dn > 128 unsupported
    fr = PgrfFrame()
    out = PgrfPassOut()
    rc = lib.pgrf_render_pass(C.byref(fr), C.c_void_p(buf.data_ptr()), 0, C.c_void_p(buf.data_ptr()),
                              C.c_void_p(buf.data_ptr()), 4, 64, C.byref(out), None, None, 0, None)
    assert rc == -1 and b'rfn' in lib.pgrf_last_error()
    with pytest.raises(PgrfError):
        lib.check(rc)
    # empty inputs are a no-op, not an error
    assert lib.pgrf_sample_coarse_depth(0.5, 15.0, 1, 0, 64, C.c_void_p(buf.data_ptr()), None) == 0


def test_product_path_refuses_to_run_without_gpu():
    from panogrf_amd import engine as E
    from panogrf_amd._lib import PgrfError
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(PgrfError):
        E.sample_coarse_depth(0.5, 15.0, 4, 64)
    with pytest.raises(PgrfError):
        E.pack_map(torch.zeros(1, 3, 4, 4))


def test_weight_packing_layout_on_host():
    """pgrf_pack_weights is host code: check the packed A operands of two layers against the
    accumulator-layout rule of pgrf_layout.h using a numpy emulation of the 32x32x2 MFMA."""
    import numpy as np
    from bench import synthetic_weights
    from panogrf_amd import engine as E
    sd = synthetic_weights(3, use_vis=True)
    blob = E.pack_pass_weights(sd, 'dist_decoder', 'agg_net', True).numpy()
    assert blob.shape[0] == E.lib.pgrf_blob_floats() and np.isfinite(blob).all()

    def rho(r, h):
        return (r & 3) + 8 * (r >> 2) + 4 * h

    def emulate(a_off, ks, mt, x_regs):
        """x_regs [ks] pairs (lo, hi) for ONE point; returns acc [mt][16 regs][2 halves] without bias"""
        ks4 = (ks + 3) // 4
        A = blob[a_off:a_off + mt * ks4 * 256].reshape(mt, ks4, 64, 4)
        acc = np.zeros((mt, 32))
        for m in range(mt):
            for s in range(ks):
                a = A[m, s // 4, :, s % 4]                  # lane -> A[i = lane & 31][k = lane >> 5]
                acc[m] += a[:32] * x_regs[s][0] + a[32:] * x_regs[s][1]
        return acc                                           # row index = output row rho

    # layer offsets follow pgrf_layout.h: DD1 first (ks 16, mt 4, bias 128), then 4 x DD2 (ks 16, mt 1, bias 32), then PE1
    off_pe1 = (4 * 4 * 256 + 128) + 4 * (4 * 256 + 32)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(34, generator=g)
    regs = [(float(x[rho(s, 0)]), float(x[rho(s, 1)])) for s in range(16)] + [(float(x[32]), float(x[33]))]
    acc = emulate(off_pe1, 17, 1, regs)[0]
    ref = sd['agg_net.prob_embed.0.weight'].double() @ x.double()
    assert np.allclose(acc, ref.numpy(), atol=1e-5)
    # fused dist-decoder layer 0: tile m = MLP m (mean, var, aw, vis)
    x32 = torch.randn(32, generator=g)
    regs = [(float(x32[rho(s, 0)]), float(x32[rho(s, 1)])) for s in range(16)]
    acc = emulate(0, 16, 4, regs)
    for m, name in enumerate(['mean', 'var', 'aw', 'vis']):
        ref = sd[f'dist_decoder.{name}_decoder.0.weight'].double() @ x32.double()
        assert np.allclose(acc[m], ref.numpy(), atol=1e-5)


def test_conv3d_packing_on_host():
    import numpy as np
    from panogrf_amd._lib import lib
    g = torch.Generator().manual_seed(1)
    cout, cin = 40, 36
    w = torch.randn(cout, cin, 3, 3, 3, generator=g)
    b = torch.randn(cout, generator=g)
    n = lib.pgrf_conv3d_packed_floats(cout, cin)
    host = torch.empty(n)
    assert lib.pgrf_pack_conv3d(C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), cout, cin, C.c_void_p(host.data_ptr())) == 0
    cot, chunks = 2, 2
    A = host[:cot * 27 * chunks * 1024].reshape(cot, 27, chunks, 4, 64, 4).numpy()
    wn = w.reshape(cout, cin, 27).numpy()
    for (ct, tap, ch, s, lane) in [(0, 0, 0, 0, 0), (1, 13, 1, 3, 37), (1, 26, 0, 15, 63), (0, 5, 1, 2, 40)]:
        co, ci = 32 * ct + (lane & 31), 32 * ch + 16 * (lane >> 5) + s
        exp = wn[co, ci, tap] if (co < cout and ci < cin) else 0.0
        assert A[ct, tap, ch, s // 4, lane, s % 4] == np.float32(exp)


def test_winograd_filter_transform_on_host():
    """pgrf_pack_conv3d_wino (host side of csrc/conv3d_wino.hip): unpack U[co][ci][f] from the packed A operands and run
    the whole F(2x2x2, 3x3x3) algorithm in numpy on a random 4x4x4 patch: out = At (U .* (Bt d)) must equal the
    27-tap correlation.  Pins the transform matrices and the (frequency, channel block, lane) layout without a GPU."""
    import numpy as np
    from panogrf_amd._lib import lib
    g = torch.Generator().manual_seed(2)
    cout, cin = 64, 32
    w = torch.randn(cout, cin, 3, 3, 3, generator=g)
    b = torch.randn(cout, generator=g)
    n = lib.pgrf_conv3d_wino_packed_floats(cout, cin)
    assert n == 2 * 64 * 1 * 1024 + 64
    host = torch.empty(n)
    assert lib.pgrf_pack_conv3d_wino(C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), cout, cin, C.c_void_p(host.data_ptr())) == 0
    A = host[:2 * 64 * 1024].reshape(2, 64, 1, 4, 64, 4).numpy()             # [cout tile][f][chunk][s/4][lane][s%4]
    U = np.zeros((cout, cin, 64), np.float64)
    for ct in range(2):
        for lane in range(64):
            for s in range(16):
                U[32 * ct + (lane & 31), 16 * (lane >> 5) + s] = A[ct, :, 0, s // 4, lane, s % 4]
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    d = torch.randn(cin, 4, 4, 4, generator=g).double().numpy()
    V = np.einsum('ai,bj,ck,nijk->nabc', Bt, Bt, Bt, d).reshape(cin, 64)
    M = np.einsum('onf,nf->of', U, V).reshape(cout, 4, 4, 4)
    out = np.einsum('ai,bj,ck,oijk->oabc', At, At, At, M)                    # [cout, 2, 2, 2]
    ref = np.zeros((cout, 2, 2, 2))
    wn = w.double().numpy()
    for oz in range(2):
        for oy in range(2):
            for ox in range(2):
                ref[:, oz, oy, ox] = np.einsum('onijk,nijk->o', wn, d[:, oz:oz + 3, oy:oy + 3, ox:ox + 3])
    assert np.abs(out - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    # bias block in accumulator order (register r of lane half h = output row (r&3) + 8*(r>>2) + 4*h)
    B = host[2 * 64 * 1024:].numpy()
    for (ct, h, r) in [(0, 0, 0), (1, 1, 5), (0, 1, 15)]:
        assert B[ct * 32 + h * 16 + r] == b[32 * ct + (r & 3) + 8 * (r >> 2) + 4 * h]
    assert lib.pgrf_conv3d_wino_supported(64, 64, 4, 4, 64) == 1 and lib.pgrf_conv3d_wino_supported(64, 64, 4, 4, 16) == 0


def test_tile_sharding_math():
    from panogrf_amd import dist as D
    for H in [512, 13, 7]:
        for world in [1, 2, 3, 8]:
            rows = [D.tile_rows(r, world, H) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == H
            assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rows]
            assert max(sizes) - min(sizes) <= 1
    c = D.tile_coords(1, 2, 4, 3)
    assert c.shape == (6, 2) and c[0].tolist() == [0.0, 2.0] and c[-1].tolist() == [2.0, 3.0]


def test_host_mirror_of_small_pose_tensors():
    """imgs_info_to_torch keeps the host values of poses / ranges next to their device copies, and engine.host_floats uses
    them only while the device tensor's version counter stands (CPU: the 'device' is torch's meta device, which cannot be
    read back -- a fallback to the device->host copy shows up as an error)."""
    import numpy as np
    from panogrf_amd.imgs_info import imgs_info_to_torch, build_render_imgs_info
    from panogrf_amd import engine as E
    pose = np.concatenate([np.eye(3), np.array([[0.25], [-0.5], [2.0]])], 1)
    info = build_render_imgs_info(pose, (8, 16), [0.5, 15.0])
    cpu = imgs_info_to_torch(info, 'cpu')
    assert not hasattr(cpu['c2w'], '_pgrf_host')                       # nothing moved, nothing mirrored
    want = [float(x) for x in cpu['c2w'].float().reshape(-1)[:12]]
    assert E.host_floats(cpu['c2w'], 12) == want
    dev = imgs_info_to_torch(info, 'meta')
    assert hasattr(dev['c2w'], '_pgrf_host') and hasattr(dev['depth_range'], '_pgrf_host')
    assert not hasattr(dev['coords'], '_pgrf_host')                    # large tensors carry no mirror
    assert E.host_floats(dev['c2w'], 12) == want
    assert E.host_floats(dev['depth_range'], 2) == [0.5, 15.0]
    shallow = dict(dev)                                                # renderer.forward copies the dicts, not the tensors
    assert E.host_floats(shallow['c2w'], 12) == want
    dev['c2w'].mul_(2.0)                                               # in-place write: the mirror is stale -> read back
    with pytest.raises((NotImplementedError, RuntimeError)):
        E.host_floats(dev['c2w'], 12)


def test_product_library_is_not_an_experiment_build():
    """pgrf_abi_version(): the header's PGRF_ABI_VERSION, positive (negative = linked from ablation / timing objects,
    csrc/pgrf_experiments.h) -- and a kernel source given an experiment switch without -DPGRF_ABLATION_BUILD does not compile,
    so the product Makefile can never ship one."""
    import subprocess
    from panogrf_amd._lib import lib, ABI_VERSION
    hdr = open(os.path.join(ROOT, 'include', 'panogrf_hip.h')).read()
    assert int(re.search(r'#define PGRF_ABI_VERSION (\d+)', hdr).group(1)) == ABI_VERSION
    assert lib.pgrf_abi_version() == ABI_VERSION
    src = os.path.join(ROOT, 'panogrf_amd', 'csrc')
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-std=c++17', '-fsyntax-only', '--cuda-host-only',
                        '-I' + os.path.join(ROOT, 'include'), '-DPGRF_NO_BARRIER', '-DPGRF_KT=2', '-DPGRF_V=2', '-DPGRF_DBG=0',
                        '-DPGRF_PREC=0', os.path.join(src, 'render_pass_inst.hip')], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'PGRF_ABLATION_BUILD' in r.stderr
    guarded = 0
    for f in os.listdir(src):
        if f.endswith('.hip') and f not in ('capi.hip', 'render_pass_inst.hip'):
            assert '#include "pgrf_experiments.h"' in open(os.path.join(src, f)).read(), f
            guarded += 1
    assert guarded >= 5 and '#include "pgrf_experiments.h"' in open(os.path.join(src, 'render_pass.h')).read()


def test_plain_bench_gpus_n_becomes_a_launcher():
    """`python bench.py --gpus 2` without torchrun must start child ranks (here they stop at 'needs a ROCm GPU'), not refuse
    with a launch hint; on a GPU box tests/test_gpu_bench_multirank.py runs the same command to completion."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip('GPU present: covered by the -m gpu test')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'needs a ROCm GPU' in r.stderr and 'launch N>1 as' not in r.stderr
