"""Host side of the 16-row narrow-layer kernels (pgrf_pack_conv3d_small16, pgrf_pack_conv3d_upbrick; csrc/conv3d.hip): the packed A
operands, unpacked from the v_mfma_f32_16x16x4_f32 lane layout and applied with plain torch indexing exactly the way the kernels walk
them, must reproduce the oracle's convolution (models/cost_reg.py:99-175 / oracle.cost_volume_oracle).  For the up-sampling-fused
form that includes the seam columns under the INTERIOR x weights with the corrected operand the kernel forms from D = (t[W-1] - t[0]) / 2.
CPU only (no kernel launch)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from oracle import cost_volume_oracle as O


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _unpack_small16(packed, Cout, Cin, ns):
    """-> A[cout group][row 16][plane tap (ns + 2) * 9][Cin] and the bias by row, following the kernels' reduction order:
    Cin >= 16: group G = chunk * ntap + tap holds channels 16 chunk + 4 g + m in lane (row, g), element m; Cin = 8: k = tap * 8 + c."""
    ntap = (ns + 2) * 9
    groups = ((ntap * Cin + 15) // 16 + 1) & ~1
    ncg = Cout // 16 if ns == 1 else 1
    blk = packed[:ncg * groups * 256].view(ncg, groups, 64, 4)
    A = torch.zeros(ncg, 16, ntap, Cin)
    for G in range(groups):
        for g in range(4):
            for m in range(4):
                k = G * 16 + 4 * g + m
                if Cin >= 16:
                    if G >= ntap * (Cin // 16):
                        assert float(blk[:, G].abs().max()) == 0.0          # padding of the last pair of groups
                        continue
                    tap, ci = G % ntap, 16 * (G // ntap) + 4 * g + m
                else:
                    tap, ci = k // Cin, k % Cin
                    if tap >= ntap:
                        continue
                A[:, :, tap, ci] = blk[:, G, g * 16:(g + 1) * 16, m]
    return A, packed[ncg * groups * 256:]


@pytest.mark.parametrize('Cin,Cout,ns', [(32, 8, 2), (16, 8, 2), (8, 8, 2), (16, 16, 1), (8, 16, 1), (64, 32, 1), (48, 16, 1)])
def test_small16_operands_reproduce_the_convolution(Cin, Cout, ns):
    from panogrf_amd._lib import lib
    g = torch.Generator().manual_seed(Cin * 10 + Cout + ns)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(Cout, generator=g) * 0.1
    n = lib.pgrf_conv3d_small16_packed_floats(Cout, Cin, ns)
    assert n > 0
    packed = torch.full((n,), float('nan'))
    lib.check(lib.pgrf_pack_conv3d_small16(_ptr(w.contiguous()), _ptr(b), Cout, Cin, ns, _ptr(packed)))
    assert bool(torch.isfinite(packed).all())
    A, bias = _unpack_small16(packed, Cout, Cin, ns)
    D, H, W = 4, 3, 5
    x = torch.randn(1, Cin, D, H, W, generator=g)
    ref = O._conv3({'c.weight': w, 'c.bias': b}, 'c', x)
    out = torch.zeros_like(ref)
    for z0 in range(0, D, ns):                               # a row = (depth shift, channel) for ns = 2, a channel for ns = 1
        for y in range(H):
            for xx in range(W):
                patch = torch.zeros((ns + 2) * 9, Cin)        # plane taps (pz, dy, dx) around output plane z0
                for tap in range((ns + 2) * 9):
                    zz, yy, xc = z0 + tap // 9 - 1, y + (tap // 3) % 3 - 1, (xx + tap % 3 - 1) % W
                    if 0 <= zz < D and 0 <= yy < H:
                        patch[tap] = x[0, :, zz, yy, xc]
                rows = (A * patch).sum((2, 3))                # [cout group][16]
                for cg in range(A.shape[0]):
                    for r in range(16):
                        delta, co = (r // 8, r % 8) if ns == 2 else (0, cg * 16 + r)
                        out[0, co, z0 + delta, y, xx] = rows[cg, r] + bias[r if ns == 2 else cg * 16 + r]
    assert float((out - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_small16_packer_rejects_what_the_kernels_cannot_do():
    from panogrf_amd._lib import lib
    f = lib.pgrf_conv3d_small16_packed_floats
    assert f(8, 32, 2) == 72 * 256 + 16 and f(16, 8, 1) == 14 * 256 + 16 and f(32, 64, 1) == 2 * 108 * 256 + 32
    assert f(8, 64, 2) == 0 and f(8, 32, 1) == 0 and f(24, 32, 1) == 0 and f(32, 8, 1) == 0 and f(16, 24, 1) == 0
    w = torch.zeros(16, 16, 27)
    assert lib.pgrf_pack_conv3d_small16(_ptr(w), None, 16, 16, 2, _ptr(torch.zeros(64))) != 0
    assert b'pgrf_pack_conv3d_small16' in lib.pgrf_last_error()


@pytest.mark.parametrize('D,H,W', [(3, 4, 33), (2, 2, 40), (4, 3, 35)])
def test_upbrick_operands_and_seam_correction_reproduce_upsample_then_conv(D, H, W):
    """Every low-res voxel: rows (z parity, channel) x four (y, x) parity sets from the class block of its (z, y) position, x always
    interior; the two seam columns with the corrected neighbour values of conv3d_upbrick_kernel."""
    from panogrf_amd._lib import lib
    Cin, Cout = 16, 8
    g = torch.Generator().manual_seed(D * 100 + H * 10 + W)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(Cout, generator=g) * 0.1
    n = lib.pgrf_conv3d_upbrick_packed_floats(Cout, Cin)
    assert n == 9 * 4 * 27 * 256 + 16
    packed = torch.full((n,), float('nan'))
    lib.check(lib.pgrf_pack_conv3d_upbrick(_ptr(w.contiguous()), _ptr(b), Cout, Cin, D, H, W, _ptr(packed)))
    blk = packed[:9 * 4 * 27 * 256].view(9, 4, 27, 64, 4)
    A = torch.zeros(9, 4, 16, 27, Cin)                       # [class][set][row][tap][ci]
    for gq in range(4):
        for m in range(4):
            A[:, :, :, :, 4 * gq + m] = blk[:, :, :, gq * 16:(gq + 1) * 16, m].permute(0, 1, 3, 2)
    bias = packed[9 * 4 * 27 * 256:]
    assert torch.equal(bias, torch.cat([b, b]))
    t = torch.randn(1, Cin, D, H, W, generator=g)
    ref = O._conv3({'c.weight': w, 'c.bias': b}, 'c', F.interpolate(t, scale_factor=2, mode='trilinear', align_corners=False))
    out = torch.zeros_like(ref)
    cls_of = lambda i, n_: 0 if i == 0 else (2 if i == n_ - 1 else 1)
    # coefficient of D per (x parity, nx): the kernel's per-lane table
    k_left = [[1.0, -1.0, 3.0], [-2.0, 0.0, 0.0]]
    k_right = [[0.0, 0.0, 2.0], [-3.0, 1.0, -1.0]]
    for z in range(D):
        for y in range(H):
            c = cls_of(z, D) * 3 + cls_of(y, H)
            for x in range(W):
                for st in range(4):
                    px = st & 1
                    patch = torch.zeros(27, Cin)
                    for nb in range(27):
                        zz, yy, nx = z + nb // 9 - 1, y + (nb // 3) % 3 - 1, nb % 3
                        if not (0 <= zz < D and 0 <= yy < H):
                            continue
                        v = t[0, :, zz, yy, (x + nx - 1) % W]
                        dd = 0.5 * (t[0, :, zz, yy, W - 1] - t[0, :, zz, yy, 0])
                        if x == 0:
                            v = v + k_left[px][nx] * dd
                        elif x == W - 1:
                            v = v + k_right[px][nx] * dd
                        patch[nb] = v
                    rows = (A[c, st] * patch).sum((1, 2)) + bias          # [16] = (z parity, channel)
                    for r in range(16):
                        out[0, r & 7, 2 * z + (r >> 3), 2 * y + (st >> 1), 2 * x + px] = rows[r]
    assert float((out - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_upbrick_packer_rejects_what_the_kernel_cannot_do():
    from panogrf_amd._lib import lib
    assert lib.pgrf_conv3d_upbrick_packed_floats(16, 16) == 0 and lib.pgrf_conv3d_upbrick_packed_floats(8, 32) == 0
    w = torch.zeros(8, 16, 27)
    out = torch.zeros(lib.pgrf_conv3d_upbrick_packed_floats(8, 16))
    assert lib.pgrf_pack_conv3d_upbrick(_ptr(w), None, 8, 16, 4, 4, 32, _ptr(out)) != 0          # W < 33
    assert lib.pgrf_pack_conv3d_upbrick(_ptr(w), None, 8, 16, 1, 4, 40, _ptr(out)) != 0          # an axis of length 1
    assert b'pgrf_pack_conv3d_upbrick' in lib.pgrf_last_error()
