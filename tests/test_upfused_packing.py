"""Host side of the up-sampling-fused convolution (pgrf_pack_conv3d_upfused, csrc/conv3d.hip): the interpolation-folded kernels
of the 27 position classes, unpacked from the MFMA operand layout and applied with plain torch indexing, must reproduce
conv(interpolate(t, scale_factor=2, mode='trilinear')) of the oracle (models/cost_reg.py:58-94 / oracle.cost_volume_oracle) on
every output voxel -- faces, edges, corners, the circular W seam, axes of length 2 included.  CPU only (no kernel launch)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from oracle import cost_volume_oracle as O


def _unpack(packed, Cout, Cin):
    steps, MT = ((27 * Cin + 31) // 32 + 1) // 2 * 2, Cout // 4
    blk = packed[:27 * MT * steps * 1024].view(27, MT, steps, 4, 64, 4)
    K = torch.zeros(27, MT * 32, 27, Cin)
    for st in range(steps):
        for s in range(16):
            for kk in range(2):
                k = st * 32 + kk * 16 + s
                tap, ci = k // Cin, k % Cin
                if tap >= 27:
                    continue
                K[:, :, tap, ci] = blk[:, :, st, s >> 2, kk * 32:(kk + 1) * 32, s & 3].reshape(27, MT * 32)
    return K, packed[27 * MT * steps * 1024:]


@pytest.mark.parametrize('Cin,Cout,D,H,W', [(16, 8, 4, 5, 6), (32, 16, 3, 4, 5), (16, 8, 2, 3, 2), (16, 16, 5, 2, 3)])
def test_folded_kernels_reproduce_upsample_then_conv(Cin, Cout, D, H, W):
    from panogrf_amd._lib import lib
    g = torch.Generator().manual_seed(Cin * 100 + Cout + D + H + W)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(Cout, generator=g) * 0.1
    n = lib.pgrf_conv3d_upfused_packed_floats(Cout, Cin)
    assert n == 27 * (Cout // 4) * (((27 * Cin + 31) // 32 + 1) // 2 * 2) * 1024 + Cout
    packed = torch.zeros(n)
    lib.check(lib.pgrf_pack_conv3d_upfused(C.c_void_p(w.contiguous().data_ptr()), C.c_void_p(b.data_ptr()), Cout, Cin, D, H, W,
                                           C.c_void_p(packed.data_ptr())))
    K, bias = _unpack(packed, Cout, Cin)
    assert torch.equal(bias, b)
    t = torch.randn(1, Cin, D, H, W, generator=g)
    ref = O._conv3({'c.weight': w, 'c.bias': b}, 'c', F.interpolate(t, scale_factor=2, mode='trilinear', align_corners=False))
    out = torch.zeros(1, Cout, 2 * D, 2 * H, 2 * W)
    cls_of = lambda i, n_: 0 if i == 0 else (2 if i == n_ - 1 else 1)
    for z in range(D):
        for y in range(H):
            for x in range(W):
                c = (cls_of(z, D) * 3 + cls_of(y, H)) * 3 + cls_of(x, W)
                patch = torch.zeros(27, Cin)
                for nb in range(27):
                    zz, yy, xx = z + nb // 9 - 1, y + (nb // 3) % 3 - 1, (x + nb % 3 - 1) % W
                    if 0 <= zz < D and 0 <= yy < H:
                        patch[nb] = t[0, :, zz, yy, xx]
                rows = (K[c, :8 * Cout] * patch.unsqueeze(0)).sum((1, 2))            # [8 * Cout]: parity-major
                for par in range(8):
                    out[0, :, 2 * z + (par >> 2), 2 * y + ((par >> 1) & 1), 2 * x + (par & 1)] = rows[par * Cout:(par + 1) * Cout] + b
    assert float((out - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_upfused_packer_rejects_what_the_kernel_cannot_do():
    from panogrf_amd._lib import lib
    assert lib.pgrf_conv3d_upfused_packed_floats(32, 64) == 0 and lib.pgrf_conv3d_upfused_packed_floats(8, 8) == 0
    w = torch.zeros(8, 16, 27)
    out = torch.zeros(lib.pgrf_conv3d_upfused_packed_floats(8, 16))
    assert lib.pgrf_pack_conv3d_upfused(C.c_void_p(w.data_ptr()), None, 8, 16, 1, 4, 4, C.c_void_p(out.data_ptr())) != 0   # an axis of length 1
    assert b'pgrf_pack_conv3d_upfused' in lib.pgrf_last_error()
