"""ERP -> cube-map conversion of the mono-depth net's input (reference:
UniFuse-Unidirectional-Fusion/UniFuse/datasets/util.py:7-111 called from network/omni_mvsnet/pipeline3_model.py:266-282).
Fixtures: tests/golden/e2c_fixtures.npz, outputs of the reference class (tests/golden/make_e2c_fixtures.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import e2c_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'e2c_fixtures.npz')
CASES = 'abc'          # c: odd face width, the only case that reaches the pole rows and the legacy-wrap seam


def _case(z, t):
    img, dep = z[t + '.img'], z[t + '.dep']
    return img, dep, img.shape[0] // 2


def test_oracle_matches_the_reference_class_bit_for_bit():
    z = np.load(GOLD)
    for t in CASES:
        img, dep, fw = _case(z, t)
        cy, cx = O.erp_coordinates(img.shape[0], img.shape[1], fw)
        assert np.array_equal(cy, z[t + '.coor_y']) and np.array_equal(cx, z[t + '.coor_x'])
        cube, cdep = O.equirec_to_cube(img, fw, dep)
        assert cube.dtype == np.float32 and np.array_equal(cube, z[t + '.cube'])
        assert np.array_equal(cdep, z[t + '.cube_dep'])
    assert float(z['c.coor_y'].min()) < 0 and float(z['c.coor_x'].max()) > z['c.img'].shape[1] - 1      # wrap exercised


def test_oracle_sampler_is_scipy_legacy_wrap():
    """The third-party sampler (scipy.ndimage.map_coordinates, mode='wrap': period len-1) restated in the oracle."""
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(0)
    im = rng.standard_normal((13, 17)).astype(np.float32)
    cy = rng.uniform(-14, 28, 4000).astype(np.float32)
    cx = rng.uniform(-20, 40, 4000).astype(np.float32)
    for order in (0, 1):
        assert np.array_equal(map_coordinates(im, [cy, cx], order=order, mode='wrap'), O.sample_wrap(im, cy, cx, order))


def test_host_tables_match_the_reference():
    from panogrf_amd.cubemap import _tables
    z = np.load(GOLD)
    for t in CASES:
        img, _, fw = _case(z, t)
        cy, cx, cos = _tables(img.shape[0], img.shape[1], fw)
        assert np.array_equal(cy, z[t + '.coor_y']) and np.array_equal(cx, z[t + '.coor_x']) and np.array_equal(cos, z[t + '.cosmap'])


@pytest.mark.gpu
def test_hip_cube_maps_vs_reference():
    """Same floats as the reference's scipy path: linear for images, nearest * cos map for depth."""
    from panogrf_amd.cubemap import Equirec2Cube
    z = np.load(GOLD)
    for t in CASES:
        img, dep, fw = _case(z, t)
        e2c = Equirec2Cube(img.shape[0], img.shape[1], fw)
        cube, cdep = e2c.run(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda())
        assert cube.shape == z[t + '.cube'].shape
        assert np.array_equal(cube.cpu().numpy(), z[t + '.cube']), float(np.abs(cube.cpu().numpy() - z[t + '.cube']).max())
        assert np.array_equal(cdep.cpu().numpy(), z[t + '.cube_dep'])
        assert np.array_equal(e2c.run(torch.from_numpy(img).cuda()).cpu().numpy(), z[t + '.cube'])


@pytest.mark.gpu
def test_hip_cube_maps_batched_channels_first_normalised():
    """[B,S,H,W,3] batches (e2c_process) and the fused NCHW + ImageNet-normalised form the MVS stage feeds the mono net."""
    from panogrf_amd.cubemap import Equirec2Cube, e2c_process
    g = torch.Generator().manual_seed(2)
    H, W, fw = 64, 128, 32
    panos = torch.rand(2, 3, H, W, 3, generator=g)
    e2c = Equirec2Cube(H, W, fw)
    got = e2c_process(panos.cuda(), e2c)
    assert got.shape == (2, 3, fw, 6 * fw, 3)
    for b in range(2):
        for s in range(3):
            assert np.array_equal(got[b, s].cpu().numpy(), O.equirec_to_cube(panos[b, s].numpy(), fw))
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    x = panos[0].permute(0, 3, 1, 2).contiguous().cuda()
    cf = e2c.to_cube(x, channels_first=True, mean=mean, std=std)
    want = (got[0].permute(0, 3, 1, 2).cpu() - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    assert cf.shape == (3, 3, fw, 6 * fw) and torch.equal(cf.cpu(), want)
    # a strided (non-contiguous) view is read in place
    xv = panos[1].cuda().permute(0, 3, 1, 2)
    assert torch.equal(e2c.to_cube(xv, channels_first=True).cpu(), got[1].permute(0, 3, 1, 2).cpu())
    with pytest.raises(Exception, match='built for'):
        e2c.to_cube(torch.zeros(1, 32, 64, 3).cuda())


@pytest.mark.gpu
def test_mvs_stage_hands_cube_maps_to_the_mono_net():
    """`d_net(erp, cube, dnet=True)` (pipeline3_model.py:657): the second argument is the normalised cube map of the same
    views at face width mono_height // 2."""
    import torch.nn as nn
    from panogrf_amd.mvs import MVSDepthNet
    from tests.mvs_fixtures import MVS_ARGS, StubMono
    seen = {}

    class Rec(StubMono):
        def forward(self, erp, cube=None, dnet=True):
            seen['erp'], seen['cube'] = erp, cube
            return super().forward(erp, cube, dnet)
    H, W = 32, 64
    args = dict(MVS_ARGS, height=H, width=W, mono_height=H, mono_width=W, depth_height=H, depth_width=W)
    net = MVSDepthNet(args, Rec()).cuda().eval()
    g = torch.Generator().manual_seed(4)
    panos = torch.rand(1, 2, H, W, 3, generator=g).cuda()
    rots = torch.eye(3).expand(1, 2, 3, 3).contiguous().cuda()
    trans = torch.tensor([[[0.3, 0.0, 0.0], [-0.3, 0.0, 0.0]]]).cuda()
    with torch.no_grad():
        net.estimate_depth_using_cost_volume(panos, rots, trans, 0.1, 10.0)
    assert seen['cube'].shape == (2, 3, H // 2, 6 * (H // 2))
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    for i, v in enumerate((1, 0)):                  # reference view first (pipeline3_model.py:655-656)
        want = (torch.from_numpy(O.equirec_to_cube(panos[0, v].cpu().numpy(), H // 2)).permute(2, 0, 1) - mean) / std
        assert torch.equal(seen['cube'][i].cpu(), want)
