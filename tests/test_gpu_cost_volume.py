"""HIP spherical cost volume / hypothesis generation vs golden vectors from the reference. -m gpu"""
import pytest
import torch

from tests import cv_fixtures as CF
from tests.fixtures import max_abs

pytestmark = pytest.mark.gpu
ARGS = dict(dataset_name='m3d', contain_dnet=True, mono_uncertainty=False)


def test_depth_hypotheses_bit_exact():
    from panogrf_amd import cost_volume as CV
    fx = CF.load()['hyp']
    k = CV.depth_sampling(5, 3)
    out = CV.build_depth_volume(fx['ref_mu'].cuda(), k, 0.5, 0.1, 10, 64)
    assert torch.equal(out.cpu(), fx['out'])          # clamp/sort/merge: no arithmetic beyond one add


def test_cost_volume_two_view():
    from panogrf_amd import cost_volume as CV
    fx = CF.load()['cv2']
    out = CV.calculate_cost_volume_erp(ARGS, fx['feats'].cuda(), None, fx['trans'].cuda(), fx['rots'].cuda(),
                                       depth_volume=fx['depth_volume'].cuda())
    assert out.shape == fx['out'].shape
    assert max_abs(out.cpu(), fx["out"]) < 2e-4   # white-noise features: |dfeat| ~ 2 x |du| (px), |du| <= 5e-5
    outd = CV.calculate_cost_volume_erp(ARGS, fx['feats'].cuda(), None, fx['trans'].cuda(), fx['rots'].cuda(),
                                        depth_volume=fx['depth_volume'][:, :4].contiguous().cuda(), cost_type='dot')
    assert max_abs(outd.cpu(), fx['out_dot']) < 2e-4


def test_cost_volume_multiview():
    from panogrf_amd import cost_volume as CV
    fx = CF.load()['cvm']
    out = CV.calculate_cost_volume_erp_multiview(ARGS, fx['feats'].cuda(), None, fx['trans'].cuda(), fx['rots'].cuda(),
                                                 depth_volume=fx['depth_volume'].cuda(), curr_idx=0)
    assert max_abs(out.cpu(), fx["out"]) < 2e-4   # white-noise features: |dfeat| ~ 2 x |du| (px), |du| <= 5e-5
    out1 = CV.calculate_cost_volume_erp_multiview(ARGS, fx['feats'].cuda(), None, fx['trans'].cuda(), fx['rots'].cuda(),
                                                  depth_volume=fx['depth_volume'][:, :2].contiguous().cuda(), curr_idx=1)
    assert max_abs(out1.cpu(), fx['out_idx1']) < 2e-4


def test_cost_volume_full_size_vs_oracle_planes():
    """BASELINE config 3 size (D=64, C=32, 128x256): identity-pose property on the whole volume +
    oracle parity on a few planes."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_volume as CV
    g = torch.Generator().manual_seed(7)
    B, h, w, Cc, D = 1, 128, 256, 32, 64
    feats = torch.randn(B, 2, h, w, Cc, generator=g)
    ref_mu = 1.0 + 7.0 * torch.rand(B, 1, h, w, generator=g)
    dv = CV.build_depth_volume(ref_mu.cuda(), CV.depth_sampling(5, 3), 0.5, 0.1, 10, 64)
    assert bool((dv[:, 1:] >= dv[:, :-1]).all())
    eye = torch.eye(3).view(1, 1, 3, 3).repeat(B, 2, 1, 1)
    trans = torch.tensor([[[0.5, 0.0, 0.0], [-0.5, 0.0, 0.0]]])
    # size-independent property: the warp (cost_type 'none') is linear in the source features
    f2 = torch.randn(B, 2, h, w, Cc, generator=g)
    wa = CV.calculate_cost_volume_erp(ARGS, feats.cuda(), None, trans.cuda(), eye.cuda(), depth_volume=dv, cost_type='none')
    wb = CV.calculate_cost_volume_erp(ARGS, f2.cuda(), None, trans.cuda(), eye.cuda(), depth_volume=dv, cost_type='none')
    wab = CV.calculate_cost_volume_erp(ARGS, (2.0 * feats - 0.5 * f2).cuda(), None, trans.cuda(), eye.cuda(),
                                       depth_volume=dv, cost_type='none')
    assert float((wab - (2.0 * wa - 0.5 * wb)).abs().max()) < 2e-5
    del wa, wb, wab
    cv = CV.calculate_cost_volume_erp(ARGS, feats.cuda(), None, trans.cuda(), eye.cuda(), depth_volume=dv)
    assert cv.shape == (B, D, h, w, Cc) and not torch.isnan(cv).any()
    sel = [0, 31, 63]
    ref = O.cost_volume_erp(feats, dv[:, sel].cpu(), eye, trans)
    err = (cv[:, sel].cpu() - ref).abs()
    # warp coordinates carry ~1e-4 px of float32 noise at w=256; features are white noise (gradient ~2/px)
    assert float(err.max()) < 5e-3 and float(err.mean()) < 2e-5


@pytest.mark.parametrize('ds', ['m3d', 'replica_test', 'residential', 'CoffeeArea'])
def test_cost_volume_dataset_conventions_vs_reference(ds):
    """args['dataset_name'] selects the spherical convention; expected volumes come from the reference itself
    (tests/golden/cv_conventions.npz)."""
    import os
    import numpy as np
    from panogrf_amd import cost_volume as CV
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cv_conventions.npz'))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    args = dict(dataset_name=ds, contain_dnet=True)
    two = CV.calculate_cost_volume_erp(args, g['feats'][:, :2].cuda(), None, g['trans'][:, :2].cuda(), g['rots'][:, :2].cuda(),
                                       depth_volume=g['depth_volume'].cuda())
    mv = CV.calculate_cost_volume_erp_multiview(args, g['feats'].cuda(), None, g['trans'].cuda(), g['rots'].cuda(),
                                                depth_volume=g['depth_volume'].cuda(), curr_idx=1)
    # 8x16 maps of N(0,1) noise: one float ulp of an angle moves a tap weight by ~1e-6 of a unit-variance difference
    assert float((two.cpu() - g[f'{ds}.two']).abs().max()) < 2e-4
    assert float((mv.cpu() - g[f'{ds}.mv']).abs().max()) < 2e-4


def _cv_env(**kw):
    import os
    keys = ('PGRF_CV_PIPE', 'PGRF_CV_TILE_W', 'PGRF_CV_DCH', 'PGRF_CV_ORDER', 'PGRF_CV_STREAM_STORES')
    old = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)
    return old


def _cv_env_restore(old):
    import os
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize('h,w,B,S', [(8, 16, 2, 2), (16, 24, 1, 2), (32, 64, 2, 2), (16, 32, 1, 4), (24, 40, 1, 2)])
def test_cost_volume_pipelined_instance_is_bit_identical_to_the_single_stage_kernel(h, w, B, S):
    """The pipelined kernel (groups of voxels in flight across group and plane boundaries, 2-D pixel patches, the geometry of plane
    d + 1 computed while plane d streams) performs the single-stage kernel's operations on the same values: equal bits for every
    patch width, pipeline depth, planes-per-wave split (incl. ragged last chunks), work-item order and store flavour; both equal
    the oracle.  Poses are random rotations + translations; D = 13 leaves ragged chunks."""
    from oracle import cost_volume_oracle as O
    from panogrf_amd import cost_volume as CV
    g = torch.Generator().manual_seed(h * 131 + w)
    D, Cc = 13, 32
    feats = torch.randn(B, S, h, w, Cc, generator=g)
    dv = torch.sort(0.3 + 9.0 * torch.rand(B, D, h, w, generator=g), dim=1)[0]
    q = torch.linalg.qr(torch.randn(B, S, 3, 3, generator=g))[0]
    rots = q * torch.sign(torch.linalg.det(q)).view(B, S, 1, 1)
    trans = 0.4 * torch.randn(B, S, 3, generator=g)

    def run():
        if S == 2:
            return CV.calculate_cost_volume_erp(ARGS, feats.cuda(), None, trans.cuda(), rots.cuda(), depth_volume=dv.cuda())
        return CV.calculate_cost_volume_erp_multiview(ARGS, feats.cuda(), None, trans.cuda(), rots.cuda(), depth_volume=dv.cuda(), curr_idx=1)
    old = _cv_env(PGRF_CV_PIPE=0, PGRF_CV_TILE_W=64)
    try:
        base = run().cpu()
        ref = O.cost_volume_erp(feats, dv, rots, trans) if S == 2 else O.cost_volume_erp_multiview(feats, dv, rots, trans, 1)
        err = (base - ref).abs()       # (a voxel within an ulp of the wrap seam may pick the other edge texel: a handful at most)
        assert float(err.mean()) < 2e-5 and float((err > 5e-3).float().mean()) < 1e-5
        for pipe in (0, 1, 2, 4, 8):
            for tw in (64, 32, 16, 8):
                for dch, order, nt in ((1, 0, 0), (3, 1, 0), (4, 0, 1), (64, 1, 1)):
                    _cv_env(PGRF_CV_PIPE=pipe, PGRF_CV_TILE_W=tw, PGRF_CV_DCH=dch, PGRF_CV_ORDER=order, PGRF_CV_STREAM_STORES=nt)
                    got = run().cpu()
                    assert torch.equal(got, base), (pipe, tw, dch, order, nt)
        _cv_env()                # the shipped defaults
        assert torch.equal(run().cpu(), base)
    finally:
        _cv_env_restore(old)
