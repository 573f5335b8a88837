"""Pin oracle/cost_volume_oracle.py against golden vectors captured from the reference. CPU only."""
import numpy as np
import torch

from oracle import cost_volume_oracle as CV
from tests import cv_fixtures as CF
from tests.fixtures import max_abs


def test_k_list_and_hypotheses():
    fx = CF.load()['hyp']
    k = CV.gaussian_k_list(5, 3)
    assert np.allclose(k, fx['k_list'], rtol=0, atol=1e-12)
    out = CV.depth_hypotheses(fx['ref_mu'], k, 0.5, 0.1, 10, 64)
    assert torch.equal(out, fx['out'])


def test_cost_volume_two_view():
    torch.set_num_threads(1)
    fx = CF.load()['cv2']
    out = CV.cost_volume_erp(fx['feats'], fx['depth_volume'], fx['rots'], fx['trans'])
    assert max_abs(out, fx['out']) < 1e-5
    outd = CV.cost_volume_erp(fx['feats'], fx['depth_volume'][:, :4], fx['rots'], fx['trans'], 'dot')
    assert max_abs(outd, fx['out_dot']) < 1e-5


def test_cost_volume_multiview_skips_ref_and_last():
    torch.set_num_threads(1)
    fx = CF.load()['cvm']
    out = CV.cost_volume_erp_multiview(fx['feats'], fx['depth_volume'], fx['rots'], fx['trans'], 0)
    assert max_abs(out, fx['out']) < 1e-5
    out1 = CV.cost_volume_erp_multiview(fx['feats'], fx['depth_volume'][:, :2], fx['rots'], fx['trans'], 1)
    assert max_abs(out1, fx['out_idx1']) < 1e-5


def test_cost_reg_net():
    torch.set_num_threads(1)
    fx = CF.load()['crn']
    w = CF.seeded_params(CF.key_shapes(fx['keys']), 41)
    out = CV.cost_reg_net(w, fx['x'])
    assert max_abs(out, fx['out']) < 2e-5


def test_unet2():
    torch.set_num_threads(1)
    fx = CF.load()['unet']
    w = CF.seeded_params(CF.key_shapes(fx['keys']), 51)
    out = CV.unet2(w, fx['x'])
    assert max_abs(out, fx['out']) < 2e-5


# ---- the other datasets' spherical conventions, pinned to the reference's own cost volumes
# (tests/golden/make_geometry_fixtures.py --cv -> cv_conventions.npz)
import os as _os

import numpy as _np
import pytest as _pytest
import torch as _torch


def _cvc():
    z = _np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'golden', 'cv_conventions.npz'))
    return {k: _torch.from_numpy(z[k]) for k in z.files}


@_pytest.mark.parametrize('ds', ['m3d', 'replica_test', 'residential', 'CoffeeArea'])
def test_oracle_cost_volume_conventions_vs_reference(ds):
    from oracle import cost_volume_oracle as O
    g = _cvc()
    two = O.cost_volume_erp(g['feats'][:, :2], g['depth_volume'], g['rots'][:, :2], g['trans'][:, :2], dataset=ds)
    mv = O.cost_volume_erp_multiview(g['feats'], g['depth_volume'], g['rots'], g['trans'], curr_idx=1, dataset=ds)
    assert float((two - g[f'{ds}.two']).abs().max()) < 1e-5
    assert float((mv - g[f'{ds}.mv']).abs().max()) < 1e-5
