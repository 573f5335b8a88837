"""Image metrics: panogrf_amd.metrics and the numpy oracle against the reference's own outputs
(tests/golden/metrics_fixtures.npz, made by tests/golden/make_metrics_fixtures.py).  CPU only."""
import os

import numpy as np
import torch

from oracle import metrics_oracle as O
from panogrf_amd import metrics as M

FX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metrics_fixtures.npz'))


def test_uint8_mapping_and_psnr_vs_reference():
    for tag in ('a', 'b'):
        ref, pred = FX[f'{tag}.ref'], FX[f'{tag}.pred']
        for x, want in ((ref, FX[f'{tag}.ref8']), (pred, FX[f'{tag}.pred8'])):
            assert np.array_equal(M.color_map_backward(x), want) and np.array_equal(O.to_uint8(x), want)
        assert want.min() == 0 and want.max() == 255                       # the clip was exercised
        got = M.compute_psnr(FX[f'{tag}.ref8'], FX[f'{tag}.pred8'])
        assert abs(got - float(FX[f'{tag}.psnr8'])) < 1e-5
        # the reference's vis-score branch has no effect on its result (metrics.py:18-28): same number
        assert float(FX[f'{tag}.psnr8_vis']) == float(FX[f'{tag}.psnr8'])
        assert M.compute_psnr(FX[f'{tag}.ref8'], FX[f'{tag}.pred8'], True, np.full(ref.shape[:2], 2.0)) == got
        assert abs(O.psnr_uint8(FX[f'{tag}.ref8'], FX[f'{tag}.pred8']) - float(FX[f'{tag}.psnr8'])) < 1e-4
        back = M.color_map_forward(FX[f'{tag}.ref8'])
        assert back.dtype == np.float32 and np.abs(back - np.clip(ref, 0, 1)).max() <= 1.0 / 255


def test_ws_psnr_vs_reference():
    ws = M.WSPSNR()
    for tag in ('a', 'b'):
        ref, pred = torch.from_numpy(FX[f'{tag}.ref'])[None], torch.from_numpy(FX[f'{tag}.pred'])[None]
        H, W = ref.shape[1:3]
        assert np.array_equal(ws.get_weights(H, W), FX[f'{tag}.weights'])
        got = ws.ws_psnr(pred, ref)
        assert np.abs(got.numpy() - FX[f'{tag}.ws']).max() < 1e-4
        got255 = ws.ws_psnr(pred * 255, ref * 255, max_val=255.0)
        assert np.abs(got255.numpy() - FX[f'{tag}.ws255']).max() < 1e-3
        assert np.abs(O.ws_psnr(pred.numpy(), ref.numpy()) - FX[f'{tag}.ws']).max() < 1e-4
        # the per-ray form on all pixels is the image form; on a subset it uses the rows' weights only
        rows = torch.arange(H).view(H, 1).expand(H, W).reshape(-1)
        full = ws.ws_psnr_rays(pred.reshape(-1, 3), ref.reshape(-1, 3), rows, H)
        assert abs(full - float(FX[f'{tag}.ws'][0])) < 1e-4
        sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(1))[: H * W // 3]
        sub = ws.ws_psnr_rays(pred.reshape(-1, 3)[sel], ref.reshape(-1, 3)[sel], rows[sel], H)
        w = np.sin((rows[sel].numpy() + 0.5) * np.pi / H)[:, None]
        e = (pred.reshape(-1, 3)[sel].double().numpy() - ref.reshape(-1, 3)[sel].double().numpy()) ** 2
        assert abs(sub - 10 * np.log10(1.0 / ((w * e).sum() / (w.sum() * 3)))) < 1e-9
        assert abs(M.psnr_float(pred, ref) - 10 * np.log10(1.0 / float(((pred.double() - ref.double()) ** 2).mean()))) < 1e-9
