"""Golden vectors for the image metrics, from the REFERENCE's own functions (build container only):
    python tests/golden/make_metrics_fixtures.py
network/metrics.py: compute_psnr, WSPSNR; utils/base_utils.py: color_map_backward.  skimage is a stub here; nothing of it runs."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402


def main():
    R.setup()
    from network.metrics import compute_psnr, WSPSNR
    from utils.base_utils import color_map_backward
    g = torch.Generator().manual_seed(23)
    arrs = {}
    for tag, (H, W) in {'a': (16, 32), 'b': (24, 48)}.items():
        ref = torch.rand(H, W, 3, generator=g)
        pred = ref + 0.05 * torch.randn(H, W, 3, generator=g)          # leaves [0,1] in places: exercises the clip
        r8, p8 = color_map_backward(ref.numpy()), color_map_backward(pred.numpy())
        ws = WSPSNR()
        arrs.update({f'{tag}.ref': ref.numpy(), f'{tag}.pred': pred.numpy(), f'{tag}.ref8': r8, f'{tag}.pred8': p8,
                     f'{tag}.psnr8': np.float64(compute_psnr(r8, p8)),
                     f'{tag}.psnr8_vis': np.float64(compute_psnr(r8, p8, True, np.full((H, W), 2.0), 1.5)),
                     f'{tag}.ws': ws.ws_psnr(pred[None], ref[None]).numpy(),
                     f'{tag}.ws255': ws.ws_psnr(pred[None] * 255, ref[None] * 255, max_val=255.0).numpy(),
                     f'{tag}.weights': ws.get_weights(H, W)})
        print(tag, arrs[f'{tag}.psnr8'], arrs[f'{tag}.ws'])
    path = os.path.join(HERE, 'metrics_fixtures.npz')
    np.savez_compressed(path, **arrs)
    print('metrics_fixtures', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
