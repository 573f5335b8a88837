"""Golden vectors for the ERP -> cube-map conversion in front of the mono-depth net, produced
by running the REFERENCE's own class (build container only):  python tests/golden/make_e2c_fixtures.py

What runs: `Equirec2Cube(H, W, H // 2)` as constructed at network/omni_mvsnet/pipeline3_model.py:149-151
(UniFuse-Unidirectional-Fusion/UniFuse/datasets/util.py:7-111), `.run(img)` and `.run(img, depth)`, on seeded inputs of
the converter's own size (its cv2.resize branch is not exercised; cv2 is a stub here).  The sampler inside is the scipy
of this image (1.15.3).  The fixture holds inputs, the coordinate tables and outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402
from make_fixtures import smooth_pano  # noqa: E402


def main():
    R.setup()
    from datasets.util import Equirec2Cube
    arrs = {}
    g = torch.Generator().manual_seed(17)
    for tag, (H, W) in {'a': (32, 64), 'b': (48, 96), 'c': (30, 60)}.items():      # c: odd face width -> pole and seam wrap
        fw = H // 2
        e2c = Equirec2Cube(H, W, fw)
        img = (smooth_pano(g, 3, H, W) + 0.05 * torch.randn(3, H, W, generator=g)).permute(1, 2, 0).contiguous().numpy()
        dep = (1.0 + 5.0 * torch.rand(H, W, 1, generator=g)).numpy()
        cube, cdep = e2c.run(img, dep)
        only = e2c.run(img)
        assert np.array_equal(only, cube)
        arrs.update({f'{tag}.img': img, f'{tag}.dep': dep, f'{tag}.cube': cube.astype(np.float32), f'{tag}.cube_dep': cdep.astype(np.float32),
                     f'{tag}.coor_x': e2c.coor_x[..., 0], f'{tag}.coor_y': e2c.coor_y[..., 0], f'{tag}.cosmap': e2c.cosmaps[..., 0]})
        print(tag, cube.shape, cube.dtype, cdep.dtype, e2c.coor_x.dtype, float(e2c.coor_x.min()), float(e2c.coor_x.max()),
              float(e2c.coor_y.min()), float(e2c.coor_y.max()))
    path = os.path.join(HERE, 'e2c_fixtures.npz')
    np.savez_compressed(path, **arrs)
    print('e2c_fixtures', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
