"""CPU restatement (numpy) of the reference's ERP -> cube-map conversion.  TEST INFRASTRUCTURE ONLY: imported by
tests/, never by the product path (panogrf_amd/ fails without the HIP library).

What it follows: `Equirec2Cube` of UniFuse-Unidirectional-Fusion/UniFuse/datasets/util.py:7-111, called once per view and
frame by `FullPipeline.e2c_process` (network/omni_mvsnet/pipeline3_model.py:266-282, call sites :644, :744, :988) to
build the cube-map input of the mono-depth net.  The sampler there is `scipy.ndimage.map_coordinates(order, mode='wrap')`
(util.py:78-79, scipy 1.7.3 in the reference's requirements_freeze.txt:176; a third-party dependency, not in
/root/reference): its algorithm for order 0 / 1 is restated in `sample_wrap` below and pinned against the scipy installed
in this image (1.15.3) and against outputs of the reference class itself (tests/golden/make_e2c_fixtures.py ->
tests/golden/e2c_fixtures.npz).

Published behaviour of the sampler that matters here (scipy ndimage, NI_GeometricTransform):
  * `mode='wrap'` is the legacy wrap with period len-1 ("d b c d | a b c d | b c a b"): a coordinate outside
    [0, len-1] is moved by multiples of (len-1), NOT len;
  * order 1: taps floor(c), floor(c)+1 with weights 1-f, f; order 0: tap floor(c + 0.5);
  * arithmetic in double: per tap ((value * w_row) * w_col), summed in tap order (r0c0, r0c1, r1c0, r1c1), then cast
    to the input dtype.
"""
import numpy as np

FACES = 'FRBLUD'


def cube_directions(face_w):
    """Points on the unit cube, faces side by side in [F R B L U D] order (util.py:27-58): [face_w, 6*face_w, 3] f32."""
    r = np.linspace(-0.5, 0.5, num=face_w, dtype=np.float32)
    u, v = np.meshgrid(r, -r)                                   # u grows to the right, v downwards-negative
    half = np.full_like(u, 0.5)
    faces = [
        (u, v, half),                    # F: z = +0.5
        (half, v, u[:, ::-1]),           # R: x = +0.5, z runs right-to-left
        (u[:, ::-1], v, -half),          # B: z = -0.5
        (-half, v, u),                   # L: x = -0.5
        (u, half, v[::-1]),              # U: y = +0.5, z takes the (flipped) vertical grid
        (u, -half, v),                   # D: y = -0.5
    ]
    return np.concatenate([np.stack(f, -1) for f in faces], 1).astype(np.float32)


def erp_coordinates(equ_h, equ_w, face_w):
    """Sub-pixel ERP coordinates of every cube-map pixel (util.py:60-70): (coor_y, coor_x), each [face_w, 6*face_w] f32."""
    xyz = cube_directions(face_w)
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    lon = np.arctan2(x, z)
    lat = np.arctan2(y, np.sqrt(x ** 2 + z ** 2))
    coor_x = (lon / (2 * np.pi) + 0.5) * equ_w - 0.5
    coor_y = (-lat / np.pi + 0.5) * equ_h - 0.5
    return coor_y.astype(np.float32), coor_x.astype(np.float32)


def cos_map(face_w):
    """Ray-distance -> z-depth factor of the cube faces (util.py:22-24): [face_w, 6*face_w] f32."""
    r = np.linspace(-0.5, 0.5, num=face_w, dtype=np.float32)
    u, v = np.meshgrid(r, -r)
    c = 1 / np.sqrt((2 * u) ** 2 + (2 * v) ** 2 + 1)
    return np.concatenate(6 * [c], 1).astype(np.float32)


def _wrap_coord(c, n):
    """legacy 'wrap': period n-1 (scipy ni_interpolation.c, map_coordinate)."""
    c = np.asarray(c, np.float64).copy()
    sz = float(n - 1)
    lo = c < 0
    c[lo] = c[lo] + sz * np.trunc(-c[lo] / sz) + sz
    hi = c > sz
    c[hi] = c[hi] - sz * np.trunc(c[hi] / sz)
    return c


def sample_wrap(img, cy, cx, order):
    """map_coordinates(img [R,C], [cy, cx], order in (0, 1), mode='wrap') -> array like cy, dtype of img."""
    R, Cn = img.shape
    y, x = _wrap_coord(cy, R), _wrap_coord(cx, Cn)
    v = img.astype(np.float64)
    if order == 0:
        iy = np.clip(np.floor(y + 0.5).astype(np.int64), 0, R - 1)
        ix = np.clip(np.floor(x + 0.5).astype(np.int64), 0, Cn - 1)
        return img[iy, ix]
    y0, x0 = np.floor(y), np.floor(x)
    fy, fx = y - y0, x - x0
    y0, x0 = y0.astype(np.int64), x0.astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, R - 1), np.minimum(x0 + 1, Cn - 1)   # weight of the clamped tap is 0 there
    t = (v[y0, x0] * (1 - fy)) * (1 - fx)
    t = t + (v[y0, x1] * (1 - fy)) * fx
    t = t + (v[y1, x0] * fy) * (1 - fx)
    t = t + (v[y1, x1] * fy) * fx
    return t.astype(img.dtype)


def pad_poles(e_img):
    """[H,W] -> [H+2,W]: last row and first row, each rolled by W/2, appended BELOW the image (util.py:73-75)."""
    w = e_img.shape[1]
    return np.concatenate([e_img, np.roll(e_img[[-1]], w // 2, 1), np.roll(e_img[[0]], w // 2, 1)], 0)


def equirec_to_cube(equ_img, face_w, equ_dep=None):
    """Equirec2Cube(H, W, face_w).run(equ_img [H,W,C], equ_dep [H,W,Cd] or None) (util.py:81-111) for an input that
    already has the converter's size (the cv2.resize branch is not restated)."""
    H, W = equ_img.shape[:2]
    cy, cx = erp_coordinates(H, W, face_w)
    cube = np.stack([sample_wrap(pad_poles(equ_img[..., i]), cy, cx, 1) for i in range(equ_img.shape[2])], -1)
    if equ_dep is None:
        return cube
    dep = np.stack([sample_wrap(pad_poles(equ_dep[..., i]), cy, cx, 0) for i in range(equ_dep.shape[2])], -1)
    return cube, dep * cos_map(face_w)[..., None]
