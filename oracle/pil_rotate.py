"""CPU restatement (numpy, float64) of the rotation augmentation of the reference -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

The reference rotates every observed image by a random angle with Pillow before inference
(/root/reference/train_galaxy.py:41-54: uint8 RGB/L images; /root/reference/train_particles.py:31-43: float32 'F'
images): `Image.rotate(angle_degrees, resample=Image.BICUBIC)`.  Pillow is a third-party dependency (requirements.txt
pins Pillow~=8.2.0; 12.2.0 is what this image has) and its resampler is C code that is not under /root/reference, so its
published algorithm is restated here:

  * Image.rotate (PIL/Image.py): angle %= 360; exact 0/90/180/270 are copies/transposes; otherwise the inverse affine
    matrix [cos a, sin a, tx, -sin a, cos a, ty] with a = -radians(angle), entries rounded to 15 decimals, centre
    (w/2, h/2); then Image.transform(size, AFFINE, matrix, BICUBIC) with zero fill.
  * libImaging/Geometry.c: affine_transform maps the output pixel centre (x+0.5, y+0.5) to (xin, yin) in doubles;
    bicubic_filter{8,32RGB,32F} reject samples outside [0, size), shift by -0.5, take a 4x4 neighbourhood with clamped
    column indices, rows outside the image repeat the previous row's value, cubic weights of the a = -1 family
    evaluated in Horner form (coefficients in the C type of the samples, see _cubic), 8-bit output clamped to
    [0, 255] and truncated ((UINT8) cast; that is what Pillow 12.2.0 does -- verified sample by sample).

Pinned against Pillow itself (tests/test_oracle_rotate.py runs both where PIL is importable) and by the `y_rot` entries of
tests/golden/galaxy_augment.npz / particles_augment.npz: the rotated batches the REFERENCE's eval_minibatch fed its encoder,
captured by tests/golden/gen_golden.py (which runs the reference, and through it Pillow, on the seeded cases).
"""
import math

import numpy as np


def pil_matrix(angle_deg, w, h):
    """The six affine coefficients Image.rotate hands to the C resampler, or None for the exact fast paths."""
    angle = angle_deg % 360.0
    if angle in (0.0, 90.0, 180.0, 270.0) and (angle in (0.0, 180.0) or w == h):
        return None
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + 0.0
    m[2] += cx
    m[5] += cy
    return m


def _cubic(v1, v2, v3, v4, d):
    """The BICUBIC macro.  The coefficient expressions are evaluated in the C type of the samples: exact integers for
    uint8 pixels, float32 arithmetic for 'F' pixels (pass float32 arrays), doubles for the second (vertical) pass;
    the Horner evaluation itself is always in doubles."""
    p1 = v2
    p2 = -v1 + v3
    p3 = 2 * (v1 - v2) + v3 - v4
    p4 = -v1 + v2 - v3 + v4
    d = d.astype(np.float64)
    return p1.astype(np.float64) + d * (p2.astype(np.float64) + d * (p3.astype(np.float64) + d * p4.astype(np.float64)))


def _floor_int(v):
    # FLOOR(v) = v < 0 ? (int)floor(v) : (int)v
    return np.where(v < 0.0, np.floor(v), np.trunc(v)).astype(np.int64)


def affine_bicubic(plane, m):
    """plane: (h, w) source samples: float64 holding uint8 values (integer arithmetic is exact there) or float32 for
    'F' images.  Returns (value, inside) with value the unclamped float64 interpolant at every output pixel and
    inside the mask of pixels the filter accepted."""
    h, w = plane.shape
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    xc, yc = xs + 0.5, ys + 0.5
    xin = m[0] * xc + m[1] * yc + m[2]
    yin = m[3] * xc + m[4] * yc + m[5]
    inside = ~((xin < 0.0) | (xin >= w) | (yin < 0.0) | (yin >= h))
    xin = xin - 0.5
    yin = yin - 0.5
    x = _floor_int(xin)
    y = _floor_int(yin)
    dx = xin - x
    dy = yin - y
    x = x - 1
    y = y - 1
    cols = [np.clip(x + k, 0, w - 1) for k in range(4)]

    def row(yy):
        r = np.clip(yy, 0, h - 1)
        return _cubic(plane[r, cols[0]], plane[r, cols[1]], plane[r, cols[2]], plane[r, cols[3]], dx)

    v1 = row(y)
    ok = (y + 1 >= 0) & (y + 1 < h)
    v2 = np.where(ok, row(y + 1), v1)
    ok = (y + 2 >= 0) & (y + 2 < h)
    v3 = np.where(ok, row(y + 2), v2)
    ok = (y + 3 >= 0) & (y + 3 < h)
    v4 = np.where(ok, row(y + 3), v3)
    return _cubic(v1, v2, v3, v4, dy), inside


def _exact(img, angle):
    k = {0.0: 0, 90.0: 1, 180.0: 2, 270.0: 3}[angle]
    return np.rot90(img, k, axes=(0, 1)).copy()          # ROTATE_90 is counter-clockwise


def rotate_u8(img, angle_deg):
    """img: (h, w) or (h, w, c) uint8.  Pillow 'L'/'RGB' BICUBIC rotate."""
    h, w = img.shape[:2]
    m = pil_matrix(angle_deg, w, h)
    if m is None:
        return _exact(img, angle_deg % 360.0)
    planes = img.reshape(h, w, -1)
    out = np.zeros_like(planes)
    for c in range(planes.shape[2]):
        v, inside = affine_bicubic(planes[:, :, c].astype(np.float64), m)
        q = np.where(v <= 0.0, 0.0, np.where(v >= 255.0, 255.0, np.trunc(v)))
        out[:, :, c] = np.where(inside, q, 0.0).astype(np.uint8)
    return out.reshape(img.shape)


def rotate_f32(img, angle_deg):
    """img: (h, w) float32.  Pillow 'F' BICUBIC rotate."""
    h, w = img.shape
    m = pil_matrix(angle_deg, w, h)
    if m is None:
        return _exact(img, angle_deg % 360.0)
    v, inside = affine_bicubic(img.astype(np.float32), m)
    return np.where(inside, v, 0.0).astype(np.float32)


def augment_galaxy(y, offset):
    """train_galaxy.py:47-54.  y: (B, n*n, C) float32 in [0,1]; offset: (B,) radians.  The image goes through uint8."""
    B, N, C = y.shape
    n = int(np.sqrt(N))
    out = np.empty_like(y)
    for i in range(B):
        im = (y[i].reshape(n, n, C) * 255).astype(np.uint8)
        if C == 1:
            im = im[:, :, 0]
        r = rotate_u8(im, 360 * offset[i] / 2 / np.pi)
        out[i] = (r.astype(float) / 255).reshape(N, C).astype(np.float32)
    return out


def augment_particles(y, offset):
    """train_particles.py:39-43.  y: (B, n*n) float32."""
    B, N = y.shape
    n = int(np.sqrt(N))
    out = np.empty_like(y)
    for i in range(B):
        out[i] = rotate_f32(y[i].reshape(n, n), 360 * offset[i] / 2 / np.pi).reshape(-1)
    return out
