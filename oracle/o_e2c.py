"""Oracle: equirectangular -> 6 cube faces.

Grid construction follows /root/reference/utils/equi_to_cube.py:12-110
(``Equi2Cube.__init__``) with the rotation matrices of
/root/reference/utils/sph_utils.py:23-38.  Sampling follows ``to_cube``
(equi_to_cube.py:112-129) = ``cv2.remap(img[:, :, c], inX, inY, INTER_LINEAR)``.
cv2 is NOT part of the reference tree nor of this image: ``remap_linear`` restates
OpenCV's published algorithm (imgproc/src/imgwarp.cpp, remap with INTER_LINEAR on
float maps: INTER_BITS = 5, BORDER_CONSTANT value 0); it is the one unpinned
boundary of the oracle (see oracle/__init__.py).
"""
import math
import numpy as np

VIEWS_DEG = [[180, 0, 0],   # back      equi_to_cube.py:17-22
             [0, -90, 0],   # bottom
             [0, 0, 0],     # front
             [-90, 0, 0],   # left
             [90, 0, 0],    # right
             [0, 90, 0]]    # top


def rotx(a):   # sph_utils.py:23-26
    return np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])


def roty(a):   # sph_utils.py:29-32
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


def rotz(a):   # sph_utils.py:35-38
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])


def _trig_luts(in_h, in_w):
    """equi_to_cube.py:49-56: monotone tables whose linear inverse stands in for
    acos / atan.  acos table: 2W samples of -cos(k*pi/2W) then 1; atan table:
    tan at half a step inside -pi/2, then tan(k*pi/2H - pi/2), then half a step
    inside +pi/2."""
    n_acos, n_atan = 2 * in_w, 2 * in_h
    d_acos, d_atan = np.pi / n_acos, np.pi / n_atan
    lut_acos = np.append(-np.cos(np.arange(0, n_acos) * d_acos), 1.)
    lut_atan = np.concatenate(([np.tan(d_atan / 2 - np.pi / 2)],
                               np.tan(np.arange(1, n_atan) * d_atan - np.pi / 2),
                               [np.tan(-d_atan / 2 + np.pi / 2)]))
    return lut_acos, d_acos, lut_atan, d_atan


def _lut_inverse(q, lut):
    """scipy ``interp1d(lut, arange(len(lut)), 'linear')`` (equi_to_cube.py:91-96)
    with its default bounds_error=True; np.interp gives the identical float64."""
    if q.size and (q.min() < lut[0] or q.max() > lut[-1]):
        raise ValueError("value outside the interpolation range")
    return np.interp(q, lut, np.arange(0, len(lut)))


def equi2cube_grids(cube_dim, in_h, in_w, vfov=90):
    """Returns (inXs, inYs): float64 [6, cube_dim*cube_dim] remap coordinates.

    Restates equi_to_cube.py:24-110 with the same per-element operation order, so
    the float64 values (and their float32 casts) are identical to the reference's
    ``self.inXs / self.inYs``:
      ray of output pixel (col j, row i): (tl0 + uv0*j, tl1 + uv1*i, 1)   :41-46,73-75
      rotate by roty(yaw) . rotx(pitch) . rotz(roll)                        :71,76
      phi from the atan table of qy/|q_xz|, theta from the acos table of
      -qz/|q_xz| negated where qx < 0, +-pi/2 where |q_xz| < 1e-9          :82-97
      inX = theta/pi * W/2 + W/2 + 1, inY = phi/(pi/2) * H/2 + H/2 + 1      :100-101
      clamp: <1 -> 1, >= size-1 -> size-1                                   :104-108
    """
    assert in_h * 2 == in_w                                   # :15
    half = math.tan((vfov * np.pi / 180) / 2)
    tl = np.array([-half * (cube_dim / cube_dim), -half, 1])
    uv = np.array([-2 * tl[0] / cube_dim, -2 * tl[1] / cube_dim, 0])
    lut_acos, d_acos, lut_atan, d_atan = _trig_luts(in_h, in_w)
    col = np.tile(np.arange(cube_dim), cube_dim)              # meshgrid X flattened (:59-62)
    row = np.repeat(np.arange(cube_dim), cube_dim)
    rays = np.stack([tl[0] + uv[0] * col, tl[1] + uv[1] * row, tl[2] + uv[2] * np.ones(col.shape[0])])
    xs = np.empty((6, col.shape[0]))
    ys = np.empty((6, col.shape[0]))
    for f, (yaw, pitch, roll) in enumerate(np.array(VIEWS_DEG) * np.pi / 180):
        q = np.dot(np.dot(np.dot(roty(yaw), rotx(pitch)), rotz(roll)), rays)
        qx, qy, qz = q[0], q[1], q[2]
        nxz = np.sqrt(qx ** 2 + qz ** 2)
        pole = nxz < 10e-10                                   # :86
        ok = ~pole
        phi = np.where(qy > 0, np.pi / 2, -np.pi / 2)          # pole value (:87-88)
        theta = np.zeros_like(nxz)
        phi[ok] = _lut_inverse(qy[ok] / nxz[ok], lut_atan) * d_atan - (np.pi / 2)
        theta[ok] = _lut_inverse(-qz[ok] / nxz[ok], lut_acos) * d_acos
        neg = ok & (qx < 0)
        theta[neg] = -theta[neg]
        x = (theta / np.pi) * (in_w / 2) + (in_w / 2) + 1
        y = (phi / (np.pi / 2)) * (in_h / 2) + (in_h / 2) + 1
        x[x < 1] = 1
        x[x >= in_w - 1] = in_w - 1
        y[y < 1] = 1
        y[y >= in_h - 1] = in_h - 1
        xs[f], ys[f] = x, y
    return xs, ys


def grids_f32(cube_dim, in_h, in_w):
    """[6, cd, cd, 2] float32 (x, y), the form ``to_cube`` hands to cv2.remap
    (equi_to_cube.py:122-125: reshape(cd, cd).astype('float32'))."""
    xs, ys = equi2cube_grids(cube_dim, in_h, in_w)
    g = np.stack([xs.reshape(6, cube_dim, cube_dim), ys.reshape(6, cube_dim, cube_dim)], axis=-1)
    return g.astype(np.float32)


def _cv_round(v):
    """cvRound: round half to even (lrint under the default FP mode)."""
    return np.rint(v).astype(np.int64)


def remap_linear(img, map_x, map_y, fixed_point=True):
    """``cv2.remap(img, map_x, map_y, cv2.INTER_LINEAR)`` restated for a 2-D or
    [H, W, C] float image and float32 maps; BORDER_CONSTANT, borderValue 0.

    fixed_point=True (OpenCV behaviour): coordinates are quantised to 1/32 px:
    ``s = cvRound(x*32); ix = s >> 5; fx = (s & 31)/32``; the four weights
    (1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx*fy are exactly representable; the
    accumulation is done in the image's own float type (float64 for the
    reference's float64 frames), taps outside the image contribute 0.
    fixed_point=False: plain floor/frac bilinear in float64 (switch kept for the
    ablation the survey asks for; not the default anywhere).
    """
    img = np.asarray(img)
    H, W = img.shape[:2]
    mx = np.asarray(map_x, dtype=np.float32)
    my = np.asarray(map_y, dtype=np.float32)
    if fixed_point:
        sx = _cv_round(mx.astype(np.float32) * np.float32(32))
        sy = _cv_round(my.astype(np.float32) * np.float32(32))
        ix, iy = sx >> 5, sy >> 5
        fx = (sx & 31).astype(np.float64) / 32.0
        fy = (sy & 31).astype(np.float64) / 32.0
    else:
        ix = np.floor(mx).astype(np.int64)
        iy = np.floor(my).astype(np.int64)
        fx = mx.astype(np.float64) - ix
        fy = my.astype(np.float64) - iy
    acc_t = np.float64 if img.dtype != np.float32 else np.float32
    src = img.astype(acc_t)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        if v.ndim == 3:
            ok = ok[..., None]
        return np.where(ok, v, 0)

    w00 = ((1 - fx) * (1 - fy)).astype(acc_t)
    w01 = (fx * (1 - fy)).astype(acc_t)
    w10 = ((1 - fx) * fy).astype(acc_t)
    w11 = (fx * fy).astype(acc_t)
    if src.ndim == 3:
        w00, w01, w10, w11 = (w[..., None] for w in (w00, w01, w10, w11))
    out = tap(iy, ix) * w00 + tap(iy, ix + 1) * w01 + tap(iy + 1, ix) * w10 + tap(iy + 1, ix + 1) * w11
    return out.astype(img.dtype if img.dtype.kind == 'f' else acc_t)


def to_cube(img, cube_dim, fixed_point=True, grids=None):
    """equi_to_cube.py:112-129. img [H, W, C] float -> ndarray [6, cd, cd, C]."""
    H, W = img.shape[:2]
    g = grids if grids is not None else grids_f32(cube_dim, H, W)
    return np.stack([remap_linear(img, g[f, :, :, 0], g[f, :, :, 1], fixed_point) for f in range(6)])


IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STD = [0.229, 0.224, 0.225]


def im_norm_batch(cubes):
    """utils/utils.py:28-33 + dataset_feat_extractor.py:148-157 +
    class_activation_model.py:55: per-channel (x-mean)/std on [6,cd,cd,3] float64,
    astype(float32), HWC -> CHW.  Returns [6, 3, cd, cd] float32."""
    c = np.array(cubes, dtype=np.float64, copy=True)
    for ch in range(3):
        c[..., ch] = (c[..., ch] - IMAGENET_MEAN[ch]) / IMAGENET_STD[ch]
    return np.ascontiguousarray(np.transpose(c.astype(np.float32), (0, 3, 1, 2)))
