"""Oracle: cube faces -> equirectangular (``Cube2Equi``).

Tables follow /root/reference/utils/cube_to_equi.py:12-35 with the helpers of
/root/reference/utils/sph_utils.py:53-77 (xy2angle, to_3dsphere, pruned_inf),
:88-102 (get_face), :114-146 (face_to_cube_coord), :149-153 (norm_to_cube).
Sampling follows ``to_equi_nn`` (cube_to_equi.py:37-66) and the channel max of
/root/reference/temporal_model/test_temporal.py:82-85.
"""
import numpy as np

FACE_B, FACE_D, FACE_F, FACE_L, FACE_R, FACE_T = 0, 1, 2, 3, 4, 5


def c2e_tables(w):
    """Returns (face_map int64 [2w,4w], out_coord float64 [2w,4w,2] (x, y) in
    [0, w-1])."""
    out_w, out_h = 4 * w, 2 * w
    XX, YY = np.meshgrid(range(out_w), range(out_h))
    # xy2angle (sph_utils.py:53-60): pixel centres
    theta = (2 * (XX + 0.5) / float(out_w) - 1) * np.pi
    phi = (1 - 2 * (YY + 0.5) / float(out_h)) * np.pi / 2
    # pruned_inf (sph_utils.py:70-77): nudge exact 0, +-pi, +-pi/2 by 1e-8
    for a in (theta, phi):
        err = 10e-9
        a[a == 0.0] = err
        a[a == np.pi] = np.pi - err
        a[a == -np.pi] = -np.pi + err
        a[a == np.pi / 2] = np.pi / 2 - err
        a[a == -np.pi / 2] = -np.pi / 2 + err
    # to_3dsphere (sph_utils.py:63-67), R = 1
    x = 1 * np.cos(phi) * np.cos(theta)
    y = 1 * np.sin(phi)
    z = 1 * np.cos(phi) * np.sin(theta)
    # get_face (sph_utils.py:88-102).  NOTE: np.maximum(|x|, |y|, |z|) passes |z|
    # as the ``out=`` argument, so the "max" is max(|x|, |y|) only - and |z| would
    # be overwritten in the reference, but np.abs(z) there is a temporary.
    eps = 10e-9
    ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
    max_arr = np.maximum(ax, ay)
    x_faces = max_arr - ax < eps
    y_faces = max_arr - ay < eps
    z_faces = max_arr - az < eps
    face = np.zeros((out_h, out_w))
    face[(x >= 0) & x_faces] = FACE_F
    face[(x <= 0) & x_faces] = FACE_B
    face[(y >= 0) & y_faces] = FACE_T
    face[(y <= 0) & y_faces] = FACE_D
    face[(z >= 0) & z_faces] = FACE_R
    face[(z <= 0) & z_faces] = FACE_L
    # face_to_cube_coord (sph_utils.py:114-146): per face (a, b, c)
    a = np.zeros_like(x)
    b = np.zeros_like(x)
    c = np.zeros_like(x)
    for f, (va, vb, vc) in {FACE_F: (z, y, x), FACE_B: (-z, y, x), FACE_T: (z, -x, y),
                            FACE_D: (z, x, y), FACE_R: (-x, y, z), FACE_L: (x, y, z)}.items():
        m = face == f
        a[m], b[m], c[m] = va[m], vb[m], vc[m]
    u = (a / np.abs(c) + 1) / 2
    v = (-b / np.abs(c) + 1) / 2
    coord = np.transpose(np.array([u, v]), (1, 2, 0))          # cube_to_equi.py:29
    # norm_to_cube (sph_utils.py:149-153)
    coord = coord * (w - 1)
    coord[coord < 0.] = 0.
    coord[coord > (w - 1)] = (w - 1)
    return face.astype(np.int64), coord


def grid_scale(out_coord):
    """cube_to_equi.py:58: M = max(gridf) over x AND y, taken on the float32 grid."""
    return np.float32(np.max(out_coord.astype(np.float32)))


def sample_pixel_coords(out_coord, w, align_corners=False):
    """Pixel-space sampling position implied by cube_to_equi.py:58-65.

    gn = (g - M/2)/(M/2) in float32, then grid_sample's un-normalisation:
    align_corners=False (what torch >= 1.3 executes today): ((gn+1)*w - 1)/2;
    align_corners=True (the torch 0.3/0.4 the reference was written for):
    (gn+1)/2*(w-1).  Returns float32 [2w,4w,2].
    """
    g = out_coord.astype(np.float32)
    M = grid_scale(out_coord)
    gn = (g - M / np.float32(2)) / (M / np.float32(2))
    if align_corners:
        return ((gn + np.float32(1)) / np.float32(2) * np.float32(w - 1)).astype(np.float32)
    return (((gn + np.float32(1)) * np.float32(w) - np.float32(1)) / np.float32(2)).astype(np.float32)


def to_equi_nn(x, face_map=None, out_coord=None, align_corners=False):
    """x: float32 [6, C, w, w] -> [1, C, 2w, 4w] (cube_to_equi.py:37-66).

    For every output pixel: pick its face, bilinear-sample that face at the pixel
    position above, taps outside [0, w) contribute zero (grid_sample
    padding_mode='zeros').  The reference computes all 6 full-size grid_samples
    and keeps the masked sixth of each; the result is the same.
    """
    x = np.asarray(x, dtype=np.float32)
    w = x.shape[2]
    if face_map is None:
        face_map, out_coord = c2e_tables(w)
    pc = sample_pixel_coords(out_coord, w, align_corners)
    px, py = pc[..., 0], pc[..., 1]
    x0 = np.floor(px)
    y0 = np.floor(py)
    fx = (px - x0).astype(np.float32)
    fy = (py - y0).astype(np.float32)
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    out = np.zeros((1, x.shape[1], 2 * w, 4 * w), dtype=np.float32)
    for f in range(6):
        m = face_map == f
        if not m.any():
            continue
        xs, ys, wx, wy = x0[m], y0[m], fx[m], fy[m]
        acc = np.zeros((x.shape[1], xs.shape[0]), dtype=np.float32)
        for dy, dx, wt in ((0, 0, (1 - wx) * (1 - wy)), (0, 1, wx * (1 - wy)),
                           (1, 0, (1 - wx) * wy), (1, 1, wx * wy)):
            yy, xx = ys + dy, xs + dx
            ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < w)
            v = x[f][:, np.clip(yy, 0, w - 1), np.clip(xx, 0, w - 1)]      # [C, npix]
            acc += np.where(ok[None, :], v, np.float32(0)) * wt[None, :].astype(np.float32)
        out[0][:, m] = acc
    return out


def saliency_from_hidden(hidden, face_map=None, out_coord=None, align_corners=False):
    """test_temporal.py:82-85: to_equi_nn -> max over channel dim -> squeeze.
    hidden float32 [6, C, w, w] -> [2w, 4w]."""
    return to_equi_nn(hidden, face_map, out_coord, align_corners)[0].max(axis=0)
