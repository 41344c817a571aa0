"""Host-side sphere / cube geometry (float64 numpy, one-time table construction).

Mirrors the public helpers of /root/reference/utils/sph_utils.py that the hot path
uses (:15-38 face ids and rotations, :53-77 angles, :88-153 face selection and cube
coordinates).  These run once per resolution; the per-frame work happens in the HIP
kernels that consume the tables.
"""
import numpy as np

FACE_B, FACE_D, FACE_F, FACE_L, FACE_R, FACE_T = 0, 1, 2, 3, 4, 5


def rotx(ang):
    c, s = np.cos(ang), np.sin(ang)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def roty(ang):
    c, s = np.cos(ang), np.sin(ang)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rotz(ang):
    c, s = np.cos(ang), np.sin(ang)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def xy2angle(XX, YY, im_w, im_h):
    """Pixel centres -> (theta in (-pi, pi), phi in (-pi/2, pi/2)); sph_utils.py:53-60."""
    return (2 * (XX + 0.5) / float(im_w) - 1) * np.pi, (1 - 2 * (YY + 0.5) / float(im_h)) * np.pi / 2


def to_3dsphere(theta, phi, R):
    return R * np.cos(phi) * np.cos(theta), R * np.sin(phi), R * np.cos(phi) * np.sin(theta)


def pruned_inf(angle):
    """Nudge angles that sit exactly on 0, +-pi, +-pi/2 by 1e-8 (sph_utils.py:70-77)."""
    err = 10e-9
    out = np.array(angle, dtype=np.float64, copy=True)
    for val, rep in ((0.0, err), (np.pi, np.pi - err), (-np.pi, -np.pi + err),
                     (np.pi / 2, np.pi / 2 - err), (-np.pi / 2, -np.pi / 2 + err)):
        out[angle == val] = rep
    return out


def get_face(x, y, z, face_map=None):
    """Face id per direction with the reference's tie-breaking (sph_utils.py:88-102).

    The reference calls np.maximum(|x|, |y|, |z|) whose third positional argument is
    ``out=``: the "largest" magnitude is max(|x|, |y|) only.  With eps = 1e-8 the net
    rule is: |z| >= max(|x|,|y|) - eps -> R/L; else |y| >= |x| - eps... evaluated as
    the same ordered overwrites the reference performs (x faces, then y, then z)."""
    eps = 10e-9
    ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
    big = np.maximum(ax, ay)
    out = np.zeros(x.shape) if face_map is None else face_map
    for cond, fid in ((((big - ax) < eps) & (x >= 0), FACE_F), (((big - ax) < eps) & (x <= 0), FACE_B),
                      (((big - ay) < eps) & (y >= 0), FACE_T), (((big - ay) < eps) & (y <= 0), FACE_D),
                      (((big - az) < eps) & (z >= 0), FACE_R), (((big - az) < eps) & (z <= 0), FACE_L)):
        out[cond] = fid
    return out


def face_to_cube_coord(face_gr, x, y, z):
    """(u, v) in [0, 1] on the selected face, top-left origin (sph_utils.py:114-146)."""
    comp = {FACE_F: (z, y, x), FACE_B: (-z, y, x), FACE_T: (z, -x, y),
            FACE_D: (z, x, y), FACE_R: (-x, y, z), FACE_L: (x, y, z)}
    a = np.zeros(face_gr.shape)
    b = np.zeros(face_gr.shape)
    c = np.zeros(face_gr.shape)
    for fid, (va, vb, vc) in comp.items():
        sel = face_gr == fid
        a[sel], b[sel], c[sel] = va[sel], vb[sel], vc[sel]
    return (a / np.abs(c) + 1) / 2, (-b / np.abs(c) + 1) / 2


def norm_to_cube(_out_coord, w):
    return np.clip(_out_coord * (w - 1), 0., w - 1)
