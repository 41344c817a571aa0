"""Oracle: CubePad as an index table (integer work -> bit-exact).

Follows /root/reference/model/cube_pad.py:95-216 (``CubePadding.forward``) and
:28-42 (``CubePad.forward`` = independent groups of 6 along dim 0).  The reference
builds the padded tensor from 24 flipped / transposed strips and 4 replicated
corners with torch.cat; here each output element's *source* (face, row, col) is
written down directly.  Faces: 0 back, 1 down, 2 front, 3 left, 4 right, 5 top
(cube_pad.py:49).
"""
import numpy as np

B, D, F, L, R, T = 0, 1, 2, 3, 4, 5


def _strip_src(side, face, a, k, n, p):
    """Source (face, row, col) for strip element.

    side 't'/'d': element [k, a] of the p x n strip (k = depth index in output
    order, a = column).  side 'l'/'r': element [a, k] of the n x p strip (a = row).
    Derived line by line from cube_pad.py:114-162.
    """
    e = n - p + k  # "last p rows/cols", k-th of them
    m = n - 1 - a  # flipped running index
    if side == 't':   # cube_pad.py:114-126
        return [(T, k, m), (F, e, a), (T, e, a), (T, a, k), (T, m, e), (B, k, m)][face]
    if side == 'd':   # cube_pad.py:127-138
        return [(D, e, m), (B, e, m), (D, k, a), (D, m, k), (D, a, e), (F, k, a)][face]
    if side == 'l':   # cube_pad.py:139-150
        return [(R, a, e), (L, e, m), (L, a, e), (B, a, e), (F, a, e), (L, k, a)][face]
    if side == 'r':   # cube_pad.py:151-162
        return [(L, a, k), (R, e, a), (R, a, k), (F, a, k), (B, a, k), (R, k, m)][face]
    raise ValueError(side)


def cubepad_table(n, p_l, p_r, p_t, p_d):
    """int32 [6, n+p_t+p_d, n+p_l+p_r]: flat source index f'*n*n + i'*n + j'.

    Corner rule (``make_cubepad_edge``, cube_pad.py:83-90,164-176): if the
    top/down depth is larger than the left/right depth the corner replicates the
    left/right strip's first (top corners) or last (bottom corners) row
    vertically, else it replicates the top/down strip's end column horizontally.
    """
    Hp, Wp = n + p_t + p_d, n + p_l + p_r
    tab = np.empty((6, Hp, Wp), dtype=np.int32)

    def flat(s):
        return s[0] * n * n + s[1] * n + s[2]

    for f in range(6):
        for i in range(Hp):
            for j in range(Wp):
                in_t, in_d = i < p_t, i >= p_t + n
                in_l, in_r = j < p_l, j >= p_l + n
                if not (in_t or in_d or in_l or in_r):
                    s = (f, i - p_t, j - p_l)
                elif (in_t or in_d) and not (in_l or in_r):
                    if in_t:
                        s = _strip_src('t', f, j - p_l, i, n, p_t)
                    else:
                        s = _strip_src('d', f, j - p_l, i - p_t - n, n, p_d)
                elif (in_l or in_r) and not (in_t or in_d):
                    if in_l:
                        s = _strip_src('l', f, i - p_t, j, n, p_l)
                    else:
                        s = _strip_src('r', f, i - p_t, j - p_l - n, n, p_r)
                else:
                    p_td = p_t if in_t else p_d
                    p_lr = p_l if in_l else p_r
                    k_td = i if in_t else i - p_t - n
                    k_lr = j if in_l else j - p_l - n
                    if p_td > p_lr:   # replicate left/right strip row vertically
                        row = 0 if in_t else n - 1
                        s = _strip_src('l' if in_l else 'r', f, row, k_lr, n, p_lr)
                    else:             # replicate top/down strip column horizontally
                        col = 0 if in_l else n - 1
                        s = _strip_src('t' if in_t else 'd', f, col, k_td, n, p_td)
                tab[f, i, j] = flat(s)
    return tab


def pads_of(lrtd_pad):
    """cube_pad.py:12-20,60-70: int -> same pad on 4 sides, else [l, r, t, d]."""
    if isinstance(lrtd_pad, (int, np.integer)):
        return (int(lrtd_pad),) * 4
    p_l, p_r, p_t, p_d = lrtd_pad
    return int(p_l), int(p_r), int(p_t), int(p_d)


def cubepad(x, lrtd_pad):
    """x: ndarray [6N, C, n, n] (any dtype) -> [6N, C, n+pt+pd, n+pl+pr].

    cube_pad.py:28-42: batch must be a multiple of 6; groups are independent.
    """
    x = np.asarray(x)
    if x.shape[0] % 6 != 0:
        raise ValueError("CubePad size mismatch: batch %d is not a multiple of 6" % x.shape[0])
    if x.shape[2] != x.shape[3]:
        raise ValueError("CubePad needs square faces")
    p_l, p_r, p_t, p_d = pads_of(lrtd_pad)
    n = x.shape[2]
    tab = cubepad_table(n, p_l, p_r, p_t, p_d)
    g = x.shape[0] // 6
    xs = x.reshape(g, 6, x.shape[1], n * n)
    xs = np.transpose(xs, (0, 2, 1, 3)).reshape(g, x.shape[1], 6 * n * n)
    out = xs[:, :, tab.reshape(-1)]                      # [g, C, 6*Hp*Wp]
    out = out.reshape(g, x.shape[1], 6, tab.shape[1], tab.shape[2])
    return np.ascontiguousarray(np.transpose(out, (0, 2, 1, 3, 4))).reshape(
        x.shape[0], x.shape[1], tab.shape[1], tab.shape[2])
