"""Oracle: the saliency metrics of /root/reference/utils/eval_saliency.py used by
test_temporal.py:101-110 - AUC_Judd (:90-146), CorrCoeff (:149-176), similarity (:179-190),
AUC_Borji (:14-87).  They are needed for the bf16 acceptance gate (AUC-Judd / CC of the
build's map within 1e-3 of the oracle map's).

Every metric first resizes both maps with ``cv2.resize(x, (240, 120), cv2.INTER_LANCZOS4)``.
The third positional argument of cv2.resize is ``dst``, not the interpolation flag, so the
call runs OpenCV's DEFAULT interpolation, INTER_LINEAR.  cv2 is absent from /root/reference
and from this image: ``resize_linear`` restates OpenCV's published bilinear resize for
float images (half-pixel centres, source index clamped at the borders, no anti-aliasing)
- UNPINNED like cv2.remap (oracle/__init__.py).
"""
import numpy as np

_trapz = getattr(np, 'trapezoid', None) or np.trapz      # np.trapz of the reference, renamed in numpy 2


def resize_linear(img, dsize):
    """cv2.resize(img, dsize=(width, height)) with INTER_LINEAR on a 2-D float array."""
    img = np.asarray(img)
    dt = img.dtype if img.dtype in (np.float32, np.float64) else np.float32
    src = img.astype(dt)
    H, W = src.shape
    dw, dh = dsize

    def axis(n_src, n_dst):
        scale = n_src / float(n_dst)
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0).astype(np.float32)
        lo = i0 < 0
        frac[lo], i0[lo] = 0.0, 0
        hi = i0 >= n_src - 1
        frac[hi], i0[hi] = 0.0, n_src - 1
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, frac

    y0, y1, fy = axis(H, dh)
    x0, x1, fx = axis(W, dw)
    fx = fx[None, :].astype(dt)
    fy = fy[:, None].astype(dt)
    top = src[y0][:, x0] * (1 - fx) + src[y0][:, x1] * fx
    bot = src[y1][:, x0] * (1 - fx) + src[y1][:, x1] * fx
    return (top * (1 - fy) + bot * fy).astype(dt)


def _prep(a, b):
    a = resize_linear(a, (240, 120))
    b = resize_linear(b, (240, 120))
    assert a.shape == b.shape
    return a, b


def auc_judd(saliency_map, fixation_map, jitter=True, rng=None):
    """eval_saliency.py:90-146.  ``rng``: np.random.RandomState for the jitter (the
    reference draws from the global, unseeded generator, :106-109)."""
    if not np.any(fixation_map):
        raise ValueError('no fixation_map')                      # reference: print + exit()
    S, F = _prep(saliency_map, fixation_map)
    if jitter:
        rng = rng or np.random
        S = S + rng.randn(S.shape[0], S.shape[1]) / 1e7
    S = (S - np.min(S)) / (np.max(S) - np.min(S))
    Sth = S[F > np.mean(F) + 2 * np.std(F)]                       # saliency at fixated pixels (:123)
    n_fix, n_pix = np.size(Sth), np.size(S)
    thr = np.sort(Sth)[::-1]
    tp = np.zeros(n_fix + 2)
    fp = np.zeros(n_fix + 2)
    tp[-1] = fp[-1] = 1.0
    # aboveth = #pixels with S >= thresh, vectorised with a sorted copy
    s_sorted = np.sort(S.reshape(-1))
    above = n_pix - np.searchsorted(s_sorted, thr, side='left')
    i = np.arange(n_fix)
    tp[1:-1] = i / n_fix                                          # as written in the reference (:135)
    fp[1:-1] = (above - i) / (n_pix - n_fix)
    return float(_trapz(tp, fp))


def corr_coeff(map1, map2):
    """eval_saliency.py:149-176."""
    a, b = _prep(map1, map2)
    a = (a - np.mean(a)) / np.std(a)
    b = (b - np.mean(b)) / np.std(b)
    am, bm = np.mean(a), np.mean(b)
    return float(np.sum((a - am) * (b - bm)) / np.sqrt(np.sum((a - am) ** 2) * np.sum((b - bm) ** 2)))


def similarity(map1, map2):
    """eval_saliency.py:179-190."""
    a, b = _prep(map1, map2)
    a = (a - np.min(a)) / (np.max(a) - np.min(a))
    a = a / np.sum(a)
    b = (b - np.min(b)) / (np.max(b) - np.min(b))
    b = b / np.sum(b)
    return float(np.sum(np.minimum(a, b)))


def auc_borji(saliency_map, fixation_map, n_splits=100, step=0.01, rng=None):
    """eval_saliency.py:14-87 (random splits drawn from ``rng``)."""
    if not np.any(fixation_map):
        raise ValueError('no fixation_map')
    rng = rng or np.random
    S, F = _prep(saliency_map, fixation_map)
    S = np.array(S, copy=True)
    S[S > np.mean(S) + 2 * np.std(S)] = 1.0                       # :37-38
    S = (S - np.min(S)) / (np.max(S) - np.min(S))
    Sf, Ff = S.flatten(), F.flatten()
    Sth = Sf[Ff > np.mean(Ff) + 2 * np.std(Ff)]
    n_fix, n_pix = np.size(Sth), np.size(Sf)
    rr = rng.randint(0, high=n_pix, size=(n_fix, n_splits))
    randfix = Sf[rr]
    aucs = []
    for ss in range(n_splits):
        cur = randfix[:, ss]
        thr = np.arange(0.0, np.max(np.append(Sth, cur)), step)[::-1]
        tp = np.zeros(len(thr) + 2)
        fp = np.zeros(len(thr) + 2)
        tp[-1] = fp[-1] = 1.0
        tp[1:-1] = [(Sth >= t).sum() / float(n_fix) for t in thr]
        fp[1:-1] = [(cur >= t).sum() / float(n_fix) for t in thr]
        aucs.append(_trapz(tp, fp))
    return float(np.mean(aucs))
