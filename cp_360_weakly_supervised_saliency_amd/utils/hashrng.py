"""Counter-based hash RNG owned by this repo.

Synthetic weights, frames and fixation maps must be bit-identical in the build
container (where the golden fixtures are generated from the reference) and on the
GPU box, independent of the numpy / torch versions installed there.  numpy's and
torch's generators give no such guarantee across versions, so everything synthetic
in this repo is drawn from the stateless integer hash below (splitmix64 finaliser
over ``seed * 2^40 + counter``), vectorised with numpy uint64 arithmetic.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _bits(seed, n, stream=0, chunk=1 << 22):
    """n uint64 words for (seed, stream); generated in chunks to bound memory."""
    out = np.empty(n, dtype=np.uint64)
    base = (np.uint64(seed) << np.uint64(44)) ^ (np.uint64(stream) << np.uint64(40))
    with np.errstate(over="ignore"):
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            ctr = np.arange(lo, hi, dtype=np.uint64)
            out[lo:hi] = _mix((base + ctr) * _GOLD + _GOLD)
    return out


def uniform(seed, shape, lo=0.0, hi=1.0, dtype=np.float32):
    """U[lo, hi) with 53 random bits per sample (float64 internally)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_bits(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).reshape(shape).astype(dtype)


_QTAB = None


def _quantile_table():
    """65536-entry table of standard-normal quantiles at (i + 0.5) / 65536
    (Acklam's rational approximation, float64, |rel err| < 1.2e-9)."""
    global _QTAB
    if _QTAB is not None:
        return _QTAB
    p = (np.arange(65536, dtype=np.float64) + 0.5) / 65536.0
    a = [-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
         1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00]
    b = [-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
         6.680131188771972e+01, -1.328068155288572e+01]
    c = [-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
         -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00]
    d = [7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
         3.754408661907416e+00]
    plow = 0.02425
    x = np.empty_like(p)
    lo = p < plow
    hi = p > 1 - plow
    mid = ~(lo | hi)
    q = np.sqrt(-2 * np.log(p[lo]))
    x[lo] = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / \
            ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1)
    q = np.sqrt(-2 * np.log(1 - p[hi]))
    x[hi] = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / \
             ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1)
    q = p[mid] - 0.5
    r = q * q
    x[mid] = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / \
             (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1)
    _QTAB = x.astype(np.float32)
    return _QTAB


def normal(seed, shape, mean=0.0, std=1.0, dtype=np.float32):
    """N(mean, std): each 64-bit hash word yields four 16-bit indices into a
    normal-quantile table (inverse-CDF sampling at 16-bit resolution).  Integer
    work plus one table gather: fast enough for the 360 M ConvLSTM parameters and
    reproducible across machines."""
    n = int(np.prod(shape)) if len(shape) else 1
    words = _bits(seed, (n + 3) // 4, stream=1)
    idx = words.view(np.uint16)[:n]
    out = np.take(_quantile_table(), idx)
    if std != 1.0:
        np.multiply(out, np.float32(std), out=out)
    if mean != 0.0:
        np.add(out, np.float32(mean), out=out)
    return out.reshape(shape).astype(dtype, copy=False)


def integers(seed, shape, lo, hi, dtype=np.int64):
    """Integers in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    span = np.uint64(hi - lo)
    return (lo + (_bits(seed, n) % span).astype(np.int64)).reshape(shape).astype(dtype)
