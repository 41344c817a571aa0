"""PIL ``Image.resize(size, resample=Image.LANCZOS)`` on uint8 RGB, restated in numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import this.

Reference call site: /root/reference/static_model/dataset_feat_extractor.py:119-142 - every decoded
frame goes through ``Image.fromarray(frame).convert('RGB').resize((cfg.equi_h, cfg.equi_w),
resample=Image.LANCZOS)`` (config.yaml:15-16: 1920 x 960, PIL sizes are (width, height)) before /255 and
the cube projection.  Pillow is a third-party dependency of the reference (no version pinned in the repo);
its resampler (src/libImaging/Resample.c, unchanged in structure since Pillow 3.4) is restated here from
its published algorithm and PINNED against the Pillow installed in this image (12.2.0), which the CPU
tests call directly: bit-exact on every case.

Algorithm (8-bit path): separable, horizontal pass first, the intermediate image is rounded to uint8,
then the vertical pass.  Per output index: ``center = (i + 0.5) * scale``, ``support = 3 * max(scale,
1)``, taps ``xmin = max(0, int(center - support + 0.5)) .. min(in, int(center + support + 0.5))``,
weights ``lanczos((x - center + 0.5) / max(scale, 1))`` normalised to sum 1 in float64, converted to
22-bit fixed point (round half away from zero), accumulated in int32 from ``1 << 21`` and shifted.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3.0)
    return 0.0


def precompute_coeffs(in_size, out_size, support=3.0, filt=_lanczos):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc: (bounds int32 [out, 2] = (xmin, count),
    kk int32 [out, ksize])."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One 8-bit resampling pass along ``axis`` of an [H, W, C] uint8 image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for i in range(bounds.shape[0]):
        x0, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = np.tensordot(kk[i, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def _bicubic(x):
    """Pillow's bicubic_filter (Resample.c): Keys kernel, a = -0.5, support 2 - Image.CUBIC of
    /root/reference/utils/utils.py:21."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_FILTERS = {'lanczos': (3.0, _lanczos), 'bicubic': (2.0, _bicubic)}


def resize_u8(img, out_hw, filter='lanczos'):
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C] exactly as PIL's Image.resize with that filter."""
    support, filt = _FILTERS[filter]
    h, w = img.shape[:2]
    oh, ow = out_hw
    cur = img
    if ow != w:
        cur = _pass(cur, *precompute_coeffs(w, ow, support, filt), axis=1)
    if oh != h:
        cur = _pass(cur, *precompute_coeffs(h, oh, support, filt), axis=0)
    return cur


def resize_lanczos_u8(img, out_hw):
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C] exactly as PIL (LANCZOS)."""
    return resize_u8(img, out_hw, 'lanczos')


def overlay(img_u8, heat, lut, alpha=0.5):
    """/root/reference/utils/utils.py:9-25 restated: min-max normalise (in the map's dtype), 256-entry colormap
    lookup as matplotlib's Colormap.__call__(bytes=True), PIL bicubic upsample, PIL.Image.blend."""
    heat = np.asarray(heat)
    heat = heat - np.min(heat)
    heat = heat / np.max(heat)
    xa = np.array(heat, copy=True)
    xa *= 256
    xa[xa == 256] = 255
    bad = np.isnan(xa)
    with np.errstate(invalid='ignore'):
        idx = np.clip(xa, 0, 255).astype(int)
    rgb = lut[idx]
    rgb[bad] = 0
    up = resize_u8(rgb.astype(np.uint8), img_u8.shape[:2], 'bicubic')
    a = img_u8.astype(np.int32)
    out = a.astype(np.float32) + np.float32(alpha) * (up.astype(np.int32) - a).astype(np.float32)
    return out.astype(np.uint8)
