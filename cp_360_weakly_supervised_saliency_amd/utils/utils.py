"""``overlay`` / ``cam_visual`` with the reference's signatures (/root/reference/utils/utils.py:9-25, :40-45),
rendered on the GPU (csrc/overlay.hip + the PIL-exact resampler of csrc/resize.hip):

    heatmap - min, / max  ->  matplotlib 'jet' colormap (bytes)  ->  PIL bicubic upsample to the image size
    ->  PIL.Image.blend(img, heatmap, alpha)

bit-exact with matplotlib 3.x + Pillow (tests compare against both).  The colormap table is rebuilt on the
host from matplotlib's published segment data (``jet_lut``), so matplotlib is not needed at run time.
"""
import numpy as np
import torch

from .._lib import check, lib, ptr, require_gpu, stream
from .resize import LanczosResize

# matplotlib's _jet_data (matplotlib/_cm.py): (x, y0, y1) breakpoints per channel
_JET = {
    'red': ((0.00, 0, 0), (0.35, 0, 0), (0.66, 1, 1), (0.89, 1, 1), (1.00, 0.5, 0.5)),
    'green': ((0.000, 0, 0), (0.125, 0, 0), (0.375, 1, 1), (0.640, 1, 1), (0.910, 0, 0), (1.000, 0, 0)),
    'blue': ((0.00, 0.5, 0.5), (0.11, 1, 1), (0.34, 1, 1), (0.65, 0, 0), (1.00, 0, 0)),
}


def _lookup_table(n, data):
    """matplotlib.colors._create_lookup_table(N, data, gamma=1.0), restated."""
    adata = np.array(data, dtype=float)
    x, y0, y1 = adata[:, 0] * (n - 1), adata[:, 1], adata[:, 2]
    xind = (n - 1) * np.linspace(0, 1, n)
    ind = np.searchsorted(x, xind)[1:-1]
    distance = (xind[1:-1] - x[ind - 1]) / (x[ind] - x[ind - 1])
    lut = np.concatenate([[y1[0]], distance * (y0[ind] - y1[ind - 1]) + y1[ind - 1], [y0[-1]]])
    return np.clip(lut, 0.0, 1.0)


def jet_lut():
    """uint8 [256, 3]: the first three columns of ``(plt.get_cmap('jet')._lut[:256] * 255).astype(np.uint8)``."""
    cols = [_lookup_table(256, _JET[c]) for c in ('red', 'green', 'blue')]
    return (np.stack(cols, axis=1) * 255).astype(np.uint8)


_LUT_DEV = {}


def _lut(cmap, device):
    if cmap != 'jet':
        raise ValueError("only the 'jet' colormap of the reference's call sites is built in")
    key = (cmap, str(device))
    if key not in _LUT_DEV:
        _LUT_DEV[key] = torch.from_numpy(jet_lut()).to(device).contiguous()
    return _LUT_DEV[key]


_RESIZERS = {}


def colorize(heatmap, cmap='jet', square=False):
    """float [h, w] (device tensor) -> uint8 [h, w, 3]: min-max normalise (float32, numpy's order), colormap."""
    require_gpu(heatmap)
    hm = heatmap.detach().to(torch.float32).contiguous()
    h, w = hm.shape
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device=hm.device)
    check(lib().cp360_overlay_colorize(ptr(hm), h, w, int(bool(square)), ptr(_lut(cmap, hm.device)), ptr(rgb), stream()))
    return rgb


def overlay(img, heatmap, cmap='jet', alpha=0.5, square=False):
    """utils/utils.py:9-25.  img: PIL image, uint8 ndarray [H, W, 3] or uint8 device tensor [H, W, 3];
    heatmap: float ndarray / tensor [h, w] (colorized first) or an RGB PIL image / uint8 [h, w, 3] array.
    Returns a PIL image (a device tensor when ``img`` was one).  ``square=True`` squares the map first
    (test_temporal.py:94 does ``equi_output ** 2`` before the call)."""
    as_tensor = torch.is_tensor(img)
    if as_tensor:
        require_gpu(img)
        im = img
    else:
        im = torch.from_numpy(np.ascontiguousarray(np.asarray(img.convert('RGB') if hasattr(img, 'convert') else img))).cuda()
    if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
        raise ValueError("img must be uint8 [H, W, 3]")
    im = im.contiguous()
    H, W = im.shape[:2]
    if hasattr(heatmap, 'convert'):
        hm = torch.from_numpy(np.ascontiguousarray(np.asarray(heatmap.convert('RGB')))).to(im.device)
    elif torch.is_tensor(heatmap) and heatmap.dtype == torch.uint8:
        hm = heatmap.to(im.device)
    elif isinstance(heatmap, np.ndarray) and heatmap.dtype == np.uint8 and heatmap.ndim == 3:
        hm = torch.from_numpy(np.ascontiguousarray(heatmap)).to(im.device)
    else:
        t = heatmap if torch.is_tensor(heatmap) else torch.from_numpy(np.ascontiguousarray(heatmap, dtype=np.float32))
        hm = colorize(t.to(im.device), cmap, square)
    h, w = hm.shape[:2]
    if (h, w) != (H, W):                               # heatmap.resize(img.size, resample=Image.CUBIC)
        key = (h, w, H, W, str(im.device))
        if key not in _RESIZERS:
            _RESIZERS[key] = LanczosResize((h, w), (H, W), device=im.device, filter='bicubic')
        hm = _RESIZERS[key](hm[None].contiguous())[0]
    out = torch.empty_like(im)
    check(lib().cp360_overlay_blend_u8(ptr(im), ptr(hm.contiguous()), ptr(out), im.numel(), float(alpha), stream()))
    if as_tensor:
        return out
    from PIL import Image
    return Image.fromarray(out.cpu().numpy())


def im_norm(in_img, mean, std):
    """utils/utils.py:28-33 - the host-side per-channel ``(x - mean) / std`` of the reference's static driver
    (dataset_feat_extractor.py:148-151), IN PLACE on the [H, W, 3] array like the reference (it returns the
    same object).  Host glue of the module boundary: the fused device path does this inside K1
    (``Equi2Cube.to_cube_batch``); this function exists so the reference's driver loop runs unchanged."""
    out_img = in_img
    for c in range(3):
        out_img[:, :, c] = (in_img[:, :, c] - mean[c]) / std[c]
    return out_img


def sigmoid(x):
    """utils/utils.py:36-37."""
    return 1 / (1 + np.exp(-x))


def cam_visual(input_equi, cam):
    """utils/utils.py:40-45: ``cam - min``, ``/ max``, ``np.uint8(255 * cam)`` in the INPUT's floating type (a
    float32 CAM is scaled and truncated in float32, as numpy does for the reference), then ``overlay``.  There the
    reference re-normalises the uint8 array (``- min`` = 0, ``/ max`` -> float64 ``k / 255``) and takes the FLOAT
    path of the colormap: entry ``int(k / 255 * 256)`` = ``k`` for k < 255 and the clipped 256 -> 255 for k = 255,
    i.e. exactly the table entry k - so the lookup below is the same RGB image without the round trip (a map
    whose uint8 maximum is below 255 is re-stretched by that second normalisation: handled the same way)."""
    cam = np.asarray(cam)
    if cam.dtype.kind != 'f':
        cam = cam.astype(np.float64)                      # numpy's true division of integer arrays
    cam = cam - np.min(cam)
    cam = cam / np.max(cam)
    idx = np.uint8(255 * cam)
    k = idx.astype(np.float64) / np.float64(idx.max()) if idx.max() != 255 else None
    if k is not None:                                     # the second normalisation is not the identity
        idx = np.minimum((k * 256).astype(np.int64), 255).astype(np.uint8)
    return overlay(input_equi, jet_lut()[idx])
