"""Deterministic synthetic weights / frames / fixation maps (hash RNG, no network).

The reference's real weights are external downloads (ImageNet ResNet-50 URL at
/root/reference/model/resnet_cubic.py:18-24; the ConvLSTM checkpoint is a
Google-Drive link in README.md:20), so benchmarks and parity tests use random-init
weights of the same architecture, generated identically on every machine.
Distributions follow SURVEY.md section 8(d): convs He-normal with
std = sqrt(2 / (k*k*Cout)) as in resnet_cubic.py:137-143 and clstm.py:84-90.
"""
import math
import numpy as np

from . import hashrng

RESNET50_LAYERS = [3, 4, 6, 3]
RESNET50_PLANES = [64, 128, 256, 512]


def _conv(seed, cout, cin, k):
    return hashrng.normal(seed, (cout, cin, k, k), 0.0, math.sqrt(2.0 / (k * k * cout)))


def _bn(seed, c):
    return {
        'weight': hashrng.uniform(seed, (c,), 0.5, 1.5),
        'bias': hashrng.normal(seed + 1, (c,), 0.0, 0.1),
        'running_mean': hashrng.normal(seed + 2, (c,), 0.0, 0.1),
        'running_var': hashrng.uniform(seed + 3, (c,), 0.5, 1.5),
    }


def resnet50_state(seed=1, num_classes=1000):
    """dict name -> float32 ndarray with the reference's / torchvision's key names
    (conv1.weight, bn1.*, layerL.B.conv{1,2,3}.weight, ...bn{1,2,3}.*,
    ...downsample.{0.weight,1.*}, fc.{weight,bias})."""
    sd = {}
    s = [seed * 100000]

    def nxt():
        s[0] += 10
        return s[0]

    def put_bn(prefix, c):
        for k, v in _bn(nxt(), c).items():
            sd[prefix + '.' + k] = v

    sd['conv1.weight'] = _conv(nxt(), 64, 3, 7)
    put_bn('bn1', 64)
    inplanes = 64
    for li, (nblk, planes) in enumerate(zip(RESNET50_LAYERS, RESNET50_PLANES), start=1):
        for b in range(nblk):
            p = 'layer%d.%d' % (li, b)
            sd[p + '.conv1.weight'] = _conv(nxt(), planes, inplanes, 1)
            put_bn(p + '.bn1', planes)
            sd[p + '.conv2.weight'] = _conv(nxt(), planes, planes, 3)
            put_bn(p + '.bn2', planes)
            sd[p + '.conv3.weight'] = _conv(nxt(), planes * 4, planes, 1)
            put_bn(p + '.bn3', planes * 4)
            if b == 0:
                sd[p + '.downsample.0.weight'] = _conv(nxt(), planes * 4, inplanes, 1)
                put_bn(p + '.downsample.1', planes * 4)
            inplanes = planes * 4
    sd['fc.weight'] = hashrng.normal(nxt(), (num_classes, 2048), 0.0, 0.01)
    sd['fc.bias'] = hashrng.normal(nxt(), (num_classes,), 0.0, 0.01)
    return sd


def clstm_state(seed=2, input_size=1000, hidden_size=1000):
    """Keys Conv1/Conv2/Gates .weight/.bias (clstm.py:28-34)."""
    h4 = 4 * hidden_size
    base = seed * 100000
    return {
        'Conv1.weight': _conv(base + 10, h4, input_size + hidden_size, 3),
        'Conv1.bias': hashrng.normal(base + 20, (h4,), 0.0, 0.05),
        'Conv2.weight': _conv(base + 30, h4, h4, 3),
        'Conv2.bias': hashrng.normal(base + 40, (h4,), 0.0, 0.05),
        'Gates.weight': _conv(base + 50, h4, h4, 3),
        'Gates.bias': hashrng.normal(base + 60, (h4,), 0.0, 0.05),
    }


def _box_blur(a, taps=9):
    """Separable box filter along H and W (wrap in W, clamp in H): makes bilinear
    sampling of the synthetic frame non-trivial without being white noise."""
    r = taps // 2
    acc = np.zeros_like(a, dtype=np.float32)
    for d in range(-r, r + 1):
        acc += np.roll(a, d, axis=1)
    a = acc / taps
    H = a.shape[0]
    acc = np.zeros_like(a, dtype=np.float32)
    for d in range(-r, r + 1):
        idx = np.clip(np.arange(H) + d, 0, H - 1)
        acc += a[idx]
    return acc / taps


def frame_u8(seed, H, W):
    """[H, W, 3] uint8 equirectangular frame: blurred hash noise + a smooth
    analytic pattern 0.5 + 0.5 sin(k theta) cos(l phi) (seam / pole checks)."""
    noise = hashrng.uniform(seed, (H, W, 3), 0.0, 1.0)
    noise = _box_blur(noise)
    noise = (noise - noise.min()) / max(float(noise.max() - noise.min()), 1e-12)
    th = (np.arange(W, dtype=np.float32) + 0.5) / W * 2 * np.pi - np.pi
    ph = np.pi / 2 - (np.arange(H, dtype=np.float32) + 0.5) / H * np.pi
    pat = 0.5 + 0.5 * np.sin(3 * th)[None, :] * np.cos(2 * ph)[:, None]
    img = 0.6 * noise + 0.4 * pat[..., None]
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)


def clip_u8(seed, T, H, W):
    """[T, H, W, 3] uint8: one base frame drifting horizontally (cheap to make,
    temporally coherent like a panning 360 camera)."""
    base = frame_u8(seed, H, W)
    return np.stack([np.roll(base, 7 * t, axis=1) for t in range(T)])


def fixation_map(seed, H=960, W=1920, sigma=15.0):
    """Zero map with 20-40 Gaussian blobs; float32 [H, W]."""
    n = int(hashrng.integers(seed, (1,), 20, 41)[0])
    ys = hashrng.integers(seed + 1, (n,), 0, H)
    xs = hashrng.integers(seed + 2, (n,), 0, W)
    yy = np.arange(H, dtype=np.float32)[:, None]
    xx = np.arange(W, dtype=np.float32)[None, :]
    m = np.zeros((H, W), dtype=np.float32)
    for y, x in zip(ys, xs):
        m += np.exp(-((yy - y) ** 2 + (xx - x) ** 2) / (2 * sigma * sigma))
    return m


def fixations_from_map(sal, seed, H=960, W=1920, n_near=24, n_far=6, top_k=12, sigma=None, jitter=0.3):
    """Fixation map [H, W] float32 that is CORRELATED with a saliency map ``sal`` [h, w]
    (normally the oracle's): ``n_near`` Gaussian blobs at the centres of its ``top_k``
    highest cells, jittered by up to ``jitter`` cells, plus ``n_far`` blobs anywhere
    (sigma defaults to 0.3 of a cell).  A map
    judged against it scores AUC-Judd ~0.8-0.9 and CC >> 0, so |dAUC|, |dCC| <= 1e-3
    discriminates (random blobs put both metrics at chance level, where any map
    passes).  Deterministic given (sal, seed): ties are broken by flat index."""
    sal = np.asarray(sal, dtype=np.float64)
    h, w = sal.shape
    if sigma is None:
        sigma = 0.3 * H / float(h)
    order = np.argsort(-sal.reshape(-1), kind='stable')[:top_k]
    pick = hashrng.integers(seed, (n_near,), 0, top_k)
    jy = hashrng.uniform(seed + 1, (n_near,), -jitter, jitter, dtype=np.float64)
    jx = hashrng.uniform(seed + 2, (n_near,), -jitter, jitter, dtype=np.float64)
    cy = (order[pick] // w + 0.5 + jy) * (H / float(h))
    cx = (order[pick] % w + 0.5 + jx) * (W / float(w))
    ys = np.concatenate([np.clip(cy, 0, H - 1), hashrng.integers(seed + 3, (n_far,), 0, H).astype(np.float64)])
    xs = np.concatenate([np.mod(cx, W), hashrng.integers(seed + 4, (n_far,), 0, W).astype(np.float64)])
    yy = np.arange(H, dtype=np.float32)[:, None]
    xx = np.arange(W, dtype=np.float32)[None, :]
    m = np.zeros((H, W), dtype=np.float32)
    for y, x in zip(ys, xs):
        m += np.exp(-((yy - np.float32(y)) ** 2 + (xx - np.float32(x)) ** 2) / np.float32(2 * sigma * sigma))
    return m


def cam_clip(seed, T, w=7, C=1000, scale=1000.0):
    """Synthetic stand-in for T consecutive cube_feat arrays [T,6,C,w,w] float32
    (smooth in t) for temporal-stage tests that do not run the ResNet."""
    a = hashrng.normal(seed, (6, C, w, w), 0.0, 1.0)
    b = hashrng.normal(seed + 1, (6, C, w, w), 0.0, 1.0)
    ts = np.linspace(0.0, 1.0, T, dtype=np.float32)
    return np.stack([(scale * ((1 - t) * a + t * b + 1.5)).astype(np.float32) for t in ts])
