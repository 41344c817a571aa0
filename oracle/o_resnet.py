"""Oracle: ResNet-50-cubic forward to layer4 and the CAM GEMM (torch-CPU fp32).

Follows /root/reference/model/resnet_cubic.py:163-175 (stem + layer1..4),
:85-106 (Bottleneck), :145-161 (_make_layer / downsample) and
/root/reference/static_model/class_activation_model.py:46-52,70-83 (CAM).
Convolutions / batch-norm / max-pool are torch-CPU calls, exactly the third-party
kernels the reference itself invokes (nn.Conv2d, nn.BatchNorm2d in eval mode,
nn.MaxPool2d); CubePad is the index table of oracle.o_cubepad.
The state dict uses the reference's (= torchvision's) key names.
"""
import numpy as np
import torch
import torch.nn.functional as Fn

from .o_cubepad import cubepad_table

LAYERS = [3, 4, 6, 3]           # resnet_cubic.py:228-237 (resnet50)
PLANES = [64, 128, 256, 512]
_TAB_CACHE = {}


def cubepad_t(x, p):
    """CubePad(p) on a torch tensor [6N, C, n, n] via the index table."""
    n = x.shape[2]
    key = (n, p)
    if key not in _TAB_CACHE:
        _TAB_CACHE[key] = torch.from_numpy(cubepad_table(n, p, p, p, p).astype(np.int64))
    tab = _TAB_CACHE[key]
    g, C = x.shape[0] // 6, x.shape[1]
    xs = x.reshape(g, 6, C, n * n).permute(0, 2, 1, 3).reshape(g, C, 6 * n * n)
    out = xs[:, :, tab.reshape(-1)].reshape(g, C, 6, tab.shape[1], tab.shape[2])
    return out.permute(0, 2, 1, 3, 4).reshape(x.shape[0], C, tab.shape[1], tab.shape[2]).contiguous()


def _bn(x, sd, prefix):
    # nn.BatchNorm2d eval mode, eps = 1e-5 (torch default; resnet_cubic.py:124)
    return Fn.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                         sd[prefix + '.weight'], sd[prefix + '.bias'], False, 0.0, 1e-5)


def _bottleneck(x, sd, prefix, stride, has_ds):
    """resnet_cubic.py:85-106.  Stride lives on the 3x3 (conv2, :76-77); the
    downsample branch reads the unpadded block input (:101-102)."""
    out = Fn.relu(_bn(Fn.conv2d(x, sd[prefix + '.conv1.weight']), sd, prefix + '.bn1'))
    out = cubepad_t(out, 1)
    out = Fn.relu(_bn(Fn.conv2d(out, sd[prefix + '.conv2.weight'], stride=stride), sd, prefix + '.bn2'))
    out = _bn(Fn.conv2d(out, sd[prefix + '.conv3.weight']), sd, prefix + '.bn3')
    res = x
    if has_ds:
        res = _bn(Fn.conv2d(x, sd[prefix + '.downsample.0.weight'], stride=stride), sd, prefix + '.downsample.1')
    return Fn.relu(out + res)


def resnet50_layer4(x, sd, return_all=False):
    """x: torch float32 [6N, 3, H, W] -> layer4 features [6N, 2048, H/32, W/32].
    resnet_cubic.py:163-175; avgpool/fc (:176-178) are skipped - their result is
    discarded by CAM (class_activation_model.py:64)."""
    feats = {}
    with torch.no_grad():
        x = cubepad_t(x, 3)
        x = Fn.relu(_bn(Fn.conv2d(x, sd['conv1.weight'], stride=2), sd, 'bn1'))
        feats['stem'] = x
        x = cubepad_t(x, 1)
        x = Fn.max_pool2d(x, kernel_size=3, stride=2, padding=0)
        feats['pool'] = x
        for li, (nblk, planes) in enumerate(zip(LAYERS, PLANES), start=1):
            for b in range(nblk):
                stride = 2 if (b == 0 and li > 1) else 1
                x = _bottleneck(x, sd, 'layer%d.%d' % (li, b), stride, b == 0)
            feats['layer%d' % li] = x
    return (x, feats) if return_all else x


def cam_weight(fc_weight):
    """class_activation_model.py:46-52: squeeze; subtract the global min only if
    it is negative.  (Returns a copy: the reference's in-place ``-=`` aliases the
    model's fc.weight on the CPU path - a side effect, not a result.)"""
    w = np.squeeze(np.array(fc_weight, dtype=np.float32, copy=True))
    if np.min(w) < 0:
        w -= np.min(w)
    return w


def cam_scores(layer4, fc_weight):
    """class_activation_model.py:70-83: per face W[1000,2048] . feat[2048, h*w].
    layer4: ndarray [6N, 2048, h, w] -> [6N, 1000, h, w] float32."""
    w = cam_weight(fc_weight)
    bz, nc, h, ww = layer4.shape
    feats = np.asarray(layer4, dtype=np.float32).reshape(bz, nc, h * ww)
    out = np.stack([w.dot(feats[i]) for i in range(bz)])
    return out.reshape(bz, w.shape[0], h, ww)


def cam_from_cubes(batch_chw, sd):
    """dataset_feat_extractor.py:160-162 -> CAM(): [6,3,cd,cd] float32 (already
    normalised, CHW) -> (cube_score [6,1000,h,w], layer4 [6,2048,h,w])."""
    feat = resnet50_layer4(torch.from_numpy(np.ascontiguousarray(batch_chw)), sd).numpy()
    return cam_scores(feat, sd['fc.weight'].numpy()), feat
