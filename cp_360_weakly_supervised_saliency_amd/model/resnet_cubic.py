"""ResNet-50 with cube-padded convolutions, module surface of
/root/reference/model/resnet_cubic.py (``resnet50(pretrained, **kw)``, ``ResNet``,
``Bottleneck``; state-dict keys = torchvision's), compute in libcp360.so.

The nn.Conv2d / nn.BatchNorm2d children exist only as parameter containers so that
``state_dict()`` / ``load_state_dict()`` / ``.cuda()`` / ``_modules.get('layer4')
.register_forward_hook`` behave exactly like the reference's model.  ``forward`` never
calls them: each block folds its eval-mode BatchNorm into per-channel scale / bias
(resnet_cubic.py:85-106 with BN eps 1e-5), packs the weights once for the MFMA
implicit-GEMM kernel (K3) and runs NHWC on the GPU with CubePad(1) fused into the 3x3
tile loader.  Packed weights are rebuilt when a parameter changes (load_state_dict,
.cuda()).  Activations between blocks are torch channels_last tensors (logical NCHW,
physical NHWC) so hooks and callers see the reference's shapes.

Inference only (the north star is the inference path): no autograd through the HIP ops.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from .._lib import PRECISIONS as _DTYPES
from .cube_pad import CubePad

__all__ = ['ResNet', 'Bottleneck', 'resnet50']

# A/B switch (tools, tests): False runs the downsample branch as its own convolution + residual add
FUSE_DOWNSAMPLE = True
FUSE_LAYER1 = True
FUSE_LAYER2 = True
FUSE_LAYER3 = True
FUSE_STEM_POOL = os.environ.get('CP360_FUSE_STEM_POOL', '1') != '0'    # A/B switch: 0 = stem kernel, then max-pool kernel
LAUNCH_ORDER = int(os.environ.get('CP360_LAUNCH_ORDER', '2'))   # 0: every launch ascending (A/B switch)
FUSE_LAYER2_NEXT = True    # the next identity block's conv1 chained onto the layer2 tail kernel (csrc/l2block.hip, NEXT)
FUSE_L2_FIRST = True       # layer2.0 after its conv1 as one launch (csrc/lfirst.hip): stride-2 conv2 -> conv3 + downsample
CHAIN_L1_L2 = True         # layer2.0's conv1 (256 -> 128) chained onto layer1's last tail kernel (csrc/l1block.hip, wide)
FUSE_L1_FIRST = os.environ.get('CP360_L1_FIRST', '1') != '0'   # layer1.0's own conv1 inside its tail kernel (csrc/l1block.hip, FIRST); 0 = its own launch



def _fold_bn(bn):
    """Eval-mode BatchNorm2d as y = x * scale + bias (scale = g / sqrt(var + eps), bias = b - mean * scale; f32, one
    rounding per operation): one-time parameter preparation by ``cp360_fold_bn`` on the device - the same folding
    ``cp360_resnet_load`` applies; the multiply into the weights happens in the pack kernel."""
    from .._lib import check, lib, ptr, require_gpu, stream
    ts = [t.detach().float().contiguous() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
    require_gpu(*ts)
    n = ts[0].numel()
    scale = torch.empty(n, dtype=torch.float32, device=ts[0].device)
    bias = torch.empty_like(scale)
    check(lib().cp360_fold_bn(ptr(ts[0]), ptr(ts[1]), ptr(ts[2]), ptr(ts[3]), float(bn.eps), ptr(scale), ptr(bias), n, stream()))
    return scale, bias


def _stamp(module, extra=()):
    ts = list(module.parameters()) + list(module.buffers())
    return tuple((t.data_ptr(), t._version) for t in ts) + tuple(extra)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, cp=True):
        super(Bottleneck, self).__init__()
        if not cp:
            raise NotImplementedError("only cube padding is supported (the reference's ZeroPad is commented out, "
                                      "cube_pad.py:219-254)")
        self.pad = CubePad(1)
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=0, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.precision = 'fp32'
        self._plan = None
        self._plan_stamp = None

    def _plans(self):
        dt = _DTYPES[self.precision]
        stamp = _stamp(self, (self.precision,))
        if self._plan is None or stamp != self._plan_stamp:
            dev = self.conv1.weight.device
            s1, b1 = _fold_bn(self.bn1)
            s2, b2 = _fold_bn(self.bn2)
            s3, b3 = _fold_bn(self.bn3)
            plan = {
                'c1': ops.Conv(self.conv1.weight, s1, b1, 1, 0, True, dt, dev),
                'c2': ops.Conv(self.conv2.weight, s2, b2, self.stride, 1, True, dt, dev),   # CubePad(1) fused
            }
            if self.downsample is not None and FUSE_DOWNSAMPLE:
                # downsample (1x1 conv, stride, + BN on the block input, resnet_cubic.py:99-104) as conv3's
                # second source: relu(W3.mid + Wd.x + b3 + bd) in one tile, the [M, 4*planes] residual tensor
                # never exists in HBM
                sd, bd = _fold_bn(self.downsample[1])
                plan['c3'] = ops.Conv(self.conv3.weight, s3, b3, 1, 0, True, dt, dev,
                                      second=(self.downsample[0].weight, sd, bd, self.stride))
            else:
                plan['c3'] = ops.Conv(self.conv3.weight, s3, b3, 1, 0, True, dt, dev)       # relu after the add
                if self.downsample is not None:
                    sd, bd = _fold_bn(self.downsample[1])
                    plan['ds'] = ops.Conv(self.downsample[0].weight, sd, bd, self.stride, 0, False, dt, dev)
            self._plan, self._plan_stamp = plan, stamp
        return self._plan

    def forward_nhwc(self, x, mid=None):
        """``mid``: this block's conv1 + bn1 + relu output when the previous layer's tail kernel already computed it."""
        p = self._plans()
        out = p['c1'](x) if mid is None else mid
        out = p['c2'](out)
        if p['c3'].second is not None:
            return p['c3'](out, x2=x)              # conv3 + bn3 + downsample(x) + relu in one tile
        res = p['ds'](x) if 'ds' in p else x
        return p['c3'](out, residual=res)          # conv3 + bn3 + residual + relu in one epilogue

    def forward(self, x):
        with torch.no_grad():
            return ops.nchw_view(self.forward_nhwc(ops.as_nhwc(x, _DTYPES[self.precision])))


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, cp=True, precision='fp32'):
        self.inplanes = 64
        super(ResNet, self).__init__()
        if not cp:
            raise NotImplementedError("only cube padding is supported")
        self.cp = cp
        self.pad3 = CubePad(3)
        self.pad1 = CubePad(1)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=0, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=0)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        self._stem = None
        self._stem_stamp = None
        self._cam = None
        self._cam_stamp = None
        self.set_precision(precision)
        import math
        for m in self.modules():            # same init as resnet_cubic.py:137-143
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def set_precision(self, precision):
        if precision not in _DTYPES:
            raise ValueError("precision must be 'fp32', 'bf16' or 'fp16'")
        self.precision = precision
        for m in self.modules():
            if isinstance(m, Bottleneck):
                m.precision = precision
        return self

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, cp=self.cp)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, cp=self.cp))
        return nn.Sequential(*layers)

    # ---- fused-path pieces
    def _stem_conv(self):
        dt = _DTYPES[self.precision]
        stamp = _stamp(self.conv1) + _stamp(self.bn1, (self.precision,))
        if self._stem is None or stamp != self._stem_stamp:
            s, b = _fold_bn(self.bn1)
            self._stem = ops.Conv(self.conv1.weight, s, b, 2, 0, True, dt, self.conv1.weight.device, stem=True)
            self._stem_stamp = stamp
        return self._stem

    def cam_conv(self):
        """CAM as a 1x1 convolution with the shifted classifier weight
        (class_activation_model.py:46-52: W -= min(W) only when min(W) < 0)."""
        dt = _DTYPES[self.precision]
        stamp = _stamp(self.fc, (self.precision,))
        if self._cam is None or stamp != self._cam_stamp:
            w = self.fc.weight.detach().float().cpu()
            mn = w.min()
            if float(mn) < 0:
                w = w - mn
            self._cam = ops.Conv(w.reshape(w.shape[0], w.shape[1], 1, 1), None, None, 1, 0, False, dt,
                                 self.fc.weight.device)
            self._cam_stamp = stamp
        return self._cam

    def stem_nhwc(self, x_nhwc4, padded=False):
        """x [6N, H, W, 4] (normalised cube faces, 4th channel 0) -> [6N, H/4, W/4, 64]:
        CubePad(3) -> conv7x7 s2 + BN + ReLU -> CubePad(1) + maxpool (resnet_cubic.py:165-170).
        padded=True: x is already the CubePad(3) output [6N, H+6, W+6, 4] (Equi2Cube layout 'nhwc4p3')."""
        xp = x_nhwc4 if padded else ops.cubepad_nhwc(x_nhwc4, 3)
        conv = self._stem_conv()
        if FUSE_STEM_POOL:
            y = conv.stem_pool(xp)               # one kernel at cube 224 / 16-bit (csrc/stem.hip, stem_pool_kernel)
            if y is not None:
                return y
        return ops.cubepad_maxpool3s2(conv(xp))

    def _layer1_fused(self):
        """ops.L1Block per Bottleneck of layer1 (K3d), rebuilt when a parameter changes."""
        dt = _DTYPES[self.precision]
        l2c = self.layer2[0]
        chain2 = bool(CHAIN_L1_L2 and tuple(l2c.conv1.weight.shape[:2]) == (128, 256) and list(self.layer1)[-1].downsample is None)
        stamp = _stamp(self.layer1, (self.precision, chain2)) + (_stamp(l2c.conv1) + _stamp(l2c.bn1) if chain2 else ())
        if getattr(self, '_l1', None) is None or stamp != self._l1_stamp:
            dev = self.conv1.weight.device
            blks = list(self.layer1)
            out = []
            for k, b in enumerate(blks):
                c = lambda conv, bn: (conv.weight,) + _fold_bn(bn)
                nxt = c(blks[k + 1].conv1, blks[k + 1].bn1) if k + 1 < len(blks) else \
                    (c(l2c.conv1, l2c.bn1) if chain2 else None)
                ds = c(b.downsample[0], b.downsample[1]) if b.downsample is not None else None
                own = c(b.conv1, b.bn1) if (k == 0 and ds is not None and tuple(b.conv1.weight.shape[:2]) == (64, 64)) else None
                out.append(ops.L1Block(c(b.conv2, b.bn2), c(b.conv3, b.bn3), ds, nxt, dt, dev, own_conv1=own))
            self._l1, self._l1_stamp = out, stamp
        return self._l1

    def layer1_nhwc(self, x, want_next=False):
        """layer1 on the fused path.  16-bit types at 56x56 faces: conv1 of the first block, then ONE launch per
        Bottleneck (conv2 -> conv3 + residual / downsample -> the next block's conv1, csrc/l1block.hip).
        want_next: return (out, mid2) with mid2 = layer2.0's conv1 + bn1 + relu output when the last tail kernel
        computed it (else None)."""
        out, mid2 = self._layer1(x)
        return (out, mid2) if want_next else out

    def _layer1(self, x):
        dt = _DTYPES[self.precision]
        blks = list(self.layer1)
        if not (FUSE_LAYER1 and dt in (torch.float16, torch.bfloat16) and x.shape[1] == x.shape[2] and x.shape[1] in (56, 128)
                and blks[0].stride == 1 and blks[0].downsample is not None
                and all(b.downsample is None for b in blks[1:])):
            for blk in blks:
                x = blk.forward_nhwc(x)
            return x, None
        fused = self._layer1_fused()
        if FUSE_L1_FIRST and x.shape[1] == 56 and fused[0].w0 is not None:
            out, mid = fused[0].first(x)               # block 0's conv1 inside its tail kernel (round 6)
        else:
            mid = blks[0]._plans()['c1'](x)
            out, mid = fused[0](mid, x_ds=x)
        for k in range(1, len(blks)):
            out, mid = fused[k](mid, residual=out)
        return out, mid                               # mid: layer2.0's conv1 output (CHAIN_L1_L2), else None

    def layer2_nhwc(self, x, mid0=None):
        """layer2 on the fused path.  16-bit types at 28x28 output faces: the identity blocks run conv1 as a
        convolution and conv2 -> conv3 + residual as ONE launch (csrc/l2block.hip); the first (stride-2,
        downsample) block stays on the per-convolution path."""
        dt = _DTYPES[self.precision]
        blks = list(self.layer2)
        b0 = blks[0]
        if (FUSE_L2_FIRST and dt in (torch.float16, torch.bfloat16) and tuple(x.shape[1:]) == (56, 56, 256) and b0.stride == 2
                and b0.downsample is not None and tuple(b0.conv2.weight.shape) == (128, 128, 3, 3)):
            # the first block after its conv1 as ONE launch (K3f): stride-2 conv2 -> conv3 + downsample(x) + relu
            b1 = blks[1]
            stamp0 = _stamp(b0, (self.precision,)) + _stamp(b1.conv1) + _stamp(b1.bn1)
            if getattr(self, '_l2f', None) is None or stamp0 != self._l2f_stamp:
                c = lambda conv, bn: (conv.weight,) + _fold_bn(bn)
                nx = c(b1.conv1, b1.bn1) if tuple(b1.conv1.weight.shape[:2]) == (128, 512) else None
                self._l2f = ops.L2First(c(b0.conv2, b0.bn2), c(b0.conv3, b0.bn3), c(b0.downsample[0], b0.downsample[1]), dt,
                                        self.conv1.weight.device, next_conv1=nx)
                self._l2f_stamp = stamp0
            r = self._l2f(b0._plans()['c1'](x) if mid0 is None else mid0, x, chain=FUSE_LAYER2 and FUSE_LAYER2_NEXT)
            x, mid1 = r if isinstance(r, tuple) else (r, None)     # mid1: layer2.1's conv1 output
        else:
            x = b0.forward_nhwc(x, mid=mid0)           # mid0: conv1 of the first block, from layer1's last tail kernel
            mid1 = None
        if not (FUSE_LAYER2 and dt in (torch.float16, torch.bfloat16) and x.shape[1] == x.shape[2] and x.shape[1] in (28, 64)
                and all(b.downsample is None and b.stride == 1 for b in blks[1:])):
            for blk in blks[1:]:
                x = blk.forward_nhwc(x)
            return x
        stamp = _stamp(self.layer2, (self.precision,))
        if getattr(self, '_l2', None) is None or stamp != self._l2_stamp:
            c = lambda conv, bn: (conv.weight,) + _fold_bn(bn)
            ident = blks[1:]
            self._l2 = [ops.L2Block(c(b.conv2, b.bn2), c(b.conv3, b.bn3), dt, self.conv1.weight.device,
                                    next_conv1=c(ident[k + 1].conv1, ident[k + 1].bn1) if k + 1 < len(ident) else None)
                        for k, b in enumerate(ident)]
            self._l2_stamp = stamp
        chain = FUSE_LAYER2_NEXT and x.shape[1] == 28          # the next block's conv1 rides on the tail (28x28 faces)
        mid = mid1
        for k, (blk, tail) in enumerate(zip(blks[1:], self._l2)):
            if mid is None:
                mid = blk._plans()['c1'](x)
            if chain and tail.w1 is not None:
                x, mid = tail(mid, x)
            else:
                x, mid = tail(mid, x, chain=False), None
        return x

    def layer3_nhwc(self, x):
        """layer3 on the fused path.  16-bit types at 14x14 output faces: the identity blocks run conv1 as a convolution
        and conv2 -> conv3 + residual as ONE launch (csrc/l2block.hip at C = 256); the first (stride-2, downsample)
        block stays on the per-convolution path."""
        dt = _DTYPES[self.precision]
        blks = list(self.layer3)
        x = blks[0].forward_nhwc(x)
        if not (FUSE_LAYER3 and dt in (torch.float16, torch.bfloat16) and x.shape[1] == x.shape[2] and x.shape[1] in (14, 32)
                and all(b.downsample is None and b.stride == 1 for b in blks[1:])):
            for blk in blks[1:]:
                x = blk.forward_nhwc(x)
            return x
        stamp = _stamp(self.layer3, (self.precision,))
        if getattr(self, '_l3', None) is None or stamp != self._l3_stamp:
            c = lambda conv, bn: (conv.weight,) + _fold_bn(bn)
            self._l3 = [ops.L3Block(c(b.conv2, b.bn2), c(b.conv3, b.bn3), dt, self.conv1.weight.device) for b in blks[1:]]
            self._l3_stamp = stamp
        for blk, tail in zip(blks[1:], self._l3):
            x = tail(blk._plans()['c1'](x), x)
        return x

    def features_nhwc(self, x_nhwc4, padded=False):
        """Fused path to layer4: [6N, H, W, 4] -> [6N, H/32, W/32, 2048]."""
        # every launch walks its work items against the order of the launch before it: the consumer starts with the
        # lines its producer wrote last, still in the 256 MB Infinity Cache (ops.launch_order, tools/mall_probe.hip)
        with ops.launch_order(LAUNCH_ORDER):
            x = self.stem_nhwc(x_nhwc4, padded)
            x, mid2 = self.layer1_nhwc(x, want_next=True)
            x = self.layer2_nhwc(x, mid2)
            x = self.layer3_nhwc(x)
            for blk in self.layer4:
                x = blk.forward_nhwc(x)
        return x

    def forward(self, x):
        """x: [6N, 3, H, W] float tensor on the GPU (the reference's batch).  Runs the
        stem, then layer1..layer4 THROUGH the nn.Module call path so forward hooks fire
        (CAM hooks 'layer4').  Returns the layer4 features [6N, 2048, H/32, W/32]: the
        reference goes on to avgpool + fc (resnet_cubic.py:176-178) but its only caller
        on this path discards that result (class_activation_model.py:64), and fc cannot
        run at all for cube sizes other than 224."""
        with torch.no_grad():
            dt = _DTYPES[self.precision]
            ops.require_gpu(x)
            xh = ops.as_nhwc(x.float(), torch.float32)                      # [6N, H, W, 3]
            x4 = ops.cubepad_nhwc(xh, 0, c_out=4)                           # channel-pad to NHWC4
            if dt != torch.float32:
                x4 = ops.nchw_to_nhwc(x4.reshape(1, 1, 1, -1), out_dtype=dt).reshape(x4.shape)   # f32 -> bf16 cast
            y = ops.nchw_view(self.stem_nhwc(x4))
            y = self.layer1(y)
            y = self.layer2(y)
            y = self.layer3(y)
            y = self.layer4(y)
            return y

    def load_pretrained_model(self, pretrained_state_dict):
        """Copy by name, skip size mismatches, KeyError on unknown keys
        (resnet_cubic.py:183-201)."""
        own = self.state_dict()
        for name, param in pretrained_state_dict.items():
            if name not in own:
                raise KeyError("unexpected key '{}' in state_dict".format(name))
            if isinstance(param, nn.Parameter):
                param = param.data
            try:
                own[name].copy_(param)
            except Exception:
                print("skip loading key '{}' due to inconsistent size".format(name))
        self.load_state_dict(own)


# torchvision's public checkpoint of the architecture whose key names this module keeps (resnet_cubic.py:18-24)
model_urls = {'resnet50': 'https://download.pytorch.org/models/resnet50-19c8e357.pth'}


def resnet50(pretrained=False, **kwargs):
    """ResNet-50-cubic.  ``pretrained=True`` loads the ImageNet weights exactly as the reference does
    (resnet_cubic.py:228-237: ``model.load_pretrained_model(model_zoo.load_url(model_urls['resnet50']))`` - torch's
    hub cache is consulted first, so a checkpoint placed in ``$TORCH_HOME/hub/checkpoints`` works offline).  Only
    when neither the cache nor the network can provide the file does it raise, naming the alternative
    (``load_pretrained_model`` / ``load_state_dict`` with a state dict)."""
    model = ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)
    if pretrained:
        import torch.utils.model_zoo as model_zoo
        try:
            state = model_zoo.load_url(model_urls['resnet50'], map_location='cpu')
        except Exception as e:                       # no network and nothing cached
            raise RuntimeError("resnet50(pretrained=True): could not fetch %s (%s: %s); put the checkpoint into torch's "
                               "hub cache or call load_pretrained_model(state_dict)"
                               % (model_urls['resnet50'], type(e).__name__, e)) from e
        model.load_pretrained_model(state)
    return model
