"""``cp360_ctx`` (include/cp360.h, csrc/ctx.hip) from Python: one C call per network stage.

The context owns the packed / BatchNorm-folded weights and ALL launch planning of a stage (which fused kernel runs
where, tiles, split-K); the shims hand it the module's parameters once (``*_load``, repeated when a parameter
changes) and then call ``cp360_resnet_forward`` / ``cp360_clstm_step`` with caller-owned activations and a torch-
allocated workspace.  ``CP360_CTX=0`` keeps the launch planning in Python (ops.Conv / model/*: the per-kernel entry
points) - both paths issue the same launches and give the same bits (tests/test_ctx.py).
"""
import ctypes as C
import os
import weakref

import torch

from ._lib import ConvBn, check, dtype_code, lib, precision_dtype, ptr, require_gpu, stream

USE_CTX = os.environ.get('CP360_CTX', '1') != '0'


# Every registration of a Parameter / buffer / submodule on ANY nn.Module (``m.weight = Parameter(..)``, ``model.fc = nn.Linear(..)``,
# ``register_buffer``) bumps this counter through torch's global registration hooks: a cached tensor list made before such an
# assignment is walked again.  (load_state_dict, .to() and .cuda() keep the objects and change data_ptr / _version instead.)
_EPOCH = [0]


def _bump(*_):
    _EPOCH[0] += 1


try:
    from torch.nn.modules import module as _tm
    _tm.register_module_parameter_registration_hook(_bump)
    _tm.register_module_buffer_registration_hook(_bump)
    _tm.register_module_module_registration_hook(_bump)
    _HOOKED = True
except (ImportError, AttributeError):                        # an older torch: walk the module on every call
    _HOOKED = False


def _sentinels(module):
    """ids of the first parameter of the module and of each direct child: replaced wholesale by an ``_apply`` under
    torch.__future__.set_overwrite_module_params_on_conversion (which bypasses the registration hooks)."""
    return tuple(id(next(c.parameters(), None)) for c in [module] + list(module._modules.values()) if c is not None)


def _stamp(module, extra=()):
    """(storage, version) of every parameter and buffer: changes when weights are loaded, moved, cast or updated in place.
    The tensor list itself is cached on the module (walking ``parameters()`` / ``buffers()`` of ResNet-50 is ~1 ms per call,
    as much as one frame's static stage).  It is rebuilt when a cached tensor has changed storage, when anything has been
    registered on any module since (``model.fc = nn.Linear(..)``, ``m.weight = Parameter(..)``: the global epoch above), or when
    the first parameters of the module / its children are other objects than at caching time."""
    d = module.__dict__
    ts = d.get('_stage_tensors')
    if ts is not None and _HOOKED and d.get('_stage_epoch') == _EPOCH[0] and d.get('_stage_sentinels') == _sentinels(module):
        st = tuple((t.data_ptr(), t._version) for t in ts)
        if st == d.get('_stage_last'):
            return st + (d.get('_stage_ids'),) + tuple(extra)
    # first call, or something changed: walk the module again (.to() / .cuda() replace the BUFFER objects)
    ts = d['_stage_tensors'] = list(module.parameters()) + list(module.buffers())
    d['_stage_epoch'], d['_stage_sentinels'] = _EPOCH[0], _sentinels(module)
    st = d['_stage_last'] = tuple((t.data_ptr(), t._version) for t in ts)
    # (object identity is part of the key: a replaced Parameter that landed on the old one's storage address must not look unchanged)
    d['_stage_ids'] = hash(tuple(id(t) for t in ts))
    return st + (d['_stage_ids'],) + tuple(extra)


class StageCtx:
    """A ``cp360_ctx`` handle bound to one device (destroyed with the object)."""

    def __init__(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError("cp360 contexts live on a GPU (HIP); there is no CPU fallback")
        self.device = device
        h = C.c_void_p()
        check(lib().cp360_create(device.index if device.index is not None else torch.cuda.current_device(), C.byref(h)))
        self.h = h
        self._ws = {}

    def close(self):
        """Free the context's packed weights (raw hipMalloc: torch's caching allocator cannot reclaim them) and drop the
        cached workspaces.  Idempotent; also runs when the object is collected."""
        h, self.h = getattr(self, 'h', None), None
        self._ws = {}
        if h:
            try:
                lib().cp360_destroy(h)
            except Exception:
                pass

    def __del__(self):
        self.close()

    def workspace(self, key, nbytes):
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = t
        return t


def _f32(t):
    t = t.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


def _load_retry(call):
    """``call()`` -> status of a cp360_*_load.  The library's only hipMallocs happen there; when one fails while torch's
    caching allocator sits on freed blocks, release them and try once more."""
    rc = call()
    if rc == -7:                                           # CP360_ERR_HIP
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        rc = call()
    check(rc)


class _Stage:
    """Common part of the two stage objects.  The module is held WEAKLY: the module keeps its stage in
    ``module.__dict__['_stage']``, so a strong back reference would be a cycle and ``del engine`` would free nothing
    (f32 parameters, the context's packed weights, the workspaces) until the cyclic collector runs."""

    def __init__(self, module):
        self._module = weakref.ref(module)
        self.ctx = None
        self.stamp = None

    def _mod(self):
        m = self._module()
        if m is None:
            raise RuntimeError("the module of this stage context is gone")
        return m

    def close(self):
        """Release the context (packed weights, workspaces) now; the next call re-packs."""
        ctx, self.ctx, self.stamp = self.ctx, None, None
        if ctx is not None:
            ctx.close()


class ResnetStage(_Stage):
    """ResNet-50-cubic + CAM of ``model`` (model/resnet_cubic.py: ResNet) behind cp360_resnet_forward."""

    model = property(_Stage._mod)

    def _load(self):
        m = self.model
        dev = m.conv1.weight.device
        require_gpu(m.conv1.weight)
        stamp = _stamp(m, (m.precision, str(dev)))
        if self.ctx is not None and stamp == self.stamp:
            return
        if self.ctx is None or self.ctx.device != dev:
            self.ctx = StageCtx(dev)
        keep = []                                          # f32 device tensors alive until the packing kernels are queued

        def cb(conv, bn):
            ts = [_f32(conv.weight), _f32(bn.weight), _f32(bn.bias), _f32(bn.running_mean), _f32(bn.running_var)]
            keep.extend(ts)
            return ConvBn(*[t.data_ptr() for t in ts])
        pairs = [cb(m.conv1, m.bn1)]
        eps = m.bn1.eps
        for layer in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in layer:
                pairs += [cb(blk.conv1, blk.bn1), cb(blk.conv2, blk.bn2), cb(blk.conv3, blk.bn3)]
                if blk.downsample is not None:
                    pairs.append(cb(blk.downsample[0], blk.downsample[1]))
        arr = (ConvBn * len(pairs))(*pairs)
        fc = _f32(m.fc.weight)
        mn = float(fc.min())                               # class_activation_model.py:51-52 (one host sync per load)
        _load_retry(lambda: lib().cp360_resnet_load(self.ctx.h, dtype_code(precision_dtype(m.precision)), arr, len(pairs), ptr(fc),
                                                    int(fc.shape[0]), mn if mn < 0 else 0.0, float(eps), stream()))
        torch.cuda.current_stream().synchronize()          # the f32 sources (possibly temporaries) may go now
        self.stamp = stamp

    def describe(self, n_img, cube_dim):
        """cp360_resnet_plan_describe: the launch plan of the static stage for ``n_img`` faces of ``cube_dim``^2 as text - one
        line per layer saying whether a fused kernel or the generic per-convolution path runs (cube sizes other than 224 /
        512 and f32 take the generic path), and the tile / split-K choice of every generic launch."""
        self._load()
        buf = C.create_string_buffer(1 << 15)
        n = lib().cp360_resnet_plan_describe(self.ctx.h, int(n_img), int(cube_dim), buf, len(buf))
        check(n if n < 0 else 0)
        return buf.value.decode()

    def forward(self, faces_p3, cam_out=None, want_feat=True):
        """faces_p3 [6N, cd+6, cd+6, 4] (model dtype) -> (cam f32 [6N, h, w, classes], layer4 [6N, h, w, 2048] or None)."""
        self._load()
        m = self.model
        dt = precision_dtype(m.precision)
        require_gpu(faces_p3, cam_out)
        if faces_p3.dtype != dt or not faces_p3.is_contiguous() or faces_p3.dim() != 4 or faces_p3.shape[3] != 4 \
                or faces_p3.shape[1] != faces_p3.shape[2]:
            raise ValueError("faces_p3 must be a contiguous %s [6N, cd+6, cd+6, 4] tensor" % dt)
        n_img, cd = faces_p3.shape[0], faces_p3.shape[1] - 6
        hw, nc = cd // 32, int(m.fc.weight.shape[0])
        need = n_img * hw * hw * nc
        if cam_out is None:
            cam_out = torch.empty((n_img, hw, hw, nc), dtype=torch.float32, device=faces_p3.device)
        elif cam_out.dtype != torch.float32 or not cam_out.is_contiguous() or cam_out.numel() < need:
            raise ValueError("cam_out must be a contiguous f32 tensor of at least %d elements" % need)
        feat = torch.empty((n_img, hw, hw, 2048), dtype=dt, device=faces_p3.device) if want_feat else None
        nbytes = lib().cp360_resnet_workspace_bytes(self.ctx.h, n_img, cd)
        if nbytes == 0:
            raise ValueError("unsupported static-stage geometry: %d faces of %d^2" % (n_img, cd))
        ws = self.ctx.workspace(('resnet', n_img, cd), nbytes)
        check(lib().cp360_resnet_forward(self.ctx.h, ptr(faces_p3), n_img, cd, ptr(cam_out), ptr(feat), ptr(ws), ws.numel(),
                                         stream()))
        return cam_out.view(-1)[:need].view(n_img, hw, hw, nc), feat


class ClstmStage(_Stage):
    """``ConvLSTMCell`` (model/clstm.py) behind cp360_clstm_step, for one face size."""

    cell = property(_Stage._mod)

    def _load(self, face):
        c = self.cell
        dev = c.Conv1.weight.device
        require_gpu(c.Conv1.weight)
        stamp = _stamp(c, (c.precision, str(dev), int(face)))
        if self.ctx is not None and stamp == self.stamp:
            return
        if self.ctx is None or self.ctx.device != dev:
            self.ctx = StageCtx(dev)
        ts = [_f32(t) for t in (c.Conv1.weight, c.Conv1.bias, c.Conv2.weight, c.Conv2.bias, c.Gates.weight, c.Gates.bias)]
        _load_retry(lambda: lib().cp360_clstm_load(self.ctx.h, dtype_code(precision_dtype(c.precision)), *[ptr(t) for t in ts],
                                                   int(c.input_size), int(c.hidden_size), int(face), stream()))
        torch.cuda.current_stream().synchronize()
        self.stamp = stamp

    def _load_wino(self, B, w):
        """The Winograd-domain filters, packed the first time a launch shape asks for them (cp360_clstm_wino_state == 2)."""
        if lib().cp360_clstm_wino_state(self.ctx.h, int(B), int(w)) != 2:
            return
        c = self.cell
        ts = [_f32(t) for t in (c.Conv1.weight, c.Conv2.weight, c.Gates.weight)]
        _load_retry(lambda: lib().cp360_clstm_load_wino(self.ctx.h, *[ptr(t) for t in ts], stream()))
        torch.cuda.current_stream().synchronize()

    def step(self, xh, c_prev, c_next, h_f32=None, x_next=None):
        """One cell update on the fused layout (see ConvLSTMCell.step_nhwc); x_next = (cam, minmax, P, clip_stride, t_next)."""
        n6, w = xh.shape[0], xh.shape[1]
        self._load(w)
        c = self.cell
        dt = precision_dtype(c.precision)
        require_gpu(xh, c_prev, c_next, h_f32)
        if xh.dtype != dt or not xh.is_contiguous() or n6 % 6 or xh.shape[3] != c.input_size + c.hidden_size:
            raise ValueError("xh must be a contiguous %s [6B, w, w, Cin + H] tensor" % dt)
        for name, t in (('c_prev', c_prev), ('c_next', c_next), ('h_f32', h_f32)):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < n6 * w * w * c.hidden_size):
                raise ValueError("%s must be a contiguous f32 [6B, w, w, H] tensor" % name)
        B = n6 // 6
        self._load_wino(B, w)
        nbytes = lib().cp360_clstm_workspace_bytes(self.ctx.h, B, w)
        ws = self.ctx.workspace(('clstm', B, w), nbytes)
        xp, mm, stride = None, None, 0
        if x_next is not None:
            cam, minmax, P, clip_stride, t_next = x_next
            require_gpu(cam, minmax)
            if cam.dtype != torch.float32 or minmax.dtype != torch.float32 or P != 6 * w * w \
                    or cam.numel() < (B - 1) * clip_stride + (t_next + 1) * P * c.hidden_size or minmax.numel() < 2 * B:
                raise ValueError("x_next: cam f32 with frame t_next of every clip, minmax f32 [B, 2]")
            xp = C.c_void_p(cam.data_ptr() + 4 * t_next * P * c.hidden_size)
            mm, stride = ptr(minmax), int(clip_stride)
        check(lib().cp360_clstm_step(self.ctx.h, ptr(xh), ptr(c_prev), ptr(c_next), ptr(h_f32), B, w, xp, mm, stride, ptr(ws),
                                     ws.numel(), stream()))

    def window(self, cam, B, T, w, xh, cells, h_out, minmax, scratch, clip_stride=0, h_all=None):
        """cp360_clstm_window: min / max, hidden = cell = frame 0, T cell updates for B windows in lock step (one C call).
        cam f32 (window b at + b * clip_stride elements; 0 = dense [B, T, P, C]); h_out f32 [6B, w, w, H] = the final hidden
        state; h_all (optional) f32 [T, 6B, w, w, H] = the hidden state after every step."""
        self._load(w)
        self._load_wino(B, w)
        c = self.cell
        dt = precision_dtype(c.precision)
        require_gpu(cam, xh, cells[0], cells[1], h_out, minmax, scratch, h_all)
        P, Cc, Hh = 6 * w * w, c.input_size, c.hidden_size
        stride = int(clip_stride) or T * P * Cc
        if cam.dtype != torch.float32 or not cam.is_contiguous() or cam.numel() < (B - 1) * stride + T * P * Cc:
            raise ValueError("cam must be a contiguous f32 tensor holding T frames of [6 w^2, Cin] per window")
        if xh.dtype != dt or not xh.is_contiguous() or tuple(xh.shape) != (6 * B, w, w, Cc + Hh):
            raise ValueError("xh must be a contiguous %s [6B, w, w, Cin + H] tensor" % dt)
        for name, t, n in (('cell0', cells[0], 6 * B * w * w * Hh), ('cell1', cells[1], 6 * B * w * w * Hh),
                           ('h_out', h_out, 6 * B * w * w * Hh), ('minmax', minmax, 2 * B), ('scratch', scratch, 512 * B),
                           ('h_all', h_all, T * 6 * B * w * w * Hh)):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < n):
                raise ValueError("%s must be a contiguous f32 tensor of at least %d elements" % (name, n))
        nbytes = lib().cp360_clstm_window_workspace_bytes(self.ctx.h, B, T, w)
        if nbytes == 0:
            raise ValueError("unsupported window geometry (needs input_size == hidden_size)")
        ws = self.ctx.workspace(('window', B, T, w), nbytes)
        check(lib().cp360_clstm_window(self.ctx.h, ptr(cam), int(clip_stride), B, T, w, ptr(xh), ptr(cells[0]), ptr(cells[1]),
                                       ptr(h_out), ptr(h_all), ptr(minmax), ptr(scratch), ptr(ws), ws.numel(), stream()))
