"""Functional layer over the C ABI (include/cp360.h).  Tensors named *_nhwc are plain
contiguous torch tensors of shape [N, H, W, C] - the layout the fused pipeline keeps in
HBM.  torch is used for allocation and stream ownership only; all arithmetic happens in
libcp360.so.  Every function here fails (RuntimeError / ImportError) without a GPU or
without the built library - there is no CPU fallback.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, check, dtype_code, lib, ptr, require_gpu, stream

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # dataset_feat_extractor.py:150-151
IMAGENET_STD = (0.229, 0.224, 0.225)


# ----------------------------------------------------------------------------- CubePad
def pads_of(lrtd_pad):
    """model/cube_pad.py:12-20,60-70: an int pads all four sides, else [l, r, t, d]."""
    if isinstance(lrtd_pad, (int, np.integer)):
        return (int(lrtd_pad),) * 4
    p_l, p_r, p_t, p_d = lrtd_pad
    return int(p_l), int(p_r), int(p_t), int(p_d)


def cubepad_table(n, lrtd_pad):
    """Host int32 [6, Hp, Wp] source-index table of the kernels (no GPU needed)."""
    pl, pr, pt, pd = pads_of(lrtd_pad)
    tab = np.empty((6, n + pt + pd, n + pl + pr), dtype=np.int32)
    check(lib().cp360_cubepad_table_host(n, pl, pr, pt, pd, tab.ctypes.data_as(C.c_void_p)))
    return tab


def cubepad_nchw(x, lrtd_pad):
    """x [6N, C, n, n] contiguous -> [6N, C, n+pt+pd, n+pl+pr] (cube_pad.py:28-42)."""
    require_gpu(x)
    if x.dim() != 4:
        raise ValueError("CubePad expects a 4-D tensor")
    if x.shape[2] != x.shape[3]:
        raise ValueError("CubePad needs square faces (cube_pad.py transposes strips)")
    pl, pr, pt, pd = pads_of(lrtd_pad)
    x = x.contiguous()
    n6, Cc, n, _ = x.shape
    y = torch.empty((n6, Cc, n + pt + pd, n + pl + pr), dtype=x.dtype, device=x.device)
    check(lib().cp360_cubepad_nchw(ptr(x), ptr(y), n6, Cc, n, pl, pr, pt, pd, x.element_size(), stream()))
    return y


def cubepad_nhwc(x, lrtd_pad, c_out=None):
    """x [6N, n, n, C] -> [6N, n+pt+pd, n+pl+pr, c_out or C] (extra channels zero)."""
    require_gpu(x)
    pl, pr, pt, pd = pads_of(lrtd_pad)
    n6, n, n2, Cc = x.shape
    if n != n2:
        raise ValueError("CubePad needs square faces")
    cy = Cc if c_out is None else int(c_out)
    y = torch.empty((n6, n + pt + pd, n + pl + pr, cy), dtype=x.dtype, device=x.device)
    check(lib().cp360_cubepad_nhwc(ptr(x), ptr(y), n6, Cc, cy, n, pl, pr, pt, pd, x.element_size(), stream()))
    return y


# ----------------------------------------------------------------------------- layout
def nchw_to_nhwc(x, out_dtype=None, out=None, coff=0):
    """[N, C, H, W] contiguous -> [N, H, W, C] (or channels [coff, coff+C) of ``out``)."""
    require_gpu(x, out)
    x = x.contiguous()
    N, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((N, H, W, Cc), dtype=out_dtype or x.dtype, device=x.device)
    ld = out.shape[3]
    check(lib().cp360_nchw_to_nhwc(ptr(x), ptr(out), N, Cc, H, W, dtype_code(x.dtype), dtype_code(out.dtype),
                                   ld, coff, stream()))
    return out


def nhwc_to_nchw(x, out_dtype=None, channels=None, coff=0):
    """[N, H, W, ld] (channels [coff, coff+channels)) -> [N, channels, H, W]."""
    require_gpu(x)
    N, H, W, ld = x.shape
    Cc = ld if channels is None else channels
    y = torch.empty((N, Cc, H, W), dtype=out_dtype or x.dtype, device=x.device)
    check(lib().cp360_nhwc_to_nchw(ptr(x), ptr(y), N, Cc, H, W, dtype_code(x.dtype), dtype_code(y.dtype),
                                   ld, coff, stream()))
    return y


def as_nhwc(x, dtype=None):
    """Accept the reference's [N, C, H, W] tensor in either memory format and return
    an [N, H, W, C] contiguous tensor of ``dtype`` (zero-copy when already so)."""
    require_gpu(x)
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous() and (dtype is None or dtype == x.dtype):
        return v
    if v.is_contiguous():
        x = x.contiguous()        # rare: physical NHWC but a dtype change is wanted
    return nchw_to_nhwc(x.contiguous(), out_dtype=dtype or x.dtype)


def nchw_view(x_nhwc):
    """Logical [N, C, H, W] view (torch channels_last strides) of an NHWC tensor."""
    return x_nhwc.permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------- K1 / K6
def equi2cube(frames, grid, cube_dim, out_dtype=torch.float32, layout='nhwc4', scale=None,
              mean=IMAGENET_MEAN, std=IMAGENET_STD, cv_fixed_point=True):
    """frames [F, H, W, 3] u8 / f32 (device), grid [6, cd, cd, 2] f32 (device).
    Returns [6F, 3, cd, cd] (layout 'nchw') or [6F, cd, cd, 4] (layout 'nhwc4').  ``grid`` may be any
    [6, n, n, 2] table of sampling points with n = cube_dim: passing the CubePad(3)-gathered grid
    ([6, cd+6, cd+6, 2], ``Equi2Cube.grid_p3``) with cube_dim = cd+6 yields the padded faces directly."""
    require_gpu(frames, grid)
    F, H, W, three = frames.shape
    if three != 3:
        raise ValueError("frames must be [F, H, W, 3]")
    frames = frames.contiguous()
    if scale is None:
        scale = 1.0 / 255.0 if frames.dtype == torch.uint8 else 1.0
    code = {'nchw': 0, 'nhwc4': 1}[layout]
    shape = (6 * F, 3, cube_dim, cube_dim) if code == 0 else (6 * F, cube_dim, cube_dim, 4)
    out = torch.empty(shape, dtype=out_dtype, device=frames.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(np.float32(1.0) / np.float32(v)) for v in std])
    check(lib().cp360_equi2cube(ptr(frames), ptr(grid), ptr(out), F, H, W, cube_dim, m, s, float(scale),
                                dtype_code(frames.dtype), dtype_code(out_dtype), code,
                                1 if cv_fixed_point else 0, stream()))
    return out


def cube2equi(x, face_map, coord, layout='nchw', want_full=True, want_max=False):
    """x f32 [6B, C, w, w] ('nchw') or [6B, w, w, C] ('nhwc'); face_map int8 [2w,4w];
    coord f32 [2w,4w,2] pixel-space sampling positions.  Returns (full, max)."""
    require_gpu(x, face_map, coord)
    # the C ABI reads raw pointers as f32 / int8 / f32: anything else would be silently reinterpreted
    if x.dtype != torch.float32:
        raise ValueError("cube2equi samples f32 features (got %s): pass the f32 hidden state" % x.dtype)
    if face_map.dtype != torch.int8 or coord.dtype != torch.float32:
        raise ValueError("face_map must be int8 and coord f32 (got %s, %s)" % (face_map.dtype, coord.dtype))
    x = x.contiguous()
    face_map, coord = face_map.contiguous(), coord.contiguous()
    if layout == 'nchw':
        n6, Cc, w, _ = x.shape
    else:
        n6, w, _, Cc = x.shape
    if n6 % 6:
        raise ValueError("batch must be a multiple of 6 faces")
    B = n6 // 6
    full = torch.empty((B, Cc, 2 * w, 4 * w), dtype=torch.float32, device=x.device) if want_full else None
    mx = torch.empty((B, 2 * w, 4 * w), dtype=torch.float32, device=x.device) if want_max else None
    check(lib().cp360_cube2equi(ptr(x), ptr(face_map), ptr(coord), ptr(full), ptr(mx), B, Cc, w,
                                0 if layout == 'nchw' else 1, stream()))
    return full, mx


# ----------------------------------------------------------------------------- convolution
def _check_buf(name, t, dtype, min_shape=None, numel=None):
    """A caller-provided destination / operand handed to the library as a raw pointer: wrong dtype,
    strides or size would be an out-of-bounds device access, so fail in Python first."""
    if t is None:
        return
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if min_shape is not None:
        if t.dim() != len(min_shape) or any(a != b for a, b in zip(t.shape[:-1], min_shape[:-1])) \
                or t.shape[-1] < min_shape[-1]:
            raise ValueError("%s has shape %s, needs %s (last dimension at least)" % (name, tuple(t.shape), tuple(min_shape)))
    if numel is not None and t.numel() < numel:
        raise ValueError("%s holds %d elements, needs %d" % (name, t.numel(), numel))


# Optional launch timer (bench.py installs one): an object with
# ``wrap(tag, flops, fn)`` that brackets the conv_forward launch with HIP events on the
# current stream.  None = no instrumentation (default).
LAUNCH_TIMER = None


class Conv:
    """One convolution of the path with its packed weights resident on the device.

    weight: f32 [c_out, c_in, kh, kw] (torch, any device); ``scale`` / ``bias``: per
    output channel f32 (BatchNorm folding happens in the pack kernel: w * scale[n]).
    pad > 0 means "preceded by CubePad(pad)" (fused into the tile loader).
    """

    def __init__(self, weight, scale=None, bias=None, stride=1, pad=0, relu=False, dtype=torch.float32,
                 device='cuda', stem=False, clip_resident=None, second=None):
        self.dtype = dtype
        self.device = torch.device(device)
        self.stride = int(stride)
        self.pad = int(pad)
        self.relu = bool(relu)
        self.stem = bool(stem)
        self.tag = 'conv'
        self._w_src = weight.detach()          # packed lazily per layout (no copy kept: the module owns it)
        self.c_out, self.c_in_w, self.kh_w, self.kw_w = weight.shape
        if stem:
            assert (self.c_in_w, self.kh_w, self.kw_w) == (3, 7, 7)
            self.c_in, self.kh, self.kw, self.pix_stride = 32, 7, 1, 4
        else:
            self.c_in, self.kh, self.kw, self.pix_stride = self.c_in_w, self.kh_w, self.kw_w, self.c_in_w
        self.bias = None if bias is None else bias.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self._scale = None if scale is None else scale.detach().to(device=self.device, dtype=torch.float32).contiguous()
        # cp360_conv_desc.clip_resident: CubePad(1) + 3x3 stride 1 with c_out >= 256 may run on the
        # clip-resident kernel when the faces turn out to be <= 7x7 at call time (ConvLSTM at cube 224,
        # layer4 conv2); it reads a channel-major packing, made on first use
        if clip_resident is None:
            clip_resident = (not stem and self.pad == 1 and self.stride == 1 and self.kh == 3 and self.kw == 3
                             and self.c_out >= 256)
        self.clip_resident_ok = bool(clip_resident)
        self.clip_only = False                  # True: never ask cp360_conv_prefer_clip (only the channel-major packing exists)
        # second source (cp360_conv_desc.c_in2): (weight2 [c_out, c_in2, 1, 1], scale2, bias2, stride2) - a 1x1
        # convolution of a SECOND tensor accumulated into the same tile (the Bottleneck's downsample branch
        # inside conv3); its bias is folded into this conv's bias
        self.second = None
        if second is not None:
            w2, s2, b2, st2 = second
            if stem or self.c_out < 256 or w2.shape[0] != self.c_out or tuple(w2.shape[2:]) != (1, 1):
                raise ValueError("a second source needs c_out >= 256 and a [c_out, c_in2, 1, 1] filter")
            self.second = (w2.detach(), None if s2 is None else s2.detach().to(device=self.device, dtype=torch.float32).contiguous(),
                           int(st2))
            self.c_in2 = int(w2.shape[1])
            if b2 is not None:
                b2 = b2.detach().to(device=self.device, dtype=torch.float32)
                self.bias = b2.contiguous() if self.bias is None else (self.bias + b2).contiguous()
            clip_resident = False
            self.clip_resident_ok = False
        self._packed = {}
        self._stem_packed = None                # resident-patch stem kernel (16-bit types, cube 224), on first use
        self._band_packed = None                # resident-band 3x3 kernel (64 -> 64 on 56x56 faces), on first use
        self._partial = None
        self._splits_cache = {}

    def packed_for(self, clip_resident):
        """Packed weights in the layout of cp360_conv_desc.clip_resident (0 / 1), built on first use."""
        t = self._packed.get(clip_resident)
        if t is None:
            d = self._desc(6, 7, 7, 1, clip_resident=clip_resident) if clip_resident else \
                self._desc(6, max(self.kh, 8), max(self.kw, 8) if not self.stem else 16, 1)
            nbytes = lib().cp360_conv_packed_bytes(C.byref(d))
            if nbytes == 0:
                raise ValueError("unsupported convolution geometry for libcp360")
            w = self._w_src.to(device=self.device, dtype=torch.float32).contiguous()
            t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            if self.second is not None:
                w2 = self.second[0].to(device=self.device, dtype=torch.float32).reshape(self.c_out, self.c_in2).contiguous()
                check(lib().cp360_conv_pack_weights2(C.byref(d), ptr(w), ptr(self._scale), ptr(w2), ptr(self.second[1]),
                                                     ptr(t), stream()))
            else:
                check(lib().cp360_conv_pack_weights(C.byref(d), ptr(w), ptr(self._scale), ptr(t), 1 if self.stem else 0,
                                                    stream()))
            self._packed[clip_resident] = t
        return t

    @property
    def packed(self):
        return self.packed_for(0)               # the tap-major packing every kernel but the clip-resident one reads

    def out_hw(self, h_in, w_in):
        p2 = 2 * self.pad
        return (h_in + p2 - self.kh) // self.stride + 1, \
               ((w_in + p2 - self.kw_w) // self.stride + 1 if self.stem else (w_in + p2 - self.kw) // self.stride + 1)

    def _desc(self, n_img, h_in, w_in, splits, ld_out=None, out_coff=0, ld_res=0, relu=None, tile_px=0,
              clip_resident=0, slab_rows=0, x2_shape=None):
        d = ConvDesc()
        d.dtype = dtype_code(self.dtype)
        d.n_img, d.h_in, d.w_in = n_img, h_in, w_in
        d.c_in, d.pix_stride = self.c_in, self.pix_stride
        d.kh, d.kw, d.sy, d.sx = self.kh, self.kw, self.stride, self.stride
        d.h_out, d.w_out = self.out_hw(h_in, w_in)
        d.c_out = self.c_out
        d.pad_mode, d.pad = (1 if self.pad > 0 else 0), self.pad
        d.ld_out = self.c_out if ld_out is None else ld_out
        d.out_coff, d.ld_res = out_coff, ld_res
        d.relu = int(self.relu if relu is None else relu)
        d.splits = splits
        d.tile_px = tile_px
        d.clip_resident = clip_resident
        d.slab_rows = slab_rows
        if self.second is not None:
            st2 = self.second[2]
            d.c_in2 = self.c_in2
            d.sy2 = d.sx2 = st2
            if x2_shape is None:                     # packing: any consistent geometry (the layout does not depend on it)
                d.h_in2, d.w_in2, d.pix_stride2 = (d.h_out - 1) * st2 + 1, (d.w_out - 1) * st2 + 1, self.c_in2
            else:
                d.h_in2, d.w_in2, d.pix_stride2 = x2_shape[1], x2_shape[2], x2_shape[3]
        return d

    def _stem_resident(self, xp):
        """cp360_stem_forward: the padded faces' rows of an 8-row output band stay in LDS (K3a)."""
        L = lib()
        code = dtype_code(self.dtype)
        if self._stem_packed is None:
            w = self._w_src.to(device=self.device, dtype=torch.float32).contiguous()
            t = torch.empty(L.cp360_stem_packed_bytes(code), dtype=torch.uint8, device=self.device)
            check(L.cp360_stem_pack_weights(code, ptr(w), ptr(self._scale), ptr(t), stream()))
            self._stem_packed = t
        n_img, cd = xp.shape[0], xp.shape[1] - 6
        out = torch.empty((n_img, cd // 2, cd // 2, 64), dtype=self.dtype, device=xp.device)
        check(L.cp360_stem_forward(code, ptr(xp.contiguous()), ptr(self._stem_packed), ptr(self.bias), ptr(out), n_img,
                                   cd, int(self.relu), stream()))
        return out

    def stem_pool(self, xp):
        """cp360_stem_pool_forward: stem + CubePad(1) + max-pool 3x3 s2 in one kernel (cube 224, 16-bit types, ReLU):
        xp [n_img, 230, 230, 4] -> [n_img, 56, 56, 64]; None when this convolution / input is not that case."""
        if not (self.stem and self.relu and self.dtype in (torch.bfloat16, torch.float16) and xp.dtype == self.dtype
                and tuple(xp.shape[1:]) == (230, 230, 4) and xp.shape[0] % 6 == 0):
            return None
        require_gpu(xp)
        L = lib()
        code = dtype_code(self.dtype)
        if self._stem_packed is None:
            w = self._w_src.to(device=self.device, dtype=torch.float32).contiguous()
            t = torch.empty(L.cp360_stem_packed_bytes(code), dtype=torch.uint8, device=self.device)
            check(L.cp360_stem_pack_weights(code, ptr(w), ptr(self._scale), ptr(t), stream()))
            self._stem_packed = t
        n_img = xp.shape[0]
        y = torch.empty((n_img, 56, 56, 64), dtype=self.dtype, device=xp.device)
        border = torch.empty(L.cp360_stem_pool_border_bytes(n_img), dtype=torch.uint8, device=xp.device)
        check(L.cp360_stem_pool_forward(code, ptr(xp.contiguous()), ptr(self._stem_packed), ptr(self.bias), ptr(y),
                                        ptr(border), n_img, 224, stream()))
        return y

    def _band_resident(self, x):
        """cp360_band3x3_forward: layer1 conv2 with the band's padded pixels resident in LDS (K3c)."""
        L = lib()
        code = dtype_code(self.dtype)
        if self._band_packed is None:
            w = self._w_src.to(device=self.device, dtype=torch.float32).contiguous()
            t = torch.empty(L.cp360_band3x3_packed_bytes(code), dtype=torch.uint8, device=self.device)
            check(L.cp360_band3x3_pack_weights(code, ptr(w), ptr(self._scale), ptr(t), stream()))
            self._band_packed = t
        out = torch.empty_like(x)
        check(L.cp360_band3x3_forward(code, ptr(x.contiguous()), ptr(self._band_packed), ptr(self.bias), ptr(out),
                                      x.shape[0], 56, 64, int(self.relu), stream()))
        return out

    def nsteps(self):
        bk = 32 if self.dtype == torch.float32 else 64
        return self.kh * self.kw * ((self.c_in + bk - 1) // bk)

    def raw_sum_f32(self, x, out=None):
        """The convolution's raw f32 sums [n_img, h_out, w_out, c_out] (no bias / activation, true channel order) - how the
        CAM scores leave (class_activation_model.py:77-83).  The library's planner may split K (one frame = 6 faces gives a
        CAM GEMM of M = 294 rows: 80 small tiles for 256 CUs); the slabs are then reduced by cp360_conv_finish with an f32
        descriptor.  ``out``: f32 buffer of >= M * c_out elements (flat), else a fresh tensor."""
        n_img, h_in, w_in, _ = x.shape
        h_out, w_out = self.out_hw(h_in, w_in)
        M = n_img * h_out * w_out
        key = (n_img, h_in, w_in, 0)
        splits = self._splits_cache.get(key)
        if splits is None:
            splits = lib().cp360_conv_suggest_splits(C.byref(self._desc(n_img, h_in, w_in, 1)))
            self._splits_cache[key] = splits
        if splits == 1:
            part, _ = self(x, raw_f32=True, splits=1, partial_buf=out, clip_resident=False)
            return part[: M * self.c_out].view(n_img, h_out, w_out, self.c_out)
        part, _ = self(x, raw_f32=True, splits=splits, clip_resident=False)
        if out is None:
            out = torch.empty(M * self.c_out, dtype=torch.float32, device=x.device)
        else:
            require_gpu(out)
            _check_buf('out', out, torch.float32, numel=M * self.c_out)
        d = self._desc(n_img, h_in, w_in, splits, relu=False)
        d.dtype = dtype_code(torch.float32)                     # the finish writes f32 whatever the convolution's type
        check(lib().cp360_conv_finish(C.byref(d), ptr(part), None, None, ptr(out), stream()))
        return out.view(-1)[: M * self.c_out].view(n_img, h_out, w_out, self.c_out)

    def __call__(self, x, residual=None, out=None, out_coff=0, raw_f32=False, splits=None, partial_buf=None,
                 tile_px=0, clip_resident=None, slab_rows=False, x2=None):
        """x [n_img, h, w, c] NHWC (for the stem: the materialised CubePad(3) output
        [n_img, h+6, w+6, 4]).  Returns [n_img, h_out, w_out, c_out] in self.dtype, or
        with raw_f32=True the (partial [splits, M, c_out] f32, splits) pair whose
        reduction / bias / activation the caller finishes (cp360_lstm_gates, CAM)."""
        require_gpu(x, residual, out, x2)
        n_img, h_in, w_in, cx = x.shape
        if (x2 is not None) != (self.second is not None):
            raise ValueError("x2 goes with a conv built with second=...")
        if (self.stem and h_in == w_in and h_in in (230, 518) and cx == 4 and residual is None and out is None and not raw_f32
                and splits is None and tile_px == 0 and self.dtype in (torch.bfloat16, torch.float16)
                and x.dtype == self.dtype):
            return self._stem_resident(x)
        if (not self.stem and self.c_in == 64 and self.c_out == 64 and self.kh == 3 and self.kw == 3 and self.pad == 1
                and self.stride == 1 and h_in == 56 and w_in == 56 and n_img % 6 == 0 and residual is None and out is None
                and not raw_f32 and splits is None and tile_px == 0 and self.dtype in (torch.bfloat16, torch.float16)
                and x.dtype == self.dtype):
            return self._band_resident(x)
        if x.dtype != self.dtype:
            raise ValueError("activation dtype %s != conv dtype %s" % (x.dtype, self.dtype))
        if not self.stem and cx != self.c_in:
            raise ValueError("input has %d channels, conv expects %d" % (cx, self.c_in))
        if self.stem and cx != 4:
            raise ValueError("stem expects an NHWC4 input")
        h_out, w_out = self.out_hw(h_in, w_in)
        M = n_img * h_out * w_out
        x = x.contiguous()
        x2s = None
        if x2 is not None:
            st2 = self.second[2]
            if x2.dtype != self.dtype or not x2.is_contiguous() or x2.dim() != 4 or x2.shape[0] != n_img \
                    or x2.shape[3] < self.c_in2 or (h_out - 1) * st2 >= x2.shape[1] or (w_out - 1) * st2 >= x2.shape[2]:
                raise ValueError("x2 must be a contiguous %s [n_img, >=%d, >=%d, >=%d] tensor"
                                 % (self.dtype, (h_out - 1) * st2 + 1, (w_out - 1) * st2 + 1, self.c_in2))
            x2s = tuple(x2.shape)
        _check_buf('out', out, self.dtype, (n_img, h_out, w_out, self.c_out + out_coff))
        _check_buf('residual', residual, self.dtype, (n_img, h_out, w_out, self.c_out))
        ld_out = self.c_out if out is None else out.shape[3]
        ld_res = 0 if residual is None else residual.shape[3]
        L = lib()
        # clip-resident kernel: whole cubes of <= 304 pixels (faces <= 7x7), half a cube of 8x8 faces, or one 16x16 face + ring per tile
        cr = self.clip_resident_ok if clip_resident is None else bool(clip_resident)
        cr = int(cr and h_in == w_in and (6 * h_in * w_in <= 304 or h_in in (8, 16)) and n_img % 6 == 0 and tile_px == 0)
        if clip_resident and not cr:
            raise ValueError("clip_resident needs CubePad(1)+3x3 stride 1 on faces of at most 7x7, 8x8 or 16x16")
        if cr and clip_resident is None and not self.clip_only:
            # both layouts can be packed: few cubes x few channels (layer4's conv2 of one frame) run better on the small
            # tap-major tiles (cp360_conv_prefer_clip; same rule in csrc/ctx.hip)
            key = ('clip', n_img, h_in, w_in)
            pref = self._splits_cache.get(key)
            if pref is None:
                pref = L.cp360_conv_prefer_clip(C.byref(self._desc(n_img, h_in, w_in, 1, clip_resident=1)))
                self._splits_cache[key] = pref
            cr = int(bool(pref))
        if splits is None:
            key = (n_img, h_in, w_in, cr)
            splits = self._splits_cache.get(key)
            if splits is None:
                splits = L.cp360_conv_suggest_splits(C.byref(self._desc(n_img, h_in, w_in, 1, clip_resident=cr, x2_shape=x2s)))
                self._splits_cache[key] = splits
        # split-K slabs read back by cp360_conv_finish (or, on request, by cp360_lstm_gates) keep the packed
        # row order: contiguous 64-byte stores (cp360_conv_desc.slab_rows)
        sr = int(self.c_out % 32 == 0 and ((splits > 1 and not raw_f32) or (raw_f32 and slab_rows)))
        if raw_f32 and slab_rows and not sr:
            raise ValueError("slab_rows needs c_out % 32 == 0")
        d = self._desc(n_img, h_in, w_in, splits, ld_out, out_coff, ld_res, tile_px=tile_px, clip_resident=cr,
                       slab_rows=sr, x2_shape=x2s)
        packed = self.packed_for(cr)
        def forward(*a):
            if x2 is not None:                       # (desc, in, in2, packed, ...)
                return L.cp360_conv_forward2(a[0], a[1], ptr(x2), *a[2:])
            if LAUNCH_TIMER is None:
                return L.cp360_conv_forward(*a)
            flops = 2.0 * M * self.c_out * self.c_in_w * self.kh_w * self.kw_w
            return LAUNCH_TIMER.wrap(self.tag, flops, lambda: L.cp360_conv_forward(*a))

        if raw_f32 or splits > 1:
            need = splits * M * self.c_out
            if partial_buf is not None:          # caller-owned destination for the raw sums
                require_gpu(partial_buf)
                _check_buf('partial_buf', partial_buf, torch.float32, numel=need)
                check(forward(C.byref(d), ptr(x), ptr(packed), None, None, None,
                                           ptr(partial_buf), stream()))
                return partial_buf, splits
            if self._partial is None or self._partial.numel() < need:
                self._partial = torch.empty(need, dtype=torch.float32, device=x.device)
            check(forward(C.byref(d), ptr(x), ptr(packed), None, None, None,
                                       ptr(self._partial), stream()))
            if raw_f32:
                return self._partial, splits
            if out is None:
                out = torch.empty((n_img, h_out, w_out, self.c_out), dtype=self.dtype, device=x.device)
            check(L.cp360_conv_finish(C.byref(d), ptr(self._partial), ptr(self.bias), ptr(residual), ptr(out),
                                      stream()))
            return out
        if out is None:
            out = torch.empty((n_img, h_out, w_out, self.c_out), dtype=self.dtype, device=x.device)
        check(forward(C.byref(d), ptr(x), ptr(packed), ptr(self.bias), ptr(residual), ptr(out),
                                   None, stream()))
        return out


def frag_pack_1x1(weight, scale, dtype, order, device):
    """1x1 filter [n_out, k(, 1, 1)] (f32) times scale[n_out] -> MFMA A fragments (cp360_frag_pack_1x1)."""
    n_out, k = int(weight.shape[0]), int(weight.shape[1])
    code = dtype_code(dtype)
    nbytes = lib().cp360_frag_packed_bytes(code, n_out, k)
    if nbytes == 0:
        raise ValueError("fragment packing needs a 16-bit type and n_out, k multiples of 32")
    w = weight.detach().to(device=device, dtype=torch.float32).reshape(n_out, k).contiguous()
    sc = None if scale is None else scale.detach().to(device=device, dtype=torch.float32).contiguous()
    t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    check(lib().cp360_frag_pack_1x1(code, ptr(w), ptr(sc), ptr(t), n_out, k, int(order), stream()))
    return t


class L1Block:
    """K3d (csrc/l1block.hip): one layer1 Bottleneck after its conv1 as ONE launch - conv2 + bn2 + relu ->
    conv3 + bn3 + (residual | downsample(x)) + relu -> optionally the next block's conv1 + bn1 + relu.
    16-bit types, 56x56 faces.  Arguments: (weight, bn scale, bn bias) triples of the convolutions."""

    def __init__(self, conv2, conv3, downsample=None, next_conv1=None, dtype=torch.float16, device='cuda', own_conv1=None):
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("L1Block runs in fp16 / bf16")
        self.dtype, self.device = dtype, torch.device(device)
        L, code = lib(), dtype_code(dtype)
        f32 = lambda t: None if t is None else t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        w2, s2, b2 = conv2
        if tuple(w2.shape) != (64, 64, 3, 3) or tuple(conv3[0].shape[:2]) != (256, 64):
            raise ValueError("L1Block is layer1's geometry: conv2 64->64 3x3, conv3 64->256 1x1")
        self.w2 = torch.empty(L.cp360_l1block_conv2_bytes(code), dtype=torch.uint8, device=self.device)
        check(L.cp360_l1block_pack_conv2(code, ptr(f32(w2)), ptr(f32(s2)), ptr(self.w2), stream()))
        self.b2 = f32(b2)
        w3, s3, b3 = conv3
        self.w3 = frag_pack_1x1(w3, s3, dtype, 0, self.device)
        self.b3 = f32(b3)
        self.wd = None
        if downsample is not None:
            wd, sd, bd = downsample
            if tuple(wd.shape[:2]) != (256, 64):
                raise ValueError("layer1's downsample is 64->256, stride 1")
            self.wd = frag_pack_1x1(wd, sd, dtype, 0, self.device)
            self.b3 = (self.b3 + f32(bd)).contiguous()
        # own_conv1 (the first block): the block's OWN conv1 (64 -> 64) for first(): it then runs inside the kernel, on the resident patch
        self.w0 = self.b0 = None
        if own_conv1 is not None:
            w0, s0, b0 = own_conv1
            if tuple(w0.shape[:2]) != (64, 64) or downsample is None:
                raise ValueError("own_conv1 is layer1.0's conv1 (64 -> 64) and goes with the downsample form")
            self.w0 = frag_pack_1x1(w0, s0, dtype, 0, self.device)
            self.b0 = f32(b0)
        self.w1 = self.b1 = None
        self.next_c = 0
        if next_conv1 is not None:
            w1, s1, b1 = next_conv1
            # 256 -> 64: the next layer1 block's conv1; 256 -> 128: layer2.0's conv1 behind the LAST (identity) block
            if tuple(w1.shape[:2]) not in ((64, 256), (128, 256)) or (w1.shape[0] == 128 and downsample is not None):
                raise ValueError("the chained conv1 is 256->64, or 256->128 behind an identity block")
            self.w1 = frag_pack_1x1(w1, s1, dtype, 1, self.device)
            self.b1 = f32(b1)
            self.next_c = int(w1.shape[0])

    def __call__(self, mid, residual=None, x_ds=None):
        """mid [n_img, n, n, 64] with n = 56 (cube 224) or 128 (cube 512); residual [n_img, n, n, 256] (identity blocks)
        or x_ds [n_img, n, n, 64] (the first block).  Returns (out [n_img, n, n, 256], next conv1's output
        [n_img, n, n, 64] or None)."""
        require_gpu(mid, residual, x_ds)
        if (self.wd is None) != (x_ds is None) or (residual is None) == (x_ds is None):
            raise ValueError("identity blocks take residual=, the downsample block takes x_ds=")
        n_img, n = mid.shape[0], mid.shape[1]
        if n not in (56, 128):
            raise ValueError("L1Block handles 56x56 and 128x128 faces")
        _check_buf('mid', mid, self.dtype, (n_img, n, n, 64))
        _check_buf('residual', residual, self.dtype, (n_img, n, n, 256))
        _check_buf('x_ds', x_ds, self.dtype, (n_img, n, n, 64))
        for t in (mid, residual, x_ds):
            if t is not None and t.shape[3] not in (64, 256):
                raise ValueError("dense NHWC tensors only")
        out = torch.empty((n_img, n, n, 256), dtype=self.dtype, device=mid.device)
        nxt = None if self.w1 is None else torch.empty((n_img, n, n, self.next_c), dtype=self.dtype, device=mid.device)
        if self.next_c == 128:
            check(lib().cp360_l1block_forward_wide(dtype_code(self.dtype), ptr(mid), ptr(self.w2), ptr(self.b2), ptr(self.w3),
                                                   ptr(self.b3), ptr(residual), ptr(out), ptr(self.w1), ptr(self.b1),
                                                   ptr(nxt), n_img, n, stream()))
            return out, nxt
        check(lib().cp360_l1block_forward(dtype_code(self.dtype), ptr(mid), ptr(self.w2), ptr(self.b2), ptr(self.w3),
                                          ptr(self.b3), ptr(residual), ptr(x_ds), ptr(self.wd), ptr(out), ptr(self.w1),
                                          ptr(self.b1), ptr(nxt), n_img, n, stream()))
        return out, nxt

    def first(self, x):
        """layer1.0 with its own conv1 inside the kernel (cp360_l1block_forward_first): x [n_img, 56, 56, 64] = the block input (conv1's
        input and the downsample source).  Returns (out, next conv1's output or None); same bits as conv1 as a launch + __call__."""
        require_gpu(x)
        if self.w0 is None or self.wd is None:
            raise ValueError("first() needs own_conv1= and downsample=")
        n_img, n = x.shape[0], x.shape[1]
        if n != 56:
            raise ValueError("the in-kernel conv1 exists for 56x56 faces")
        _check_buf('x', x, self.dtype, (n_img, n, n, 64))
        out = torch.empty((n_img, n, n, 256), dtype=self.dtype, device=x.device)
        nxt = None if self.w1 is None else torch.empty((n_img, n, n, self.next_c), dtype=self.dtype, device=x.device)
        check(lib().cp360_l1block_forward_first(dtype_code(self.dtype), ptr(x), ptr(self.w0), ptr(self.b0), ptr(self.w2), ptr(self.b2),
                                                ptr(self.w3), ptr(self.b3), ptr(self.wd), ptr(out), ptr(self.w1), ptr(self.b1),
                                                ptr(nxt), n_img, n, stream()))
        return out, nxt


class L2Block:
    """K3e (csrc/l2block.hip): the tail of a layer2 identity Bottleneck as ONE launch - conv2 + bn2 + relu ->
    conv3 + bn3 + residual + relu (-> the next identity block's conv1 + bn1 + relu when `next_conv1` is given, 28x28
    faces).  16-bit types, 28x28 / 64x64 faces.  (weight, bn scale, bn bias) triples."""

    def __init__(self, conv2, conv3, dtype=torch.float16, device='cuda', next_conv1=None):
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("L2Block runs in fp16 / bf16")
        self.dtype, self.device = dtype, torch.device(device)
        L, code = lib(), dtype_code(dtype)
        f32 = lambda t: None if t is None else t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        w2, s2, b2 = conv2
        w3, s3, b3 = conv3
        if tuple(w2.shape) != (128, 128, 3, 3) or tuple(w3.shape[:2]) != (512, 128):
            raise ValueError("L2Block is layer2's geometry: conv2 128->128 3x3, conv3 128->512 1x1")
        self.w2 = torch.empty(L.cp360_l2block_packed_bytes(code), dtype=torch.uint8, device=self.device)
        check(L.cp360_l2block_pack_weights(code, ptr(f32(w2)), ptr(f32(s2)), ptr(self.w2), stream()))
        self.b2 = f32(b2)
        self.w3 = frag_pack_1x1(w3, s3, dtype, 0, self.device)
        self.b3 = f32(b3)
        self.w1 = self.b1 = None
        if next_conv1 is not None:
            w1, s1, b1 = next_conv1
            if tuple(w1.shape[:2]) != (128, 512):
                raise ValueError("the chained conv1 is 512->128")
            self.w1 = frag_pack_1x1(w1, s1, dtype, 0, self.device)
            self.b1 = f32(b1)

    def __call__(self, mid, residual, chain=True):
        """mid [n_img, n, n, 128], residual [n_img, n, n, 512] -> out [n_img, n, n, 512]; n = 28 (cube 224) or 64 (cube 512).
        With `next_conv1` (and 28x28 faces, chain=True): returns (out, next conv1's output [n_img, 28, 28, 128])."""
        require_gpu(mid, residual)
        n_img, n = mid.shape[0], mid.shape[1]
        if n not in (28, 64):
            raise ValueError("L2Block handles 28x28 and 64x64 faces")
        _check_buf('mid', mid, self.dtype, (n_img, n, n, 128))
        _check_buf('residual', residual, self.dtype, (n_img, n, n, 512))
        if mid.shape[3] != 128 or residual.shape[3] != 512:
            raise ValueError("dense NHWC tensors only")
        out = torch.empty((n_img, n, n, 512), dtype=self.dtype, device=mid.device)
        if self.w1 is not None and chain:
            if n != 28:
                raise ValueError("the chained conv1 exists for 28x28 faces")
            nxt = torch.empty((n_img, n, n, 128), dtype=self.dtype, device=mid.device)
            check(lib().cp360_l2block_forward_next(dtype_code(self.dtype), ptr(mid), ptr(self.w2), ptr(self.b2), ptr(self.w3),
                                                   ptr(self.b3), ptr(residual), ptr(out), ptr(self.w1), ptr(self.b1), ptr(nxt),
                                                   n_img, n, stream()))
            return out, nxt
        check(lib().cp360_l2block_forward(dtype_code(self.dtype), ptr(mid), ptr(self.w2), ptr(self.b2), ptr(self.w3),
                                          ptr(self.b3), ptr(residual), ptr(out), n_img, n, stream()))
        return out


class L2First:
    """K3f (csrc/lfirst.hip): layer2's first Bottleneck after its conv1 as ONE launch - conv2 (CubePad(1) + 3x3 stride 2) +
    bn2 + relu -> conv3 + bn3 + downsample(x) + relu.  16-bit types, 56x56 -> 28x28 faces.  (weight, bn scale, bn bias) triples."""

    def __init__(self, conv2, conv3, downsample, dtype=torch.float16, device='cuda', next_conv1=None):
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("L2First runs in fp16 / bf16")
        self.dtype, self.device = dtype, torch.device(device)
        L, code = lib(), dtype_code(dtype)
        f32 = lambda t: None if t is None else t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        (w2, s2, b2), (w3, s3, b3), (wd, sd, bd) = conv2, conv3, downsample
        if tuple(w2.shape) != (128, 128, 3, 3) or tuple(w3.shape[:2]) != (512, 128) or tuple(wd.shape[:2]) != (512, 256):
            raise ValueError("L2First is layer2.0's geometry: conv2 128->128 3x3 s2, conv3 128->512, downsample 256->512 s2")
        self.w2 = torch.empty(L.cp360_l2block_packed_bytes(code), dtype=torch.uint8, device=self.device)
        check(L.cp360_l2block_pack_weights(code, ptr(f32(w2)), ptr(f32(s2)), ptr(self.w2), stream()))
        self.b2 = f32(b2)
        self.w3d = torch.empty(L.cp360_l2first_w3d_bytes(code), dtype=torch.uint8, device=self.device)
        check(L.cp360_l2first_pack_w3d(code, ptr(f32(w3).reshape(512, 128)), ptr(f32(s3)), ptr(f32(wd).reshape(512, 256)),
                                       ptr(f32(sd)), ptr(self.w3d), stream()))
        self.b3d = (f32(b3) + f32(bd)).contiguous()
        self.w1 = self.b1 = None
        if next_conv1 is not None:                       # layer2.1's conv1 (512 -> 128) chained onto this launch
            w1, s1, b1 = next_conv1
            if tuple(w1.shape[:2]) != (128, 512):
                raise ValueError("the chained conv1 is 512->128")
            self.w1 = frag_pack_1x1(w1, s1, dtype, 0, self.device)
            self.b1 = f32(b1)

    def __call__(self, mid, x, chain=True):
        """mid [n_img, 56, 56, 128] (conv1 output), x [n_img, 56, 56, 256] (block input) -> out [n_img, 28, 28, 512]
        (with ``next_conv1`` and chain=True: (out, the next block's conv1 output [n_img, 28, 28, 128]))."""
        require_gpu(mid, x)
        n_img = mid.shape[0]
        _check_buf('mid', mid, self.dtype, (n_img, 56, 56, 128))
        _check_buf('x', x, self.dtype, (n_img, 56, 56, 256))
        if mid.shape[3] != 128 or x.shape[3] != 256:
            raise ValueError("dense NHWC tensors only")
        out = torch.empty((n_img, 28, 28, 512), dtype=self.dtype, device=mid.device)
        nxt = torch.empty((n_img, 28, 28, 128), dtype=self.dtype, device=mid.device) if (self.w1 is not None and chain) else None
        check(lib().cp360_l2first_forward(dtype_code(self.dtype), ptr(mid), ptr(self.w2), ptr(self.b2), ptr(self.w3d),
                                          ptr(self.b3d), ptr(x), ptr(out), ptr(self.w1) if nxt is not None else None,
                                          ptr(self.b1) if nxt is not None else None, ptr(nxt), n_img, 28, stream()))
        return out if nxt is None else (out, nxt)


class L3Block:
    """K3e at layer3's geometry (csrc/l2block.hip, C = 256): the tail of a layer3 identity Bottleneck as ONE launch -
    conv2 + bn2 + relu -> conv3 + bn3 + residual + relu.  16-bit types, 14x14 faces.  (weight, bn scale, bn bias) triples."""

    def __init__(self, conv2, conv3, dtype=torch.float16, device='cuda'):
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("L3Block runs in fp16 / bf16")
        self.dtype, self.device = dtype, torch.device(device)
        L, code = lib(), dtype_code(dtype)
        f32 = lambda t: None if t is None else t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        w2, s2, b2 = conv2
        w3, s3, b3 = conv3
        if tuple(w2.shape) != (256, 256, 3, 3) or tuple(w3.shape[:2]) != (1024, 256):
            raise ValueError("L3Block is layer3's geometry: conv2 256->256 3x3, conv3 256->1024 1x1")
        self.w2 = torch.empty(L.cp360_l3block_packed_bytes(code), dtype=torch.uint8, device=self.device)
        check(L.cp360_l3block_pack_weights(code, ptr(f32(w2)), ptr(f32(s2)), ptr(self.w2), stream()))
        self.b2 = f32(b2)
        self.w3 = frag_pack_1x1(w3, s3, dtype, 0, self.device)
        self.b3 = f32(b3)

    def __call__(self, mid, residual):
        """mid [n_img, n, n, 256], residual [n_img, n, n, 1024] -> out [n_img, n, n, 1024]; n = 14 (cube 224) or 32 (cube 512)."""
        require_gpu(mid, residual)
        n_img, n = mid.shape[0], mid.shape[1]
        if n not in (14, 32):
            raise ValueError("L3Block handles 14x14 and 32x32 faces")
        _check_buf('mid', mid, self.dtype, (n_img, n, n, 256))
        _check_buf('residual', residual, self.dtype, (n_img, n, n, 1024))
        if mid.shape[3] != 256 or residual.shape[3] != 1024:
            raise ValueError("dense NHWC tensors only")
        out = torch.empty((n_img, n, n, 1024), dtype=self.dtype, device=mid.device)
        check(lib().cp360_l3block_forward(dtype_code(self.dtype), ptr(mid), ptr(self.w2), ptr(self.b2), ptr(self.w3),
                                          ptr(self.b3), ptr(residual), ptr(out), n_img, n, stream()))
        return out


class launch_order:
    """``with ops.launch_order(2): ...`` - work-item order of the launches issued inside (cp360_set_launch_order: 0
    ascending, 1 descending, 2 alternating).  A performance hint: a consumer that walks its input against the order its
    producer wrote it in starts with the lines still resident in the 256 MB Infinity Cache."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        self.old = lib().cp360_set_launch_order(self.mode)
        return self

    def __exit__(self, *exc):
        lib().cp360_set_launch_order(self.old)
        return False


def cubepad_maxpool3s2(x):
    """CubePad(1) + MaxPool2d(3, 2, 0) on NHWC (resnet_cubic.py:169-170)."""
    require_gpu(x)
    n6, n, _, Cc = x.shape
    ho = (n + 2 - 3) // 2 + 1
    y = torch.empty((n6, ho, ho, Cc), dtype=x.dtype, device=x.device)
    check(lib().cp360_cubepad_maxpool3s2(ptr(x), ptr(y), n6, n, Cc, dtype_code(x.dtype), stream()))
    return y


def lstm_gates(partial, splits, bias, c_prev, c_next, h_out, h_coff, h_f32, M, Hc, slab_rows=False, x_next=None):
    """model/clstm.py:68-80.  ``x_next`` = (cam, minmax, x_coff, P, clip_stride, t_next): also write the window-
    normalised frame t_next of every clip into channels [x_coff, x_coff + Hc) of ``h_out`` (the next step's input)."""
    require_gpu(partial, bias, c_prev, c_next, h_out, h_f32)
    _check_buf('gates partial', partial, torch.float32, numel=splits * M * 4 * Hc)
    _check_buf('gates bias', bias, torch.float32, numel=4 * Hc)
    _check_buf('c_prev', c_prev, torch.float32, numel=M * Hc)
    _check_buf('c_next', c_next, torch.float32, numel=M * Hc)
    _check_buf('h_f32', h_f32, torch.float32, numel=M * Hc)
    if not h_out.is_contiguous() or h_out.numel() < M * h_out.shape[-1] or h_coff + Hc > h_out.shape[-1]:
        raise ValueError("h_out must be contiguous [.., ld] with M pixels and h_coff + Hc <= ld")
    if x_next is not None:
        cam, minmax, x_coff, P, clip_stride, t_next = x_next
        require_gpu(cam, minmax)
        _check_buf('cam', cam, torch.float32, numel=(M // P - 1) * clip_stride + (t_next + 1) * P * Hc)
        _check_buf('minmax', minmax, torch.float32, numel=2 * (M // P))
        xp = C.c_void_p(cam.data_ptr() + 4 * t_next * P * Hc)
        check(lib().cp360_lstm_gates_next(ptr(partial), splits, ptr(bias), ptr(c_prev), ptr(c_next), ptr(h_out),
                                          dtype_code(h_out.dtype), h_out.shape[-1], h_coff, ptr(h_f32), M, Hc,
                                          int(slab_rows), xp, ptr(minmax), x_coff, P, clip_stride, stream()))
        return
    check(lib().cp360_lstm_gates(ptr(partial), splits, ptr(bias), ptr(c_prev), ptr(c_next), ptr(h_out),
                                 dtype_code(h_out.dtype), h_out.shape[-1], h_coff, ptr(h_f32), M, Hc,
                                 int(slab_rows), stream()))


class WinoConv:
    """CubePad(1) + 3x3 convolution in the Winograd F(2x2, 3x3) domain (csrc/wino.hip, include/cp360.h "K5w"): the 16-bit
    form of the ConvLSTM convolutions (model/clstm.py:56-64) when the launch has enough tiles to fill 384-row GEMM tiles.
    weight f32 [c_out, c_in, 3, 3], bias f32 [c_out] or None.  ``preferred(n_img, face)`` = the library's planner."""

    # one V / M workspace pair per (device, dtype, HIP stream), shared by every convolution launched on that stream: V is dead
    # once the GEMM has run and M once its output transform has (stream order - which only holds within ONE stream, hence the key)
    _ws = {}

    def __init__(self, weight, bias=None, relu=False, dtype=torch.bfloat16, device='cuda'):
        if dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("the Winograd path runs in bf16 / fp16 (f32 stays on the direct kernels)")
        if tuple(weight.shape[2:]) != (3, 3):
            raise ValueError("a [c_out, c_in, 3, 3] filter")
        self.dtype, self.device, self.relu = dtype, torch.device(device), bool(relu)
        self.c_out, self.c_in = int(weight.shape[0]), int(weight.shape[1])
        self._w_src = weight.detach()
        self.bias = None if bias is None else bias.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self._packed = None
        self.tag = 'wino'

    def desc(self, n_img, face, pix_stride=None, ld_out=0, out_coff=0, relu=None):
        d = _lib.WinoDesc()
        d.dtype = dtype_code(self.dtype)
        d.n_img, d.face, d.c_in, d.c_out = n_img, face, self.c_in, self.c_out
        d.pix_stride = self.c_in if pix_stride is None else pix_stride
        d.ld_out, d.out_coff = ld_out, out_coff
        d.relu = int(self.relu if relu is None else relu)
        return d

    def preferred(self, n_img, face):
        return bool(lib().cp360_wino_preferred(C.byref(self.desc(n_img, face))))

    @property
    def packed(self):
        if self._packed is None:
            d = self.desc(6, 8)
            w = self._w_src.to(device=self.device, dtype=torch.float32).contiguous()
            t = torch.empty(lib().cp360_wino_packed_bytes(C.byref(d)), dtype=torch.uint8, device=self.device)
            check(lib().cp360_wino_pack_weights(C.byref(d), ptr(w), ptr(t), stream()))
            self._packed = t
        return self._packed

    def workspace(self, d):
        key = (self.device, self.dtype, int(torch.cuda.current_stream(self.device).cuda_stream))
        need_v, need_m = lib().cp360_wino_v_bytes(C.byref(d)), lib().cp360_wino_m_bytes(C.byref(d))
        v, m = WinoConv._ws.get(key, (None, None))
        if v is None or v.numel() < need_v:
            v = torch.empty(need_v, dtype=torch.uint8, device=self.device)
        if m is None or m.numel() * 4 < need_m:
            m = torch.empty(need_m // 4, dtype=torch.float32, device=self.device)
        WinoConv._ws[key] = (v, m)
        return v, m

    def _check_in(self, x):
        require_gpu(x)
        if x.dim() != 4 or x.dtype != self.dtype or not x.is_contiguous() or x.shape[1] != x.shape[2] or x.shape[3] < self.c_in \
                or x.shape[0] % 6:
            raise ValueError("x must be a contiguous %s [6B, w, w, >= %d] tensor" % (self.dtype, self.c_in))

    def input(self, x):
        """Input transform of x: returns (V workspace, desc)."""
        self._check_in(x)
        d = self.desc(x.shape[0], x.shape[1], pix_stride=x.shape[3])
        v, _ = self.workspace(d)
        check(lib().cp360_wino_input(C.byref(d), ptr(x), ptr(v), stream()))
        return v, d

    def gemm(self, v, d):
        """The 16 GEMMs on a transformed input: returns the M workspace (f32 [16, m_pad, c_out])."""
        _, m = self.workspace(d)
        L = lib()
        if LAUNCH_TIMER is None:
            check(L.cp360_wino_gemm(C.byref(d), ptr(v), ptr(self.packed), ptr(m), stream()))
        else:
            flops = 2.0 * d.n_img * d.face * d.face * self.c_out * self.c_in * 9       # quoted on the DIRECT form's flops
            packed = self.packed
            check(LAUNCH_TIMER.wrap(self.tag, flops, lambda: L.cp360_wino_gemm(C.byref(d), ptr(v), ptr(packed), ptr(m), stream())))
        return m

    def sums(self, x):
        """Input transform + the 16 GEMMs: returns (M workspace, desc) for an output transform."""
        v, d = self.input(x)
        return self.gemm(v, d), d

    def output(self, m, d, out=None, out_coff=0):
        n_img, w = d.n_img, d.face
        if out is None:
            out = torch.empty((n_img, w, w, self.c_out), dtype=self.dtype, device=self.device)
        require_gpu(out)
        _check_buf('out', out, self.dtype, (n_img, w, w, self.c_out + out_coff))
        d.ld_out, d.out_coff = out.shape[3], out_coff
        check(lib().cp360_wino_output(C.byref(d), ptr(m), ptr(self.bias), ptr(out), stream()))
        return out

    def output_input(self, m, d, nxt):
        """This convolution's output transform fused with ``nxt``'s input transform (cp360_wino_output_input): returns
        (V workspace, nxt's desc), or None where the fused kernel does not apply (faces above 9 x 9)."""
        if nxt.c_in != self.c_out or nxt.dtype != self.dtype:
            raise ValueError("the next convolution reads this one's output")
        dn = nxt.desc(d.n_img, d.face)
        v, _ = self.workspace(dn)
        rc = lib().cp360_wino_output_input(C.byref(d), ptr(m), ptr(self.bias), ptr(v), stream())
        if rc == -8:                                           # CP360_ERR_UNSUPPORTED
            return None
        check(rc)
        return v, dn

    def __call__(self, x, out=None, out_coff=0):
        m, d = self.sums(x)
        return self.output(m, d, out, out_coff)

    def gates_from(self, m, d, bias, c_prev, c_next, h_out, h_coff, h_f32=None, x_next=None):
        """The gate epilogue (clstm.py:68-80) on the Gates convolution's M; x_next as ops.lstm_gates."""
        n_img, w = d.n_img, d.face
        M, Hc = n_img * w * w, self.c_out // 4
        require_gpu(bias, c_prev, c_next, h_out, h_f32)
        _check_buf('gates bias', bias, torch.float32, numel=4 * Hc)
        _check_buf('c_prev', c_prev, torch.float32, numel=M * Hc)
        _check_buf('c_next', c_next, torch.float32, numel=M * Hc)
        _check_buf('h_f32', h_f32, torch.float32, numel=M * Hc)
        if h_out.dtype != self.dtype or not h_out.is_contiguous() or h_out.numel() < M * h_out.shape[-1] or h_coff + Hc > h_out.shape[-1]:
            raise ValueError("h_out must be a contiguous %s [.., ld] tensor with M pixels and h_coff + Hc <= ld" % self.dtype)
        xp, mm, x_coff, stride = None, None, 0, 0
        if x_next is not None:
            cam, minmax, x_coff, P, stride, t_next = x_next
            require_gpu(cam, minmax)
            _check_buf('cam', cam, torch.float32, numel=(M // P - 1) * stride + (t_next + 1) * P * Hc)
            _check_buf('minmax', minmax, torch.float32, numel=2 * (M // P))
            xp, mm = C.c_void_p(cam.data_ptr() + 4 * t_next * P * Hc), ptr(minmax)
        check(lib().cp360_wino_output_gates(C.byref(d), ptr(m), ptr(bias), ptr(c_prev), ptr(c_next), ptr(h_out), h_out.shape[-1],
                                            h_coff, ptr(h_f32), xp, mm, x_coff, stride, stream()))

    def gates(self, x, bias, c_prev, c_next, h_out, h_coff, h_f32=None, x_next=None):
        """The Gates convolution + the cell update (clstm.py:68-80) in the output transform; x_next as ops.lstm_gates."""
        m, d = self.sums(x)
        self.gates_from(m, d, bias, c_prev, c_next, h_out, h_coff, h_f32, x_next)


def window_minmax(x, B, per_clip, minmax, scratch, clip_stride=0):
    require_gpu(x, minmax, scratch)
    _check_buf('x', x, torch.float32, numel=(B - 1) * (clip_stride or per_clip) + per_clip)
    _check_buf('minmax', minmax, torch.float32, numel=2 * B)
    _check_buf('scratch', scratch, torch.float32, numel=B * 256 * 2)
    check(lib().cp360_window_minmax(ptr(x), ptr(minmax), ptr(scratch), B, per_clip, clip_stride, stream()))


def window_normalize(x, minmax, y, y_coff, y2, B, T, t, P, Cc, clip_stride=0):
    require_gpu(x, minmax, y, y2)
    _check_buf('x', x, torch.float32, numel=(B - 1) * (clip_stride or T * P * Cc) + (t + 1) * P * Cc)
    _check_buf('minmax', minmax, torch.float32, numel=2 * B)
    _check_buf('y2', y2, torch.float32, numel=B * P * Cc)
    if not y.is_contiguous() or y.numel() < B * P * y.shape[-1] or y_coff + Cc > y.shape[-1]:
        raise ValueError("y must be contiguous [.., ld] with B*P pixels and y_coff + C <= ld")
    check(lib().cp360_window_normalize(ptr(x), ptr(minmax), ptr(y), dtype_code(y.dtype), y.shape[-1], y_coff,
                                       ptr(y2), B, T, t, P, Cc, clip_stride, stream()))


def held_clock_ghz(device='cuda', ms=6.0, launches=3):
    """Diagnostic (never on the hot path): the shader clock the chip holds under a dense bf16 MFMA load on random operands -
    cp360_clock_probe (csrc/misc.hip), one wave per SIMD on every CU, median over the waves of the last of ``launches``
    back-to-back launches of about ``ms`` milliseconds each.  MI355X boxes differ in the clock they hold (DVFS), which moves
    a bench line by more than a round of kernel work: bench.py prints this next to its numbers."""
    dev = torch.device(device)
    n_wg = torch.cuda.get_device_properties(dev).multi_processor_count
    stamps = torch.zeros((n_wg * 4, 2), dtype=torch.int64, device=dev)
    iters = max(1, int(ms * 1e-3 * 1.8e9 / (16 * 16)))            # 16 MFMAs of 16 cycles per iteration at about 1.8 GHz
    with torch.cuda.device(dev):
        for _ in range(launches):
            check(lib().cp360_clock_probe(ptr(stamps), n_wg, iters, stream()))
        torch.cuda.synchronize()
    s = stamps.cpu().double()
    ok = s[:, 1] > 0
    if not bool(ok.any()):
        return None
    ghz = 0.1 * s[ok, 0] / s[ok, 1]
    return round(float(ghz.median()), 4)
