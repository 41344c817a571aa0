"""CPU tests of the C-ABI library: it loads, exports every symbol the header declares,
its host-side CubePad index function is bit-identical to the oracle (the same inline
function drives every device kernel), and argument errors come back as status codes."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import o_cubepad
from cp_360_weakly_supervised_saliency_amd import _lib, ops

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(REPO, 'include', 'cp360.h')).read()
    internal = open(os.path.join(REPO, 'include', 'cp360_internal.h')).read()
    declared = set(re.findall(r'\b(cp360_[a-z0-9_]+)\s*\(', hdr))
    declared_internal = set(re.findall(r'\b(cp360_[a-z0-9_]+)\s*\(', internal))
    assert declared and declared_internal, "no declarations parsed"
    assert not (declared & declared_internal)
    L = _lib.lib()
    for name in sorted(declared | declared_internal):
        assert hasattr(L, name), "libcp360.so does not export %s" % name
    # the documented boundary (cp360.h) and the internal fused-kernel entry points (cp360_internal.h) are bound separately
    assert declared == set(_lib.PUBLIC_SYMBOLS)
    assert declared_internal == set(_lib.INTERNAL_SYMBOLS)
    # the internal ones are exactly the shape-specific kernels + the launch-order hint + the bare GEMM of tools/wino_probe.py: nothing a binder of the reference needs
    assert all(re.match(r'cp360_(stem|band3x3|frag|l1block|l2block|l2first|l3block|set_launch_order|wino_gemm_raw)', n) for n in declared_internal)
    hv = int(re.search(r'#define\s+CP360_VERSION\s+(\d+)', hdr).group(1))
    assert L.cp360_version() == hv == _lib.ABI_VERSION           # header, library and binding agree
    assert L.cp360_conv_desc_bytes() == C.sizeof(_lib.ConvDesc)
    assert L.cp360_strerror(-2).decode().startswith('CubePad size mismatch')


def test_no_kernel_uses_scratch_memory():
    """Every kernel of libcp360.so keeps its working set in registers: ScratchSize 0 in the resource reports the build writes
    beside the objects (csrc/*.rpt).  A spill in a hand-tiled kernel is a silent 2-5x slowdown, not a warning."""
    import __graft_entry__ as g
    assert g.scratch_report() == {}


@pytest.mark.parametrize('n', [1, 2, 4, 5, 7])
def test_cubepad_table_matches_oracle_all_pad_combinations(n):
    for pl in range(0, min(n, 3) + 1):
        for pr in range(0, min(n, 3) + 1):
            for pt in range(0, min(n, 3) + 1):
                for pd in range(0, min(n, 3) + 1):
                    got = ops.cubepad_table(n, [pl, pr, pt, pd])
                    want = o_cubepad.cubepad_table(n, pl, pr, pt, pd)
                    assert np.array_equal(got, want), (n, pl, pr, pt, pd)


@pytest.mark.parametrize('n,p', [(224, 3), (112, 1), (56, 1), (28, 1), (14, 1), (7, 1), (16, 1), (9, 2)])
def test_cubepad_table_network_sizes(n, p):
    assert np.array_equal(ops.cubepad_table(n, p), o_cubepad.cubepad_table(n, p, p, p, p))


def test_status_codes_without_gpu():
    L = _lib.lib()
    dummy = C.c_void_p(16)
    assert L.cp360_cubepad_nchw(dummy, dummy, 5, 1, 4, 1, 1, 1, 1, 4, None) == -2      # batch % 6
    assert L.cp360_cubepad_nchw(None, dummy, 6, 1, 4, 1, 1, 1, 1, 4, None) == -5       # null
    assert L.cp360_cubepad_nchw(dummy, dummy, 6, 1, 4, 1, 1, 1, 1, 3, None) == -4      # elem size
    assert L.cp360_cubepad_nchw(dummy, dummy, 6, 1, 4, -1, 1, 1, 1, 4, None) == -1
    d = _lib.ConvDesc()
    assert L.cp360_conv_packed_bytes(C.byref(d)) == 0
    with pytest.raises(ValueError):
        _lib.check(-2)
    with pytest.raises(_lib.Cp360Error):
        _lib.check(-7)


def test_conv_packed_size_and_desc_checks():
    import torch
    L = _lib.lib()
    d = _lib.ConvDesc()
    for k, v in dict(dtype=0, n_img=6, h_in=7, w_in=7, c_in=2000, pix_stride=2000, kh=3, kw=3, sy=1, sx=1,
                     h_out=7, w_out=7, c_out=4000, pad_mode=1, pad=1, ld_out=4000, out_coff=0, ld_res=0,
                     relu=1, splits=1).items():
        setattr(d, k, v)
    # f32: K per tap padded to 32 -> 2016; rows padded to 128 -> 4096
    assert L.cp360_conv_packed_bytes(C.byref(d)) == 4096 * 9 * 2016 * 4
    assert 1 <= L.cp360_conv_suggest_splits(C.byref(d)) <= 32
    d.dtype = 1   # bf16: K per tap padded to 64 -> 2048
    assert L.cp360_conv_packed_bytes(C.byref(d)) == 4096 * 9 * 2048 * 2
    d.n_img = 5
    assert L.cp360_conv_packed_bytes(C.byref(d)) == 0          # fused CubePad needs 6N images
    d.n_img, d.splits = 6, 4
    assert L.cp360_conv_partial_bytes(C.byref(d)) == 4 * 294 * 4000 * 4
    assert torch is not None
    # a FORCED small tile (tile_px 6464) is held to what the planner requires of conv_small_kernel: the 16-bit kernel moves
    # residuals / outputs in 16-byte pieces (ld_out, out_coff, ld_res % 8), its weight offsets are 32-bit
    one = C.c_void_p(16)
    d.splits, d.tile_px, d.c_out, d.ld_out = 1, 6464, 4000, 4004
    assert L.cp360_conv_forward(C.byref(d), one, one, None, None, one, None, None) == -6          # ld_out % 8 (bf16): ALIGN
    d.dtype = 0
    assert L.cp360_conv_packed_bytes(C.byref(d)) != 0                                             # f32: 4-element alignment is enough
    d.dtype, d.ld_out, d.out_coff = 1, 4008, 4
    assert L.cp360_conv_forward(C.byref(d), one, one, None, None, one, None, None) == -6          # out_coff % 8
    d.ld_out, d.out_coff, d.c_in, d.pix_stride = 4000, 0, 40000000, 40000000                      # 64 rows x K bytes >= 2^32
    d.n_img, d.h_in, d.w_in, d.h_out, d.w_out, d.kh, d.kw, d.pad_mode, d.pad = 6, 1, 1, 1, 1, 1, 1, 0, 0
    assert L.cp360_conv_forward(C.byref(d), one, one, None, None, one, None, None) == -8          # UNSUPPORTED, not a wrapped offset


def test_ops_refuse_cpu_tensors():
    import torch
    with pytest.raises(RuntimeError):
        ops.cubepad_nchw(torch.zeros(6, 1, 4, 4), 1)


def test_stem_and_resize_entry_points_validate_without_gpu():
    """Argument validation happens before any launch: error codes on the CPU box."""
    import ctypes as C
    L = _lib.lib()
    assert L.cp360_stem_packed_bytes(_lib.BF16) == 7 * 64 * 64 and L.cp360_stem_packed_bytes(_lib.F32) == 0
    one = C.c_void_p(16)
    assert L.cp360_stem_forward(_lib.BF16, None, one, None, one, 6, 224, 1, None) == -5          # NULL
    assert L.cp360_stem_forward(_lib.BF16, one, one, None, one, 6, 256, 1, None) == -8           # cube sizes other than 224 / 512
    assert L.cp360_stem_forward(_lib.BF16, one, one, None, one, 0, 224, 1, None) == -1           # bad shape
    assert L.cp360_band3x3_packed_bytes(_lib.F16) == 9 * 64 * 128
    assert L.cp360_band3x3_forward(_lib.BF16, one, one, None, one, 6, 28, 128, 1, None) == -8     # other shapes
    assert L.cp360_band3x3_forward(_lib.BF16, one, one, None, one, 7, 56, 64, 1, None) == -2      # not 6N
    assert L.cp360_resize_ksize(3840, 1920) == 13 and L.cp360_resize_ksize(100, 200) == 7
    assert L.cp360_resize_ksize(0, 5) == -1
    assert L.cp360_resize_lanczos_u8(None, one, None, 1, 4, 4, 2, 2, None, None, 0, None, None, 0, None) == -5


def test_round2_entry_points_validate_without_gpu():
    """The fused kernels added in round 2: sizes and argument checks (no launch on the CPU box); the launch-order hint
    is host state and returns the previous mode."""
    import ctypes as C
    L = _lib.lib()
    one = C.c_void_p(16)
    assert L.cp360_l2block_packed_bytes(_lib.F16) == 9 * 128 * 128 * 2 and L.cp360_l3block_packed_bytes(_lib.BF16) == 9 * 256 * 256 * 2
    assert L.cp360_l2block_packed_bytes(_lib.F32) == 0
    assert L.cp360_l2block_forward(_lib.F16, one, one, None, one, one, one, one, 6, 14, None) == -8      # layer2: faces 28 / 64
    assert L.cp360_l3block_forward(_lib.F16, one, one, None, one, one, one, one, 6, 28, None) == -8      # layer3: faces 14
    assert L.cp360_l3block_forward(_lib.F16, one, one, None, one, one, one, one, 5, 14, None) == -2      # not 6N
    assert L.cp360_l2block_forward_next(_lib.F16, one, one, None, one, one, one, one, one, None, one, 6, 64, None) == -8   # chained conv1: 28x28 only
    assert L.cp360_l2block_forward_next(_lib.F16, one, one, None, one, one, one, one, None, None, one, 6, 28, None) == -5
    assert L.cp360_stem_pool_border_bytes(6) == 6 * 4 * 112 * 64 * 2 and L.cp360_stem_pool_border_bytes(0) == 0
    assert L.cp360_stem_pool_forward(_lib.BF16, one, one, None, one, one, 6, 512, None) == -8            # cube 224 only
    assert L.cp360_stem_pool_forward(_lib.BF16, one, one, None, one, None, 6, 224, None) == -5
    assert L.cp360_stem_pool_forward(_lib.F32, one, one, None, one, one, 6, 224, None) != 0
    old = L.cp360_set_launch_order(2)
    assert L.cp360_set_launch_order(1) == 2 and L.cp360_set_launch_order(7) == 1 and L.cp360_set_launch_order(old) == 0


def _desc(dtype, n_img, face, c_in, c_out, k, stride=1, clip_resident=0):
    d = _lib.ConvDesc()
    pad = 1 if k == 3 else 0
    ho = (face + 2 * pad - k) // stride + 1
    for key, v in dict(dtype=dtype, n_img=n_img, h_in=face, w_in=face, c_in=c_in, pix_stride=c_in, kh=k, kw=k, sy=stride, sx=stride,
                       h_out=ho, w_out=ho, c_out=c_out, pad_mode=1 if pad else 0, pad=pad, ld_out=c_out, out_coff=0, ld_res=0,
                       relu=1, splits=1, clip_resident=clip_resident).items():
        setattr(d, key, v)
    return d


@pytest.mark.parametrize('cube', [224, 256, 512])
def test_launch_planner_over_the_network_shapes(cube):
    """The host-side launch planner (cp360_conv_suggest_splits / cp360_conv_plan_describe, csrc/conv_igemm.hip) on every
    convolution shape of the path (resnet_cubic.py:85-175, clstm.py:56-64) x 1 / 4 / 16 / 64 frames x the three dtypes: a plan
    exists, its split count is a valid one, the workspace it implies is what cp360_conv_partial_bytes reports, and the
    description names a kernel.  No GPU involved: this is the part of the library that decides what gets launched."""
    L = _lib.lib()
    f1 = cube // 4
    shapes = [(f1, 64, 64, 1, 1), (f1, 64, 64, 3, 1), (f1, 64, 256, 1, 1), (f1, 256, 64, 1, 1),
              (f1, 256, 128, 1, 1), (f1, 128, 128, 3, 2), (f1 // 2, 128, 512, 1, 1), (f1 // 2, 512, 128, 1, 1),
              (f1 // 2, 128, 128, 3, 1), (f1 // 2, 512, 256, 1, 1), (f1 // 2, 256, 256, 3, 2), (f1 // 4, 256, 1024, 1, 1),
              (f1 // 4, 1024, 256, 1, 1), (f1 // 4, 256, 256, 3, 1), (f1 // 4, 1024, 512, 1, 1), (f1 // 4, 512, 512, 3, 2),
              (f1 // 8, 512, 2048, 1, 1), (f1 // 8, 2048, 512, 1, 1), (f1 // 8, 512, 512, 3, 1), (f1 // 8, 2048, 1000, 1, 1),
              (f1 // 8, 2000, 4000, 3, 1), (f1 // 8, 4000, 4000, 3, 1)]
    buf = C.create_string_buffer(256)
    seen = set()
    for frames in (1, 4, 16, 64):
        for dtype in (_lib.F32, _lib.BF16, _lib.F16):
            for face, cin, cout, k, s in shapes:
                d = _desc(dtype, 6 * frames, face, cin, cout, k, s)
                sp = L.cp360_conv_suggest_splits(C.byref(d))
                assert 1 <= sp <= 32, (frames, dtype, face, cin, cout, k, s, sp)
                n = L.cp360_conv_plan_describe(C.byref(d), buf, 256)
                text = buf.value.decode()
                assert n > 0 and text.startswith('conv_'), text
                assert ('split-K %d' % sp) in text, (text, sp)
                seen.add(text.split(',')[0])
                d.splits = sp
                ho = d.h_out
                assert L.cp360_conv_partial_bytes(C.byref(d)) == (sp * 6 * frames * ho * ho * cout * 4 if sp > 1 else 0)
                assert L.cp360_conv_packed_bytes(C.byref(d)) > 0
    assert len(seen) >= 3, seen                                 # small tiles, ring tiles, narrow tiles all occur somewhere


@pytest.mark.parametrize('face,tile,clips,want', [(7, 'one cube', 4, 4), (8, 'half a cube of 8x8 faces', 4, 2),
                                                   (16, 'one 16x16 face', 1, 5), (7, 'one cube', 1, 16)])
def test_clip_resident_planner(face, tile, clips, want):
    """The clip-resident kernel's tiles (a cube at up to 7x7 faces, half a cube at 8x8, a face at 16x16) x split-K fill the
    256 CUs: the ConvLSTM's Conv2 / Gates shape (clstm.py:59-64) at the batch sizes the bench lines use."""
    L = _lib.lib()
    d = _desc(_lib.BF16, 6 * clips, face, 4000, 4000, 3, 1, clip_resident=1)
    buf = C.create_string_buffer(256)
    assert L.cp360_conv_plan_describe(C.byref(d), buf, 256) > 0
    text = buf.value.decode()
    assert tile in text, text
    sp = L.cp360_conv_suggest_splits(C.byref(d))
    assert sp == want, (text, sp)
    tiles = 16 * {7: clips, 8: 2 * clips, 16: 6 * clips}[face]
    assert ('%d workgroups' % (tiles * sp)) in text, text
    assert 200 <= tiles * sp <= 512                             # one or two rounds of the chip, never a fraction of it
    for bad in (9, 12):                                         # other face sizes: UNSUPPORTED, not a silent generic launch
        d2 = _desc(_lib.BF16, 6 * clips, bad, 4000, 4000, 3, 1, clip_resident=1)
        assert L.cp360_conv_plan_describe(C.byref(d2), buf, 256) == -8
    assert L.cp360_conv_prefer_clip(C.byref(d)) in (0, 1)
