"""ctypes binding of csrc/libcp360.so (the C ABI declared in include/cp360.h).

No CPU fallback: ``lib()`` raises if the shared library has not been built
(``make -C cp_360_weakly_supervised_saliency_amd/csrc`` or
``__graft_entry__.build()``), and every op checks that its tensors live on a GPU.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CP360_LIB: an alternative build of the SAME library (kernel experiments, tools/exp_build.sh)
LIB_PATH = os.environ.get('CP360_LIB') or os.path.join(_HERE, 'csrc', 'libcp360.so')

F32, BF16, F16, U8 = 0, 1, 2, 3

# precision names accepted by the model wrappers -> torch dtype of activations / packed weights
PRECISIONS = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}


def precision_dtype(precision):
    try:
        return PRECISIONS[precision]
    except KeyError:
        raise ValueError("precision must be one of %s" % sorted(PRECISIONS))
OK = 0
# must equal CP360_VERSION of include/cp360.h (checked against the loaded library in lib())
ABI_VERSION = 305


# every exported symbol of include/cp360.h - the documented boundary (checked by tests/test_abi.py)
PUBLIC_SYMBOLS = [
    'cp360_strerror', 'cp360_version', 'cp360_conv_desc_bytes', 'cp360_cubepad_table_host', 'cp360_cubepad_nchw',
    'cp360_cubepad_nhwc', 'cp360_nchw_to_nhwc', 'cp360_nhwc_to_nchw', 'cp360_equi2cube', 'cp360_cube2equi',
    'cp360_conv_packed_bytes', 'cp360_conv_partial_bytes', 'cp360_conv_suggest_splits', 'cp360_conv_pack_weights',
    'cp360_conv_pack_weights2', 'cp360_conv_forward', 'cp360_conv_forward2', 'cp360_conv_finish',
    'cp360_cubepad_maxpool3s2', 'cp360_lstm_gates', 'cp360_lstm_gates_next', 'cp360_window_minmax',
    'cp360_window_normalize', 'cp360_resize_ksize', 'cp360_resize_coeffs_host', 'cp360_resize_lanczos_u8',
    'cp360_resize_linear_f32', 'cp360_metric_work_bytes', 'cp360_metric_auc_prepare', 'cp360_metric_auc_judd',
    'cp360_metric_auc_borji', 'cp360_metric_cc_sim', 'cp360_resize_ksize2', 'cp360_resize_coeffs_host2',
    'cp360_overlay_colorize', 'cp360_overlay_blend_u8', 'cp360_fold_bn', 'cp360_create', 'cp360_destroy',
    'cp360_resnet_load', 'cp360_resnet_workspace_bytes', 'cp360_resnet_forward', 'cp360_clstm_load', 'cp360_clstm_workspace_bytes',
    'cp360_clstm_step', 'cp360_conv_finish_add', 'cp360_window_normalize_frames',
    'cp360_clstm_window_workspace_bytes', 'cp360_clstm_window', 'cp360_clock_probe', 'cp360_conv_prefer_clip',
    'cp360_conv_plan_describe', 'cp360_resnet_plan_describe', 'cp360_wino_packed_bytes', 'cp360_wino_v_bytes',
    'cp360_wino_m_bytes', 'cp360_wino_preferred', 'cp360_wino_pack_weights', 'cp360_wino_input', 'cp360_wino_gemm',
    'cp360_wino_output', 'cp360_wino_output_gates', 'cp360_wino_forward', 'cp360_wino_output_input', 'cp360_clstm_wino_state', 'cp360_clstm_load_wino',
]
# ... and of include/cp360_internal.h: the shape-specific fused kernels the stage contexts are built from (exported for
# tests and the CP360_CTX=0 planner; not part of the boundary)
INTERNAL_SYMBOLS = [
    'cp360_stem_packed_bytes', 'cp360_stem_pack_weights', 'cp360_stem_forward', 'cp360_band3x3_packed_bytes',
    'cp360_band3x3_pack_weights', 'cp360_band3x3_forward', 'cp360_frag_packed_bytes', 'cp360_frag_pack_1x1',
    'cp360_l1block_forward', 'cp360_l1block_forward_wide', 'cp360_l1block_forward_first', 'cp360_l1block_conv2_bytes', 'cp360_l1block_pack_conv2',
    'cp360_l2block_packed_bytes', 'cp360_l2block_pack_weights', 'cp360_l2block_forward',
    'cp360_l2block_forward_next', 'cp360_set_launch_order', 'cp360_stem_pool_border_bytes',
    'cp360_stem_pool_forward', 'cp360_l3block_packed_bytes', 'cp360_l3block_pack_weights', 'cp360_l3block_forward',
    'cp360_l2first_w3d_bytes', 'cp360_l2first_pack_w3d', 'cp360_l2first_forward', 'cp360_wino_gemm_raw',
]
SYMBOLS = PUBLIC_SYMBOLS + INTERNAL_SYMBOLS


class ConvDesc(C.Structure):
    """Mirror of ``cp360_conv_desc`` (include/cp360.h)."""
    _fields_ = [(n, C.c_int) for n in (
        'dtype', 'n_img', 'h_in', 'w_in', 'c_in', 'pix_stride', 'kh', 'kw', 'sy', 'sx',
        'h_out', 'w_out', 'c_out', 'pad_mode', 'pad', 'ld_out', 'out_coff', 'ld_res',
        'relu', 'splits', 'tile_px', 'clip_resident', 'slab_rows',
        'c_in2', 'pix_stride2', 'h_in2', 'w_in2', 'sy2', 'sx2')]


class WinoDesc(C.Structure):
    """Mirror of ``cp360_wino_desc`` (include/cp360.h)."""
    _fields_ = [(n, C.c_int) for n in ('dtype', 'n_img', 'face', 'c_in', 'pix_stride', 'c_out', 'ld_out', 'out_coff', 'relu')]


class ConvBn(C.Structure):
    """Mirror of ``cp360_conv_bn`` (include/cp360.h): device f32 pointers of one convolution + its BatchNorm."""
    _fields_ = [(n, C.c_void_p) for n in ('weight', 'bn_weight', 'bn_bias', 'bn_mean', 'bn_var')]


class Cp360Error(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libcp360.so not found at %s - the HIP library must be built "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    pd = C.POINTER(ConvDesc)
    L.cp360_strerror.restype = C.c_char_p
    L.cp360_strerror.argtypes = [i]
    L.cp360_version.restype = i
    L.cp360_conv_desc_bytes.restype = sz
    L.cp360_cubepad_table_host.argtypes = [i, i, i, i, i, vp]
    L.cp360_cubepad_nchw.argtypes = [vp, vp, i, i, i, i, i, i, i, i, vp]
    L.cp360_cubepad_nhwc.argtypes = [vp, vp, i, i, i, i, i, i, i, i, i, vp]
    L.cp360_nchw_to_nhwc.argtypes = [vp, vp, i, i, i, i, i, i, i, i, vp]
    L.cp360_nhwc_to_nchw.argtypes = [vp, vp, i, i, i, i, i, i, i, i, vp]
    L.cp360_equi2cube.argtypes = [vp, vp, vp, i, i, i, i, C.POINTER(f), C.POINTER(f), f, i, i, i, i, vp]
    L.cp360_cube2equi.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
    L.cp360_conv_packed_bytes.restype = sz
    L.cp360_conv_packed_bytes.argtypes = [pd]
    L.cp360_conv_partial_bytes.restype = sz
    L.cp360_conv_partial_bytes.argtypes = [pd]
    L.cp360_conv_suggest_splits.argtypes = [pd]
    L.cp360_conv_pack_weights.argtypes = [pd, vp, vp, vp, i, vp]
    L.cp360_conv_forward.argtypes = [pd, vp, vp, vp, vp, vp, vp, vp]
    L.cp360_conv_pack_weights2.argtypes = [pd, vp, vp, vp, vp, vp, vp]
    L.cp360_conv_forward2.argtypes = [pd, vp, vp, vp, vp, vp, vp, vp, vp]
    L.cp360_conv_finish.argtypes = [pd, vp, vp, vp, vp, vp]
    L.cp360_cubepad_maxpool3s2.argtypes = [vp, vp, i, i, i, i, vp]
    L.cp360_lstm_gates.argtypes = [vp, i, vp, vp, vp, vp, i, i, i, vp, i, i, i, vp]
    L.cp360_lstm_gates_next.argtypes = [vp, i, vp, vp, vp, vp, i, i, i, vp, i, i, i, vp, vp, i, i, sz, vp]
    L.cp360_window_minmax.argtypes = [vp, vp, vp, i, sz, sz, vp]
    L.cp360_window_normalize.argtypes = [vp, vp, vp, i, i, i, vp, i, i, i, i, i, sz, vp]
    L.cp360_resize_ksize.argtypes = [i, i]
    L.cp360_resize_coeffs_host.argtypes = [i, i, vp, vp]
    L.cp360_resize_lanczos_u8.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, i, vp, vp, i, vp]
    L.cp360_stem_packed_bytes.restype = sz
    L.cp360_stem_packed_bytes.argtypes = [i]
    L.cp360_stem_pack_weights.argtypes = [i, vp, vp, vp, vp]
    L.cp360_stem_forward.argtypes = [i, vp, vp, vp, vp, i, i, i, vp]
    L.cp360_band3x3_packed_bytes.restype = sz
    L.cp360_band3x3_packed_bytes.argtypes = [i]
    L.cp360_band3x3_pack_weights.argtypes = [i, vp, vp, vp, vp]
    L.cp360_band3x3_forward.argtypes = [i, vp, vp, vp, vp, i, i, i, i, vp]
    L.cp360_l1block_conv2_bytes.restype = sz
    L.cp360_l1block_conv2_bytes.argtypes = [i]
    L.cp360_l1block_pack_conv2.argtypes = [i, vp, vp, vp, vp]
    L.cp360_frag_packed_bytes.restype = sz
    L.cp360_frag_packed_bytes.argtypes = [i, i, i]
    L.cp360_frag_pack_1x1.argtypes = [i, vp, vp, vp, i, i, i, vp]
    L.cp360_l1block_forward.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.cp360_l1block_forward_wide.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.cp360_l1block_forward_first.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.cp360_l2block_packed_bytes.restype = sz
    L.cp360_l2block_packed_bytes.argtypes = [i]
    L.cp360_l2block_pack_weights.argtypes = [i, vp, vp, vp, vp]
    L.cp360_l2block_forward.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.cp360_l3block_packed_bytes.restype = sz
    L.cp360_l3block_packed_bytes.argtypes = [i]
    L.cp360_l3block_pack_weights.argtypes = [i, vp, vp, vp, vp]
    L.cp360_l3block_forward.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.cp360_set_launch_order.argtypes = [i]
    L.cp360_stem_pool_border_bytes.restype = sz
    L.cp360_stem_pool_border_bytes.argtypes = [i]
    L.cp360_stem_pool_forward.argtypes = [i, vp, vp, vp, vp, vp, i, i, vp]
    L.cp360_l2block_forward_next.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.cp360_resize_ksize2.argtypes = [i, i, i]
    L.cp360_resize_coeffs_host2.argtypes = [i, i, i, vp, vp]
    L.cp360_overlay_colorize.argtypes = [vp, i, i, i, vp, vp, vp]
    L.cp360_overlay_blend_u8.argtypes = [vp, vp, vp, C.c_longlong, f, vp]
    L.cp360_resize_linear_f32.argtypes = [vp, i, i, vp, i, i, vp]
    L.cp360_metric_work_bytes.restype = sz
    L.cp360_metric_work_bytes.argtypes = [i]
    L.cp360_metric_auc_prepare.argtypes = [vp, vp, vp, i, i, vp, vp]
    L.cp360_metric_auc_judd.argtypes = [i, vp, vp, vp]
    L.cp360_metric_auc_borji.argtypes = [i, i, vp, C.c_double, vp, vp, vp]
    L.cp360_metric_cc_sim.argtypes = [vp, vp, i, vp, vp]
    L.cp360_l2first_w3d_bytes.restype = sz
    L.cp360_l2first_w3d_bytes.argtypes = [i]
    L.cp360_l2first_pack_w3d.argtypes = [i, vp, vp, vp, vp, vp, vp]
    L.cp360_l2first_forward.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.cp360_fold_bn.argtypes = [vp, vp, vp, vp, f, vp, vp, i, vp]
    L.cp360_create.argtypes = [i, C.POINTER(vp)]
    L.cp360_destroy.argtypes = [vp]
    L.cp360_destroy.restype = None
    L.cp360_resnet_load.argtypes = [vp, i, C.POINTER(ConvBn), i, vp, i, f, f, vp]
    L.cp360_resnet_workspace_bytes.restype = sz
    L.cp360_resnet_workspace_bytes.argtypes = [vp, i, i]
    L.cp360_resnet_forward.argtypes = [vp, vp, i, i, vp, vp, vp, sz, vp]
    L.cp360_clstm_load.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, i, i, i, vp]
    L.cp360_clstm_workspace_bytes.restype = sz
    L.cp360_clstm_workspace_bytes.argtypes = [vp, i, i]
    L.cp360_clstm_step.argtypes = [vp, vp, vp, vp, vp, i, i, vp, vp, sz, vp, sz, vp]
    L.cp360_conv_finish_add.argtypes = [pd, vp, vp, vp, vp, vp, vp]
    L.cp360_window_normalize_frames.argtypes = [vp, vp, vp, i, i, i, i, i, sz, vp]
    L.cp360_clstm_window_workspace_bytes.restype = sz
    L.cp360_clstm_window_workspace_bytes.argtypes = [vp, i, i, i]
    L.cp360_clstm_window.argtypes = [vp, vp, sz, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.cp360_clock_probe.argtypes = [vp, i, i, vp]
    L.cp360_conv_prefer_clip.argtypes = [pd]
    L.cp360_conv_plan_describe.argtypes = [pd, C.c_char_p, sz]
    L.cp360_resnet_plan_describe.argtypes = [vp, i, i, C.c_char_p, sz]
    pw = C.POINTER(WinoDesc)
    for fn in ('cp360_wino_packed_bytes', 'cp360_wino_v_bytes', 'cp360_wino_m_bytes'):
        getattr(L, fn).restype = sz
        getattr(L, fn).argtypes = [pw]
    L.cp360_wino_preferred.argtypes = [pw]
    L.cp360_wino_pack_weights.argtypes = [pw, vp, vp, vp]
    L.cp360_wino_input.argtypes = [pw, vp, vp, vp]
    L.cp360_wino_gemm.argtypes = [pw, vp, vp, vp, vp]
    L.cp360_wino_output.argtypes = [pw, vp, vp, vp, vp]
    L.cp360_wino_output_input.argtypes = [pw, vp, vp, vp, vp]
    L.cp360_wino_output_gates.argtypes = [pw, vp, vp, vp, vp, vp, i, i, vp, vp, vp, i, sz, vp]
    L.cp360_wino_forward.argtypes = [pw, vp, vp, vp, vp, vp, vp, vp]
    L.cp360_wino_gemm_raw.argtypes = [i, vp, vp, vp, i, i, i, i, i, vp]
    L.cp360_clstm_wino_state.argtypes = [vp, i, i]
    L.cp360_clstm_load_wino.argtypes = [vp, vp, vp, vp, vp]
    for name in SYMBOLS:
        getattr(L, name)          # AttributeError here = header and library disagree
    if L.cp360_version() != ABI_VERSION or L.cp360_conv_desc_bytes() != C.sizeof(ConvDesc):
        raise ImportError("%s is a stale build: version %d / cp360_conv_desc %d bytes, the binding expects %d / %d "
                          "- rebuild it (__graft_entry__.build())"
                          % (LIB_PATH, L.cp360_version(), L.cp360_conv_desc_bytes(), ABI_VERSION, C.sizeof(ConvDesc)))
    _lib = L
    return L


def check(status):
    if status != OK:
        msg = lib().cp360_strerror(int(status)).decode()
        if status in (-1, -2, -3, -4, -6):
            raise ValueError("cp360: %s (status %d)" % (msg, status))
        raise Cp360Error("cp360: %s (status %d)" % (msg, status))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream():
    """The HIP stream torch is currently queueing on (the library is stream-ordered)."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("cp360 ops run on the GPU only (HIP); got a %s tensor - there is no CPU fallback"
                               % t.device)


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    if dt == torch.uint8:
        return U8
    raise ValueError("unsupported dtype %s" % dt)
