"""``CAM()`` with the reference's signature and return values
(/root/reference/static_model/class_activation_model.py:13-85) on the HIP path.

The reference hooks ``layer4``, copies the features to the host and does the CAM GEMM
``(W - min W)[1000 x 2048] . feat[2048 x h*w]`` per face in numpy.  Here the GEMM is a
1x1 MFMA convolution on the device (K4) with the shifted weight packed once; only the
results the caller asked for cross PCIe.
"""
import numpy as np
import torch

from .. import _lib, ops, stage_ctx


def cam_device(input_cubemap_nhwc4, model, out=None, padded=False, want_feat=True):
    """Fused device path: normalised cube faces [6N, H, W, 4] (NHWC4, the output of
    ``Equi2Cube.to_cube_batch``) -> (cube_score f32 [6N, h, w, 1000] NHWC,
    layer4 features [6N, h, w, 2048] NHWC in the model's compute dtype)."""
    if stage_ctx.USE_CTX:
        # ONE C call for the whole static stage (cp360_resnet_forward, csrc/ctx.hip): the library owns the packed weights
        # and the launch planning; CP360_CTX=0 plans the same launches from Python (below)
        stage = model.__dict__.get('_stage')
        if stage is None:
            stage = model.__dict__['_stage'] = stage_ctx.ResnetStage(model)
        xp = input_cubemap_nhwc4 if padded else ops.cubepad_nhwc(input_cubemap_nhwc4, 3)
        return stage.forward(xp, cam_out=None if out is None else out.view(-1), want_feat=want_feat)
    feat = model.features_nhwc(input_cubemap_nhwc4, padded)      # padded: the faces already carry their CubePad(3) ring
    cam = model.cam_conv()
    n6, h, w, _ = feat.shape
    # raw f32 output: the CAM has no bias / activation, and its scores feed the f32 window
    # normalisation, so they are kept in f32 even on the bf16 path.  ``out`` (f32, >= n6*h*w*1000
    # elements) lets the caller place them (e.g. straight into the clip buffer the ConvLSTM stage reads).
    score = cam.raw_sum_f32(feat, None if out is None else out.view(-1))
    return score, feat


def CAM(input_cubemap, input_equi, model, feature_layer_name, weight_layer_name,
        use_gpu=True, class_const=False, num_class=1000):
    """Compute the Class Activation Map of a cube batch.

    Args (as in the reference):
        input_cubemap (np.array): normalised cube faces, shape 6N x H x W x C (float32)
        input_equi: unused (kept for signature compatibility)
        model: ``resnet50()`` from this package, on the GPU
        feature_layer_name: must be 'layer4'; weight_layer_name: must be 'fc.weight'
    Returns:
        cube_score  np.float32 [6N, num_class, h, w]
        cubic_feature np.float32 [6N, 2048, h, w]   (the hooked layer4 output)
        weight_softmax np.float32 [num_class, 2048] (fc weight, shifted by its min if < 0)
    """
    if feature_layer_name != 'layer4' or weight_layer_name != 'fc.weight':
        raise ValueError("this build accelerates the ResNet-50 CAM path only (layer4 / fc.weight)")
    if class_const:
        raise NotImplementedError("class_const needs imagenet_labeldict.npz, which the reference does not ship")
    if model.training:                                   # (a full .eval() walk per call is 0.8 ms of this per-frame path)
        model.eval()
    dev = model.fc.weight.device
    if dev.type != 'cuda':
        raise RuntimeError("CAM runs on the GPU only (HIP kernels); move the model with .cuda()")
    dt = _lib.precision_dtype(model.precision)
    with torch.no_grad():
        img = torch.as_tensor(np.ascontiguousarray(input_cubemap, dtype=np.float32)).to(dev)   # H2D, [6N,H,W,3]
        x4 = ops.cubepad_nhwc(img, 0, c_out=4)                       # NHWC3 -> NHWC4 (pad 0 = channel pad only)
        if dt != torch.float32:
            x4 = ops.nchw_to_nhwc(x4.reshape(1, 1, 1, -1), out_dtype=dt).reshape(x4.shape)
        score, feat = cam_device(x4, model)
        cube_score = ops.nhwc_to_nchw(score).cpu().numpy()
        cubic_feature = ops.nhwc_to_nchw(feat, out_dtype=torch.float32).cpu().numpy()
    # weight_softmax: fc.weight, shifted by its minimum when that is negative (class_activation_model.py:46-52).  8 MB of
    # device -> host copy plus two passes over it per call: cached per (storage, version) of the parameter.  The reference
    # hands out a fresh array every call; this one is shared between calls and therefore read-only.
    wt = model.fc.weight
    key = (wt.data_ptr(), wt._version, str(wt.device))
    cached = model.__dict__.get('_cam_weight_softmax')
    if cached is None or cached[0] != key:
        w = wt.detach().float().cpu().numpy()
        weight_softmax = np.squeeze(np.array(w, copy=True))
        if np.min(weight_softmax) < 0:
            weight_softmax -= np.min(weight_softmax)
        weight_softmax.flags.writeable = False
        model.__dict__['_cam_weight_softmax'] = cached = (key, weight_softmax)
    return cube_score[:, :num_class], cubic_feature, cached[1]
