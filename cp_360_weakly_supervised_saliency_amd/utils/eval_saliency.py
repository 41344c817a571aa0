"""Saliency metrics with the reference's names and call signatures
(/root/reference/utils/eval_saliency.py: AUC_Judd :90-146, AUC_Borji :14-87, CorrCoeff :149-176,
similarity :179-190), computed on the GPU by csrc/metrics.hip (K8).

Maps may be numpy arrays (as in temporal_model/test_temporal.py:101-110) or device tensors; they are resized
to 120 x 240 the way the reference's ``cv2.resize(x, (240, 120), cv2.INTER_LANCZOS4)`` actually runs (the flag
sits in the ``dst`` slot: default bilinear).  Random draws come from numpy's GLOBAL generator exactly where
the reference draws them (``np.random.randn`` for AUC_Judd's jitter, ``np.random.randint`` for AUC_Borji's
splits), so ``np.random.seed(s)`` in front of a call gives the reference's value for the same seed.
No CPU fallback: the functions need the HIP library and a GPU.

Types: the maps the path produces are float32 and are resized in float32 (cv2.resize keeps the input type); a float64
input (e.g. a ground-truth map stored as float64) is cast to float32 first - the only consumer of its values is the
``F > mean(F) + 2 std(F)`` fixation threshold, evaluated in float64 from those float32 values.  AUC_Judd normalises in
float64 with jitter (the float64 ``randn`` term promotes the map, as in numpy) and in float32 with ``jitter=False``.
"""
import ctypes as C

import numpy as np
import torch

from .._lib import check, lib, ptr, require_gpu, stream

SIZE = (240, 120)            # (width, height) of the reference's cv2.resize


def _dev(a):
    if torch.is_tensor(a):
        require_gpu(a)
        t = a.detach().to(torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    if t.dim() != 2:
        raise ValueError("metrics take 2-D maps")
    return t.contiguous()


def resize_linear(a, dsize=SIZE):
    """cv2.resize(a, dsize=(width, height)) with the default INTER_LINEAR, on the device."""
    t = _dev(a)
    dw, dh = dsize
    out = torch.empty((dh, dw), dtype=torch.float32, device=t.device)
    check(lib().cp360_resize_linear_f32(ptr(t), t.shape[0], t.shape[1], ptr(out), dh, dw, stream()))
    return out


def _prepare(saliency_map, fixation_map, jitter, mode):
    if not np.any(fixation_map.detach().cpu().numpy() if torch.is_tensor(fixation_map) else fixation_map):
        raise ValueError('no fixation_map')                       # the reference prints this and exit()s
    S, F = resize_linear(saliency_map), resize_linear(fixation_map)
    n = S.numel()
    work = torch.empty(lib().cp360_metric_work_bytes(n), dtype=torch.uint8, device=S.device)
    jit = None
    if jitter is not None:
        jit = torch.from_numpy(np.ascontiguousarray(jitter, dtype=np.float64)).to(S.device)
    check(lib().cp360_metric_auc_prepare(ptr(S), ptr(F), ptr(jit), n, mode, ptr(work), stream()))
    return n, work


def AUC_Judd(saliency_map, fixation_map, jitter=True, to_plot=False):
    """eval_saliency.py:90-146."""
    jit = np.random.randn(SIZE[1], SIZE[0]) / 1e7 if jitter else None      # drawn where the reference draws it (:106-109)
    n, work = _prepare(saliency_map, fixation_map, jit, 0)
    out = torch.empty(2, dtype=torch.float64, device=work.device)
    check(lib().cp360_metric_auc_judd(n, ptr(work), ptr(out), stream()))
    return float(out[0].item())


def AUC_Borji(saliency_map, fixation_map, Nsplits=100, stepSize=0.01, to_plot=False):
    """eval_saliency.py:14-87."""
    n, work = _prepare(saliency_map, fixation_map, None, 1)
    n_fix = int(work[:8].view(torch.float64).item())               # the splits' shape depends on it (:53)
    rr = np.random.randint(0, high=n, size=(n_fix, Nsplits))
    rr_d = torch.from_numpy(np.ascontiguousarray(rr, dtype=np.int32)).to(work.device)
    aucs = torch.empty(Nsplits, dtype=torch.float64, device=work.device)
    check(lib().cp360_metric_auc_borji(n, int(Nsplits), ptr(rr_d), C.c_double(float(stepSize)), ptr(work), ptr(aucs),
                                       stream()))
    return float(aucs.mean().item())


def _cc_sim(map1, map2):
    a, b = resize_linear(map1), resize_linear(map2)
    out = torch.empty(2, dtype=torch.float64, device=a.device)
    check(lib().cp360_metric_cc_sim(ptr(a), ptr(b), a.numel(), ptr(out), stream()))
    return out.cpu().numpy()


def CorrCoeff(map1, map2):
    """eval_saliency.py:149-176."""
    return float(_cc_sim(map1, map2)[0])


def similarity(map1, map2):
    """eval_saliency.py:179-190."""
    return float(_cc_sim(map1, map2)[1])
