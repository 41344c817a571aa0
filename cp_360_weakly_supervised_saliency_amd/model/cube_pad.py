"""``CubePad`` / ``CubePadding`` with the reference's signatures
(/root/reference/model/cube_pad.py:23-42,45-70) running the HIP kernel
``cp360_cubepad_nchw`` / ``cp360_cubepad_nhwc`` (K2): one launch instead of a Python
loop over frames with 33 ``torch.cat`` + 8 ``index_select`` per group of 6 faces.
"""
import numpy as np
import torch.nn as nn

from .. import ops


def get_pad_size(lrtd_pad):
    return ops.pads_of(lrtd_pad)


class CubePad(nn.Module):
    """Input [6N, C, H, W] -> [6N, C, H + pt + pd, W + pl + pr]; the batch must be a
    multiple of 6 (faces back, down, front, left, right, top per frame).  The
    reference print()s and exit()s on a bad batch (cube_pad.py:33-35); this raises
    ValueError.  ``use_gpu`` is kept for signature compatibility: the op is GPU-only."""

    def __init__(self, lrtd_pad, use_gpu=True):
        super(CubePad, self).__init__()
        self.lrtd_pad = lrtd_pad if isinstance(lrtd_pad, (int, np.integer)) else [int(v) for v in lrtd_pad]
        self.use_gpu = use_gpu

    def forward(self, x):
        if x.size()[0] % 6 != 0:
            raise ValueError('CubePad size mismatch!')
        v = x.permute(0, 2, 3, 1)
        if v.is_contiguous() and not x.is_contiguous():
            # channels_last tensor (the fused pipeline's layout): pad pixel vectors
            return ops.nchw_view(ops.cubepad_nhwc(v, self.lrtd_pad))
        return ops.cubepad_nchw(x, self.lrtd_pad)


class CubePadding(CubePad):
    """The reference's single-frame variant ([6, C, H, W]); same kernel."""

    def forward(self, x):
        if x.size()[0] != 6:
            raise ValueError('CubePadding expects exactly 6 faces')
        return super(CubePadding, self).forward(x)
