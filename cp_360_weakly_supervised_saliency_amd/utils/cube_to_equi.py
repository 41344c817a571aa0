"""``Cube2Equi`` with the reference's constructor / ``to_equi_nn`` signature
(/root/reference/utils/cube_to_equi.py:12-66), sampling done by the HIP kernel
``cp360_cube2equi`` (K6): one pass that picks each output pixel's face and does the
4-tap bilinear (the reference runs six full-size grid_samples and keeps a masked sixth
of each).  ``saliency()`` adds the fused channel max of test_temporal.py:83-84.
"""
import numpy as np
import torch

from .. import ops
from .sph_utils import xy2angle, pruned_inf, to_3dsphere, get_face, face_to_cube_coord, norm_to_cube


def sample_positions(out_coord, w, align_corners=False):
    """Pixel-space sampling position per output pixel, float32 [2w, 4w, 2].

    cube_to_equi.py:58 normalises the float32 grid with ONE data-dependent scalar
    M = max(grid): gn = (g - M/2) / (M/2); grid_sample then maps gn back to pixels:
    align_corners=False (what torch >= 1.3 executes for the reference's call, and this
    build's default): ((gn + 1) * w - 1) / 2;  True (torch 0.3/0.4 era):
    (gn + 1) / 2 * (w - 1).  Same float32 operation order as torch."""
    g = np.asarray(out_coord, dtype=np.float32)
    M = np.float32(g.max())
    gn = (g - M / np.float32(2)) / (M / np.float32(2))
    if align_corners:
        return ((gn + np.float32(1)) / np.float32(2) * np.float32(w - 1)).astype(np.float32)
    return (((gn + np.float32(1)) * np.float32(w) - np.float32(1)) / np.float32(2)).astype(np.float32)


class Cube2Equi:
    def __init__(self, input_w, align_corners=False, device='cuda'):
        w = int(input_w)
        out_w, out_h = w * 4, w * 2
        XX, YY = np.meshgrid(range(out_w), range(out_h))
        theta, phi = xy2angle(XX, YY, out_w, out_h)
        theta, phi = pruned_inf(theta), pruned_inf(phi)
        _x, _y, _z = to_3dsphere(theta, phi, 1)
        face_map = get_face(_x, _y, _z, np.zeros((out_h, out_w)))
        x_o, y_o = face_to_cube_coord(face_map, _x, _y, _z)
        out_coord = np.transpose(np.array([x_o, y_o]), (1, 2, 0))      # h x w x 2
        self.out_coord = norm_to_cube(out_coord, w)
        self.face_map = face_map
        self.input_w = w
        self.align_corners = bool(align_corners)
        self.device = torch.device(device)
        self._dev = None

    def _tables(self):
        if self._dev is None:
            fm = torch.from_numpy(self.face_map.astype(np.int8)).to(self.device)
            pc = torch.from_numpy(sample_positions(self.out_coord, self.input_w, self.align_corners)).to(self.device)
            self._dev = (fm.contiguous(), pc.contiguous())
        return self._dev

    def to_equi_nn(self, input_data):
        """input_data: 6 x c x w x w float tensor on the GPU -> 1 x c x 2w x 4w."""
        fm, pc = self._tables()
        full, _ = ops.cube2equi(input_data.float(), fm, pc, layout='nchw', want_full=True, want_max=False)
        return full

    def saliency(self, hidden, layout='nchw'):
        """Final hidden state(s) [6B, C, w, w] (or NHWC [6B, w, w, C]) -> [B, 2w, 4w]:
        to_equi_nn + max over channels + squeeze (test_temporal.py:82-85), fused."""
        fm, pc = self._tables()
        _, mx = ops.cube2equi(hidden, fm, pc, layout=layout, want_full=False, want_max=True)
        return mx
