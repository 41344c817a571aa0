"""``Equi2Cube`` with the reference's constructor / ``to_cube`` signature
(/root/reference/utils/equi_to_cube.py:12-129), sampling done by the HIP kernel
``cp360_equi2cube`` (K1) instead of 18 ``cv2.remap`` calls per frame.

The sampling grids are built once on the host in float64 with the reference's
recipe (edge-aligned pixel rays, yaw/pitch/roll rotation, table-interpolated
atan / acos, +1 offset, clamps, no seam wrap) and uploaded as [6, cd, cd, 2] f32 -
the same float32 maps the reference hands to cv2.remap (:122-125).
"""
import math

import numpy as np
import torch

from .. import ops
from .sph_utils import rotx, roty, rotz

# numpy dtypes uploaded as they are and converted to float32 on the device (anything else: one numpy pass on the host)
_TORCH_CASTABLE = (np.dtype(np.float64), np.dtype(np.float32), np.dtype(np.float16), np.dtype(np.uint8), np.dtype(np.int16),
                   np.dtype(np.int32), np.dtype(np.int64))

# yaw, pitch, roll in degrees for back, bottom, front, left, right, top (:17-22)
_VIEWS = ((180, 0, 0), (0, -90, 0), (0, 0, 0), (-90, 0, 0), (90, 0, 0), (0, 90, 0))


def build_grids(output_width, input_height, input_width, vfov=90):
    """Returns (inXs, inYs): lists of six float64 arrays [cd*cd] (0-based remap
    coordinates), identical to the reference's ``self.inXs / self.inYs``."""
    cd = int(output_width)
    H, W = int(input_height), int(input_width)
    half = math.tan((vfov * np.pi / 180) / 2)
    tl = np.array([-half * (cd / cd), -half, 1])                       # topLeft (:41-42)
    uv = np.array([-2 * tl[0] / cd, -2 * tl[1] / cd, 0])               # per-pixel step (:45-46)
    # monotone lookup tables replacing acos / atan (:49-56)
    step_acos, step_atan = np.pi / (2 * W), np.pi / (2 * H)
    tab_acos = np.append(-np.cos(np.arange(0, 2 * W) * step_acos), 1.)
    tab_atan = np.append(np.append(np.tan(step_atan / 2 - np.pi / 2),
                                   np.tan(np.arange(1, 2 * H) * step_atan - np.pi / 2)),
                         np.tan(-step_atan / 2 + np.pi / 2))
    jj, ii = np.meshgrid(range(cd), range(cd))
    jj, ii = jj.flatten(), ii.flatten()
    pts = np.concatenate((np.concatenate((tl[0] + uv[0] * jj[None, :], tl[1] + uv[1] * ii[None, :]), axis=0),
                          tl[2] + uv[2] * np.ones((1, jj.shape[0]))), axis=0)
    inXs, inYs = [], []
    for yaw, pitch, roll in np.array(_VIEWS) * np.pi / 180:
        rot = np.dot(np.dot(roty(yaw), rotx(pitch)), rotz(roll))       # :71
        px, py, pz = np.dot(rot, pts)
        nxz = np.sqrt(px ** 2 + pz ** 2)
        at_pole = nxz < 10e-10
        rest = np.logical_not(at_pole)
        phi = np.zeros(jj.shape[0])
        theta = np.zeros(jj.shape[0])
        phi[at_pole & (py > 0)] = np.pi / 2
        phi[at_pole & (py <= 0)] = -np.pi / 2
        # linear inverse of the tables (scipy interp1d in the reference, :91-96)
        phi[rest] = np.interp(py[rest] / nxz[rest], tab_atan, np.arange(0, 2 * H + 1)) * step_atan - (np.pi / 2)
        theta[rest] = np.interp(-pz[rest] / nxz[rest], tab_acos, np.arange(0, 2 * W + 1)) * step_acos
        flip = rest & (px < 0)
        theta[flip] = -theta[flip]
        gx = (theta / np.pi) * (W / 2) + (W / 2) + 1                    # :100
        gy = (phi / (np.pi / 2)) * (H / 2) + (H / 2) + 1                # :101
        gx[gx < 1] = 1                                                 # :104-108
        gx[gx >= W - 1] = W - 1
        gy[gy < 1] = 1
        gy[gy >= H - 1] = H - 1
        inXs.append(gx)
        inYs.append(gy)
    return inXs, inYs


def grids_f32(inXs, inYs, cube_dim):
    """[6, cd, cd, 2] float32 (x, y) = the maps ``to_cube`` passes to cv2.remap."""
    xs = np.stack(inXs).reshape(6, cube_dim, cube_dim)
    ys = np.stack(inYs).reshape(6, cube_dim, cube_dim)
    return np.ascontiguousarray(np.stack([xs, ys], axis=-1).astype(np.float32))


class Equi2Cube:
    def __init__(self, output_width, in_image, vfov=90, device='cuda', cv_fixed_point=True):
        """output_width: cube face size; in_image: an [H, W, C] array (only its shape
        is used, as in the reference) or an (H, W) tuple."""
        shape = in_image if isinstance(in_image, (tuple, list)) else in_image.shape
        assert shape[0] * 2 == shape[1]                                # :15
        self.out = {}
        self.output_width = self.output_height = int(output_width)
        self.input_height, self.input_width = int(shape[0]), int(shape[1])
        self.inXs, self.inYs = build_grids(output_width, shape[0], shape[1], vfov)
        self.grid_host = grids_f32(self.inXs, self.inYs, self.output_width)
        self.device = torch.device(device)
        self.cv_fixed_point = bool(cv_fixed_point)
        self._grid_dev = None
        self._grid_p3_dev = None
        self._stage = {}                                               # pinned host buffer of to_cube's D2H copy

    @property
    def grid(self):
        if self._grid_dev is None:
            self._grid_dev = torch.from_numpy(self.grid_host).to(self.device)
        return self._grid_dev

    @property
    def grid_p3(self):
        """[6, cd+6, cd+6, 2]: the sampling grid gathered through CubePad(3) (the library's host table,
        model/cube_pad.py:95-216).  A padded pixel is a copy of a cube pixel, so sampling it at that
        pixel's grid point gives the CubePad(3) output of resnet_cubic.py:165 without a second pass."""
        if self._grid_p3_dev is None:
            tab = ops.cubepad_table(self.output_width, 3)                     # [6, cd+6, cd+6] flat source index
            g = self.grid_host.reshape(-1, 2)[tab.reshape(-1)].reshape(tab.shape + (2,))
            self._grid_p3_dev = torch.from_numpy(np.ascontiguousarray(g)).to(self.device)
        return self._grid_p3_dev

    def to_cube(self, in_image):
        """[H, W, C=3] float array -> {0..5: [cd, cd, 3]} like the reference (:112-129):
        plain bilinear remap, no normalisation.  H2D + kernel + D2H."""
        img = np.asarray(in_image)
        # The frame crosses PCIe in the dtype it arrives in (the reference's driver hands in the float64 ``np.array(img) /
        # 255.0``) and is rounded to float32 on the device - the same IEEE rounding numpy's cast does, without a
        # single-threaded pass over 6 M elements on the host.  Read-only / byte-swapped / negatively strided (``img[..., ::-1]``, a
        # flipped frame: torch.from_numpy refuses those) / exotic inputs: numpy converts.
        if img.flags.writeable and img.dtype.isnative and img.dtype in _TORCH_CASTABLE and all(st >= 0 for st in img.strides):
            x = torch.from_numpy(img).to(self.device)[None].to(torch.float32)
        else:
            x = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))[None].to(self.device)
        cubes = ops.equi2cube(x, self.grid, self.output_width, torch.float32, 'nchw', scale=1.0,
                              mean=(0., 0., 0.), std=(1., 1., 1.), cv_fixed_point=self.cv_fixed_point)
        back = self._staging('out', (6, self.output_height, self.output_width, img.shape[2]), torch.float32)
        back.copy_(cubes.permute(0, 2, 3, 1), non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()           # the faces are in the (reused) pinned buffer
        host = back.numpy().astype(img.dtype if img.dtype.kind == 'f' else np.float32)   # fresh arrays per call, as :117-118
        for idx in range(6):
            self.out[idx] = host[idx]
        return self.out

    def _staging(self, name, shape, dtype):
        buf = self._stage.get(name)
        if buf is None or tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
            buf = self._stage[name] = torch.empty(shape, dtype=dtype).pin_memory()
        return buf

    def to_cube_batch(self, frames, out_dtype=torch.float32, layout='nhwc4', normalize=True):
        """Device path: frames [F, H, W, 3] u8 / f32 tensor on the GPU -> the
        normalised network input of dataset_feat_extractor.py:142-157 in one kernel
        ([6F, cd, cd, 4] NHWC4, [6F, 3, cd, cd] NCHW, or with layout 'nhwc4p3' the CubePad(3)-padded
        NHWC4 faces [6F, cd+6, cd+6, 4] the stem reads)."""
        if layout == 'nhwc4p3':
            if not normalize:
                raise ValueError("the padded layout is the network input: normalised only")
            return ops.equi2cube(frames, self.grid_p3, self.output_width + 6, out_dtype, 'nhwc4',
                                 cv_fixed_point=self.cv_fixed_point)
        if normalize:
            return ops.equi2cube(frames, self.grid, self.output_width, out_dtype, layout,
                                 cv_fixed_point=self.cv_fixed_point)
        return ops.equi2cube(frames, self.grid, self.output_width, out_dtype, layout, scale=1.0,
                             mean=(0., 0., 0.), std=(1., 1., 1.), cv_fixed_point=self.cv_fixed_point)
