"""Frame resize of /root/reference/static_model/dataset_feat_extractor.py:119-142
(``Image.fromarray(frame).convert('RGB').resize((equi_h, equi_w), resample=Image.LANCZOS)``) on the
device: uint8 frames in HBM -> uint8 frames in HBM, bit-exact with Pillow (K0, csrc/resize.hip).
The coefficient tables come from the library's host routine (Pillow's double-precision recipe)."""
import ctypes as C

import numpy as np
import torch

from .._lib import check, lib, ptr, require_gpu, stream


FILTERS = {'lanczos': 0, 'bicubic': 1}       # PIL.Image.LANCZOS / PIL.Image.BICUBIC (= Image.CUBIC)


def pil_tables(in_size, out_size, filter='lanczos'):
    """Pillow's coefficient tables of one axis: (bounds int32 [out, 2], kk int32 [out, ksize]) - host, no GPU."""
    L = lib()
    f = FILTERS[filter]
    ksize = L.cp360_resize_ksize2(int(in_size), int(out_size), f)
    if ksize < 0:
        check(ksize)
    bounds = np.empty((out_size, 2), dtype=np.int32)
    kk = np.empty((out_size, ksize), dtype=np.int32)
    rc = L.cp360_resize_coeffs_host2(int(in_size), int(out_size), f, bounds.ctypes.data_as(C.c_void_p),
                                     kk.ctypes.data_as(C.c_void_p))
    if rc < 0:
        check(rc)
    return bounds, kk


def lanczos_tables(in_size, out_size):
    return pil_tables(in_size, out_size, 'lanczos')


class LanczosResize:
    """``LanczosResize((h_in, w_in), (h_out, w_out))(frames)``: frames u8 [F, h_in, w_in, 3] on the GPU
    -> u8 [F, h_out, w_out, 3].  Note the reference passes PIL's (width, height) = (cfg.equi_h,
    cfg.equi_w) = (1920, 960) (config.yaml:15-16): out_hw here is (960, 1920).  ``filter='bicubic'`` gives
    Pillow's Image.CUBIC (the overlay's heat-map upsample, utils/utils.py:21)."""

    def __init__(self, in_hw, out_hw, device='cuda', filter='lanczos'):
        self.in_hw = (int(in_hw[0]), int(in_hw[1]))
        self.out_hw = (int(out_hw[0]), int(out_hw[1]))
        self.device = torch.device(device)
        self.h = self.v = None
        if self.in_hw[1] != self.out_hw[1]:
            b, k = pil_tables(self.in_hw[1], self.out_hw[1], filter)
            self.h = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), k.shape[1])
        if self.in_hw[0] != self.out_hw[0]:
            b, k = pil_tables(self.in_hw[0], self.out_hw[0], filter)
            self.v = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), k.shape[1])

    def __call__(self, frames, out=None):
        require_gpu(frames, out)
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
            raise ValueError("frames must be uint8 [F, H, W, 3]")
        if tuple(frames.shape[1:3]) != self.in_hw:
            raise ValueError("resizer built for %s frames, got %s" % (self.in_hw, tuple(frames.shape[1:3])))
        frames = frames.contiguous()
        F = frames.shape[0]
        if out is None:
            out = torch.empty((F,) + self.out_hw + (3,), dtype=torch.uint8, device=frames.device)
        tmp = None
        if self.h is not None and self.v is not None:
            tmp = torch.empty((F, self.in_hw[0], self.out_hw[1], 3), dtype=torch.uint8, device=frames.device)
        hb, hk, hks = self.h if self.h is not None else (None, None, 0)
        vb, vk, vks = self.v if self.v is not None else (None, None, 0)
        check(lib().cp360_resize_lanczos_u8(ptr(frames), ptr(out), ptr(tmp), F, self.in_hw[0], self.in_hw[1],
                                            self.out_hw[0], self.out_hw[1], ptr(hb), ptr(hk), hks, ptr(vb), ptr(vk),
                                            vks, stream()))
        return out
