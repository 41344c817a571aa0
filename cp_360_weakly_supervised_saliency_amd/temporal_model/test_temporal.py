"""Clip semantics of /root/reference/temporal_model/test_temporal.py:57-85 on the
device: per-window min / max, ``hidden = cell = normalised first frame``, T ConvLSTM
steps (the first frame fed again), cube -> equi, channel max.  File I/O, overlays and
the GT metrics of the reference's ``test()`` / ``main()`` are outside the hot path.
"""
import torch

from .. import _lib, ops


class ClipRunner:
    """Runs B windows of T frames each through a ``ConvLSTMCell`` in lock step
    (the recurrence is serial in t; batching clips raises the GEMM M to 294*B)."""

    def __init__(self, cell, c2e, B, T, w=7):
        self.cell, self.c2e, self.B, self.T, self.w = cell, c2e, int(B), int(T), int(w)
        dev = next(cell.parameters()).device
        dt = _lib.precision_dtype(cell.precision)
        cin, ch = cell.input_size, cell.hidden_size
        n6 = 6 * self.B
        self.P = 6 * w * w
        self.xh = torch.empty((n6, w, w, cin + ch), dtype=dt, device=dev)
        self.c = [torch.empty((n6, w, w, ch), dtype=torch.float32, device=dev) for _ in range(2)]
        self.h_f32 = torch.empty((n6, w, w, ch), dtype=torch.float32, device=dev)
        self.a = [torch.empty((n6, w, w, 4 * ch), dtype=dt, device=dev) for _ in range(2)]
        self.minmax = torch.empty((self.B, 2), dtype=torch.float32, device=dev)
        self.scratch = torch.empty((self.B * 256 * 2,), dtype=torch.float32, device=dev)

    def run(self, cam, return_hidden=False, sliding=False):
        """cam: f32 [B, T, 6*w*w, C] (NHWC cube_feat of every frame of every window,
        frame-major) on the device.  Returns saliency f32 [B, 2w, 4w].

        sliding=True: ``cam`` is ONE feature sequence [B+T-1, 6*w*w, C] and window b covers
        frames b .. b+T-1 - the reference's stride-1 sliding window (test_temporal.py:57-65)
        run as B windows in lock step without copying the overlapping frames."""
        B, T, P = self.B, self.T, self.P
        cin = self.cell.input_size
        if cin != self.cell.hidden_size:
            raise ValueError("hidden = cell = first frame needs input_size == hidden_size (test_temporal.py:70-73)")
        if sliding and cam.shape[0] != B + T - 1:
            raise ValueError("a sliding run over %d windows of %d frames needs %d frames" % (B, T, B + T - 1))
        stride = P * cin if sliding else 0
        ops.window_minmax(cam, B, T * P * cin, self.minmax, self.scratch, stride)
        # hidden = cell = (frame0 - mn) / (mx - mn)
        ops.window_normalize(cam, self.minmax, self.xh, cin, self.c[0], B, T, 0, P, cin, stride)
        cur = 0
        for t in range(T):
            ops.window_normalize(cam, self.minmax, self.xh, 0, None, B, T, t, P, cin, stride)
            self.cell.step_nhwc(self.xh, self.c[cur], self.c[cur ^ 1],
                                self.h_f32 if t == T - 1 else None, bufs=self.a)
            cur ^= 1
        sal = self.c2e.saliency(self.h_f32, layout='nhwc')
        return (sal, self.h_f32) if return_hidden else sal
