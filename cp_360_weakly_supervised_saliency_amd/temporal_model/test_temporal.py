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
        # (with the stage context the two intermediate activations live in its workspace)
        from .. import stage_ctx
        self.a = None if stage_ctx.USE_CTX else [torch.empty((n6, w, w, 4 * ch), dtype=dt, device=dev) for _ in range(2)]
        self.minmax = torch.empty((self.B, 2), dtype=torch.float32, device=dev)
        # measurement hook (bench.py stage_split, never set on the hot path): a list that receives three HIP events of the
        # launch stream - before the window, after its last cell update, after cube -> equi + channel max
        self.stage_events = None
        self.scratch = torch.empty((self.B * 256 * 2,), dtype=torch.float32, device=dev)

    def run(self, cam, return_hidden=False, sliding=False, return_all_steps=False):
        """cam: f32 [B, T, 6*w*w, C] (NHWC cube_feat of every frame of every window,
        frame-major) on the device.  Returns saliency f32 [B, 2w, 4w].

        return_all_steps=True: the map of the hidden state after EVERY step of the window,
        f32 [B, T, 2w, 4w] - ``[:, t]`` is test_temporal.py:82-85 applied to ``hidden`` after step t of the loop
        :76-79 (the reference keeps only the last one; ``[:, T-1]`` is that map).

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
        from .. import stage_ctx

        def mark():
            if self.stage_events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                self.stage_events.append(e)
        mark()
        if stage_ctx.USE_CTX:
            # the whole window in ONE C call (cp360_clstm_window, csrc/ctx.hip): min / max, initial state, T cell updates -
            # and, with at most two windows per call, Conv1's x half batched over all T frames
            stage = self.cell.__dict__.get('_stage')
            if stage is None:
                stage = self.cell.__dict__['_stage'] = stage_ctx.ClstmStage(self.cell)
            h_all = None
            if return_all_steps:
                h_all = torch.empty((T,) + tuple(self.h_f32.shape), dtype=torch.float32, device=self.h_f32.device)
            stage.window(cam, B, T, self.w, self.xh, self.c, self.h_f32, self.minmax, self.scratch, clip_stride=stride, h_all=h_all)
            mark()
            if return_all_steps:
                sal = torch.stack([self.c2e.saliency(h_all[t], layout='nhwc') for t in range(T)], dim=1)
            else:
                sal = self.c2e.saliency(self.h_f32, layout='nhwc')
            mark()
            return (sal, self.h_f32) if return_hidden else sal
        ops.window_minmax(cam, B, T * P * cin, self.minmax, self.scratch, stride)
        # hidden = cell = (frame0 - mn) / (mx - mn)
        ops.window_normalize(cam, self.minmax, self.xh, cin, self.c[0], B, T, 0, P, cin, stride)
        cur = 0
        per_clip = stride if sliding else T * P * cin
        ops.window_normalize(cam, self.minmax, self.xh, 0, None, B, T, 0, P, cin, stride)
        steps = [] if return_all_steps else None
        for t in range(T):
            # frame t+1's normalisation rides on step t's gate kernel (its x half of xh is free by then)
            nxt = (cam, self.minmax, P, per_clip, t + 1) if t + 1 < T else None
            self.cell.step_nhwc(self.xh, self.c[cur], self.c[cur ^ 1],
                                self.h_f32 if (t == T - 1 or return_all_steps) else None, bufs=self.a, x_next=nxt)
            cur ^= 1
            if return_all_steps:
                steps.append(self.c2e.saliency(self.h_f32, layout='nhwc'))
        mark()
        sal = torch.stack(steps, dim=1) if return_all_steps else self.c2e.saliency(self.h_f32, layout='nhwc')
        mark()
        return (sal, self.h_f32) if return_hidden else sal


def infer_video_dir(cell, c2e, indir, vid_name, output_dir=None, num_subseq=5, max_windows=64):
    """File-based counterpart of the reference's ``test()`` loop (test_temporal.py:41-88): read the
    ``<indir>/<vid_name>/cube_feat/*.npy`` files its static stage wrote ([6, C, w, w] float32 per frame),
    run every stride-1 window of ``num_subseq`` frames through the ConvLSTM on the GPU (``max_windows`` windows per
    launch group, zero-copy over one feature sequence) and, with ``output_dir``, save each window's map as
    ``<output_dir>/<vid_name>/{:05}.npy`` named after the window's LAST frame index (:86-88).
    Returns float32 [n_windows, 2w, 4w] (host).  The reference skips the last window of a sequence
    (``idx >= len(seq) - num_subseq: continue``, :60-61); so does this."""
    import os
    import numpy as np
    from ..utils import npy_io
    feat_dir = os.path.join(indir, vid_name, 'cube_feat')
    seq = sorted(f for f in os.listdir(feat_dir) if f.endswith('.npy'))
    n_win = len(seq) - num_subseq                     # idx = 0 .. len(seq) - num_subseq - 1
    if n_win <= 0:
        return np.zeros((0, 0, 0), dtype=np.float32)
    feats = np.stack([npy_io.cube_feat_to_cam_nhwc(np.load(os.path.join(feat_dir, f))) for f in seq])
    w = int(round((feats.shape[1] // 6) ** 0.5))
    dev = next(cell.parameters()).device
    cam = torch.from_numpy(feats).to(dev)
    maps = []
    for lo in range(0, n_win, max_windows):
        nb = min(max_windows, n_win - lo)
        runner = ClipRunner(cell, c2e, nb, num_subseq, w)
        maps.append(runner.run(cam[lo:lo + nb + num_subseq - 1], sliding=True).cpu().numpy())
    maps = np.concatenate(maps)
    if output_dir is not None:
        for idx in range(n_win):
            npy_io.save_saliency(output_dir, vid_name, idx + num_subseq - 1, maps[idx])
    return maps
