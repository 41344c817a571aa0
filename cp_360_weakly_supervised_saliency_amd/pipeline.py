"""End-to-end engine: equirectangular frames in HBM -> saliency maps in HBM.

    frames u8 [B, T, H, W, 3]   (or [B, T, Hs, Ws, 3] decoded frames when source_hw is given:
      K0 PIL-exact Lanczos resize to H x W              dataset_feat_extractor.py:131-133)
      K1 equi -> cube (+/255, ImageNet normalise)        dataset_feat_extractor.py:138-157
      K2 CubePad(3) -> K3 ResNet-50-cubic -> layer4      model/resnet_cubic.py:163-175
      K4 CAM (shifted fc weight as 1x1 conv)             class_activation_model.py:46-83
      K7 window min/max + normalise                      test_temporal.py:66-73
      K5 T x ConvLSTM step                               model/clstm.py:42-82
      K6 cube -> equi + channel max                      test_temporal.py:82-85
    saliency f32 [B, 2w, 4w]

It is the fused counterpart of the reference's two drivers chained through .npy files
(inference.sh): nothing leaves the device between the stages.  Each clip is one
window (non-overlapping clips; the reference's stride-1 sliding window repeats the
ConvLSTM work seq_len times and is available through ``ClipRunner`` directly).
"""
import torch

from . import _lib, ops
from .model.resnet_cubic import resnet50
from .model.clstm import ConvLSTMCell
from .static_model.class_activation_model import cam_device
from .temporal_model.test_temporal import ClipRunner
from .utils.cube_to_equi import Cube2Equi
from .utils.equi_to_cube import Equi2Cube
from .utils.resize import LanczosResize


class SaliencyEngine:
    def __init__(self, resnet_state, clstm_state, equi_hw=(1024, 2048), cube_dim=224, clips=1, frames=16,
                 precision='fp32', device='cuda', align_corners=False, cv_fixed_point=True,
                 input_size=1000, hidden_size=1000, frame_chunk=None, source_hw=None, static_precision=None,
                 return_all_steps=False):
        self.device = torch.device(device)
        # ``precision`` is the arithmetic type of the temporal stage (the ConvLSTM: 81 % of the flops);
        # ``static_precision`` that of the static stage (cube projection output, ResNet-50, CAM conv).  The
        # stages meet at the f32 CAM scores.  A bf16 engine runs the static stage in fp16 by default: same
        # MFMA rate and bytes, 3 more mantissa bits - measured on a 1024x2048 T=16 clip against the oracle
        # (tests/probe_precision_split.py): bf16 ResNet -> map max|d| 3.4e-3, dCC 1.2e-3 whatever the ConvLSTM
        # runs in; fp16 ResNet + bf16 ConvLSTM -> 5.2e-4, dCC 2.2e-4.
        self.precision = precision
        self.static_precision = static_precision or ('fp16' if precision == 'bf16' else precision)
        # fp16 has a 5-bit exponent (max 65504) where bf16 has f32's range (the reference runs fp32,
        # class_activation_model.py:55-64).  When the fp16 static stage was chosen BY DEFAULT for a bf16 engine, EVERY batch
        # is checked: the window min / max kernel - which reads every CAM score anyway - poisons a window's (min, max) pair
        # with NaN when it meets an inf / NaN (csrc/misc.hip), and the engine copies those 8 bytes per clip to pinned host
        # memory behind the batch.
        #   * the FIRST batch is waited for (a checkpoint whose folded-BN scales / activations exceed fp16's range shows on
        #     any input): non-finite -> the static stage switches to bf16 for good and the batch is recomputed;
        #   * every LATER batch is looked at WITHOUT waiting, when a following call finds its copy complete (a blocking
        #     read-back per batch cost 0.3 ms of a 16 ms step and 2 ms of a 6 ms one - the host could no longer queue the
        #     next batch while the GPU worked): a non-finite pair then switches the stage to bf16 for every batch not yet
        #     queued, records the batch in ``overflow_batches`` and warns; ``check()`` waits for everything in flight and
        #     raises FloatingPointError if any batch overflowed.  The maps of such a batch are NaN - loud, never plausible.
        # ``static_precision`` / ``fp16_fallback`` say what runs, and bench.py reports them.  An explicit
        # static_precision='fp16' (or precision='fp16') is honoured as given and not checked; ``nonfinite()`` lets a caller
        # check any batch.
        self._guard = static_precision is None and precision == 'bf16'
        self._first_checked = False
        self._pending = []                                 # [batch index, event, pinned [B, 2] copy of the window min / max]
        self._free_hosts = []
        self._batch = 0
        self.overflow_batches = []
        self.fp16_fallback = False
        self.dtype = _lib.precision_dtype(self.static_precision)
        self.B, self.T = int(clips), int(frames)
        # True: one map per ConvLSTM step, [B, T, 2w, 4w]; False: the window's final map [B, 2w, 4w]
        self.return_all_steps = bool(return_all_steps)
        self.H, self.W = equi_hw
        self.cube_dim = int(cube_dim)
        self.w = self.cube_dim // 32
        # frames per static-stage launch group (bounds activation memory; None = all)
        self.frame_chunk = frame_chunk or self.B * self.T

        self.resnet = resnet50(precision=self.static_precision)
        missing, unexpected = self.resnet.load_state_dict(_to_tensors(resnet_state), strict=False)
        bad = [k for k in missing if not k.endswith('num_batches_tracked')]
        if bad or unexpected:
            raise KeyError("resnet state dict mismatch: missing %s unexpected %s" % (bad, unexpected))
        self.resnet.to(self.device).eval()
        self.cell = ConvLSTMCell(input_size, hidden_size, precision=precision)
        self.cell.load_state_dict(_to_tensors(clstm_state))           # strict, as test_temporal.py:149
        self.cell.to(self.device).eval()
        # decoded frames of another size are first resized as the reference does (PIL LANCZOS, K0)
        self.source_hw = None if source_hw is None else (int(source_hw[0]), int(source_hw[1]))
        self.resize = None
        if self.source_hw is not None and self.source_hw != (self.H, self.W):
            self.resize = LanczosResize(self.source_hw, (self.H, self.W), device=self.device)
        self.e2c = Equi2Cube(self.cube_dim, (self.H, self.W), device=self.device, cv_fixed_point=cv_fixed_point)
        self.c2e = Cube2Equi(self.w, align_corners=align_corners, device=self.device)
        self.runner = ClipRunner(self.cell, self.c2e, self.B, self.T, self.w)
        self.cam = torch.empty((self.B, self.T, 6 * self.w * self.w, input_size), dtype=torch.float32,
                               device=self.device)
        self._mm_host = None                               # pinned copy of the runner's [B, 2] window min / max
        self._cam2, self._side = None, None                # stream(): second CAM buffer and the static stage's stream
        if self._guard:                                    # pinned flag buffers up front: hipHostMalloc inside a timed step is a 1-2 ms hiccup
            self._free_hosts = [torch.empty((self.B, 2), dtype=torch.float32).pin_memory() for _ in range(4)]

    def static_stage(self, frames, cam=None):
        """frames u8/f32 [F, H, W, 3] on the device -> CAM f32 [F, 6*w*w, 1000] written
        into self.cam (frame-major, NHWC; ``cam``: another buffer of the same shape)."""
        F = frames.shape[0]
        cam = self.cam if cam is None else cam
        cam_flat = cam.view(self.B * self.T, 6 * self.w * self.w, -1)
        for lo in range(0, F, self.frame_chunk):
            hi = min(F, lo + self.frame_chunk)
            chunk = frames[lo:hi]
            if self.resize is not None:
                chunk = self.resize(chunk)
            # K1 writes the CubePad(3)-padded faces directly (one pass instead of project + pad)
            x4 = self.e2c.to_cube_batch(chunk, out_dtype=self.dtype, layout='nhwc4p3')
            cam_device(x4, self.resnet, out=cam_flat[lo:hi], padded=True, want_feat=False)     # CAM conv writes the clip buffer directly
        return cam

    def temporal_stage(self, cam=None):
        return self.runner.run(self.cam if cam is None else cam, return_all_steps=self.return_all_steps)

    def stream(self, batches):
        """Software-pipelined form for a STREAM of batches (serving): a generator that takes an iterable of frame batches
        ([B, T, H, W, 3] each) and yields one saliency tensor per batch, in order - the static stage of batch k+1 runs on a
        second HIP stream beside the ConvLSTM of batch k (the clip kernel owns every CU's LDS, so the static stage's
        workgroups get its tails and launch gaps: +2-3 % throughput, tools/overlap_probe.py).  Results are the same bits as
        ``__call__`` gives batch by batch (same kernels, same order inside each stage; the two CAM buffers alternate).  Each
        yielded tensor is overwritten two batches later - consume or clone it.  The fp16 range guard polls as in ``__call__``
        (a switch to bf16 takes effect for the batches whose static stage has not been queued yet)."""
        if self._cam2 is None:
            self._cam2 = torch.empty_like(self.cam)
            self._side = torch.cuda.Stream(device=self.device)
        cams = (self.cam, self._cam2)
        main = torch.cuda.current_stream(self.device)
        side = self._side
        out2 = [None, None]
        pending = None                                     # (index, cam buffer, event: its static stage is done)
        temporal_done = [None, None]                       # events: the ConvLSTM that read cam buffer i is done
        # (no ``with torch.no_grad()`` around the loop: a context manager held across ``yield`` would leak into the consumer)
        for i, frames in enumerate(batches):
            B, T = frames.shape[:2]
            if (B, T) != (self.B, self.T):
                raise ValueError("engine built for %dx%d clips x frames" % (self.B, self.T))
            if self._guard_on() and self._poll():
                import warnings
                warnings.warn("fp16 static stage overflowed on batch(es) %s (their maps are NaN): switching the static stage "
                              "to bf16" % self.overflow_batches)
                main.synchronize()
                side.synchronize()
                self._fallback_to_bf16()
            flat = frames.reshape((B * T,) + tuple(frames.shape[2:]))
            # the batch was allocated on the caller's stream and is READ on the side stream: tell the caching allocator, or the
            # block could be handed to the caller's next allocation (the H2D of batch i+1) while the static stage still reads it
            if frames.is_cuda:
                frames.record_stream(side)
            buf = cams[i & 1]
            side.wait_stream(main)                         # the frames (and everything the caller queued) are ready
            if temporal_done[i & 1] is not None:
                side.wait_event(temporal_done[i & 1])      # batch i-2's ConvLSTM has read this buffer
            with torch.no_grad(), torch.cuda.stream(side):
                self.static_stage(flat, cam=buf)
                ev = torch.cuda.Event()
                ev.record(side)
            if pending is not None:
                yield self._stream_temporal(pending, temporal_done, out2, main)
            pending = (i, buf, ev)
        if pending is not None:
            yield self._stream_temporal(pending, temporal_done, out2, main)

    def _stream_temporal(self, pending, temporal_done, out2, main):
        i, buf, ev = pending
        main.wait_event(ev)
        with torch.no_grad():
            sal = self.temporal_stage(buf)
            if out2[i & 1] is None:
                out2[i & 1] = torch.empty_like(sal)
            out2[i & 1].copy_(sal)                         # (the runner's map buffer is reused by the next window)
        if self._guard_on():
            if not self._first_checked:
                self._first_checked = True
                if self.nonfinite():
                    raise FloatingPointError("fp16 static stage overflowed on the first batch of the stream: build the engine with "
                                             "static_precision='bf16' (or run one batch through __call__ first, which repairs it)")
            else:
                self._queue_flag()
        self._batch += 1
        done = torch.cuda.Event()
        done.record(main)
        temporal_done[i & 1] = done
        return out2[i & 1]

    # ---- hipGraph replay (launch-bound small configurations: one frame / one clip)
    def capture(self, frames):
        """Record the whole path over ``frames`` (u8/f32 [B, T, H, W, 3] on the device) into a HIP
        graph.  Later ``__call__``s replay it: ~70 + 7*T kernel launches become one graph launch, which
        is what bounds single-frame / single-clip latency (BASELINE configs C2, C3).  The graph reads the
        engine's OWN copy of the input (``frames`` is never written): every call copies its frames into that
        static buffer first.  The returned maps alias the graph's output buffer and are overwritten by the
        next call - clone them to keep them."""
        with torch.no_grad():
            self._graph = None
            self._forward(frames)                       # warm-up: packs weights, sizes every buffer
            torch.cuda.synchronize()
            self._graph_in = frames.clone()             # engine-owned static input (the caller's tensor stays untouched)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._graph_out = self._forward(self._graph_in)
            self._graph = g
        return self

    def close(self):
        """Release the device memory the engine owns outside torch's allocator (the stage contexts' packed weights) and
        its cached workspaces; the engine re-packs on the next call.  ``del engine`` does the same through the objects'
        finalisers (the stage objects hold their modules weakly, so there is no cycle to wait for)."""
        self._graph = None
        self._pending = []
        for mod in (self.resnet, self.cell):
            st = mod.__dict__.get('_stage')
            if st is not None:
                st.close()

    def nonfinite(self):
        """True when the last batch's CAM scores held an inf / NaN: the window min / max pairs of that batch (poisoned
        with NaN by the reduction kernel, csrc/misc.hip) read back - 8 bytes per clip and a sync.  With an fp16 static
        stage this is how activations beyond 65504 show."""
        if self._mm_host is None:
            self._mm_host = torch.empty(tuple(self.runner.minmax.shape), dtype=torch.float32).pin_memory()
        self._mm_host.copy_(self.runner.minmax, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return not bool(torch.isfinite(self._mm_host).all())

    def _fallback_to_bf16(self):
        self.resnet.set_precision('bf16')                  # plans are re-packed on the next call (the stamp changed)
        self.static_precision = 'bf16'
        self.dtype = _lib.precision_dtype('bf16')
        self.fp16_fallback = True
        self._guard = False
        self._graph = None                                 # (captured launches are the fp16 ones)

    def _guard_on(self):
        return self._guard and self.static_precision == 'fp16'

    def _queue_flag(self):
        """Copy this batch's window min / max pairs to pinned memory behind the batch (8 bytes per clip, no wait)."""
        host = self._free_hosts.pop() if self._free_hosts else \
            torch.empty(tuple(self.runner.minmax.shape), dtype=torch.float32).pin_memory()
        host.copy_(self.runner.minmax, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((self._batch, ev, host))

    def _poll(self, wait=False):
        """Look at the flags whose copies are complete (all of them with ``wait``); True when a batch overflowed."""
        bad = False
        while self._pending and (wait or self._pending[0][1].query()):
            idx, ev, host = self._pending.pop(0)
            if wait:
                ev.synchronize()
            if not bool(torch.isfinite(host).all()):
                bad = True
                self.overflow_batches.append(idx)
            self._free_hosts.append(host)
        if len(self._pending) > 8:                         # the GPU is far behind: do not let the queue grow without bound
            return self._poll(wait=True) or bad
        return bad

    def check(self):
        """Wait for every batch queued so far and raise FloatingPointError if the fp16 static stage overflowed on any of
        them (their maps are NaN); the engine has switched to bf16 by then, so re-running those batches gives finite maps."""
        if self._poll(wait=True) and self._guard_on():
            self._fallback_to_bf16()
        if self.overflow_batches:
            raise FloatingPointError("fp16 static stage overflowed (non-finite CAM scores) on batch(es) %s; the engine now runs "
                                     "the static stage in bf16 - re-run those batches" % self.overflow_batches)

    def _forward(self, frames):
        B, T = frames.shape[:2]
        if (B, T) != (self.B, self.T):
            raise ValueError("engine built for %dx%d clips x frames" % (self.B, self.T))
        flat = frames.reshape((B * T,) + tuple(frames.shape[2:]))
        capturing = torch.cuda.is_current_stream_capturing()
        if self._guard_on() and not capturing and self._poll():
            import warnings
            warnings.warn("fp16 static stage overflowed on batch(es) %s (their maps are NaN): switching the static stage to bf16"
                          % self.overflow_batches)
            self._fallback_to_bf16()
        self.static_stage(flat)
        sal = self.temporal_stage()
        if self._guard_on() and not capturing:             # see __init__
            if not self._first_checked:
                self._first_checked = True
                if self.nonfinite():                       # the first batch is waited for and repaired in place
                    self._fallback_to_bf16()
                    self.static_stage(flat)
                    sal = self.temporal_stage()
            else:
                self._queue_flag()
        self._batch += 1
        return sal

    def __call__(self, frames):
        """frames [B, T, H, W, 3] (u8 or f32, device) -> saliency f32 [B, 2w, 4w] ([B, T, 2w, 4w] with
        ``return_all_steps``)."""
        with torch.no_grad():
            if getattr(self, '_graph', None) is not None:
                if frames.shape != self._graph_in.shape or frames.dtype != self._graph_in.dtype:
                    raise ValueError("graph captured for %s %s frames" % (tuple(self._graph_in.shape), self._graph_in.dtype))
                if self._guard_on() and self._poll():      # an earlier replay overflowed: back to eager launches, in bf16
                    import warnings
                    warnings.warn("fp16 static stage overflowed on batch(es) %s (their maps are NaN): switching the static "
                                  "stage to bf16 (eager launches; capture again for graph replay)" % self.overflow_batches)
                    self._fallback_to_bf16()
                    return self._forward(frames)
                self._graph_in.copy_(frames)
                self._graph.replay()
                if self._guard_on():
                    self._queue_flag()
                self._batch += 1
                return self._graph_out
            return self._forward(frames)


def _to_tensors(sd):
    return {k: (v if torch.is_tensor(v) else torch.from_numpy(v)) for k, v in sd.items()}
