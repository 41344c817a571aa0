"""Benchmark of the hot path on N MI355X GPUs of one node (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json config C4's per-GPU shard = 4x config C3): each rank owns `--clips`
clips of `--frames` synthetic HxW u8 equirectangular frames, resident in HBM before the timed region.
One step = the whole path over that batch: equi->cube (K1), CubePad + ResNet-50-cubic (K2/K3), CAM
(K4), window normalise (K7), T ConvLSTM steps (K5), cube->equi + channel max (K6), and for N > 1 one
RCCL all-gather of the saliency maps.  Metric: frames/s = N * clips * frames * K / max-over-ranks
wall time.  The K steps are issued as a stream of batches through the engine's two-stage software pipeline
(SaliencyEngine.stream: the static stage of batch k+1 on a second HIP stream beside the ConvLSTM of batch k; the
pipeline starts empty and is drained INSIDE the timed region; every batch's maps are the bits the one-by-one form
gives); the same K steps issued one by one are timed right after and reported as `sequential` (--sequential makes
them the headline).  The timed region carries NO instrumentation; the per-kernel roofline of the dominant
kernel (the ConvLSTM convolution: its Winograd-domain GEMM) is measured with HIP events in a separate short pass
after it.  Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`,
`cpu_baseline` (the oracle on the host cores, bounded sample, N = 1 only) and `secondary`: the other
precisions / BASELINE configs timed by the same invocation (N = 1 only; --no-secondary skips them).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from cp_360_weakly_supervised_saliency_amd import dist as cpdist          # noqa: E402
from cp_360_weakly_supervised_saliency_amd import ops                      # noqa: E402
from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine  # noqa: E402
from cp_360_weakly_supervised_saliency_amd.utils import synth              # noqa: E402

PEAK = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3}     # dense MFMA TFLOP/s, MI355X_MICROARCH.md
DTYPE = {'bf16': 'bf16', 'fp16': 'f16', 'fp32': 'f32'}


class LaunchTimer:
    """HIP-event timer around tagged kernel launches on torch's current stream (the stream libcp360
    launches on).  Only used in the roofline pass, never inside the timed region."""

    TAGS = ('clstm.Conv2', 'clstm.Gates')     # the dominant kernel: K = 9 * 4000 ConvLSTM convolutions

    def __init__(self):
        self.active = False
        self.records = []          # (tag, flops, start_event, end_event)

    def wrap(self, tag, flops, fn):
        if not self.active or tag not in self.TAGS:
            return fn()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn()
        b.record()
        self.records.append((tag, flops, a, b))
        return rc

    def summary(self):
        n, ms, flops = 0, 0.0, 0.0
        for tag, fl, a, b in self.records:
            n += 1
            ms += a.elapsed_time(b)
            flops = fl
        return n, ms, flops


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _safe(fn):
    """A diagnostic value, or None when collecting it fails (never an exception after the timed region)."""
    try:
        return fn()
    except Exception:                               # noqa: BLE001
        return None


def sync(dev):
    if torch.device(dev).type == 'cuda':
        torch.cuda.synchronize()


def timed(step, warmup, steps, dev, per_rank=False):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.
    per_rank: also this rank's own time up to its synchronize, BEFORE the closing barrier (diagnosis of N > 1 runs)."""
    out = None
    for _ in range(warmup):
        out = step()
    sync(dev)
    cpdist.barrier()
    sync(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync(dev)
    mine = time.perf_counter() - t0
    cpdist.barrier()
    sync(dev)
    total = cpdist.max_over_ranks(time.perf_counter() - t0, dev)
    return ((total, mine) if per_rank else total), out


def timed_pipelined(eng, frames, finish, warmup, steps, dev):
    """The same contract for the engine's software-pipelined form (SaliencyEngine.stream: the static stage of batch k+1 on a
    second HIP stream beside the ConvLSTM of batch k): W untimed batches (their own fill and drain), then exactly K batches
    between barrier + synchronize on both sides - the pipeline starts EMPTY inside the timed region and is drained inside
    it, so every one of the K batches does all of its work there.  ``finish(maps)`` = the all-gather of a batch's maps."""
    out = None
    for sal in eng.stream(frames for _ in range(warmup)):
        out = finish(sal)
    sync(dev)
    cpdist.barrier()
    sync(dev)
    t0 = time.perf_counter()
    n = 0
    for sal in eng.stream(frames for _ in range(steps)):
        out = finish(sal)
        n += 1
    sync(dev)
    mine = time.perf_counter() - t0
    cpdist.barrier()
    sync(dev)
    total = cpdist.max_over_ranks(time.perf_counter() - t0, dev)
    assert n == steps
    return (total, mine), out


class StubEngine:
    """--stub-engine: stands in for SaliencyEngine on a box WITHOUT a GPU so that this file's own main() - launcher
    contract, sharding, the all-gather, max-over-ranks timing, the JSON line - runs under
    ``torch.distributed.run --nproc-per-node N`` on gloo (tests/test_distributed_cpu.py).  It computes no saliency: a
    clip's "map" is a fixed function of its frames, so the gathered result can be checked clip for clip.  Its numbers
    mean nothing and the JSON line says so (``"stub_engine": true``)."""

    def __init__(self, cube_dim, clips, frames):
        self.B, self.T, self.w = int(clips), int(frames), int(cube_dim) // 32
        self.static_precision, self.fp16_fallback = 'fp32', False

    def __call__(self, frames):
        B = frames.shape[0]
        m = frames.reshape(B, -1).float().mean(dim=1)
        ramp = torch.arange(2 * self.w * 4 * self.w, dtype=torch.float32).reshape(1, 2 * self.w, 4 * self.w)
        return m.reshape(B, 1, 1) + ramp / ramp.numel()

    def stream(self, batches):
        for frames in batches:
            yield self(frames)

    def close(self):
        pass


def roofline_pass(eng, frames, precision, steps=2):
    """Dominant kernel = the K = 36000 ConvLSTM convolutions (Conv2 / Gates have the same shape):
    algorithmic flops of one launch / its mean duration over `steps` extra steps, every launch
    bracketed by HIP events on the launch stream."""
    # the timed region runs each stage through ONE C call (cp360_resnet_forward / cp360_clstm_step); to bracket single
    # launches with events this pass plans the SAME launch sequence from Python (CP360_CTX=0 path: same kernels, same
    # descriptors, bit-identical results - tests/test_ctx.py)
    from cp_360_weakly_supervised_saliency_amd import stage_ctx
    timer = LaunchTimer()
    ops.LAUNCH_TIMER = timer
    old_ctx, stage_ctx.USE_CTX = stage_ctx.USE_CTX, False
    try:
        eng(frames)                                    # packs the Python-side plan (untimed)
        torch.cuda.synchronize()
        timer.active = True
        for _ in range(steps):
            eng(frames)
        torch.cuda.synchronize()
    finally:
        timer.active = False
        ops.LAUNCH_TIMER = None
        stage_ctx.USE_CTX = old_ctx
    n, ms, flops = timer.summary()
    if not n:
        return None
    ms /= n
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = None, None
    # PMC counters cannot be read inside the run: the per-launch HBM bytes come from the tracked rocprofv3 summary of
    # the SAME launch shape (precision, clips per GPU, face size), when one has been collected
    tp = os.path.join(REPO, 'profiles', 'traffic_%s_b%d_w%d.json' % (precision, eng.B, eng.w))
    if not os.path.exists(tp) and (precision, eng.B, eng.w) == ('bf16', 4, 7):
        tp = os.path.join(REPO, 'profiles', 'traffic_bf16.json')
    wino = precision != 'fp32' and eng.cell.uses_winograd(6 * eng.B, eng.w)
    if os.path.exists(tp):
        t = json.load(open(tp))
        # (a summary collected for the OTHER kernel of this shape - the direct clip-resident one before round 5 - is not quoted)
        if t.get('kernel', 'conv_clip_kernel').startswith('wino_gemm' if wino else 'conv_clip'):
            traffic, src = t.get('conv_igemm_clstm_bytes_per_launch'), 'profiles/' + t.get('source', '')
    M = 6 * eng.B * eng.w * eng.w
    H4 = 4 * eng.cell.hidden_size
    res = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK[precision], 'unit': 'TFLOP/s', 'frac': round(ach / PEAK[precision], 4),
           'traffic': traffic, 'traffic_source': src and (src + ' (rocprofv3 PMC passes, (2*FETCH_SIZE+WRITE_SIZE)*1024 per launch; not re-measured in this run)'),
           'avg_launch_ms': round(ms, 4), 'launches_timed': n, 'flops_per_launch': flops}
    if wino:
        # the launch computes the SAME convolution in the Winograd domain.  `achieved` / `frac` = the flops the matrix pipe
        # EXECUTES (16 GEMMs over the padded tile count) / launch time (/ peak): a hardware fraction, never above 1.  The
        # convolution's algorithmic flops (2 M N 9 C, the direct form) over the same time are reported apart as
        # `achieved_effective` / `effective_frac` (they can exceed the peak: the transform removes multiplies)
        th = (eng.w + 1) // 2
        tiles = -(-6 * eng.B * th * th // 384) * 384
        mf = 2.0 * 16 * tiles * H4 * H4
        res['kernel'] = ('wino_gemm_kernel (ConvLSTM Conv2/Gates as Winograd F(2x2,3x3): 16 GEMMs of %d tiles x N=%d x K=%d; '
                         'direct form M=%d N=%d K=%d)' % (tiles, H4, H4, M, H4, 9 * H4))
        res['achieved_effective'] = res['achieved']
        res['effective_frac'] = res['frac']
        res['algorithmic_flops_per_launch'] = flops
        res['flops_per_launch'] = mf
        res['achieved'] = round(mf / (ms * 1e-3) / 1e12, 2)
        res['frac'] = round(res['achieved'] / PEAK[precision], 4)
        res['note'] = ('achieved / frac = the flops the matrix pipe executes (%.2f of the direct form) / launch time (/ peak); '
                       'achieved_effective / effective_frac = the convolution\'s algorithmic (direct-form) flops / the same time; '
                       'the input / output transform kernels are separate launches' % (mf / flops))
    else:
        res['kernel'] = ('conv_clip_kernel<%s> (ConvLSTM Conv2/Gates, M=%d N=%d K=%d)' % ('face tile' if eng.w > 7 else 'clip tile', M, H4, 9 * H4))
    return res


def pmc_clock(precision, B, w):
    tp = os.path.join(REPO, 'profiles', 'pmc_clock_%s_b%d_w%s.json' % (precision, B, w))
    if not os.path.exists(tp):
        return None
    t = json.load(open(tp))
    return {'kernel': t.get('kernel'), 'clock_ghz': t.get('clock_ghz'), 'mfma_pipe_busy': t.get('mfma_pipe_busy'),
            'source': 'profiles/' + t.get('source', '')}


STATIC_GFLOP = {224: 49.05 + 1.20, 256: 64.06 + 1.57, 512: 256.24 + 6.29}     # ResNet-50-cubic + CAM, per frame


def stage_split(eng, frames, reps=3):
    """Where one step's time goes (never inside the timed region): the static stage (K1 + ResNet-50-cubic + CAM), the
    ConvLSTM window (min / max, T cell updates) and the cube -> equi + channel max, each bracketed by one HIP event pair on
    the launch stream; median of ``reps`` extra steps.  With it two driver runs on different boxes can be compared stage
    by stage."""
    B, T = frames.shape[:2]
    flat = frames.reshape((B * T,) + tuple(frames.shape[2:]))
    rows = []
    with torch.no_grad():
        for _ in range(reps):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            marks = []
            ev[0].record()
            eng.static_stage(flat)
            ev[1].record()
            eng.runner.stage_events = marks
            try:
                eng.temporal_stage()
            finally:
                eng.runner.stage_events = None
            torch.cuda.synchronize()
            rows.append((ev[0].elapsed_time(ev[1]), marks[0].elapsed_time(marks[1]), marks[1].elapsed_time(marks[2])))
    med = lambda k: round(float(np.median([r[k] for r in rows])), 3)
    return {'static': med(0), 'convlstm': med(1), 'c2e': med(2)}


def static_roofline(ms_per_step, B, T, cd, precision):
    """BASELINE config C2 (static path only): algorithmic flops of the stage / the measured time per pass
    against the MFMA peak of the arithmetic type; the per-launch picture (slowest launch, workgroups per launch) is the
    tracked timeline of the same command, when one has been collected."""
    gf = STATIC_GFLOP.get(cd)
    if gf is None:
        return None
    ach = gf * B * T / (ms_per_step * 1e-3) / 1e3
    r = {'bound': 'mfma', 'kernel': 'static stage: K1 + %d conv launches of ResNet-50-cubic + CAM, %d faces per pass' % (54, 6 * B * T),
         'achieved': round(ach, 2), 'peak': PEAK[precision], 'unit': 'TFLOP/s', 'frac': round(ach / PEAK[precision], 4),
         'flops_per_pass': gf * B * T * 1e9, 'traffic': None}
    tp = os.path.join(REPO, 'profiles', 'c2_static_%s_f%d.json' % (precision, B * T))
    if os.path.exists(tp):
        t = json.load(open(tp))
        r['timeline'] = 'profiles/' + t.get('source', '')
        r['slowest_launch'] = t.get('slowest_launch')
        r['launches'] = t.get('launches')
    return r


def run_workload(dev, rank, world, H, W, cd, B, T, precision, steps, warmup, graph=False, static_only=False,
                 frame_chunk=None, source_hw=None, want_roofline=False, static_precision=None, all_steps=False,
                 f32_input=False, stub=False, want_split=False, want_plan=False, clock_after=False, pipelined=False):
    if stub:
        eng = StubEngine(cd, B, T)
    else:
        rs = synth.resnet50_state(seed=1)
        cs = synth.clstm_state(seed=2)
        eng = SaliencyEngine(rs, cs, (H, W), cd, clips=B, frames=T, precision=precision, device=dev,
                             frame_chunk=frame_chunk, source_hw=source_hw, static_precision=static_precision,
                             return_all_steps=all_steps)
        del rs, cs
    fh, fw = source_hw if source_hw else (H, W)
    # this rank's clips (global clip id = rank*B + b), resident in HBM before timing
    frames = torch.stack([torch.from_numpy(synth.clip_u8(3 + rank * B + b, T, fh, fw)) for b in range(B)]).to(dev)
    if f32_input:
        # SURVEY 8(d), "uint8 and fp32 variants": the reference feeds np.array(img) / 255.0 (dataset_feat_extractor.py:142);
        # here as f32 [H, W, 3] in [0, 1] resident in HBM (4x the input bytes of the u8 variant, K1 scale 1.0)
        frames = frames.to(torch.float32) / 255.0
    n_clips = world * B
    if graph and not static_only:
        eng.capture(frames)
    static_graph = None
    if graph and static_only:
        # one frame is ~60 launches of a few microseconds each: replay them from a HIP graph
        flat = frames.reshape((B * T,) + tuple(frames.shape[2:]))
        with torch.no_grad():
            eng.static_stage(flat)
            torch.cuda.synchronize()
            static_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(static_graph):
                eng.static_stage(flat)

    def step():
        if static_only:
            if static_graph is not None:
                static_graph.replay()
                return eng.cam.view(B, -1)[:, :8]
            with torch.no_grad():
                cam = eng.static_stage(frames.reshape((B * T,) + tuple(frames.shape[2:])))
            return cam.view(B, -1)[:, :8].float()
        sal = eng(frames)
        return cpdist.gather_maps(sal, n_clips, rank, world, reuse=True)

    pipelined = bool(pipelined and not graph and not static_only)
    if pipelined:
        elapsed, out = timed_pipelined(eng, frames, lambda sal: cpdist.gather_maps(sal, n_clips, rank, world, reuse=True), warmup, steps, dev)
    else:
        elapsed, out = timed(step, warmup, steps, dev, per_rank=True)
    elapsed, mine = elapsed
    assert (static_only or out.shape[0] == n_clips) and bool(torch.isfinite(out).all())
    res = {'value': round(world * B * T * steps / elapsed, 3), 'ms_per_step': round(1000.0 * elapsed / steps, 3),
           'roofline': None, 'pipelined': pipelined}
    if pipelined:
        # the same K steps one batch at a time (engine(frames): nothing of batch k+1 starts before batch k's maps are out) -
        # the form every round before this one quoted as the headline; reported beside it, never as `value`
        out_p = out.clone()                                # (gather_maps reuses its result buffer)
        el2, out2 = timed(step, 1, steps, dev, per_rank=True)
        res['sequential'] = {'value': round(world * B * T * steps / el2[0], 3), 'ms_per_step': round(1000.0 * el2[0] / steps, 3),
                             'what': 'engine(frames) batch by batch on one stream, same K steps, measured right after the timed region',
                             'same_maps': bool(torch.equal(out_p, out2))}
    if clock_after and not stub:
        # the register-only MFMA clock probe, launched right behind the timed region's last step (the chip still warm)
        res['held_clock_ghz_after'] = _safe(lambda: ops.held_clock_ghz(dev))
        res['timed_region_s'] = round(elapsed, 3)
    # diagnosis of the N > 1 runs (never part of `value`): every rank's own ms per step and the all-gather alone
    res['ms_per_step_per_rank'] = [round(1000.0 * v / steps, 3) for v in cpdist.all_ranks(mine, dev)]
    if not static_only:
        sal = eng(frames)
        sync(dev)
        cpdist.barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            cpdist.gather_maps(sal, n_clips, rank, world, reuse=True)
        sync(dev)
        res['allgather_ms'] = round(cpdist.max_over_ranks(time.perf_counter() - t0, dev) * 100.0, 4)
        res['allgather_bytes_per_rank'] = int(sal.numel() * sal.element_size())
        res['map_shape'] = list(out.shape)
    # diagnostics after the timed region: a failure here is reported in the line, it never costs the measured value
    if want_split and not static_only and not stub:
        if graph:
            eng._graph = None
        try:
            res['stage_ms'] = stage_split(eng, frames)
        except Exception as e:                      # noqa: BLE001
            res['stage_ms'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    if want_roofline and not static_only and not stub:
        if graph:
            eng._graph = None                      # the roofline pass needs eager launches to bracket
        try:
            res['roofline'] = roofline_pass(eng, frames, precision)
        except Exception as e:                      # noqa: BLE001
            res['roofline'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    if static_only and not stub:
        res['roofline'] = static_roofline(res['ms_per_step'], B, T, cd, eng.static_precision)
    if want_plan and not stub:
        # which path the static stage takes at this cube size (cp360_resnet_plan_describe): fused kernels exist for cube 224 / 512
        # in the 16-bit types, anything else runs convolution by convolution
        try:
            text = eng.resnet.__dict__['_stage'].describe(6 * B * T, cd)
            lines = [ln for ln in text.splitlines() if ln and not ln.startswith('  ')]
            res['static_plan'] = {'generic_layers': sum('GENERIC' in ln for ln in lines), 'fused_layers': sum('GENERIC' not in ln for ln in lines),
                                  'lines': lines[:12]}
            res['convlstm_winograd'] = bool(precision != 'fp32' and eng.cell.uses_winograd(6 * B, eng.w))
        except Exception as e:                      # noqa: BLE001
            res['static_plan'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    res['w'] = eng.w
    res['static_dtype'] = DTYPE[eng.static_precision]
    res['fp16_fallback'] = bool(eng.fp16_fallback)      # fp16 static stage overflowed on the first batch -> bf16 (pipeline.py)
    eng.close()                                          # the contexts' packed weights are raw hipMalloc: release them now
    del eng, frames
    torch.cuda.empty_cache()
    return res


def workload_name(static_only, B, T, H, W, cd, w, src_hw=None, all_steps=False, f32_input=False):
    inp = 'f32 [0,1]' if f32_input else 'u8'
    if static_only:
        return 'C2 static path only: %d x %d frames %dx%d %s equi -> 6x%d^2 cube -> CubePad ResNet-50 -> CAM' % (B, T, H, W, inp, cd)
    s = ('%d clips x %d frames %dx%d %s equi -> 6x%d^2 cube -> CubePad ResNet-50 -> CAM -> ConvLSTM x%d -> '
         'cube_to_equi saliency %dx%d' % (B, T, H, W, inp, cd, T, 2 * w, 4 * w))
    if all_steps:
        s += ' after EVERY step (return_all_steps: [clips, %d, %d, %d] gathered)' % (T, 2 * w, 4 * w)
    if src_hw:
        s += ' (frames decoded at %dx%d, PIL-exact Lanczos resize included)' % src_hw
    return s


def level1_bench(dev, precision='fp32', reps=5):
    """The drop-in (module-boundary) path timed as the reference's drivers use it - numpy in, numpy out, one call per
    frame / per ConvLSTM step, H2D + D2H copies and the per-call layout conversions included (never the headline):
      static   dataset_feat_extractor.py:145-162: to_cube(img) dict -> im_norm -> concatenate -> CAM(...) per frame
      temporal test_temporal.py:63-85: window min/max, 5 x ``hidden, cell = model(frame, [hidden, cell])``,
               ``c2e.to_equi_nn`` + ``torch.max`` -> one map per window (seq_len 5, config.yaml:34)."""
    from cp_360_weakly_supervised_saliency_amd.model.clstm import ConvLSTMCell
    from cp_360_weakly_supervised_saliency_amd.model.resnet_cubic import resnet50
    from cp_360_weakly_supervised_saliency_amd.static_model.class_activation_model import CAM
    from cp_360_weakly_supervised_saliency_amd.utils.cube_to_equi import Cube2Equi
    from cp_360_weakly_supervised_saliency_amd.utils.equi_to_cube import Equi2Cube
    from cp_360_weakly_supervised_saliency_amd.utils.utils import im_norm
    H, W, cd, T = 1024, 2048, 224, 5
    rs = synth.resnet50_state(seed=1)
    cs = synth.clstm_state(seed=2)
    model = resnet50(precision=precision)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rs.items()}, strict=False)
    model = model.to(dev).eval()
    cell = ConvLSTMCell(1000, 1000, precision=precision)
    cell.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in cs.items()})
    cell = cell.to(dev).eval()
    del rs, cs
    frame = synth.frame_u8(31, H, W)
    input_img = np.array(frame) / 255.0
    e2c = Equi2Cube(cd, input_img, device=dev)

    def static_frame():
        cubes = e2c.to_cube(input_img)
        batch = np.concatenate([np.expand_dims(im_norm(cubes[i], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]), 0)
                                for i in range(6)], axis=0).astype(np.float32)
        return CAM(batch, None, model, 'layer4', 'fc.weight', use_gpu=True)[0]

    c2e = Cube2Equi(7, device=dev)
    subseq = [f for f in synth.cam_clip(6005, T)]

    def temporal_window():
        mx, mn = np.max(subseq), np.min(subseq)
        init = (subseq[0] - mn) / (mx - mn)
        cst = torch.FloatTensor(init).to(dev)
        hidden = torch.FloatTensor(init).to(dev)
        for f in subseq:
            f = torch.FloatTensor((f - mn) / (mx - mn)).to(dev)
            hidden, cst = cell(f, [hidden, cst])
        return torch.squeeze(torch.max(c2e.to_equi_nn(hidden), 1)[0]).cpu().numpy()

    def rate(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    ts, tw = rate(static_frame), rate(temporal_window)
    # where the temporal window's time is (tools/level1_breakdown.py --temporal): the reference's host loop ~1 ms, five f32 cell
    # updates of ONE cube ~1.7 ms each = 0.79 of the f32 MFMA peak (f32 matrix rate = 1/16 of the 16-bit one) - device work, not
    # shim overhead.  The same loop with the cell in bf16 (set_precision: the one-cube launches stay on the direct kernels):
    cell.set_precision('bf16')
    tw16 = rate(temporal_window)
    cell.set_precision(precision)
    del model, cell
    torch.cuda.empty_cache()
    return {'name': 'Level-1 drop-in path (reference driver loops on the shims, numpy in / numpy out, %s)' % precision,
            'static_frames_per_s': round(1.0 / ts, 2), 'static_ms_per_frame': round(1000 * ts, 2),
            'temporal_maps_per_s': round(1.0 / tw, 2), 'temporal_ms_per_window': round(1000 * tw, 2),
            'temporal_ms_per_window_bf16_cell': round(1000 * tw16, 2),
            'temporal_breakdown': 'fp32: ~1 ms host loop of the reference (numpy min / max / normalise, 7 pageable H2D copies) + 5 cell updates of '
                                  '~1.7 ms each on the device (211.7 GFLOP at 0.79 of the 157 TFLOP/s f32 MFMA peak) + 0.05 ms output; '
                                  'tools/level1_breakdown.py --temporal',
            'window': 'seq_len 5 (config.yaml:34): 5 ConvLSTMCell calls + to_equi_nn + max',
            'what': 'to_cube dict -> im_norm -> CAM() per 1024x2048 frame; 5 x model(frame, [hidden, cell]) + '
                    'to_equi_nn + torch.max per window; PCIe copies and NCHW<->NHWC conversions of every call included'}


def sliding_bench(dev, precision='bf16', n_frames=64, seq_len=5, reps=5):
    """The reference's temporal workload as it runs it (temporal_model/test_temporal.py:57-62, config.yaml:34): a stride-1
    sliding window of seq_len 5 over ONE video's cube_feat sequence - every window is its own min / max normalisation,
    hidden = cell = first frame and 5 ConvLSTM steps, so a video of n frames costs 5 (n - 5) cell updates.  Here all windows of a
    64-frame video run in lock step (ClipRunner.run(sliding=True): 60 windows over 64 frames, zero-copy over one feature
    sequence, GEMM M = 60 * 294) on CAM features resident in HBM; the static stage is not part of this line."""
    from cp_360_weakly_supervised_saliency_amd.model.clstm import ConvLSTMCell
    from cp_360_weakly_supervised_saliency_amd.temporal_model.test_temporal import ClipRunner
    from cp_360_weakly_supervised_saliency_amd.utils.cube_to_equi import Cube2Equi
    cs = synth.clstm_state(seed=2)
    cell = ConvLSTMCell(1000, 1000, precision=precision)
    cell.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in cs.items()})
    cell = cell.to(dev).eval()
    n_win = n_frames - seq_len + 1
    feats = synth.cam_clip(7001, n_frames)                                   # [n, 6, 1000, 7, 7]
    cam = torch.from_numpy(np.ascontiguousarray(feats.transpose(0, 1, 3, 4, 2)).reshape(n_frames, 294, 1000)).to(dev)
    runner = ClipRunner(cell, Cube2Equi(7, device=dev), n_win, seq_len)
    with torch.no_grad():
        runner.run(cam, sliding=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            maps = runner.run(cam, sliding=True)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert tuple(maps.shape) == (n_win, 14, 28) and bool(torch.isfinite(maps).all())
    st = cell.__dict__.get('_stage')
    if st is not None:
        st.close()
    del cell, runner
    torch.cuda.empty_cache()
    return {'name': 'reference temporal workload: stride-1 sliding window, seq_len 5, one 64-frame video, %s' % precision,
            'maps_per_s': round(n_win / dt, 1), 'ms_per_video': round(1000 * dt, 3), 'windows': n_win, 'cell_updates': n_win * seq_len,
            'cell_updates_per_s': round(n_win * seq_len / dt, 1), 'dtype': DTYPE[precision],
            'what': 'ClipRunner.run(sliding=True): %d windows x %d steps in lock step over one resident cube_feat sequence '
                    '(test_temporal.py:57-62); temporal stage only' % (n_win, seq_len)}


def cpu_baseline(precision, dev, static_precision=None, clips=4):
    """the oracle (numpy / torch-CPU restatement of the reference, oracle/) on the host
    cores, timed on config C1 (one 960x1920 frame, static stage) and on ONE 16-frame 1024x2048 clip end to
    end - the bounded sample `value` is quoted on.  The same clip then goes through the HIP path at the
    bench precision and the oracle acts as the checker for the second half of the metric: AUC-Judd / CC of
    both maps against fixations sampled from the oracle map, and CC(build, oracle): reported under "check"."""
    from tests.parity_helpers import oracle_pipeline, oracle_cam_frames
    from oracle import o_metrics
    # 32 threads: fastest setting measured on the GPU box's host (tests/probe_cpu_threads.py); using all
    # 256 hardware threads of the EPYC host makes oneDNN ~100x slower on these small convolutions
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    rs = synth.resnet50_state(seed=1)
    cs = synth.clstm_state(seed=2)
    t0 = time.time()
    oracle_cam_frames(synth.frame_u8(31, 960, 1920)[None], rs, 224)
    c1 = time.time() - t0
    H, W, cd, T = 1024, 2048, 224, 16
    clip = synth.clip_u8(3, T, H, W)
    t0 = time.time()
    ref = oracle_pipeline(clip, rs, cs, cd)
    dt = time.time() - t0
    # the check runs the TIMED path: the headline's launch shape (`clips` clips per GPU -> for 4 clips of 7x7 faces the
    # ConvLSTM convolutions run in the Winograd domain, which one clip alone would not), the oracle's sample clip as clip 0
    # beside the bench's other clips; clip 0's map is compared with the oracle
    batch = np.stack([clip] + [synth.clip_u8(3 + b, T, H, W) for b in range(1, clips)])
    eng = SaliencyEngine(rs, cs, (H, W), cd, clips=clips, frames=T, precision=precision, device=dev,
                         static_precision=static_precision)
    wino = bool(precision != 'fp32' and eng.cell.uses_winograd(6 * clips, eng.w))
    got = eng(torch.from_numpy(batch).to(dev)).float().cpu().numpy()[0]
    eng.close()
    del eng
    fix = synth.fixations_from_map(ref, 200, H // 2, W // 2)
    rng = lambda: np.random.RandomState(0)
    check = {'max_abs_diff': float(np.max(np.abs(got - ref))),
             'auc_judd': [round(o_metrics.auc_judd(ref, fix, rng=rng()), 6), round(o_metrics.auc_judd(got, fix, rng=rng()), 6)],
             'cc': [round(o_metrics.corr_coeff(ref, fix), 6), round(o_metrics.corr_coeff(got, fix), 6)],
             'cc_build_vs_oracle': round(o_metrics.corr_coeff(got, ref), 6),
             'clips_in_batch': clips, 'convlstm_winograd': wino,
             'what': 'oracle vs HIP (%s) saliency of the T=16 sample clip, run as clip 0 of a %d-clip batch = the launch shape '
                     'of the timed region (ConvLSTM convolutions: %s); [oracle, hip] metrics vs fixations sampled from the '
                     'oracle map' % (precision, clips, 'Winograd-domain GEMM' if wino else 'direct kernel')}
    return {'value': round(T / dt, 4), 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'cpu_model': cpu_model(), 'host_threads': os.cpu_count(),
            'sample': 'config C3 shape: 1 clip x 16 frames %dx%d -> 6x%d^2 end to end, oracle fp32 (torch-CPU conv, '
                      'numpy remap), %.1f s' % (H, W, cd, dt),
            'c1_static_frame_s': round(c1, 3),
            'c1_sample': 'config C1: one 960x1920 frame -> 6x224^2 -> CubePad ResNet-50 -> CAM (static stage), oracle fp32',
            'check': check}


def cpu_anchor(reps=5):
    """BASELINE.md section 4, "sanity anchor": the oracle on all host cores, on the two pieces BASELINE.md section 2
    timed with the REFERENCE code in the 8-vCPU survey container - CAM() of one cube (6x3x224x224: 0.43 s, min 0.35)
    and one ConvLSTMCell step (6x1000x7x7: 0.62 s, min 0.54).  Part of the cpu_baseline leg (the only place outside
    tests/ that runs the oracle); needs no GPU."""
    from oracle import o_resnet, o_clstm
    torch.set_num_threads(os.cpu_count() or 1)
    rs = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.resnet50_state(seed=1).items()}
    cs = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.clstm_state(seed=2).items()}
    from cp_360_weakly_supervised_saliency_amd.utils import hashrng
    cube = hashrng.uniform(41, (6, 3, 224, 224), -1.0, 1.0).astype(np.float32)
    x = torch.from_numpy(hashrng.uniform(42, (6, 1000, 7, 7), 0.0, 1.0).astype(np.float32))
    h = torch.zeros_like(x)
    c = torch.zeros_like(x)

    def med(fn):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return round(float(np.median(ts)), 3), round(float(np.min(ts)), 3)
    with torch.no_grad():
        cam = med(lambda: o_resnet.cam_from_cubes(cube, rs))
        step = med(lambda: o_clstm.clstm_step(x, h, c, cs))
    return {'cpu_anchor': {'cam_one_cube_s': {'median': cam[0], 'min': cam[1], 'reference_in_survey': [0.43, 0.35]},
                           'clstm_step_s': {'median': step[0], 'min': step[1], 'reference_in_survey': [0.62, 0.54]},
                           'threads': torch.get_num_threads(), 'cpu_model': cpu_model(), 'kind': 'port (oracle/)',
                           'what': 'BASELINE.md section 2 pieces: CAM() of 6x3x224x224, ConvLSTMCell step 6x1000x7x7, fp32'}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--precision', default=os.environ.get('CP360_PRECISION', 'bf16'), choices=['fp32', 'bf16', 'fp16'])
    ap.add_argument('--static-precision', default='', choices=['', 'fp32', 'bf16', 'fp16'],
                    help='arithmetic type of the static stage (ResNet-50 + CAM); default: fp16 under --precision bf16 '
                         '(same MFMA rate, 3 more mantissa bits: DESIGN.md section 4), else --precision')
    ap.add_argument('--clips', type=int, default=4, help='clips per GPU')
    ap.add_argument('--frames', type=int, default=16, help='frames per clip')
    ap.add_argument('--equi', default='1024x2048')
    ap.add_argument('--cube', type=int, default=224)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary precisions / configs (N = 1)')
    ap.add_argument('--frame-chunk', type=int, default=0, help='frames per static-stage group (0 = all)')
    ap.add_argument('--graph', action='store_true', help='replay the step from a HIP graph (launch-bound small configs)')
    ap.add_argument('--sequential', action='store_true',
                    help='time engine(frames) batch by batch on one stream instead of the software-pipelined stream of batches '
                         '(SaliencyEngine.stream); without it the sequential figure is reported beside the headline')
    ap.add_argument('--source', default='', help='decoded frame size HxW: include the PIL-exact Lanczos resize (K0) in the step')
    ap.add_argument('--all-steps', action='store_true',
                    help='return_all_steps: one map per ConvLSTM step, [clips, T, 2w, 4w] gathered instead of [clips, 2w, 4w]')
    ap.add_argument('--f32-input', action='store_true', help='frames resident as f32 [H, W, 3] in [0, 1] instead of u8 (SURVEY 8(d))')
    ap.add_argument('--static-only', action='store_true',
                    help='BASELINE config C2: the static path only (equi -> cube -> ResNet-50 -> CAM); roofline = stage flops / time')
    ap.add_argument('--stub-engine', action='store_true',
                    help='no GPU: run main() itself (launcher contract, sharding, all-gather, timing, JSON line) on CPU tensors '
                         'over gloo with a stand-in engine; the numbers mean nothing (tests/test_distributed_cpu.py)')
    ap.add_argument('--cpu-anchor', action='store_true',
                    help='no GPU: time the oracle on the pieces BASELINE.md section 2 timed with the reference itself '
                         '(CAM of one cube, one ConvLSTM step) - the sanity anchor of the cpu_baseline leg')
    args = ap.parse_args()
    if args.cpu_anchor:
        print(json.dumps(cpu_anchor()))
        return

    rank, world, local = cpdist.init_from_env(backend='gloo' if args.stub_engine else None)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if args.stub_engine:
        dev = torch.device('cpu')
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    H, W = (int(v) for v in args.equi.split('x'))
    B, T = args.clips, args.frames
    src_hw = tuple(int(v) for v in args.source.split('x')) if args.source else None

    head = run_workload(dev, rank, world, H, W, args.cube, B, T, args.precision, args.steps, args.warmup,
                        graph=args.graph, static_only=args.static_only, frame_chunk=args.frame_chunk or None,
                        source_hw=src_hw, want_roofline=(rank == 0), static_precision=args.static_precision or None,
                        all_steps=args.all_steps, f32_input=args.f32_input, stub=args.stub_engine, want_split=(rank == 0),
                        pipelined=not args.sequential)

    if rank == 0:
        line = {
            'metric': 'frames/sec end-to-end %dx%d equi->saliency' % (H, W),
            'value': head['value'], 'unit': 'frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': head['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            # the arithmetic type(s) the path computes in: temporal stage (ConvLSTM, 81 % of the flops) + static stage
            # (ResNet-50 / CAM) when they differ - the default 16-bit engine is bf16 + f16
            'dtype': DTYPE[args.precision] if head['static_dtype'] == DTYPE[args.precision]
                     else '%s+%s' % (DTYPE[args.precision], head['static_dtype']),
            'mixed_precision': head['static_dtype'] != DTYPE[args.precision],
            'data': 'synthetic',
            'config': {'workload': ('' if args.static_only else 'C3/C4 per-GPU shard: ')
                                   + workload_name(args.static_only, B, T, H, W, args.cube, head['w'], src_hw,
                                                   args.all_steps, args.f32_input),
                       'clips_per_gpu': B, 'frames_per_clip': T, 'equi': [H, W], 'cube_dim': args.cube,
                       'graph_replay': bool(args.graph),
                       'temporal_stage_dtype': DTYPE[args.precision], 'static_stage_dtype': head['static_dtype'],
                       'static_stage_fp16_fallback': head['fp16_fallback'],
                       'parallelism': 'clips sharded over %d GPU(s), 1 all-gather of maps' % world,
                       # how the K timed steps are issued: as a stream of batches through the engine's two-stage software pipeline
                       # (static stage of batch k+1 on a second HIP stream beside the ConvLSTM of batch k; the pipeline starts empty
                       # and is drained INSIDE the timed region; per-batch maps bit-identical to the sequential form), or one by one
                       'pipelining': ('SaliencyEngine.stream: two HIP streams, fill + drain inside the timed region'
                                      if head.get('pipelined') else 'none: engine(frames) batch by batch'),
                       # (inside `config` so that the driver's record keeps them) the same K steps WITHOUT pipelining - the form
                       # rounds 1-4 quoted as `value` -, where one step's time goes, and (N = 1, filled in below) the headline
                       # workload sustained for 250 steps
                       'sequential': head.get('sequential'), 'stage_ms': head.get('stage_ms'), 'sustained': None},
            'roofline': head['roofline'],
            'cpu_baseline': None,
            # the same K steps WITHOUT pipelining (the form rounds 1-4 quoted as `value`)
            'sequential': head.get('sequential'),
            # diagnosis of multi-GPU runs (not part of `value`): each rank's own ms per step before the closing barrier,
            # and the single collective of the path - the all-gather of the maps - timed alone after the timed region
            'ms_per_step_per_rank': head.get('ms_per_step_per_rank'),
            'allgather_ms': head.get('allgather_ms'), 'allgather_bytes_per_rank': head.get('allgather_bytes_per_rank'),
            'map_shape': head.get('map_shape'),
            # attribution of one box's number (measured after the timed region): where a step's time goes, and the shader
            # clock this chip holds under a dense bf16 MFMA load (boxes differ: MI355X_MICROARCH.md, DVFS give-back)
            'stage_ms': head.get('stage_ms'),
            'held_clock_ghz': None if args.stub_engine else _safe(lambda: ops.held_clock_ghz(dev)),
            # ... that probe is a register-only MFMA loop: it ranks boxes, it is NOT the clock the dominant kernel holds (LDS, DMA
            # and HBM traffic beside the matrix pipe draw power too).  The counter-derived clock of that kernel, from the tracked
            # PMC summary of the same launch shape (SQ_BUSY_CYCLES / 32 / launch time; not re-measured in this run):
            'dominant_kernel_clock': pmc_clock(args.precision, B, head.get('w')),
        }
        if args.stub_engine:
            line['stub_engine'] = True
            line['data'] = 'stub'
        if world == 1 and not args.no_secondary and not args.stub_engine:
            sec = []

            def guarded(name, fn):
                """A secondary line must never cost the headline its JSON line: a failure is recorded, not raised."""
                try:
                    sec.append(fn())
                except Exception as e:                      # noqa: BLE001 - reported in the line
                    sec.append({'name': name, 'error': '%s: %s' % (type(e).__name__, str(e)[:300])})
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()

            def add(name, H2, W2, cd2, B2, T2, prec, steps, warmup, **kw):
                def run():
                    r = run_workload(dev, 0, 1, H2, W2, cd2, B2, T2, prec, steps, warmup, want_roofline=True, **kw)
                    return {'name': name, 'workload': workload_name(kw.get('static_only', False), B2, T2, H2, W2, cd2, r['w'],
                                                                    None, kw.get('all_steps', False), kw.get('f32_input', False)),
                            'dtype': DTYPE[prec], 'static_stage_dtype': r['static_dtype'], 'graph_replay': bool(kw.get('graph')),
                            'value': r['value'], 'unit': 'frames/s', 'ms_per_step': r['ms_per_step'], 'steps': steps, 'warmup': warmup,
                            'roofline': r['roofline'], 'pipelined': bool(r.get('pipelined')),
                            **{k: r[k] for k in ('static_plan', 'convlstm_winograd', 'held_clock_ghz_after', 'timed_region_s', 'sequential')
                               if k in r}}
                guarded(name, run)

            # the headline workload held for >= 3 s of timed region (MI355X_MICROARCH.md, DVFS give-back: a clock is believed after
            # >= 2 s of back-to-back load; the headline's own region is steps x ~13 ms)
            add('C4 per-GPU shard, SUSTAINED: the headline workload for 250 steps (>= 3 s of timed region)', 1024, 2048, 224, 4, 16,
                args.precision, 250, 5, clock_after=True, pipelined=not args.sequential)
            add('C4 per-GPU shard, fp32 (the 1e-3 parity precision)', 1024, 2048, 224, 4, 16, 'fp32', 3, 1)
            add('C4 per-GPU shard, bf16 in BOTH stages (static stage bf16 instead of fp16)', 1024, 2048, 224, 4, 16, 'bf16', 3, 1,
                static_precision='bf16')
            add('C3 literal: one 16-frame clip, bf16, eager launches', 1024, 2048, 224, 1, 16, 'bf16', 10, 3)
            add('C3 literal: one 16-frame clip, bf16, hipGraph replay', 1024, 2048, 224, 1, 16, 'bf16', 10, 3, graph=True)
            add('C2: one frame (6 faces), fp32, static path only', 1024, 2048, 224, 1, 1, 'fp32', 20, 5, static_only=True)
            add('C2: one frame (6 faces), fp32, static path only, hipGraph replay', 1024, 2048, 224, 1, 1, 'fp32', 20, 5,
                static_only=True, graph=True)
            add('C5 per-GPU shard: one 16-frame 2048x4096 clip, 6x512^2 faces, fp16', 2048, 4096, 512, 1, 16, 'fp16', 3, 1)
            # "C2 ... cd = 256 optional variant" (the reference's own smoke-test size, model/cube_pad.py:256-261): layer4 is
            # 8x8, the map 16x32; no fused static-stage kernel is specialised to it (static_plan says which path every layer takes)
            add('cube 256 variant: 4 clips x 16 frames 1024x2048, 6x256^2 faces -> 8x8 -> 16x32 map', 1024, 2048, 256, 4, 16,
                args.precision, 5, 2, want_plan=True)
            add('C4 per-GPU shard, frames resident as f32 [0,1] instead of u8 (SURVEY 8(d) fp32-input variant)',
                1024, 2048, 224, 4, 16, args.precision, 5, 2, f32_input=True)
            add('C4 per-GPU shard, return_all_steps: a map after every ConvLSTM step ([4, 16, 14, 28])',
                1024, 2048, 224, 4, 16, args.precision, 5, 2, all_steps=True)
            guarded('reference temporal workload: sliding window', lambda: sliding_bench(dev))
            guarded('Level-1 drop-in path', lambda: level1_bench(dev))
            line['secondary'] = sec
            sus = sec[0] if sec and 'SUSTAINED' in sec[0].get('name', '') and 'value' in sec[0] else None
            if sus:
                line['config']['sustained'] = {'value': sus['value'], 'ms_per_step': sus['ms_per_step'], 'steps': sus['steps'],
                                               'timed_region_s': sus.get('timed_region_s'), 'pipelined': sus.get('pipelined'),
                                               'sequential': (sus.get('sequential') or {}).get('value'),
                                               'held_clock_ghz_after': sus.get('held_clock_ghz_after')}
        if world == 1 and not args.no_cpu_baseline and not args.stub_engine:
            try:
                line['cpu_baseline'] = cpu_baseline(args.precision, dev, args.static_precision or None, clips=B)
            except Exception as e:                          # noqa: BLE001 - the headline line is printed whatever happens here
                line['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        print(json.dumps(line))
    cpdist.barrier()
    cpdist.release_gather_buffers()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
