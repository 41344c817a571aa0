#!/usr/bin/env python3
"""Benchmark of the hot path on N MI355X GPUs of one node (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json config C4's per-GPU shard = 4x config C3): each rank owns `--clips`
clips of `--frames` synthetic HxW u8 equirectangular frames, resident in HBM before the timed region.
One step = the whole path over that batch: equi->cube (K1), CubePad + ResNet-50-cubic (K2/K3), CAM
(K4), window normalise (K7), T ConvLSTM steps (K5), cube->equi + channel max (K6), and for N > 1 one
RCCL all-gather of the saliency maps.  Metric: frames/s = N * clips * frames * K / max-over-ranks
wall time.  The timed region carries NO instrumentation; the per-kernel roofline of the dominant
kernel (the ConvLSTM implicit-GEMM convolution) is measured with HIP events in a separate short pass
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


def timed(step, warmup, steps, dev):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
    out = None
    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    cpdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    cpdist.barrier()
    torch.cuda.synchronize()
    return cpdist.max_over_ranks(time.perf_counter() - t0, dev), out


def roofline_pass(eng, frames, precision, steps=2):
    """Dominant kernel = the K = 36000 ConvLSTM convolutions (Conv2 / Gates have the same shape):
    algorithmic flops of one launch / its mean duration over `steps` extra steps, every launch
    bracketed by HIP events on the launch stream."""
    timer = LaunchTimer()
    ops.LAUNCH_TIMER = timer
    timer.active = True
    for _ in range(steps):
        eng(frames)
    torch.cuda.synchronize()
    timer.active = False
    ops.LAUNCH_TIMER = None
    n, ms, flops = timer.summary()
    if not n:
        return None
    ms /= n
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = None, None
    tp = os.path.join(REPO, 'profiles', 'traffic_%s.json' % precision)
    if os.path.exists(tp) and (eng.B, eng.w) == (4, 7):          # counters were collected on this launch shape only
        t = json.load(open(tp))
        traffic, src = t.get('conv_igemm_clstm_bytes_per_launch'), 'profiles/' + t.get('source', '')
    M = 6 * eng.B * eng.w * eng.w
    return {'bound': 'mfma', 'kernel': 'conv_clip_kernel<%s> (ConvLSTM Conv2/Gates, M=%d N=%d K=%d)'
            % ('face tile' if eng.w > 7 else 'clip tile', M, 4 * eng.cell.hidden_size, 36 * eng.cell.hidden_size),
            'achieved': round(ach, 2), 'peak': PEAK[precision], 'unit': 'TFLOP/s', 'frac': round(ach / PEAK[precision], 4),
            'traffic': traffic, 'traffic_source': src and (src + ' (rocprofv3 PMC passes, (2*FETCH_SIZE+WRITE_SIZE)*1024 per launch; not re-measured in this run)'),
            'avg_launch_ms': round(ms, 4), 'launches_timed': n, 'flops_per_launch': flops}


def run_workload(dev, rank, world, H, W, cd, B, T, precision, steps, warmup, graph=False, static_only=False,
                 frame_chunk=None, source_hw=None, want_roofline=False, static_precision=None):
    rs = synth.resnet50_state(seed=1)
    cs = synth.clstm_state(seed=2)
    eng = SaliencyEngine(rs, cs, (H, W), cd, clips=B, frames=T, precision=precision, device=dev,
                         frame_chunk=frame_chunk, source_hw=source_hw, static_precision=static_precision)
    del rs, cs
    fh, fw = source_hw if source_hw else (H, W)
    # this rank's clips (global clip id = rank*B + b), resident in HBM before timing
    frames = torch.stack([torch.from_numpy(synth.clip_u8(3 + rank * B + b, T, fh, fw)) for b in range(B)]).to(dev)
    n_clips = world * B
    if graph and not static_only:
        eng.capture(frames)

    def step():
        if static_only:
            with torch.no_grad():
                cam = eng.static_stage(frames.reshape((B * T,) + tuple(frames.shape[2:])))
            return cam.view(B, -1)[:, :8].float()
        sal = eng(frames)
        return cpdist.gather_maps(sal, n_clips, rank, world)

    elapsed, out = timed(step, warmup, steps, dev)
    assert (static_only or out.shape[0] == n_clips) and bool(torch.isfinite(out).all())
    res = {'value': round(world * B * T * steps / elapsed, 3), 'ms_per_step': round(1000.0 * elapsed / steps, 3),
           'roofline': None}
    if want_roofline and not static_only:
        if graph:
            eng._graph = None                      # the roofline pass needs eager launches to bracket
        res['roofline'] = roofline_pass(eng, frames, precision)
    res['w'] = eng.w
    res['static_dtype'] = DTYPE[eng.static_precision]
    del eng, frames
    torch.cuda.empty_cache()
    return res


def workload_name(static_only, B, T, H, W, cd, w, src_hw=None):
    if static_only:
        return 'C2 static path only: %d x %d frames %dx%d u8 equi -> 6x%d^2 cube -> CubePad ResNet-50 -> CAM' % (B, T, H, W, cd)
    s = ('%d clips x %d frames %dx%d u8 equi -> 6x%d^2 cube -> CubePad ResNet-50 -> CAM -> ConvLSTM x%d -> '
         'cube_to_equi saliency %dx%d' % (B, T, H, W, cd, T, 2 * w, 4 * w))
    if src_hw:
        s += ' (frames decoded at %dx%d, PIL-exact Lanczos resize included)' % src_hw
    return s


def cpu_baseline(precision, dev, static_precision=None):
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
    eng = SaliencyEngine(rs, cs, (H, W), cd, clips=1, frames=T, precision=precision, device=dev,
                         static_precision=static_precision)
    got = eng(torch.from_numpy(clip[None]).to(dev)).float().cpu().numpy()[0]
    del eng
    fix = synth.fixations_from_map(ref, 200, H // 2, W // 2)
    rng = lambda: np.random.RandomState(0)
    check = {'max_abs_diff': float(np.max(np.abs(got - ref))),
             'auc_judd': [round(o_metrics.auc_judd(ref, fix, rng=rng()), 6), round(o_metrics.auc_judd(got, fix, rng=rng()), 6)],
             'cc': [round(o_metrics.corr_coeff(ref, fix), 6), round(o_metrics.corr_coeff(got, fix), 6)],
             'cc_build_vs_oracle': round(o_metrics.corr_coeff(got, ref), 6),
             'what': 'oracle vs HIP (%s) saliency of the T=16 sample clip; [oracle, hip] metrics vs fixations '
                     'sampled from the oracle map' % precision}
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
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
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
    ap.add_argument('--source', default='', help='decoded frame size HxW: include the PIL-exact Lanczos resize (K0) in the step')
    ap.add_argument('--static-only', action='store_true',
                    help='BASELINE config C2: the static path only (equi -> cube -> ResNet-50 -> CAM); no roofline object')
    ap.add_argument('--cpu-anchor', action='store_true',
                    help='no GPU: time the oracle on the pieces BASELINE.md section 2 timed with the reference itself '
                         '(CAM of one cube, one ConvLSTM step) - the sanity anchor of the cpu_baseline leg')
    args = ap.parse_args()
    if args.cpu_anchor:
        print(json.dumps(cpu_anchor()))
        return

    rank, world, local = cpdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    H, W = (int(v) for v in args.equi.split('x'))
    B, T = args.clips, args.frames
    src_hw = tuple(int(v) for v in args.source.split('x')) if args.source else None

    head = run_workload(dev, rank, world, H, W, args.cube, B, T, args.precision, args.steps, args.warmup,
                        graph=args.graph, static_only=args.static_only, frame_chunk=args.frame_chunk or None,
                        source_hw=src_hw, want_roofline=(rank == 0), static_precision=args.static_precision or None)

    if rank == 0:
        line = {
            'metric': 'frames/sec end-to-end %dx%d equi->saliency' % (H, W),
            'value': head['value'], 'unit': 'frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': head['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': DTYPE[args.precision], 'data': 'synthetic',
            'config': {'workload': ('' if args.static_only else 'C3/C4 per-GPU shard: ')
                                   + workload_name(args.static_only, B, T, H, W, args.cube, head['w'], src_hw),
                       'clips_per_gpu': B, 'frames_per_clip': T, 'equi': [H, W], 'cube_dim': args.cube,
                       'graph_replay': bool(args.graph),
                       'temporal_stage_dtype': DTYPE[args.precision], 'static_stage_dtype': head['static_dtype'],
                       'parallelism': 'clips sharded over %d GPU(s), 1 all-gather of maps' % world},
            'roofline': head['roofline'],
            'cpu_baseline': None,
        }
        if world == 1 and not args.no_secondary:
            sec = []

            def add(name, H2, W2, cd2, B2, T2, prec, steps, warmup, **kw):
                r = run_workload(dev, 0, 1, H2, W2, cd2, B2, T2, prec, steps, warmup, want_roofline=not kw.get('static_only'), **kw)
                sec.append({'name': name, 'workload': workload_name(kw.get('static_only', False), B2, T2, H2, W2, cd2, r['w']),
                            'dtype': DTYPE[prec], 'static_stage_dtype': r['static_dtype'], 'graph_replay': bool(kw.get('graph')), 'value': r['value'],
                            'unit': 'frames/s', 'ms_per_step': r['ms_per_step'], 'steps': steps, 'warmup': warmup,
                            'roofline': r['roofline']})

            add('C4 per-GPU shard, fp32 (the 1e-3 parity precision)', 1024, 2048, 224, 4, 16, 'fp32', 3, 1)
            add('C4 per-GPU shard, bf16 in BOTH stages (static stage bf16 instead of fp16)', 1024, 2048, 224, 4, 16, 'bf16', 3, 1,
                static_precision='bf16')
            add('C3 literal: one 16-frame clip, bf16, eager launches', 1024, 2048, 224, 1, 16, 'bf16', 10, 3)
            add('C3 literal: one 16-frame clip, bf16, hipGraph replay', 1024, 2048, 224, 1, 16, 'bf16', 10, 3, graph=True)
            add('C2: one frame (6 faces), fp32, static path only', 1024, 2048, 224, 1, 1, 'fp32', 20, 5, static_only=True)
            add('C5 per-GPU shard: one 16-frame 2048x4096 clip, 6x512^2 faces, fp16', 2048, 4096, 512, 1, 16, 'fp16', 3, 1)
            line['secondary'] = sec
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.precision, dev, args.static_precision or None)
        print(json.dumps(line))
    cpdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
