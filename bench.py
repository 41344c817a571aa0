#!/usr/bin/env python3
"""Benchmark of the hot path on N MI355X GPUs of one node (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs C3/C4 per-GPU shard): each rank owns `--clips` clips of
`--frames` synthetic 1024x2048 u8 equirectangular frames, resident in HBM before the
timed region.  One step = the whole path over that batch: equi->cube (K1), CubePad +
ResNet-50-cubic (K2/K3), CAM (K4), window normalise (K7), T ConvLSTM steps (K5),
cube->equi + channel max (K6), and for N > 1 one RCCL all-gather of the saliency maps.
Metric: frames/s = N * clips * frames * K / max-over-ranks wall time.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (the ConvLSTM implicit-GEMM convolution, timed live with HIP events) and
`cpu_baseline` (the oracle on the host cores, bounded sample, rank 0 at N = 1 only).
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


class LaunchTimer:
    """HIP-event timer around tagged kernel launches on torch's current stream (the
    stream libcp360 launches on).  Events are resolved after the timed region."""

    TAGS = ('clstm.Conv2', 'clstm.Gates')     # the dominant kernel: K = 36000 ConvLSTM convolutions

    def __init__(self, every=4):
        self.active = False
        self.records = []          # (tag, flops, start_event, end_event)
        # an event pair costs ~6 us of stream bubble per launch (measured: profiles/), so only every
        # `every`-th launch of the dominant kernel is bracketed: 8 of the 32 per 16-frame window
        self.every = every
        self.count = 0

    def wrap(self, tag, flops, fn):
        if not self.active or tag not in self.TAGS:
            return fn()
        self.count += 1
        if self.count % self.every:
            return fn()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn()
        b.record()
        self.records.append((tag, flops, a, b))
        return rc

    def summary(self):
        by = {}
        for tag, flops, a, b in self.records:
            ms = a.elapsed_time(b)
            d = by.setdefault(tag, {'n': 0, 'ms': 0.0, 'flops': flops})
            d['n'] += 1
            d['ms'] += ms
        return by


def cpu_baseline(H, W, cd, precision, dev):
    """The oracle (numpy / torch-CPU restatement of the reference, oracle/) on the host
    cores for a bounded sample: ONE clip of 2 frames at the benchmark resolution through
    the whole path (static stage x2, ConvLSTM x2, cube->equi).  The same sample then goes
    through the HIP path at the benchmark precision and the oracle acts as the checker for
    the second half of the metric (AUC-Judd / CC of both maps against a synthetic fixation
    map): reported under "check"."""
    from tests.parity_helpers import oracle_pipeline
    from oracle import o_metrics
    # 32 threads: fastest setting measured on the GPU box's host (tools/cpu_threads_probe.py); using all
    # 256 hardware threads of the EPYC host makes oneDNN ~100x slower on these small convolutions
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    rs = synth.resnet50_state(seed=1)
    cs = synth.clstm_state(seed=2)
    clip = synth.clip_u8(3, 2, H, W)
    t0 = time.time()
    ref = oracle_pipeline(clip, rs, cs, cd)
    dt = time.time() - t0
    eng = SaliencyEngine(rs, cs, (H, W), cd, clips=1, frames=2, precision=precision, device=dev)
    got = eng(torch.from_numpy(clip[None]).to(dev)).float().cpu().numpy()[0]
    fix = synth.fixation_map(103, H // 2, W // 2)
    rng = lambda: np.random.RandomState(0)
    check = {'max_abs_diff': float(np.max(np.abs(got - ref))),
             'auc_judd': [round(o_metrics.auc_judd(ref, fix, rng=rng()), 6), round(o_metrics.auc_judd(got, fix, rng=rng()), 6)],
             'cc': [round(o_metrics.corr_coeff(ref, fix), 6), round(o_metrics.corr_coeff(got, fix), 6)],
             'what': 'oracle vs HIP (%s) saliency of the sample clip; [oracle, hip] metrics vs a synthetic fixation map' % precision}
    return {'value': round(2.0 / dt, 4), 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '1 clip x 2 frames %dx%d -> 6x%d^2, oracle fp32 (torch-CPU conv, numpy remap), %.1f s'
                      % (H, W, cd, dt), 'check': check}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--precision', default=os.environ.get('CP360_PRECISION', 'bf16'), choices=['fp32', 'bf16', 'fp16'])
    ap.add_argument('--clips', type=int, default=4, help='clips per GPU')
    ap.add_argument('--frames', type=int, default=16, help='frames per clip')
    ap.add_argument('--equi', default='1024x2048')
    ap.add_argument('--cube', type=int, default=224)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--frame-chunk', type=int, default=0, help='frames per static-stage group (0 = all)')
    ap.add_argument('--graph', action='store_true',
                    help='replay the step from a HIP graph (launch-bound small configs; no per-kernel roofline timing)')
    ap.add_argument('--source', default='', help='decoded frame size HxW: include the PIL-exact Lanczos resize (K0) in the step')
    ap.add_argument('--static-only', action='store_true',
                    help='BASELINE config C2: the static path only (equi -> cube -> ResNet-50 -> CAM); no roofline object')
    args = ap.parse_args()

    rank, world, local = cpdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    H, W = (int(v) for v in args.equi.split('x'))
    B, T = args.clips, args.frames

    rs = synth.resnet50_state(seed=1)
    cs = synth.clstm_state(seed=2)
    src_hw = tuple(int(v) for v in args.source.split('x')) if args.source else None
    eng = SaliencyEngine(rs, cs, (H, W), args.cube, clips=B, frames=T, precision=args.precision, device=dev,
                         frame_chunk=args.frame_chunk or None, source_hw=src_hw)
    del rs, cs
    # this rank's clips (global clip id = rank*B + b), resident in HBM before timing
    fh, fw = src_hw if src_hw else (H, W)
    frames = torch.stack([torch.from_numpy(synth.clip_u8(3 + rank * B + b, T, fh, fw)) for b in range(B)]).to(dev)
    n_clips = world * B

    timer = LaunchTimer()
    if args.graph and not args.static_only:
        eng.capture(frames)
    else:
        ops.LAUNCH_TIMER = timer

    def step():
        if args.static_only:
            with torch.no_grad():
                cam = eng.static_stage(frames.reshape((B * T,) + tuple(frames.shape[2:])))
            return cam.view(B, -1)[:, :8].float()
        sal = eng(frames)
        return cpdist.gather_maps(sal, n_clips, rank, world)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    cpdist.barrier()
    torch.cuda.synchronize()
    timer.active = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    cpdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.active = False
    elapsed = cpdist.max_over_ranks(elapsed, dev)
    assert (args.static_only or out.shape[0] == n_clips) and bool(torch.isfinite(out).all())

    if rank == 0:
        frames_total = world * B * T * args.steps
        ksum = timer.summary()
        # dominant kernel: the K = 36000 ConvLSTM convolutions (Conv2 / Gates have the same shape)
        dom = [ksum[k] for k in ('clstm.Conv2', 'clstm.Gates') if k in ksum]
        roof = None
        if dom:
            n = sum(d['n'] for d in dom)
            ms = sum(d['ms'] for d in dom) / n
            flops = dom[0]['flops']
            ach = flops / (ms * 1e-3) / 1e12
            traffic = None
            tp = os.path.join(REPO, 'profiles', 'traffic_%s.json' % args.precision)
            if os.path.exists(tp):
                traffic = json.load(open(tp)).get('conv_igemm_clstm_bytes_per_launch')
            roof = {'bound': 'mfma', 'kernel': '%s (ConvLSTM Conv2/Gates, M=%d N=4000 K=36000)'
                    % ('conv_clip_kernel' if eng.w <= 7 else 'conv_igemm_ring_kernel', 6 * B * eng.w * eng.w), 'achieved': round(ach, 2), 'peak': PEAK[args.precision],
                    'unit': 'TFLOP/s', 'frac': round(ach / PEAK[args.precision], 4), 'traffic': traffic,
                    'avg_launch_ms': round(ms, 4), 'launches_timed': n, 'flops_per_launch': flops}
        line = {
            'metric': 'frames/sec end-to-end 1024x2048 equi->saliency',
            'value': round(frames_total / elapsed, 3), 'unit': 'frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1000.0 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'bf16': 'bf16', 'fp16': 'f16', 'fp32': 'f32'}[args.precision], 'data': 'synthetic',
            'config': {'workload': ('C2 static path only: %d x %d frames %dx%d u8 equi -> 6x%d^2 cube -> CubePad ResNet-50 -> CAM'
                                    % (B, T, H, W, args.cube)) if args.static_only else
                                   ('C3/C4 per-GPU shard: %d clips x %d frames %dx%d u8 equi -> 6x%d^2 cube -> '
                                    'CubePad ResNet-50 -> CAM -> ConvLSTM x%d -> cube_to_equi saliency %dx%d'
                                    % (B, T, H, W, args.cube, T, 2 * eng.w, 4 * eng.w))
                                   + ((' (frames decoded at %dx%d, PIL-exact Lanczos resize included)' % (fh, fw)) if src_hw else ''),
                       'clips_per_gpu': B, 'frames_per_clip': T, 'equi': [H, W], 'cube_dim': args.cube,
                       'parallelism': 'clips sharded over %d GPU(s), 1 all-gather of maps' % world},
            'roofline': roof,
            'cpu_baseline': None,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(H, W, args.cube, args.precision, dev)
        print(json.dumps(line))
    cpdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
