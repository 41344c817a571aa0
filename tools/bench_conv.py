"""Micro-benchmark of the implicit-GEMM convolution on the shapes of the path
(GPU box).  Prints avg launch time (HIP events over back-to-back launches) and TFLOP/s.

    python tools/bench_conv.py [--precision bf16] [--clips 4] [--frames 64]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp_360_weakly_supervised_saliency_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='bf16')
ap.add_argument('--clips', type=int, default=4)
ap.add_argument('--frames', type=int, default=64)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--only', default='')
ap.add_argument('--splits', type=int, default=0)
ap.add_argument('--residual', action='store_true', help='add a residual input (ResNet conv3)')
ap.add_argument('--tile-px', type=int, default=0)
ap.add_argument('--clip-resident', type=int, default=-1, help='1/0 force the clip-resident kernel on/off')
args = ap.parse_args()
dt = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[args.precision]
dev = 'cuda'
B, F = args.clips, args.frames
# (name, n_img, cin, cout, n, k, stride, pad)
shapes = [
    ('clstm.Conv2 ', 6 * B, 4000, 4000, 7, 3, 1, 1),
    ('clstm.Conv1 ', 6 * B, 2000, 4000, 7, 3, 1, 1),
    ('l1.conv1 1x1', 6 * F, 256, 64, 56, 1, 1, 0),
    ('l1.conv2 3x3', 6 * F, 64, 64, 56, 3, 1, 1),
    ('l1.conv3 1x1', 6 * F, 64, 256, 56, 1, 1, 0),
    ('l2.conv1 1x1', 6 * F, 512, 128, 28, 1, 1, 0),
    ('l2.conv2 3x3', 6 * F, 128, 128, 28, 3, 1, 1),
    ('l2.conv3 1x1', 6 * F, 128, 512, 28, 1, 1, 0),
    ('l3.conv2 3x3', 6 * F, 256, 256, 14, 3, 1, 1),
    ('l3.conv3 1x1', 6 * F, 256, 1024, 14, 1, 1, 0),
    ('l4.conv2 3x3', 6 * F, 512, 512, 7, 3, 1, 1),
    ('l4.conv3 1x1', 6 * F, 512, 2048, 7, 1, 1, 0),
    ('cam 1x1     ', 6 * F, 2048, 1000, 7, 1, 1, 0),
    ('l3.conv1 1x1', 6 * F, 1024, 256, 14, 1, 1, 0),
    ('l4.conv1 1x1', 6 * F, 2048, 512, 7, 1, 1, 0),
    ('l3.0conv1 1x1', 6 * F, 512, 256, 28, 1, 1, 0),
    ('l4.0conv1 1x1', 6 * F, 1024, 512, 14, 1, 1, 0),
    ('c5face.Conv2', 6 * B, 4000, 4000, 16, 3, 1, 1),      # config C5: 16x16 faces (use --clips 1)
    ('c5face.Conv1', 6 * B, 2000, 4000, 16, 3, 1, 1),
    ('w8.Conv2    ', 6 * B, 4000, 4000, 8, 3, 1, 1),        # cube 256: 8x8 faces
    ('w8.Conv1    ', 6 * B, 2000, 4000, 8, 3, 1, 1),
]
for name, n_img, cin, cout, n, k, s, pad in shapes:
    if args.only and args.only not in name:
        continue
    w = torch.randn(cout, cin, k, k) * (2.0 / (k * k * cin)) ** 0.5
    conv = ops.Conv(w, None, torch.zeros(cout), s, pad, True, dt, dev)
    x = torch.randn(n_img, n, n, cin, device=dev).to(dt)
    kw = {'splits': args.splits} if args.splits else {}
    if args.tile_px:
        kw['tile_px'] = args.tile_px
    if args.clip_resident >= 0 and k == 3 and (n <= 7 or n in (8, 16)) and cout >= 256:
        kw['clip_resident'] = bool(args.clip_resident)
    if args.residual:
        ho_ = (n + 2 * pad - k) // s + 1
        kw['residual'] = torch.randn(n_img, ho_, ho_, cout, device=dev).to(dt)
    y = conv(x, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        y = conv(x, **kw)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.iters
    ho = y.shape[1]
    flops = 2.0 * n_img * ho * ho * cout * cin * k * k
    M = n_img * ho * ho
    sp = args.splits or list(conv._splits_cache.values())[-1]
    print('%s M=%7d N=%4d K=%5d splits=%d  %8.3f ms  %7.1f TFLOP/s' % (name, M, cout, cin * k * k, sp, ms, flops / ms / 1e9),
          flush=True)
