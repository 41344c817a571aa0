"""HBM-bound kernels of the path, one by one, at the shapes of the benchmark step (4 clips x 16 frames of
1024x2048, cube 224, 16-bit static stage): algorithmic bytes, microseconds, GB/s and the fraction of the 8 TB/s
HBM3E peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a float4 copy reaches).  Times are HIP-event averages over
back-to-back launches on the launch stream.  Kernels whose working set fits the 256 MiB Infinity Cache say so:
their rate is a cache rate, which is also how they run inside the pipeline (producer -> consumer).

    python tools/hbm_kernels.py [--md profiles/r02_hbm_kernels.md] [--json profiles/r02_hbm_kernels.json]

The same script under rocprofv3 (`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/hbm_kernels.py`, then the
same with WRITE_SIZE: separate passes, as tools/collect_profile.sh does for bench.py) gives the counter-side bytes per
kernel.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cp_360_weakly_supervised_saliency_amd import ops                                  # noqa: E402
from cp_360_weakly_supervised_saliency_amd.model.cube_pad import CubePad                # noqa: E402
from cp_360_weakly_supervised_saliency_amd.utils import synth, hashrng                  # noqa: E402
from cp_360_weakly_supervised_saliency_amd.utils.cube_to_equi import Cube2Equi          # noqa: E402
from cp_360_weakly_supervised_saliency_amd.utils.equi_to_cube import Equi2Cube          # noqa: E402
from cp_360_weakly_supervised_saliency_amd.utils.resize import LanczosResize            # noqa: E402

PEAK = 8000.0     # GB/s


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default='')
    ap.add_argument('--json', default='')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--frames', type=int, default=64)
    args = ap.parse_args()
    dev = torch.device('cuda')
    F, H, W, cd, B, T = args.frames, 1024, 2048, 224, 4, 16
    n6 = 6 * F
    h16 = torch.float16
    rows = []

    def add(name, fn, nbytes, note=''):
        us = timed(fn, args.reps)
        gbs = nbytes / us / 1e3
        rows.append({'kernel': name, 'bytes': int(nbytes), 'us': round(us, 2), 'GBps': round(gbs, 1),
                     'frac_of_8TBps': round(gbs / PEAK, 3), 'note': note})
        print('%-58s %9.1f MB %9.1f us %8.1f GB/s  %.2f  %s' % (name, nbytes / 1e6, us, gbs, gbs / PEAK, note))

    # ---- K1 equi -> cube (padded NHWC4, u8 in, f16 out)
    frames = torch.from_numpy(np.stack([synth.frame_u8(3, H, W)] * 1)).to(dev).expand(F, H, W, 3).contiguous()
    e2c = Equi2Cube(cd, (H, W), device=dev)
    npx = 6 * (cd + 6) ** 2
    add('K1 equi2cube u8 -> f16 NHWC4 (CubePad(3) fused), %d frames' % F,
        lambda: e2c.to_cube_batch(frames, out_dtype=h16, layout='nhwc4p3'),
        F * (min(H * W * 3, 4 * npx * 3) + npx * 8) + npx * 8)
    # ---- K0 PIL-exact Lanczos resize 1080x2160 -> 1024x2048 (8 frames)
    src = torch.from_numpy(hashrng.uniform(5, (8, 1080, 2160, 3), 0, 256).astype(np.uint8)).to(dev)
    rz = LanczosResize((1080, 2160), (H, W), device=dev)
    add('K0 resize_lanczos u8 1080x2160 -> 1024x2048, 8 frames (2 passes)', lambda: rz(src),
        8 * 3 * (1080 * 2160 + 2 * 1080 * 2048 + 1024 * 2048))
    # ---- K2 standalone CubePad
    for (C, n, p, dt, lab) in ((3, 224, 3, torch.float32, 'f32'), (64, 112, 1, h16, 'f16'), (64, 56, 1, h16, 'f16'),
                               (256, 56, 1, h16, 'f16'), (128, 28, 1, h16, 'f16'), (256, 14, 1, h16, 'f16')):
        x = torch.randn((n6, C, n, n), device=dev).to(dt)
        pad = CubePad(p)
        s = x.element_size()
        add('K2 CubePad(%d) NCHW %s [%d,%d,%d,%d]' % (p, lab, n6, C, n, n), lambda x=x, pad=pad: pad(x),
            s * n6 * C * (n * n + (n + 2 * p) ** 2))
        xn = x.permute(0, 2, 3, 1).contiguous()
        if (C * s) % 4 == 0:
            add('K2 CubePad(%d) NHWC %s [%d,%d,%d,%d]' % (p, lab, n6, n, n, C), lambda xn=xn, p=p: ops.cubepad_nhwc(xn, p),
                s * n6 * C * (n * n + (n + 2 * p) ** 2))
        del x, xn
    xl = torch.randn((6 * B, 2000, 7, 7), device=dev)
    padl = CubePad(1)
    add('K2 CubePad(1) NCHW f32 ConvLSTM [24,2000,7,7]', lambda: padl(xl), 4 * 24 * 2000 * (49 + 81), 'fits the Infinity Cache')
    # ---- stem + max-pool
    xs = torch.randn((n6, 112, 112, 64), device=dev).to(h16)
    add('K3b cubepad_maxpool3s2 f16 [%d,112,112,64]' % n6, lambda: ops.cubepad_maxpool3s2(xs), 2 * n6 * 64 * (112 * 112 + 56 * 56))
    del xs
    # ---- K7 window min/max + normalise, conv_finish, lstm_gates (ConvLSTM glue at B = 4)
    cam = torch.randn((B, T, 294, 1000), device=dev)
    minmax = torch.empty((B, 2), device=dev)
    scratch = torch.empty((B * 512,), device=dev)
    add('K7 window_minmax f32 [4,16,294,1000]', lambda: ops.window_minmax(cam, B, T * 294 * 1000, minmax, scratch),
        4 * B * T * 294 * 1000, 'fits the Infinity Cache')
    xh = torch.empty((6 * B, 7, 7, 2000), device=dev, dtype=torch.bfloat16)
    add('K7 window_normalize f32 -> bf16 (one step, 4 clips)', lambda: ops.window_normalize(cam, minmax, xh, 0, None, B, T, 3, 294, 1000),
        B * 294 * 1000 * (4 + 2), 'fits the Infinity Cache')
    M, N4 = 6 * B * 49, 4000
    part = torch.randn((4, M, N4), device=dev)
    bias = torch.randn((N4,), device=dev)
    conv = ops.Conv(torch.zeros((N4, 32, 3, 3)), None, bias, 1, 1, True, torch.bfloat16, dev)
    outb = torch.empty((6 * B, 7, 7, N4), device=dev, dtype=torch.bfloat16)
    from cp_360_weakly_supervised_saliency_amd._lib import lib, ptr, stream, check
    import ctypes as C
    d = conv._desc(6 * B, 7, 7, 4, slab_rows=1)
    add('conv_finish: 4 f32 split-K slabs -> bias + ReLU -> bf16 [1176, 4000]',
        lambda: check(lib().cp360_conv_finish(C.byref(d), ptr(part), ptr(bias), None, ptr(outb), stream())),
        4 * 4 * M * N4 + 2 * M * N4, 'fits the Infinity Cache')
    c0 = torch.randn((M, 1000), device=dev)
    c1 = torch.empty_like(c0)
    hf = torch.empty_like(c0)
    add('lstm_gates: 4 slabs [1176, 4000] + cell -> hidden / cell',
        lambda: ops.lstm_gates(part, 4, bias, c0, c1, xh, 1000, hf, M, 1000, slab_rows=True),
        4 * 4 * M * N4 + M * 1000 * (4 + 4 + 2 + 4), 'fits the Infinity Cache')
    # ---- K6 cube -> equi + channel max
    c2e = Cube2Equi(7, device=dev)
    hid = torch.randn((6 * B, 7, 7, 1000), device=dev)
    add('K6 cube2equi + channel max f32 [24,7,7,1000] -> [4,14,28]', lambda: c2e.saliency(hid, layout='nhwc'),
        4 * 24 * 49 * 1000 + B * 14 * 28 * 13, 'fits the Infinity Cache; launch-bound')
    # ---- reference point: a plain copy of 616 MB
    big = torch.empty(308 * 1024 * 1024, device=dev, dtype=h16)
    big2 = torch.empty_like(big)
    add('(reference) torch copy_ of 616 MB f16', lambda: big2.copy_(big), 2 * big.numel() * 2)

    if args.json:
        json.dump(rows, open(args.json, 'w'), indent=1)
    if args.md:
        with open(args.md, 'w') as f:
            f.write('| kernel | algorithmic MB | us | GB/s | of 8 TB/s | note |\n|---|---|---|---|---|---|\n')
            for r in rows:
                f.write('| %s | %.1f | %.1f | %.0f | %.2f | %s |\n' % (r['kernel'], r['bytes'] / 1e6, r['us'], r['GBps'],
                                                                    r['frac_of_8TBps'], r['note']))


if __name__ == '__main__':
    main()
