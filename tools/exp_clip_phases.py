"""Where does a launch of the dominant kernel (conv_clip_kernel, ConvLSTM Conv2 / Gates: M = 1176, N = 4000, K = 36000, 4 K-splits,
256 workgroups) spend its time OUTSIDE the steady K loop?  Runs on the diagnostic build `tools/exp_build.sh clip_phases`
(CP360_LIB=...): every workgroup stamps s_memrealtime (100 MHz, chip-wide) at kernel entry, before its first sub-step, after its
last one and after its last slab store has completed.  Prints the distribution over the 256 workgroups of the LAST launch:
dispatch skew (entry - earliest entry), prologue, K loop, epilogue, and the launch window (earliest entry -> latest exit).

    CP360_LIB=/tmp/exp_clip_phases/libcp360.so python3 tools/exp_clip_phases.py [out.md]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cp_360_weakly_supervised_saliency_amd import ops, _lib

L = _lib.lib()
ORDER = int(os.environ.get('CP360_ORDER', '0'))       # 1: descending work-item order (does the slow XCD follow the work or the silicon?)
L.cp360_set_launch_order(ORDER)
w = torch.randn(4000, 4000, 3, 3) * 0.01
conv = ops.Conv(w, None, torch.zeros(4000), 1, 1, True, torch.bfloat16, 'cuda')
x = torch.randn(24, 7, 7, 4000, device='cuda').to(torch.bfloat16)
for _ in range(3):
    y = conv(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
N = 20
for _ in range(N):
    y = conv(x)
b.record()
torch.cuda.synchronize()
us_launch = a.elapsed_time(b) * 1000.0 / N            # conv + its split-K finish
nwg = 256
buf = (C.c_ulonglong * (nwg * 8))()
L.cp360_debug_phases.argtypes = [C.c_void_p, C.c_int]
assert L.cp360_debug_phases(buf, nwg * 8) == 0
t = np.array(list(buf), dtype=np.float64).reshape(nwg, 2, 4) / 100.0      # microseconds
t0 = t[:, :, 0].min()
lines = []
say = lines.append
say('conv_clip_kernel<bf16, clip tile>, ConvLSTM Conv2 shape (M = 1176, N = 4000, K = 36000, 4 K-splits, 256 workgroups), back-to-back')
say('launches: %.1f us per convolution incl. its split-K finish (HIP events over %d launches).  Stamps: s_memrealtime (10 ns), per' % (us_launch, N))
say('workgroup (wave 0 = leading half, wave 4 = trailing half), of the last launch.')
say('')
say('| quantity (us) | min | median | p90 | max |')
say('|---|---|---|---|---|')


def row(name, v):
    v = np.asarray(v).reshape(-1)
    say('| %s | %.2f | %.2f | %.2f | %.2f |' % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))


row('dispatch skew: workgroup entry - earliest entry', t[:, 0, 0] - t0)
row('prologue: entry -> first sub-step (source-row table, first 5 weight stages + activation tile requested, first tile landed)', t[:, :, 1] - t[:, :, 0])
row('K loop: first sub-step -> last MFMA issued (%d sub-steps)' % (9 * 4000 * 2 // 64 // 4), t[:, :, 2] - t[:, :, 1])
row('epilogue: last MFMA issued -> slab stores complete (vmcnt 0)', t[:, :, 3] - t[:, :, 2])
row('workgroup lifetime: entry -> exit', t[:, :, 3].max(axis=1) - t[:, :, 0].min(axis=1))
say('')
win = t[:, :, 3].max() - t0
loop = np.median(t[:, :, 2] - t[:, :, 1])
say('launch window (earliest entry -> latest exit): %.2f us; median K loop %.2f us = %.1f %% of it; prologue + epilogue + skew = the rest'
    % (win, loop, 100.0 * loop / win))
say('latest exit - median exit: %.2f us (the tail during which some CUs already idle)' % (t[:, :, 3].max() - np.median(t[:, :, 3].max(axis=1))))
# who is slow?  blockIdx -> XCD (blockIdx & 7) and, through the kernel's XCD-contiguous mapping, -> (clip, channel tile, split)
bid = np.arange(nwg)
xcd = bid & 7
w_item = xcd * (nwg // 8) + (bid >> 3)
if ORDER == 1:
    w_item = nwg - 1 - w_item
clip_i, tile_i, split_i = w_item % 4, (w_item // 4) % 16, w_item // 64
kl = (t[:, :, 2] - t[:, :, 1]).max(axis=1)
say('')
say('K loop (us, the slower half of each workgroup), mean by XCD:   ' + '  '.join('%d: %.1f' % (k, kl[xcd == k].mean()) for k in range(8)))
say('K loop mean by clip (the four workgroups that share a weight stream): ' + '  '.join('%d: %.1f' % (k, kl[clip_i == k].mean()) for k in range(4)))
say('K loop mean by K split: ' + '  '.join('%d: %.1f' % (k, kl[split_i == k].mean()) for k in range(4)))
say('K loop mean by channel tile: ' + '  '.join('%d: %.1f' % (k, kl[tile_i == k].mean()) for k in range(16)))
say('spread inside an XCD (max - min), mean over XCDs: %.1f us; between XCD means: %.1f us'
    % (np.mean([kl[xcd == k].max() - kl[xcd == k].min() for k in range(8)]),
       max(kl[xcd == k].mean() for k in range(8)) - min(kl[xcd == k].mean() for k in range(8))))
text = '\n'.join(lines) + '\n'
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], 'w').write(text)
