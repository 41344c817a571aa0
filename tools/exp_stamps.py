"""Where does a sub-step of the clip kernel spend its cycles?  Runs the ConvLSTM Conv2 shape on a library built with
`tools/exp_build.sh clip_stamps` (CP360_LIB=...) and prints, per wave, the mean s_memtime cycles of the five
intervals of a sub-step: [vmcnt wait + loop] [barrier 1] [first half: HEAD (waves 0-3) / TAIL (4-7)] [barrier 2]
[second half: TAIL / HEAD].  Each stamp costs ~40 cycles itself and drains the LDS queue (cdna_hip_programming.md 7)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cp_360_weakly_supervised_saliency_amd import ops, _lib

L = _lib.lib()
w = torch.randn(4000, 4000, 3, 3) * 0.01
conv = ops.Conv(w, None, torch.zeros(4000), 1, 1, True, torch.bfloat16, 'cuda')
x = torch.randn(24, 7, 7, 4000, device='cuda').to(torch.bfloat16)
y = conv(x)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
L.cp360_debug_stamps.argtypes = [C.c_void_p, C.c_int]
L.cp360_debug_stamps(buf, 1)
for _ in range(5):
    y = conv(x)
torch.cuda.synchronize()
L.cp360_debug_stamps(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(8, 8)
names = ['vmcnt+loop', 'barrier1', 'half1', 'barrier2', 'half2']
print('wave   ' + '  '.join('%10s' % n for n in names) + '       total   (cycles per sub-step, mean over all workgroups)')
for wv in range(8):
    n = a[wv, 7]
    if n == 0:
        continue
    v = a[wv, :5] / n
    print('%4d   ' % wv + '  '.join('%10.0f' % t for t in v) + '  %10.0f' % v.sum())
