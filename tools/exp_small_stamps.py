"""Where does a K step of conv_small_kernel spend its cycles?  Runs layer3's conv2 of ONE frame (f32, M = 1176, K = 2304: 72 steps,
76 workgroups) on the diagnostic build of tools/exp_small_stamps.sh (CP360_LIB=tools/_exp/libcp360_small_stamps.so) and prints,
per wave, the mean s_memtime cycles per EVEN step of: barrier wait | trailing waves' MFMA block | load block | leading waves'
MFMA block.  Every stamp drains the LDS queue (s_waitcnt lgkmcnt(0)) and costs ~40 cycles: read the numbers as proportions."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cp_360_weakly_supervised_saliency_amd import ops, _lib

L = _lib.lib()
cin, cout, n, k = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (256, 256, 14, 3)))
w = torch.randn(cout, cin, k, k) * 0.01
conv = ops.Conv(w, None, torch.zeros(cout), 1, 1 if k == 3 else 0, True, torch.float32, 'cuda')
x = torch.randn(6, n, n, cin, device='cuda')
y = conv(x, tile_px=6464, splits=1)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
L.cp360_debug_small_stamps.argtypes = [C.c_void_p, C.c_int]
L.cp360_debug_small_stamps(buf, 1)
for _ in range(5):
    y = conv(x, tile_px=6464, splits=1)
torch.cuda.synchronize()
L.cp360_debug_small_stamps(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(8, 8)
print('wave   cycles per K step   held clock (d s_memtime / d s_memrealtime x 0.1 GHz)   us per step')
for wv in range(8):
    steps = a[wv, 7]
    if steps == 0:
        continue
    cyc, rt = a[wv, 0], a[wv, 1]
    print('%4d   %18.0f   %10.3f GHz %40.3f' % (wv, cyc / steps, 0.1 * cyc / rt, rt / 100.0 / steps))
