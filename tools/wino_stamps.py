"""In-kernel timeline of the Winograd-domain GEMM (diagnostic build -DWINO_STAMPS of csrc/wino.hip, see tools/wino_stamps.sh):
s_memtime stamps of waves 0 (leading group) and 4 (lagging group) of every workgroup around the phases of 8 consecutive sub-steps.
Prints the median cycles per phase.  Stamp k: 0 loop top, 1 after the DMA wait (vmcnt), 2 after barrier 1, 3 after the first phase
(HEAD for wave 0, TAIL for wave 4), 4 after barrier 2, 5 after the second phase.
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cp_360_weakly_supervised_saliency_amd import _lib

L = _lib.lib()
dt = torch.bfloat16
nsub, nt, mt, cout = 125, 16, 1, 4000
U = (torch.randn(16 * nt * nsub * 256 * 32, device='cuda') * 0.05).to(dt)
V = torch.randn(16 * nsub * 384 * 32, device='cuda').to(dt)
M = torch.zeros(16 * 384 * cout, device='cuda')
for _ in range(int(os.environ.get('ITERS', 6))):
    _lib.check(L.cp360_wino_gemm_raw(_lib.BF16, _lib.ptr(U), _lib.ptr(V), _lib.ptr(M), nsub, nt, mt, cout, cout, _lib.stream()))
torch.cuda.synchronize()
buf = np.zeros(256 * 2 * 8 * 6, dtype=np.uint64)
L.cp360_wino_stamps_read.argtypes = [C.c_void_p]
assert L.cp360_wino_stamps_read(buf.ctypes.data) == 0
s = buf.reshape(256, 2, 8, 6).astype(np.int64)
names = ['DMA wait (vmcnt)', 'barrier 1', 'phase A (HEAD lead / TAIL lag)', 'barrier 2', 'phase B (TAIL lead / HEAD lag)']
for g, label in ((0, 'wave 0 (leading)'), (1, 'wave 4 (lagging)')):
    d = np.diff(s[:, g], axis=2)                       # [wg, substep, 5]
    step = s[:, g, 1:, 0] - s[:, g, :-1, 0]            # loop top to loop top
    print('%s: sub-step median %d cycles (p10 %d, p90 %d)' % (label, np.median(step), np.percentile(step, 10), np.percentile(step, 90)))
    for k, n in enumerate(names):
        print('   %-34s median %5d  p90 %5d' % (n, np.median(d[..., k]), np.percentile(d[..., k], 90)))
