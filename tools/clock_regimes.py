"""Which shader clock does the chip hold in the regimes bench.py runs in?  cp360_clock_probe (a bare bf16 MFMA loop stamped with
s_memtime / s_memrealtime, csrc/misc.hip) as (a) long back-to-back launches on every CU - the dense regime of the 4 x 16
headline - and (b) launches of ~30 us on a part of the CUs separated by idle gaps - the regime of one frame (BASELINE C2).

    python3 tools/clock_regimes.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cp_360_weakly_supervised_saliency_amd import _lib

L = _lib.lib()
dev = torch.device('cuda')


def probe(n_wg, iters, launches, gap_s=0.0):
    st = torch.zeros((n_wg * 4, 2), dtype=torch.int64, device=dev)
    out = []
    for _ in range(launches):
        _lib.check(L.cp360_clock_probe(_lib.ptr(st), n_wg, iters, _lib.stream()))
        if gap_s:
            torch.cuda.synchronize()
            time.sleep(gap_s)
    torch.cuda.synchronize()
    s = st.cpu().double()
    ghz = 0.1 * s[:, 0] / s[:, 1]
    us = s[:, 1] / 100.0
    return float(ghz.median()), float(us.median())


for name, n_wg, iters, launches, gap in [
        ('dense: 256 WGs, ~6 ms launches back to back', 256, 40000, 4, 0.0),
        ('256 WGs, ~30 us launches back to back', 256, 200, 200, 0.0),
        ('148 WGs, ~30 us launches back to back', 148, 200, 200, 0.0),
        ('148 WGs, ~30 us launches, 2 ms idle between', 148, 200, 20, 0.002),
        ('36 WGs, ~30 us launches back to back', 36, 200, 200, 0.0)]:
    ghz, us = probe(n_wg, iters, launches, gap)
    print('%-48s held clock %.3f GHz (median over waves of the LAST launch), loop %.1f us' % (name, ghz, us))
