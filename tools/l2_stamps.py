"""In-kernel timeline of l2block_kernel<256, 14x14> (diagnostic build -DL2_STAMPS of csrc/l2block.hip, see tools/l2_stamps.sh): s_memtime stamps of
wave 0 of every workgroup at the phase boundaries of layer3's LAST tail (the launch that runs last with this kernel in a static pass).
Prints the median length of every phase in s_memtime ticks (100 MHz: 10 ns) and as a share of a workgroup's life."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cp_360_weakly_supervised_saliency_amd import _lib
from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
from cp_360_weakly_supervised_saliency_amd.utils import synth

dev = 'cuda'
H, W, cd, B, T = 1024, 2048, 224, 4, 16
eng = SaliencyEngine(synth.resnet50_state(seed=1), synth.clstm_state(seed=2), (H, W), cd, clips=B, frames=T, precision='bf16', device=dev)
frames = torch.stack([torch.from_numpy(synth.clip_u8(3 + b, T, H, W)) for b in range(B)]).to(dev)
flat = frames.reshape((B * T,) + tuple(frames.shape[2:]))
with torch.no_grad():
    for _ in range(3):
        eng.static_stage(flat)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(8192 * 16, dtype=np.uint64)
L.cp360_l2_stamps_read.argtypes = [C.c_void_p]
assert L.cp360_l2_stamps_read(buf.ctypes.data) == 0
NWG = 768
s = buf.reshape(8192, 16)[:NWG].astype(np.int64)
names = ['patch gather + wait + barrier', 'conv2, channel half 0 (36 steps)', 'conv2, channel half 1', 'stage 2 + first fragments / residual + barrier'] + \
        ['stage 3 pass %d (+ residual, store)' % p for p in range(8)] + ['end']
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15]
life = s[:, 15] - s[:, 0]
tick = 'ticks'
print('workgroups %d; life median %d %s (p10 %d, p90 %d)' % (len(s), np.median(life), tick, np.percentile(life, 10), np.percentile(life, 90)))
span = s[:, 15].max() - s[:, 0].min()
print('first start to last end: %d ticks = %.2f workgroup lives (768 workgroups on 512 slots = 1.5 rounds)' % (span, span / np.median(life)))
for k, n in enumerate(names):
    d = s[:, idx[k + 1]] - s[:, idx[k]]
    print('   %-56s median %6d  p10 %6d  p90 %6d  %5.1f %%' % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90), 100.0 * np.median(d) / np.median(life)))
# how many workgroups are alive over time (first round vs tail)
t0 = s[:, 0].min()
ends = np.sort(s[:, 15] - t0)
starts = np.sort(s[:, 0] - t0)
for frac in (0.25, 0.5, 0.75, 0.9, 1.0):
    t = int(span * frac)
    alive = int((starts <= t).sum() - (ends <= t).sum())
    print('   at %3d %% of the launch: %4d workgroups started, %4d finished, %4d alive' % (int(frac * 100), int((starts <= t).sum()), int((ends <= t).sum()), alive))
