"""In-kernel timeline of l1block_kernel (diagnostic build -DL1_STAMPS of csrc/l1block.hip, see tools/l1_stamps.sh): s_memtime stamps of
wave 0 of every workgroup at the phase boundaries of layer1's second block (identity residual + chained conv1: the launch that runs
last with this kernel in a static pass).  Prints the median length of every phase in s_memtime ticks and as a share of a workgroup's life.
"""
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
L.cp360_l1_stamps_read.argtypes = [C.c_void_p]
assert L.cp360_l1_stamps_read(buf.ctypes.data) == 0
s = buf.reshape(8192, 16)[:5376].astype(np.int64)
names = ['patch gather + wait + barrier', 'conv2 (9 taps)', 'barrier + fragment DMA + stage 2 + first residual wait'] + \
        ['stage 3 pass %d' % p for p in range(8)] + ['chained conv1 epilogue']
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
life = s[:, 12] - s[:, 0]
print('workgroups %d; life median %d ticks (p10 %d, p90 %d)' % (len(s), np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
span = s[:, 12].max() - s[:, 0].min()
print('first start to last end: %d ticks = %.1f workgroup lives (5376 workgroups on 512 slots = 10.5 rounds)' % (span, span / np.median(life)))
for k, n in enumerate(names):
    d = s[:, idx[k + 1]] - s[:, idx[k]]
    print('   %-56s median %6d  p90 %6d  %5.1f %%' % (n, np.median(d), np.percentile(d, 90), 100.0 * np.median(d) / np.median(life)))
