"""Stand-alone NCHW CubePad timings on the network's large shapes (GPU box): python tools/cubepad_bench.py <tag>.
A/B switches (read once per process by csrc/cubepad.hip): CP360_CUBEPAD_NOLDS6, CP360_CUBEPAD_NOCHANNEL,
CP360_CUBEPAD_STRIP_V1, CP360_CUBEPAD_ELEMENTWISE, CP360_CUBEPAD_NOCUBE.  TB/s = (input + output bytes) / time."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from cp_360_weakly_supervised_saliency_amd.model.cube_pad import CubePad
cases = [((384, 64, 112, 112), torch.float16, 1), ((384, 3, 224, 224), torch.float32, 3), ((384, 256, 56, 56), torch.float16, 1),
         ((96, 64, 256, 256), torch.float16, 1)]
for shp, dt, p in cases:
    x = torch.randn(shp, device='cuda').to(dt)
    m = CubePad(p)
    for _ in range(3): y = m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = m(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (x.numel() + y.numel()) * x.element_size() / 1e6
    print(sys.argv[1], shp, dt, 'p', p, '%.1f us  %.2f TB/s' % (us, mb / us))
