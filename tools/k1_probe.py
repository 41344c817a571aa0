"""K1 (equi2cube, 64 frames 1024x2048 u8 -> padded f16 faces) alone, a few launches: for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
passes with the CP360_E2C_FU / CP360_E2C_CAP switches.  Prints the HIP-event time per launch.

Under the profiler put the interpreter itself after ``--`` (never ``env``, a shell or this file as an executable: the profiler's
preloaded library has initialised the GPU by then and such a hop is an exec the pool forbids):

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/k1_fetch -- python3 $R/tools/k1_probe.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp_360_weakly_supervised_saliency_amd.utils.equi_to_cube import Equi2Cube
from cp_360_weakly_supervised_saliency_amd.utils import synth
F = int(os.environ.get('K1_FRAMES', '64'))
frames = torch.stack([torch.from_numpy(synth.frame_u8(3 + i % 4, 1024, 2048)) for i in range(F)]).cuda()
e2c = Equi2Cube(224, (1024, 2048))
for _ in range(2):
    y = e2c.to_cube_batch(frames, out_dtype=torch.float16, layout='nhwc4p3')
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    y = e2c.to_cube_batch(frames, out_dtype=torch.float16, layout='nhwc4p3')
b.record(); torch.cuda.synchronize()
print('K1 FU=%s CAP=%s: %.1f us per launch' % (os.environ.get('CP360_E2C_FU', '4'), os.environ.get('CP360_E2C_CAP', '64'), a.elapsed_time(b) * 200))
