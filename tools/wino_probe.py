"""Round-5 gate (b): the bare Winograd-domain GEMM (16 positions x [384 tiles x c_in] x [c_in x 4000]) on the GPU box - checks
the kernel against torch on a sample of positions and times back-to-back launches (HIP events), beside the direct clip-resident
kernel on the same convolution.

    python3 tools/wino_probe.py [--precision bf16] [--cin 4000] [--iters 30]
"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp_360_weakly_supervised_saliency_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='bf16')
ap.add_argument('--cin', type=int, default=4000)
ap.add_argument('--cout', type=int, default=4000)
ap.add_argument('--mt', type=int, default=1)
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--no-check', action='store_true', help='timing only (ablation builds compute garbage)')
ap.add_argument('--zeros', action='store_true', help='all-zero operands: the same cycles at the clock the chip holds without data toggling (DVFS check)')
ap.add_argument('--zero-unneeded', action='store_true', help='zero the V rows of (position, tile) pairs whose M is never read at odd faces (7x7: last tile row / column)')
args = ap.parse_args()
dt = {'bf16': torch.bfloat16, 'fp16': torch.float16}[args.precision]
L = _lib.lib()
dev = 'cuda'
nsub = (args.cin + 31) // 32
nt = (args.cout + 255) // 256
mpad = 384 * args.mt
torch.manual_seed(0)
Ul = (torch.randn(16, nt * 256, nsub * 32, device=dev) * 0.05).to(dt)
Vl = torch.randn(16, mpad, nsub * 32, device=dev).to(dt)
if args.zeros:
    Ul.zero_()
    Vl.zero_()
if args.zero_unneeded:
    t = torch.arange(mpad, device=dev)
    for pos in range(16):
        if pos >> 2 == 3:
            Vl[pos, (t // 4) % 4 == 3] = 0
        if pos & 3 == 3:
            Vl[pos, t % 4 == 3] = 0
U = Ul.view(16, nt, 256, nsub, 32).permute(0, 1, 3, 2, 4).contiguous()
V = Vl.view(16, mpad, nsub, 32).permute(0, 2, 1, 3).contiguous()
M = torch.zeros(16, mpad, args.cout, device=dev, dtype=torch.float32)


def run():
    _lib.check(L.cp360_wino_gemm_raw(_lib.dtype_code(dt), _lib.ptr(U), _lib.ptr(V), _lib.ptr(M), nsub, nt, args.mt, args.cout,
                                     args.cout, _lib.stream()))


run()
torch.cuda.synchronize()
worst = 0.0
for pos in (0, 5, 15):
    want = Vl[pos].float() @ Ul[pos, :args.cout].float().t()
    worst = max(worst, float((M[pos] - want).abs().max() / want.abs().max()))
print('wino gemm max relative error vs torch (3 positions): %.2e' % worst)
assert args.no_check or args.zeros or worst < 1e-3
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
times = []
for rep in range(7):
    a.record()
    for _ in range(args.iters):
        run()
    b.record()
    torch.cuda.synchronize()
    times.append(a.elapsed_time(b) / args.iters * 1e3)
fl = 2.0 * 16 * mpad * args.cout * args.cin
times.sort()
print('wino gemm  M=%d N=%d K=%d x16: min %.1f median %.1f max %.1f us  (median: %.0f TFLOP/s Winograd-domain, U stream %.2f TB/s)'
      % (mpad, args.cout, args.cin, times[0], times[3], times[-1], fl / times[3] / 1e6, U.numel() * 2 / times[3] / 1e6), flush=True)
