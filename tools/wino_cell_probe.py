"""The seven launches of one ConvLSTM cell update in the Winograd domain (csrc/wino.hip), on random operands, through the C ABI:

    wino_in(xh) -> GEMM K=2000 -> out_in -> GEMM K=4000 -> out_in -> GEMM K=4000 -> gates

Times the whole sequence with HIP events (back to back, --iters sequences per sample); run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split, and with CP360_LIB=<variant .so> for same-box A/B of a kernel.

    python3 tools/wino_cell_probe.py [--clips 4] [--face 7] [--iters 20]
"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp_360_weakly_supervised_saliency_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='bf16')
ap.add_argument('--clips', type=int, default=4)
ap.add_argument('--face', type=int, default=7)
ap.add_argument('--hidden', type=int, default=1000)
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
dt = {'bf16': torch.bfloat16, 'fp16': torch.float16}[args.precision]
L, dev = _lib.lib(), 'cuda'
n6, w, Hc = 6 * args.clips, args.face, args.hidden
cx, c4 = 2 * Hc, 4 * Hc
code = _lib.dtype_code(dt)


def desc(c_in, c_out, relu):
    d = _lib.WinoDesc()
    d.dtype, d.n_img, d.face, d.c_in, d.pix_stride, d.c_out, d.ld_out, d.out_coff, d.relu = code, n6, w, c_in, c_in, c_out, c_out, 0, relu
    return d


d1, d2, dg = desc(cx, c4, 1), desc(c4, c4, 1), desc(c4, c4, 0)
torch.manual_seed(0)
P = n6 * w * w
xh = torch.rand(P, cx, device=dev).to(dt)
us = []
for d in (d1, d2, dg):
    wt = torch.randn(d.c_out, d.c_in, 3, 3, device=dev) * (2.0 / (9 * d.c_out)) ** 0.5
    u = torch.empty(L.cp360_wino_packed_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    _lib.check(L.cp360_wino_pack_weights(C.byref(d), _lib.ptr(wt), _lib.ptr(u), _lib.stream()))
    us.append(u)
    del wt
v = torch.zeros(L.cp360_wino_v_bytes(C.byref(d2)), dtype=torch.uint8, device=dev)
m = torch.zeros(L.cp360_wino_m_bytes(C.byref(d2)) // 4, dtype=torch.float32, device=dev)
b1, b2, bg = (torch.randn(c4, device=dev) * 0.01 for _ in range(3))
c_prev, c_next, h32 = torch.rand(P, Hc, device=dev), torch.empty(P, Hc, device=dev), torch.empty(P, Hc, device=dev)
p, s = _lib.ptr, _lib.stream


def cell():
    _lib.check(L.cp360_wino_input(C.byref(d1), p(xh), p(v), s()))
    _lib.check(L.cp360_wino_gemm(C.byref(d1), p(v), p(us[0]), p(m), s()))
    _lib.check(L.cp360_wino_output_input(C.byref(d1), p(m), p(b1), p(v), s()))
    _lib.check(L.cp360_wino_gemm(C.byref(d2), p(v), p(us[1]), p(m), s()))
    _lib.check(L.cp360_wino_output_input(C.byref(d2), p(m), p(b2), p(v), s()))
    _lib.check(L.cp360_wino_gemm(C.byref(dg), p(v), p(us[2]), p(m), s()))
    _lib.check(L.cp360_wino_output_gates(C.byref(dg), p(m), p(bg), p(c_prev), p(c_next), p(xh), cx, Hc, p(h32), None, None, 0, 0, s()))


cell()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
times = []
for rep in range(5):
    a.record()
    for _ in range(args.iters):
        cell()
    b.record()
    torch.cuda.synchronize()
    times.append(a.elapsed_time(b) / args.iters * 1e3)
times.sort()
print('wino cell  %d clips %dx%d faces: min %.1f median %.1f max %.1f us per cell update   (checksum %.6e)'
      % (args.clips, w, w, times[0], times[2], times[-1], float(h32.double().sum())), flush=True)
