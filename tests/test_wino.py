"""The Winograd F(2x2, 3x3) form of CubePad(1) + 3x3 convolution (csrc/wino.hip, include/cp360.h "K5w") - the 16-bit ConvLSTM
convolutions of /root/reference/model/clstm.py:56-64 - through the C ABI against torch-CPU on the oracle's CubePad:
pack (U = G g G^T), input transform, the 16 GEMMs, output transform (+ bias, ReLU) and the gate epilogue (clstm.py:68-80).

Tolerances: U and V are rounded once to the 16-bit type on top of the direct form's operand rounding; the bound is still the
direct kernels' (tests/test_gpu_parity.py _TOL), plus an RMS bound that a wrong tile / position would break at once.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from oracle import o_resnet
from cp_360_weakly_supervised_saliency_amd import _lib, ops
from cp_360_weakly_supervised_saliency_amd.utils import hashrng

DEV = 'cuda'
_TDT = {'bf16': torch.bfloat16, 'fp16': torch.float16}
_TOL = {'bf16': 1.2e-2, 'fp16': 1.5e-3}        # max |d| / max |want|: the direct kernels' bound (measured 5e-3 / 7e-4)
_RMS = {'bf16': 7e-3, 'fp16': 9e-4}            # rms(d) / rms(want) (measured 4.6e-3 / 5.7e-4)


def _rb(a, dt):
    return torch.from_numpy(a).to(dt).float()


def _ref(x, w, bias, dt, relu):
    y = Fn.conv2d(o_resnet.cubepad_t(_rb(x, dt), 1), _rb(w, dt), torch.from_numpy(bias))
    return (Fn.relu(y) if relu else y).numpy()


def _errs(got, want):
    d = got - want
    return float(np.max(np.abs(d)) / np.max(np.abs(want))), float(np.sqrt(np.mean(d * d)) / np.sqrt(np.mean(want * want)))


# ------------------------------------------------------------------ host-side planner and sizes (no GPU)
def test_wino_planner_and_sizes():
    L = _lib.lib()
    d = _lib.WinoDesc()
    d.dtype, d.c_in, d.pix_stride, d.c_out = _lib.BF16, 4000, 4000, 4000
    want = {(6, 7): 0, (12, 7): 0, (18, 7): 1, (24, 7): 1, (30, 7): 0, (48, 7): 1, (6, 16): 1, (6, 8): 0, (24, 8): 1}
    import os
    if os.environ.get('CP360_WINO') is None:
        for (n, f), p in want.items():
            d.n_img, d.face = n, f
            assert L.cp360_wino_preferred(C.byref(d)) == p, (n, f)
    d.n_img, d.face = 24, 7
    assert L.cp360_wino_packed_bytes(C.byref(d)) == 16 * 16 * 125 * 256 * 64          # 16 positions x 16 channel tiles x 125 sub-steps
    assert L.cp360_wino_v_bytes(C.byref(d)) == 16 * 125 * 384 * 64
    assert L.cp360_wino_m_bytes(C.byref(d)) == 16 * 384 * 4000 * 4
    d.n_img = 30                                                                       # 480 tiles: two 384-row blocks
    assert L.cp360_wino_v_bytes(C.byref(d)) == 16 * 125 * 768 * 64
    # errors: f32, n_img not 6n, misaligned channels
    d.dtype = _lib.F32
    assert L.cp360_wino_packed_bytes(C.byref(d)) == 0 and L.cp360_wino_preferred(C.byref(d)) == 0
    d.dtype, d.n_img = _lib.BF16, 7
    assert L.cp360_wino_gemm(C.byref(d), None, None, None, None) == -2
    d.n_img, d.c_in = 6, 12
    assert L.cp360_wino_gemm(C.byref(d), None, None, None, None) == -6


def test_wino_entry_points_validate_without_gpu():
    """Argument validation of every cp360_wino_* entry point happens before any launch (error codes on the CPU box)."""
    L = _lib.lib()
    one = C.c_void_p(16)
    d = _lib.WinoDesc()
    d.dtype, d.n_img, d.face, d.c_in, d.pix_stride, d.c_out, d.ld_out, d.out_coff, d.relu = _lib.BF16, 6, 7, 64, 64, 64, 0, 0, 1
    assert L.cp360_wino_input(C.byref(d), None, one, None) == -5 and L.cp360_wino_input(C.byref(d), one, None, None) == -5
    assert L.cp360_wino_gemm(C.byref(d), one, one, None, None) == -5
    assert L.cp360_wino_output(C.byref(d), None, None, one, None) == -5
    assert L.cp360_wino_output_input(C.byref(d), one, None, None, None) == -5
    assert L.cp360_wino_pack_weights(C.byref(d), None, one, None) == -5
    assert L.cp360_wino_output_gates(C.byref(d), one, None, one, one, one, 128, 64, None, None, None, 0, 0, None) == -5   # bias is required
    d.c_out = 72                                                             # the gate epilogue splits c_out into 4 gates of Hc % 4 == 0
    assert L.cp360_wino_output_gates(C.byref(d), one, one, one, one, one, 128, 64, None, None, None, 0, 0, None) == -6
    d.c_out = 64
    assert L.cp360_wino_output_gates(C.byref(d), one, one, one, one, one, 64, 56, None, None, None, 0, 0, None) == -1     # h_coff + Hc > ld_h
    assert L.cp360_wino_output_gates(C.byref(d), one, one, one, one, one, 128, 64, None, one, None, 0, 0, None) == -1     # x_next without minmax
    d.pix_stride = 32
    assert L.cp360_wino_input(C.byref(d), one, one, None) == -1                                                            # pixel stride < c_in
    d.pix_stride, d.ld_out = 64, 32
    assert L.cp360_wino_output(C.byref(d), one, None, one, None) == -1                                                     # ld_out < c_out
    d.ld_out, d.face = 0, 1
    assert L.cp360_wino_packed_bytes(C.byref(d)) == 0                                                                      # faces of at least 2 x 2
    d.face = 64
    assert L.cp360_wino_input(C.byref(d), one, one, None) == -8 and L.cp360_wino_output_input(C.byref(d), one, None, one, None) == -8
    # the cell context: queries on a context without a loaded cell, and the f32 cell
    assert L.cp360_clstm_wino_state(None, 4, 7) == 0 and L.cp360_clstm_load_wino(None, one, one, one, None) == -5


# ------------------------------------------------------------------ the convolution
@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('n,n_img', [(7, 24), (7, 30), (8, 12), (16, 6), (5, 6), (4, 12), (3, 6), (2, 6)])
def test_wino_conv_matches_torch_cpu(n, n_img, prec):
    """c_in = 104: the K tail ends inside a 64-byte sub-step; c_out = 264: a ragged second channel tile; 30 faces of 7x7 = 480
    tiles: two 384-row tile blocks, the second ragged; odd faces: the last tile row / column hangs over the face."""
    dt = _TDT[prec]
    cin, cout = 104, 264
    x = hashrng.normal(7400 + n, (n_img, cin, n, n))
    w = hashrng.normal(7401, (cout, cin, 3, 3), 0, (2.0 / (9 * cin)) ** 0.5)
    bias = hashrng.normal(7403, (cout,), 0, 0.1)
    want = _ref(x, w, bias, dt, True)
    conv = ops.WinoConv(torch.from_numpy(w), torch.from_numpy(bias), True, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    got = ops.nhwc_to_nchw(conv(xt), out_dtype=torch.float32).cpu().numpy()
    mx, rms = _errs(got, want)
    print('wino %s n=%d n_img=%d: max %.2e rms %.2e' % (prec, n, n_img, mx, rms))
    assert mx <= _TOL[prec] and rms <= _RMS[prec], (mx, rms)


@pytest.mark.gpu
def test_wino_conv_strided_pixels_and_output_offset():
    """The ConvLSTM's layouts: the input is the first c_in channels of a wider pixel (the fused [x | h] buffer), the output goes
    to a channel offset of a wider pixel; no ReLU, no bias."""
    dt = torch.bfloat16
    n, n_img, cin, cout, ld_in, ld_out, coff = 7, 24, 96, 256, 160, 320, 40
    x = hashrng.normal(7500, (n_img, ld_in, n, n))
    w = hashrng.normal(7501, (cout, cin, 3, 3), 0, (2.0 / (9 * cin)) ** 0.5)
    want = _ref(x[:, :cin], w, np.zeros(cout, np.float32), dt, False)
    conv = ops.WinoConv(torch.from_numpy(w), None, False, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    out = torch.full((n_img, n, n, ld_out), 7.0, dtype=dt, device=DEV)
    conv(xt, out=out, out_coff=coff)
    o = out.float().cpu().numpy()
    assert np.all(o[..., :coff] == 7.0) and np.all(o[..., coff + cout:] == 7.0)      # nothing outside its channels
    got = o[..., coff:coff + cout].transpose(0, 3, 1, 2)
    mx, rms = _errs(got, want)
    assert mx <= _TOL['bf16'] and rms <= _RMS['bf16'], (mx, rms)


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('n', [7, 16])
def test_wino_gates_matches_torch_cpu(n, prec):
    """Gates convolution + cell update in the output transform (clstm.py:68-80) with the next frame's window normalisation
    written into the x half, against torch-CPU gate arithmetic on the torch-CPU convolution of the same rounded operands."""
    dt = _TDT[prec]
    B, Hc, cin = (4 if n == 7 else 1), 16, 64
    n_img, P, M = 6 * B, 6 * n * n, 6 * B * n * n
    a2 = hashrng.normal(7600 + n, (n_img, cin, n, n))
    w = hashrng.normal(7601, (4 * Hc, cin, 3, 3), 0, (2.0 / (9 * cin)) ** 0.5)
    bias = hashrng.normal(7602, (4 * Hc,), 0, 0.2)
    cprev = hashrng.normal(7603, (n_img, Hc, n, n))
    cam = hashrng.normal(7604, (B, 2, P, Hc), 5.0, 2.0)                     # two frames per clip, pixel-major
    mm = np.stack([cam.reshape(B, -1).min(1), cam.reshape(B, -1).max(1)], 1).astype(np.float32)
    g = _ref(a2, w, bias, dt, False)
    i_g, f_g, o_g, c_g = [torch.from_numpy(t) for t in np.split(g, 4, 1)]
    cn = torch.sigmoid(f_g) * torch.from_numpy(cprev) + torch.sigmoid(i_g) * torch.tanh(c_g)
    hn = torch.sigmoid(o_g) * torch.tanh(cn)
    conv = ops.WinoConv(torch.from_numpy(w), None, False, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(a2).to(DEV), out_dtype=dt)
    cp = ops.nchw_to_nhwc(torch.from_numpy(cprev).to(DEV))
    c_next = torch.empty_like(cp)
    h_f32 = torch.empty_like(cp)
    xh = torch.zeros((n_img, n, n, 2 * Hc), dtype=dt, device=DEV)
    camt, mmt = torch.from_numpy(cam).to(DEV), torch.from_numpy(mm).to(DEV)
    conv.gates(xt, torch.from_numpy(bias).to(DEV), cp, c_next, xh, Hc, h_f32, x_next=(camt, mmt, 0, P, 2 * P * Hc, 1))
    got_c = ops.nhwc_to_nchw(c_next).cpu().numpy()
    got_h = ops.nhwc_to_nchw(h_f32).cpu().numpy()
    tol = 2.5e-2 if prec == 'bf16' else 3e-3
    assert np.max(np.abs(got_c - cn.numpy())) <= tol and np.max(np.abs(got_h - hn.numpy())) <= tol
    xhc = xh.float().cpu().numpy()
    assert np.max(np.abs(xhc[..., Hc:].transpose(0, 3, 1, 2) - got_h)) <= (8e-3 if prec == 'bf16' else 1e-3)   # h half = rounded h
    xn = (cam[:, 1] - mm[:, :1, None]) / (mm[:, 1:, None] - mm[:, :1, None])                                    # [B, P, Hc]
    assert np.max(np.abs(xhc[..., :Hc].reshape(B, P, Hc) - xn)) <= (4e-3 if prec == 'bf16' else 5e-4)


# ------------------------------------------------------------------ the cell in the Winograd domain
@pytest.mark.gpu
@pytest.mark.parametrize('prec,w,B', [('bf16', 7, 4), ('bf16', 8, 4), ('fp16', 16, 1)])
def test_wino_cell_window_matches_oracle(prec, w, B):
    """A whole window (test_temporal.py:63-85: min / max, hidden = cell = frame 0, T updates, cube -> equi, channel max) for the
    launch shapes the planner runs in the Winograd domain - 4 clips of 7x7 faces (BASELINE C4's shard), 4 clips of 8x8 faces
    (cube 256), one clip of 16x16 faces (C5's shard) - against the oracle's window on every clip."""
    from oracle import o_clstm
    from cp_360_weakly_supervised_saliency_amd.model.clstm import ConvLSTMCell
    from cp_360_weakly_supervised_saliency_amd.temporal_model.test_temporal import ClipRunner
    from cp_360_weakly_supervised_saliency_amd.utils.cube_to_equi import Cube2Equi
    from cp_360_weakly_supervised_saliency_amd.utils import synth
    sd = synth.clstm_state(seed=2)
    cell = ConvLSTMCell(1000, 1000, precision=prec)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    cell = cell.to(DEV).eval()
    assert cell.uses_winograd(6 * B, w)
    T, P = 3, 6 * w * w
    cams = [synth.cam_clip(7700 + 10 * w + b, T, w=w) for b in range(B)]
    pack = lambda f: np.ascontiguousarray(f.transpose(0, 1, 3, 4, 2)).reshape(T, P, 1000)
    sal = ClipRunner(cell, Cube2Equi(w), B, T, w).run(torch.from_numpy(np.stack([pack(f) for f in cams])).to(DEV)).cpu().numpy()
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    tol = 3e-3 if prec == 'bf16' else 6e-4                      # measured 1.1e-3 / 2.0e-4
    for b in range(B):
        want = o_clstm.window_saliency(cams[b], sdt)
        err = float(np.max(np.abs(sal[b] - want)))
        print('wino cell %s w=%d clip %d: map max|d| %.2e (range %.3f..%.3f)' % (prec, w, b, err, want.min(), want.max()))
        assert err <= tol, (b, err)


@pytest.mark.gpu
def test_clstm_wino_lazy_load_through_the_c_abi():
    """INTEGRATION.md's two-line binder change with raw ctypes: cp360_clstm_wino_state says 2 for a shape that would run in the
    Winograd domain, the step runs on the direct kernels until cp360_clstm_load_wino has packed U, then in the Winograd domain
    (state 1, another workspace size) - the two results agree within the 16-bit rounding; one clip stays direct (state 0)."""
    L = _lib.lib()
    dt, code = torch.bfloat16, _lib.BF16
    Cin = Hc = 256
    B, w = 4, 7
    n6, c4 = 6 * B, 4 * Hc
    g = lambda seed, shape, std: torch.from_numpy(hashrng.normal(seed, shape, 0, std)).to(DEV)
    w1, w2, wg = g(7801, (c4, Cin + Hc, 3, 3), (2.0 / (9 * c4)) ** 0.5), g(7802, (c4, c4, 3, 3), (2.0 / (9 * c4)) ** 0.5), \
        g(7803, (c4, c4, 3, 3), (2.0 / (9 * c4)) ** 0.5)
    b1, b2, bg = g(7804, (c4,), 0.05), g(7805, (c4,), 0.05), g(7806, (c4,), 0.05)
    h = C.c_void_p()
    _lib.check(L.cp360_create(torch.cuda.current_device(), C.byref(h)))
    try:
        p = _lib.ptr
        _lib.check(L.cp360_clstm_load(h, code, p(w1), p(b1), p(w2), p(b2), p(wg), p(bg), Cin, Hc, w, _lib.stream()))
        assert L.cp360_clstm_wino_state(h, 1, w) == 0 and L.cp360_clstm_wino_state(h, B, w) == 2
        outs, sizes = [], []
        for phase in range(2):
            if phase == 1:
                _lib.check(L.cp360_clstm_load_wino(h, p(w1), p(w2), p(wg), _lib.stream()))
                assert L.cp360_clstm_wino_state(h, B, w) == 1
            nb = L.cp360_clstm_workspace_bytes(h, B, w)
            sizes.append(nb)
            ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
            xh = torch.from_numpy(hashrng.uniform(7810, (n6, w, w, Cin + Hc), 0.0, 1.0)).to(DEV).to(dt)
            c0 = torch.from_numpy(hashrng.uniform(7811, (n6, w, w, Hc), 0.0, 1.0)).to(DEV)
            c1, hf = torch.empty_like(c0), torch.empty_like(c0)
            _lib.check(L.cp360_clstm_step(h, p(xh), p(c0), p(c1), p(hf), B, w, None, None, 0, p(ws), nb, _lib.stream()))
            torch.cuda.synchronize()
            outs.append((c1.cpu().numpy(), hf.cpu().numpy()))
        assert min(sizes) > 0                                      # (V + M workspaces replace the split-K slabs: query again after the load)
        for a, b in zip(outs[0], outs[1]):
            assert np.all(np.isfinite(a)) and np.max(np.abs(a - b)) <= 2e-2 and not np.array_equal(a, b)
    finally:
        L.cp360_destroy(h)


@pytest.mark.gpu
def test_clstm_wino_step_with_input_wider_than_three_hidden():
    """cp360_clstm_step in the Winograd domain when Conv1's K (input + hidden = 1280) exceeds Conv2's (4 * hidden = 1024): the
    shared V workspace has to hold Conv1's transformed input (round-5 advisor finding: it was sized from Conv2 alone, so Conv1's V
    ran into M).  The Winograd step against the direct step of the same context, and the reported workspace is large enough for
    Conv1's V + M.  A cell whose Conv1 is not a Winograd shape (input + hidden not a multiple of 8) reports state 0."""
    L = _lib.lib()
    dt, code = torch.bfloat16, _lib.BF16
    Cin, Hc = 1024, 256
    B, w = 4, 7
    n6, c4, cx = 6 * B, 4 * Hc, Cin + Hc
    g = lambda seed, shape, std: torch.from_numpy(hashrng.normal(seed, shape, 0, std)).to(DEV)
    w1, w2, wg = g(7821, (c4, cx, 3, 3), (2.0 / (9 * c4)) ** 0.5), g(7822, (c4, c4, 3, 3), (2.0 / (9 * c4)) ** 0.5), \
        g(7823, (c4, c4, 3, 3), (2.0 / (9 * c4)) ** 0.5)
    b1, b2, bg = g(7824, (c4,), 0.05), g(7825, (c4,), 0.05), g(7826, (c4,), 0.05)
    h = C.c_void_p()
    _lib.check(L.cp360_create(torch.cuda.current_device(), C.byref(h)))
    try:
        p = _lib.ptr
        _lib.check(L.cp360_clstm_load(h, code, p(w1), p(b1), p(w2), p(b2), p(wg), p(bg), Cin, Hc, w, _lib.stream()))
        assert L.cp360_clstm_wino_state(h, B, w) == 2
        outs = []
        for phase in range(2):
            if phase == 1:
                _lib.check(L.cp360_clstm_load_wino(h, p(w1), p(w2), p(wg), _lib.stream()))
                assert L.cp360_clstm_wino_state(h, B, w) == 1
            nb = L.cp360_clstm_workspace_bytes(h, B, w)
            if phase == 1:
                d1 = _lib.WinoDesc(code, n6, w, cx, cx, c4, c4, 0, 1)
                assert nb >= L.cp360_wino_v_bytes(C.byref(d1)) + L.cp360_wino_m_bytes(C.byref(d1))
            ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
            xh = torch.from_numpy(hashrng.uniform(7830, (n6, w, w, cx), 0.0, 1.0)).to(DEV).to(dt)
            c0 = torch.from_numpy(hashrng.uniform(7831, (n6, w, w, Hc), 0.0, 1.0)).to(DEV)
            c1, hf = torch.empty_like(c0), torch.empty_like(c0)
            _lib.check(L.cp360_clstm_step(h, p(xh), p(c0), p(c1), p(hf), B, w, None, None, 0, p(ws), nb, _lib.stream()))
            torch.cuda.synchronize()
            outs.append((c1.cpu().numpy(), hf.cpu().numpy()))
        for a, b in zip(outs[0], outs[1]):
            assert np.all(np.isfinite(a)) and np.max(np.abs(a - b)) <= 2e-2 and not np.array_equal(a, b)
    finally:
        L.cp360_destroy(h)
    # Conv1 not a Winograd shape: input + hidden = 1276 is no multiple of 8 -> the whole cell stays on the direct kernels
    h = C.c_void_p()
    _lib.check(L.cp360_create(torch.cuda.current_device(), C.byref(h)))
    try:
        Cin2 = 1020
        w1b = g(7827, (c4, Cin2 + Hc, 3, 3), (2.0 / (9 * c4)) ** 0.5)
        rc = L.cp360_clstm_load(h, code, p(w1b), p(b1), p(w2), p(b2), p(wg), p(bg), Cin2, Hc, w, _lib.stream())
        if rc == 0:
            assert L.cp360_clstm_wino_state(h, B, w) == 0
    finally:
        L.cp360_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('n,n_img', [(7, 24), (8, 12), (5, 6), (7, 30), (9, 6)])
def test_wino_fused_output_input_equals_the_two_kernels(n, n_img, prec):
    """cp360_wino_output_input (one launch between two convolutions of the ConvLSTM) against cp360_wino_output followed by
    cp360_wino_input: the same V, bit for bit, on every valid tile row - c_out = 264 ends inside a 32-channel block (the
    next V's zero padding), 30 faces span two 384-row tile blocks; 9x9 is the largest face it takes."""
    dt = _TDT[prec]
    cin, cmid, cout = 40, 264, 64
    x = hashrng.normal(7900 + n, (n_img, cin, n, n))
    w1 = hashrng.normal(7901, (cmid, cin, 3, 3), 0, (2.0 / (9 * cin)) ** 0.5)
    b1 = hashrng.normal(7902, (cmid,), 0, 0.1)
    w2 = hashrng.normal(7903, (cout, cmid, 3, 3), 0, (2.0 / (9 * cmid)) ** 0.5)
    c1 = ops.WinoConv(torch.from_numpy(w1), torch.from_numpy(b1), True, dt, DEV)
    c2 = ops.WinoConv(torch.from_numpy(w2), None, False, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    m, d = c1.sums(xt)
    m = m.clone()                                                      # (the M workspace is shared; keep this convolution's sums)
    th = (n + 1) // 2
    tiles, nsub = n_img * th * th, (cmid + 31) // 32
    m_pad = -(-tiles // 384) * 384
    view = lambda v: v[: 16 * nsub * m_pad * 64].view(dt).view(16, nsub, m_pad, 32)[:, :, :tiles].clone()
    a1 = c1.output(m, d)
    v_two, _ = c2.input(a1)
    want = view(v_two)
    v_two.zero_()
    v_one, d2 = c1.output_input(m, c1.desc(n_img, n), c2)
    got = view(v_one)
    assert d2.c_in == cmid and torch.equal(got, want)
    assert float(want.float().abs().max()) > 0.1 and bool((want[:, -1, :, 8:] == 0).all())     # channels 264 .. 287 of the last block: zeros
    # and the convolution that follows gives the torch-CPU result of conv2(relu(conv1(x)))
    y = ops.nhwc_to_nchw(c2.output(c2.gemm(v_one, d2), d2), out_dtype=torch.float32).cpu().numpy()
    mid = _ref(x, w1, b1, dt, True)
    ref = _ref(_rb(mid, dt).numpy(), w2, np.zeros(cout, np.float32), dt, False)
    mx, rms = _errs(y, ref)
    assert mx <= 2 * _TOL[prec] and rms <= 2 * _RMS[prec], (mx, rms)


@pytest.mark.gpu
@pytest.mark.parametrize('n,n_img', [(7, 24), (7, 30), (5, 6), (9, 6), (3, 6)])
def test_wino_never_read_rows_of_m_reach_no_output(n, n_img):
    """Odd faces: the GEMM does not store transform-domain row 3 / column 3 of the tiles in a face's last tile row / column
    (csrc/wino.hip, wino_dead_row / wino_dead_col) and no output transform may read them.  The M workspace is filled with NaN
    before the GEMM: the rows it skips keep the NaN, and every consumer of M - the plain output transform, the fused output +
    next-input transform, the gate epilogue - must still give finite results equal to a run on a zero-filled workspace, bit for
    bit; the GEMM must really have skipped rows (NaN left in M), and exactly the (tile, position) pairs the rule names."""
    dt = torch.bfloat16
    cin, cout, Hc = 40, 64, 16
    x = hashrng.normal(8100 + n, (n_img, cin, n, n))
    w = hashrng.normal(8101, (cout, cin, 3, 3), 0, (2.0 / (9 * cin)) ** 0.5)
    b = hashrng.normal(8102, (cout,), 0, 0.1)
    conv = ops.WinoConv(torch.from_numpy(w), torch.from_numpy(b), True, dt, DEV)
    nxt = ops.WinoConv(torch.zeros(32, cout, 3, 3), None, False, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    cp = torch.rand((n_img, n, n, Hc), device=DEV)
    bias_g = torch.from_numpy(b).to(DEV)

    def run(fill):
        v, d = conv.input(xt)
        _, m = conv.workspace(d)
        m.fill_(fill)
        conv.gemm(v, d)
        mm = m.clone()
        out = conv.output(mm, d).clone()
        c_next, h32 = torch.empty_like(cp), torch.empty_like(cp)
        xh = torch.zeros((n_img, n, n, 2 * Hc), dtype=dt, device=DEV)
        conv.gates_from(mm, d, bias_g, cp, c_next, xh, Hc, h32)
        vn = conv.output_input(mm, conv.desc(n_img, n), nxt)
        vn = None if vn is None else vn[0].clone()
        return mm, out, c_next, h32, vn

    m_nan, *got = run(float('nan'))
    m_zero, *want = run(0.0)
    th = (n + 1) // 2
    tiles = n_img * th * th
    m_pad = -(-tiles // 384) * 384
    mv = m_nan[: 16 * m_pad * cout].view(16, m_pad, cout)[:, :tiles]
    dead = torch.isnan(mv).all(dim=2).cpu().numpy()                       # [pos, tile]: the row was not stored
    assert not torch.isnan(mv).any(dim=2).cpu().numpy()[~dead].any()      # a row is skipped whole or not at all
    t = np.arange(tiles) % (th * th)
    ty, tx = t // th, t % th
    pos = np.arange(16)[:, None]
    rule = ((pos >> 2) == 3) & (ty == th - 1)[None] | ((pos & 3) == 3) & (tx == th - 1)[None]
    assert np.array_equal(dead, rule) and dead.sum() == n_img * (2 * 4 * th - 1)   # per face: th tiles x 4 positions, twice, minus the corner's overlap
    for g, wnt in zip(got, want):
        if g is None:
            assert wnt is None
            continue
        assert bool(torch.isfinite(g.float()).all()) and torch.equal(g, wnt)


@pytest.mark.gpu
def test_wino_fused_output_input_refuses_large_faces():
    c1 = ops.WinoConv(torch.zeros(32, 32, 3, 3), None, False, torch.float16, DEV)
    xt = torch.zeros((6, 16, 16, 32), dtype=torch.float16, device=DEV)           # a cube's 32-channel image: 98 KB of LDS
    m, d = c1.sums(xt)
    assert c1.output_input(m, d, c1) is None
