"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel, reached through
the Python shims -> ctypes -> the C ABI of libcp360.so, against the CPU oracle and the
reference-generated golden fixtures.

Tolerances: index / copy work (CubePad, face_map) bit-exact; fp32 paths 1e-5 relative
per op and 1e-3 absolute on saliency maps / window-normalised CAM (north star); bf16
paths are checked against their own looser bounds (AUC/CC gate: tests/test_metrics.py).
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from oracle import o_cubepad, o_e2c, o_c2e, o_resnet, o_clstm
from cp_360_weakly_supervised_saliency_amd import ops
from cp_360_weakly_supervised_saliency_amd.model.cube_pad import CubePad
from cp_360_weakly_supervised_saliency_amd.model.resnet_cubic import resnet50
from cp_360_weakly_supervised_saliency_amd.model.clstm import ConvLSTMCell
from cp_360_weakly_supervised_saliency_amd.static_model.class_activation_model import CAM
from cp_360_weakly_supervised_saliency_amd.temporal_model.test_temporal import ClipRunner
from cp_360_weakly_supervised_saliency_amd.utils.equi_to_cube import Equi2Cube
from cp_360_weakly_supervised_saliency_amd.utils.cube_to_equi import Cube2Equi
from cp_360_weakly_supervised_saliency_amd.utils import hashrng, synth
from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
from tests.golden import make_golden as mg
from tests import parity_helpers as ph

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rel_err(got, want):
    return float(np.max(np.abs(got - want)) / max(float(np.max(np.abs(want))), 1e-30))


# ------------------------------------------------------------------ K2 CubePad
def test_cubepad_golden_small_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, 'cubepad_small.npz'))
    for k in range(len(mg.CUBEPAD_SMALL)):
        x, y, pad = z['x%d' % k], z['y%d' % k], [int(v) for v in z['pad%d' % k]]
        got = CubePad(pad)(torch.from_numpy(x).to(DEV)).cpu().numpy()
        assert got.shape == y.shape and np.array_equal(got, y), (k, pad)


def test_cubepad_golden_hashed_bit_exact(golden_dir):
    want = json.load(open(os.path.join(golden_dir, 'cubepad_sha256.json')))
    for k, (n, p, C) in enumerate(mg.CUBEPAD_HASHED):
        x = mg.cubepad_input(n, C, 1, 2000 + k)
        got = CubePad(p)(torch.from_numpy(x).to(DEV)).cpu().numpy()
        assert sha(got) == want['%d_%d_%d' % (n, p, C)], (n, p, C)


def test_cubepad_reference_smoke(golden_dir):
    """The reference's own smoke test as a GPU test (model/cube_pad.py:256-261: ``CubePad(2)`` on a [12, 64, 256, 256] CUDA
    tensor, printing the output size [12, 64, 260, 260]) - through the drop-in module, against the hash of the reference's
    output for the same seeded input, and as the reference runs it (zeros in, size out)."""
    want = json.load(open(os.path.join(golden_dir, 'cubepad_sha256.json')))
    n, p, C, groups = mg.CUBEPAD_SMOKE
    cp = CubePad(p)
    got = cp(torch.from_numpy(mg.cubepad_input(n, C, groups, 2100)).to(DEV))
    assert tuple(got.size()) == (12, 64, 260, 260)
    assert sha(got.cpu().numpy()) == want['smoke_%d_%d_%d_x%d' % mg.CUBEPAD_SMOKE]
    aa = torch.FloatTensor(np.zeros([12, 64, 256, 256])).to(DEV)
    out = cp(aa)
    assert tuple(out.size()) == (12, 64, 260, 260) and not bool(out.any())


@pytest.mark.parametrize('dtype', [torch.uint8, torch.bfloat16, torch.float32, torch.float64])
def test_cubepad_every_element_size(dtype):
    x = hashrng.integers(77, (12, 5, 9, 9), 0, 200).astype(np.float64)
    xt = torch.from_numpy(x).to(dtype)
    bits = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[xt.element_size()]
    want = o_cubepad.cubepad(xt.view(bits).numpy(), [2, 1, 3, 0])          # compare raw bit patterns
    got = CubePad([2, 1, 3, 0])(xt.to(DEV)).cpu()
    assert got.dtype == dtype and np.array_equal(got.view(bits).numpy(), want)


@pytest.mark.parametrize('shape,pad,dtype', [
    ((12, 16, 7, 7), 1, torch.float32),          # ConvLSTM faces (clstm.py:38: CubePad(1) on [6B, C, 7, 7])
    ((24, 2000, 7, 7), 1, torch.float32),        # ... at the cell's real width
    ((6, 256, 14, 14), 1, torch.bfloat16),       # layer3 (resnet_cubic.py:85-106)
    ((12, 128, 28, 28), 1, torch.float16),       # layer2
    ((6, 8, 8, 8), [1, 2, 3, 1], torch.float32), # asymmetric pads (cube_pad.py:60-70): corners from the deeper strip
    ((6, 32, 7, 7), 3, torch.uint8),
    ((6, 4, 32, 32), 3, torch.float64),
    ((6, 6, 9, 9), 2, torch.float32),            # plane bytes not a multiple of 16 for most channel counts: fallback or CH = 4
])
def test_cubepad_nchw_small_faces_whole_cube_kernel(shape, pad, dtype):
    """csrc/cubepad.hip, cubepad_nchw_cube_kernel: (cube, channel range) items staged through LDS, the CubePad map as a
    per-workgroup table - bit-exact against the oracle (model/cube_pad.py:95-216) for the network's small faces, asymmetric
    pads, every element size; and identical to the element-per-lane kernel (CP360 A/B path: channels_last input takes
    the NHWC kernel, so the comparison partner is the oracle)."""
    x = hashrng.integers(81, shape, 0, 250).astype(np.float64)
    xt = torch.from_numpy(x).to(dtype)
    bits = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[xt.element_size()]
    want = o_cubepad.cubepad(xt.view(bits).numpy(), pad)
    got = CubePad(pad)(xt.to(DEV)).cpu()
    assert got.dtype == dtype and np.array_equal(got.view(bits).numpy(), want)


def test_cubepad_nchw_randomised_geometry_sweep():
    """Seeded sweep over the NCHW kernels' dispatch space (csrc/cubepad.hip launch_nchw: whole-cube kernel for small
    faces, lds6 when the six padded planes fit the LDS, row bands for long rows, channel / plane / strip kernels and the
    element-per-lane kernel otherwise): random face size, asymmetric pads (each <= n, cube_pad.py:60-70), channel count,
    element size, 1-2 cubes, and a base pointer offset by a few elements so the 16-byte alignment checks see misaligned
    planes; then shapes with enough (cube, channel) items for the read-once kernels, odd plane sizes (chunks shared between
    planes), pads of 0 on some sides.  Bit-exact against the oracle (cube_pad.py:95-216).  tests/parity_helpers.py."""
    ph.cubepad_sweep(DEV)


@pytest.mark.parametrize('env', [
    {'CP360_CUBEPAD_NOLDS6': '1', 'CP360_CUBEPAD_NOBAND': '1', 'CP360_CUBEPAD_CHANNEL_MIN': '1'},      # channel kernel
    {'CP360_CUBEPAD_NOLDS6': '1', 'CP360_CUBEPAD_NOBAND': '1', 'CP360_CUBEPAD_NOCHANNEL': '1'},        # plane kernel
    {'CP360_CUBEPAD_NOLDS6': '1', 'CP360_CUBEPAD_NOBAND': '1', 'CP360_CUBEPAD_STRIP_V1': '1'},         # round-2 strip kernel
    {'CP360_CUBEPAD_NOLDS6': '1', 'CP360_CUBEPAD_NOCUBE': '1', 'CP360_CUBEPAD_BAND_MINROW': '1'},      # row bands everywhere
    {'CP360_CUBEPAD_NOCUBE': '1', 'CP360_CUBEPAD_LDS6_MIN': '1'},                                       # lds6 wherever it fits
    {'CP360_CUBEPAD_ELEMENTWISE': '1', 'CP360_CUBEPAD_NOCUBE': '1'},                                    # element-per-lane kernel
], ids=['channel', 'plane', 'strip_v1', 'band', 'lds6', 'elementwise'])
def test_cubepad_nchw_sweep_with_each_kernel_forced(env):
    """The same sweep with the dispatch pinned to each NCHW kernel through its A/B switch (the switches are read once per
    process, hence a child process per setting): every kernel is bit-exact on every geometry it accepts, not only on the
    ones the default dispatch hands it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    e['PYTHONPATH'] = root + os.pathsep + e.get('PYTHONPATH', '')
    r = subprocess.run([sys.executable, '-c', 'from tests import parity_helpers as ph; ph.cubepad_sweep("cuda"); print("SWEEP_OK")'],
                       cwd=root, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'SWEEP_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_cubepad_nhwc_and_channel_padding():
    x = hashrng.integers(78, (6, 7, 14, 14), 0, 1 << 20).astype(np.float32)       # NCHW
    want = o_cubepad.cubepad(x, 1)
    xt = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(DEV)   # NHWC, C = 7 (odd word count)
    got = ops.cubepad_nhwc(xt, 1).cpu().numpy().transpose(0, 3, 1, 2)
    assert np.array_equal(got, want)
    got8 = ops.cubepad_nhwc(xt, 1, c_out=8).cpu().numpy()
    assert np.array_equal(got8[..., :7].transpose(0, 3, 1, 2), want) and not got8[..., 7].any()
    # channels_last tensors take the NHWC kernel through the module too
    cl = torch.from_numpy(x).to(DEV).contiguous(memory_format=torch.channels_last)
    assert np.array_equal(CubePad(1)(cl).cpu().numpy(), want)


@pytest.mark.parametrize('C,dtype,c_out,n,pad', [
    (3, torch.float32, None, 224, 3),        # conv1's input pad in NHWC (pixel-per-thread kernel, 3 words)
    (3, torch.float32, 4, 57, [2, 0, 1, 3]), # ... with the channel padding 3 -> 4 of the fused path, asymmetric pads
    (1, torch.float32, None, 9, 1), (2, torch.float32, 8, 16, 2), (4, torch.int32, None, 14, 1),
    (6, torch.float16, None, 28, 1),         # 3 words of 2-byte channels
    (8, torch.bfloat16, None, 28, 1),        # 4 words: the 16-byte vector kernel
    (64, torch.float16, None, 56, 1), (7, torch.float32, None, 33, [1, 2, 3, 0]),
])
def test_cubepad_nhwc_pixel_widths(C, dtype, c_out, n, pad):
    """csrc/cubepad.hip NHWC kernels (a thread per pixel for 1-4 word pixels, 4 / 8 / 16-byte vectors with 32-bit index
    arithmetic otherwise), optional zero-filled channel padding: bit-exact against the oracle (cube_pad.py:95-216)."""
    x = hashrng.integers(91, (12, C, n, n), 0, 250).astype(np.float64)
    xt = torch.from_numpy(x).to(dtype)
    bits = {2: torch.int16, 4: torch.int32}[xt.element_size()]
    want = o_cubepad.cubepad(xt.view(bits).numpy(), pad)                       # NCHW bit patterns
    xn = xt.permute(0, 2, 3, 1).contiguous().to(DEV)
    got = ops.cubepad_nhwc(xn, pad, c_out=c_out) if c_out else ops.cubepad_nhwc(xn, pad)
    g = got.cpu().view(bits).numpy()
    assert np.array_equal(g[..., :C].transpose(0, 3, 1, 2), want)
    if c_out:
        assert g.shape[-1] == c_out and not g[..., C:].any()


def test_cubepad_errors():
    with pytest.raises(ValueError):
        CubePad(1)(torch.zeros(5, 2, 4, 4, device=DEV))
    with pytest.raises(ValueError):
        CubePad(1)(torch.zeros(6, 2, 4, 5, device=DEV))
    y = CubePad(0)(torch.ones(6, 1, 3, 3, device=DEV))
    assert y.shape == (6, 1, 3, 3) and bool((y == 1).all())


def test_layout_round_trip_and_slices():
    x = hashrng.normal(79, (4, 37, 5, 9))
    xt = torch.from_numpy(x).to(DEV)
    nhwc = ops.nchw_to_nhwc(xt)
    assert np.array_equal(nhwc.cpu().numpy(), x.transpose(0, 2, 3, 1))
    assert np.array_equal(ops.nhwc_to_nchw(nhwc).cpu().numpy(), x)
    wide = torch.zeros((4, 5, 9, 80), device=DEV)
    ops.nchw_to_nhwc(xt, out=wide, coff=40)
    w = wide.cpu().numpy()
    assert np.array_equal(w[..., 40:77], x.transpose(0, 2, 3, 1)) and not w[..., :40].any() and not w[..., 77:].any()
    assert np.array_equal(ops.nhwc_to_nchw(wide, channels=37, coff=40).cpu().numpy(), x)
    bf = ops.nchw_to_nhwc(xt, out_dtype=torch.bfloat16)
    assert torch.equal(bf.cpu(), torch.from_numpy(x.transpose(0, 2, 3, 1).copy()).to(torch.bfloat16))


# ------------------------------------------------------------------ K1 equi -> cube
@pytest.mark.parametrize('H,W,cd', [(64, 128, 16), (256, 512, 64), (960, 1920, 224)])
def test_equi2cube_matches_oracle(H, W, cd):
    frame = synth.frame_u8(5, H, W)
    e = Equi2Cube(cd, (H, W))
    want = ph.oracle_cubes(frame, cd)                                   # [6,3,cd,cd] f32, fixed point
    ft = torch.from_numpy(frame[None]).to(DEV)
    got = e.to_cube_batch(ft, layout='nchw').cpu().numpy()
    assert np.max(np.abs(got - want)) <= 2e-5
    got4 = e.to_cube_batch(ft, layout='nhwc4').cpu().numpy()
    assert np.max(np.abs(got4[..., :3].transpose(0, 3, 1, 2) - want)) <= 2e-5 and not got4[..., 3].any()
    # float input, reference-style to_cube (no normalisation), dict of 6 HWC faces
    img = frame.astype(np.float64) / 255.0
    faces = e.to_cube(img)
    ref = o_e2c.to_cube(img, cd)
    for f in range(6):
        assert faces[f].shape == (cd, cd, 3) and np.max(np.abs(faces[f] - ref[f])) <= 2e-6
    # plain-float bilinear switch
    e2 = Equi2Cube(cd, (H, W), cv_fixed_point=False)
    got_f = e2.to_cube_batch(ft, layout='nchw').cpu().numpy()
    want_f = ph.oracle_cubes(frame, cd, fixed_point=False)
    assert np.max(np.abs(got_f - want_f)) <= 2e-4      # fp32 fractional weights vs float64
    bf = e.to_cube_batch(ft, out_dtype=torch.bfloat16, layout='nhwc4').float().cpu().numpy()
    assert np.max(np.abs(bf[..., :3].transpose(0, 3, 1, 2) - want)) <= 2e-2


def test_to_cube_staging_keeps_the_reference_semantics():
    """``Equi2Cube.to_cube`` (equi_to_cube.py:112-129) through its pinned staging buffers: fresh arrays per call (a later
    call does not change what an earlier one returned), the input's floating dtype, and the same faces for a float64, a
    float32, a read-only, a non-contiguous and a negatively strided view of the same frame (the last ones take the numpy conversion
    path)."""
    H, W, cd = 256, 512, 64
    e = Equi2Cube(cd, (H, W))
    a = synth.frame_u8(5, H, W).astype(np.float64) / 255.0
    b = synth.frame_u8(6, H, W).astype(np.float64) / 255.0
    fa = dict(e.to_cube(a))
    keep = {k: v.copy() for k, v in fa.items()}
    fb = e.to_cube(b)
    for f in range(6):
        assert fa[f].dtype == np.float64 and np.array_equal(fa[f], keep[f]) and not np.array_equal(fb[f], keep[f])
        assert np.max(np.abs(keep[f] - o_e2c.to_cube(a, cd)[f])) <= 2e-6
    f32 = dict(e.to_cube(a.astype(np.float32)))                 # (the dict itself is the object's, as in the reference)
    ro = a.copy()
    ro.flags.writeable = False
    fro = dict(e.to_cube(ro))
    wide = np.zeros((H, W, 4))
    wide[..., :3] = a
    fnc = e.to_cube(wide[..., :3])
    for f in range(6):
        assert f32[f].dtype == np.float32 and np.max(np.abs(f32[f] - keep[f])) <= 1e-6
        assert np.array_equal(fro[f], keep[f]) and np.array_equal(fnc[f], keep[f])
    # negative strides (BGR -> RGB by a[..., ::-1], a vertically flipped frame): torch.from_numpy refuses them, the
    # reference's cv2.remap of per-channel slices takes any strides
    rev = dict(e.to_cube(a[..., ::-1]))
    want_rev = dict(e.to_cube(np.ascontiguousarray(a[..., ::-1])))
    flip = dict(e.to_cube(a[::-1]))
    want_flip = dict(e.to_cube(np.ascontiguousarray(a[::-1])))
    for f in range(6):
        assert np.array_equal(rev[f], want_rev[f]) and np.array_equal(rev[f], keep[f][..., ::-1])
        assert np.array_equal(flip[f], want_flip[f])


# ------------------------------------------------------------------ K6 cube -> equi
@pytest.mark.parametrize('w', [4, 7, 8, 16])
def test_cube2equi_golden(golden_dir, w):
    z = np.load(os.path.join(golden_dir, 'c2e.npz'))
    c = Cube2Equi(w)
    x = torch.from_numpy(z['nn_in_%d' % w]).to(DEV)
    got = c.to_equi_nn(x).cpu().numpy()
    assert got.shape == (1, 5, 2 * w, 4 * w)
    assert np.max(np.abs(got - z['nn_out_%d' % w])) <= 2e-6
    sal = c.saliency(x).cpu().numpy()[0]
    assert np.max(np.abs(sal - z['nn_out_%d' % w][0].max(axis=0))) <= 2e-6


def test_cube2equi_nhwc_batched_max_and_align_corners():
    w, C, B = 7, 1000, 3
    x = hashrng.normal(81, (6 * B, C, w, w))
    for ac in (False, True):
        c = Cube2Equi(w, align_corners=ac)
        xt = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(DEV)
        sal = c.saliency(xt, layout='nhwc').cpu().numpy()
        for b in range(B):
            want = o_c2e.saliency_from_hidden(x[6 * b:6 * b + 6], align_corners=ac)
            assert np.max(np.abs(sal[b] - want)) <= 5e-6, (ac, b)


# ------------------------------------------------------------------ K3 convolution
def _conv_ref(x_nchw, w, scale, bias, stride, pad, relu, res=None):
    xt = torch.from_numpy(x_nchw)
    if pad:
        xt = o_resnet.cubepad_t(xt, pad)
    wt = torch.from_numpy(w)
    if scale is not None:
        wt = wt * torch.from_numpy(scale)[:, None, None, None]
    y = Fn.conv2d(xt, wt, None if bias is None else torch.from_numpy(bias), stride=stride)
    if res is not None:
        y = y + torch.from_numpy(res)
    return (Fn.relu(y) if relu else y).numpy()


CONV_CASES = [
    # n_img, cin, cout, n, k, stride, pad, relu, res, splits
    (6, 64, 64, 12, 1, 1, 0, True, False, None),       # narrow tile (c_out <= 64)
    (6, 64, 256, 12, 1, 1, 0, True, True, None),       # residual epilogue
    (12, 128, 128, 10, 3, 2, 1, True, False, None),    # CubePad fused, stride on the 3x3
    (6, 256, 512, 8, 1, 2, 0, False, False, None),     # strided 1x1 (downsample)
    (6, 40, 72, 7, 3, 1, 1, True, False, 3),           # K tail (40 % 32), split-K + finish
    (6, 2000, 136, 7, 3, 1, 1, True, False, 4),        # ConvLSTM-like K (2000 per tap)
    (18, 32, 1000, 7, 1, 1, 0, False, False, None),    # CAM-like c_out = 1000 (N tail)
]


_TDT = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}
# one output rounding: 2^-8 relative for bf16, 2^-11 for fp16 (f32 accumulate in both)
_TOL = {'fp32': 2e-5, 'bf16': 1.2e-2, 'fp16': 1.5e-3}


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('prec', ['fp32', 'bf16', 'fp16'])
def test_conv_matches_torch_cpu(case, prec):
    n_img, cin, cout, n, k, stride, pad, relu, use_res, splits = case
    dt = _TDT[prec]
    seed = 9000 + cin + cout
    x = hashrng.normal(seed, (n_img, cin, n, n))
    w = hashrng.normal(seed + 1, (cout, cin, k, k), 0, (2.0 / (k * k * cin)) ** 0.5)
    scale = hashrng.uniform(seed + 2, (cout,), 0.5, 1.5)
    bias = hashrng.normal(seed + 3, (cout,), 0, 0.1)
    ho = (n + 2 * pad - k) // stride + 1
    res = hashrng.normal(seed + 4, (n_img, cout, ho, ho)) if use_res else None
    if prec != 'fp32':     # the oracle sees the same rounded operands; accumulation stays f32
        rb = lambda a: torch.from_numpy(a).to(dt).float().numpy()
        x_ref, res_ref = rb(x), (None if res is None else rb(res))
        w_ref = (torch.from_numpy(w) * torch.from_numpy(scale)[:, None, None, None]).to(dt).float().numpy()
        want = _conv_ref(x_ref, w_ref, None, bias, stride, pad, relu, res_ref)
    else:
        want = _conv_ref(x, w, scale, bias, stride, pad, relu, res)
    conv = ops.Conv(torch.from_numpy(w), torch.from_numpy(scale), torch.from_numpy(bias), stride, pad, relu, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    rt = None if res is None else ops.nchw_to_nhwc(torch.from_numpy(res).to(DEV), out_dtype=dt)
    got = ops.nhwc_to_nchw(conv(xt, residual=rt, splits=splits), out_dtype=torch.float32).cpu().numpy()
    assert got.shape == want.shape
    assert rel_err(got, want) <= _TOL[prec], rel_err(got, want)


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('prec', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('raw', [False, True])
def test_conv_small_tile_matches_torch_cpu(case, prec, raw):
    """tile_px = 6464: the 64 x 64-tile small-M kernel (csrc/conv_small.hip; what BASELINE config C2 - one frame = 6 faces,
    fp32 - runs layers 1-4 and the CAM on) over the same cases as the generic kernels: narrow / ragged channel tiles, the
    residual epilogue, fused CubePad with stride, strided 1x1, K tails, split-K slabs (packed-row column order) + finish,
    c_out = 1000; raw: the f32 sums in true channel order (how the CAM scores leave, class_activation_model.py:77-83)."""
    n_img, cin, cout, n, k, stride, pad, relu, use_res, splits = case
    dt = _TDT[prec]
    seed = 9000 + cin + cout
    x = hashrng.normal(seed, (n_img, cin, n, n))
    w = hashrng.normal(seed + 1, (cout, cin, k, k), 0, (2.0 / (k * k * cin)) ** 0.5)
    scale = hashrng.uniform(seed + 2, (cout,), 0.5, 1.5)
    bias = hashrng.normal(seed + 3, (cout,), 0, 0.1)
    ho = (n + 2 * pad - k) // stride + 1
    res = hashrng.normal(seed + 4, (n_img, cout, ho, ho)) if use_res and not raw else None
    rb = (lambda a: torch.from_numpy(a).to(dt).float().numpy()) if prec != 'fp32' else (lambda a: a)
    w_ref = rb((w * scale[:, None, None, None]).astype(np.float32))
    want = _conv_ref(rb(x), w_ref, None, None if raw else bias, stride, pad, relu and not raw, None if res is None else rb(res))
    conv = ops.Conv(torch.from_numpy(w), torch.from_numpy(scale), torch.from_numpy(bias), stride, pad, relu, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    rt = None if res is None else ops.nchw_to_nhwc(torch.from_numpy(res).to(DEV), out_dtype=dt)
    if raw:
        part, ns = conv(xt, raw_f32=True, splits=splits or 1, tile_px=6464)
        got = part[:ns * n_img * ho * ho * cout].view(ns, n_img, ho, ho, cout).sum(0).permute(0, 3, 1, 2).cpu().numpy()
        tol = 2e-5 if prec == 'fp32' else 1e-4                     # no output rounding: only the summation order differs
    else:
        got = ops.nhwc_to_nchw(conv(xt, residual=rt, splits=splits, tile_px=6464), out_dtype=torch.float32).cpu().numpy()
        tol = _TOL[prec]
    assert got.shape == want.shape
    assert rel_err(got, want) <= tol, rel_err(got, want)


def test_conv_small_tile_is_what_one_frame_fp32_runs():
    """The launch planner picks the small tile by itself for fp32 launches that cannot fill the chip (M = 294 .. 4704 pixels,
    one frame), not for the 64-frame batches: same result as the forced tile bit for bit at 6 faces, and the suggested split
    count differs between 6 and 384 faces (different kernels were planned)."""
    cin, cout, n = 1024, 256, 14
    w = hashrng.normal(9701, (cout, cin, 1, 1), 0, (2.0 / cin) ** 0.5)
    conv = ops.Conv(torch.from_numpy(w), None, torch.zeros(cout), 1, 0, True, torch.float32, DEV)
    x = torch.from_numpy(hashrng.normal(9700, (6, n, n, cin))).to(DEV)
    auto, forced = conv(x), conv(x, tile_px=6464, splits=conv._splits_cache[(6, n, n, 0)])
    assert torch.equal(auto, forced)


def _pw64_case(prec, n_img, n, order):
    """One 1x1, 64 -> 64 convolution + bn + relu on seeded operands (shared by the test below and its CP360_PW64=0 child)."""
    dt = _TDT[prec]
    w = hashrng.normal(9801, (64, 64, 1, 1), 0, (2.0 / 64) ** 0.5)
    scale = hashrng.normal(9802, (64,), 1.0, 0.1)
    bias = hashrng.normal(9803, (64,), 0, 0.2)
    conv = ops.Conv(torch.from_numpy(w), torch.from_numpy(scale), torch.from_numpy(bias), 1, 0, True, dt, DEV)
    xt = torch.from_numpy(hashrng.normal(9800 + n, (n_img, n, n, 64))).to(DEV).to(dt)
    with ops.launch_order(order):
        got = conv(xt)
    wr = (torch.from_numpy(w[:, :, 0, 0]) * torch.from_numpy(scale)[:, None]).to(dt).float()
    want = torch.relu(xt.float().cpu() @ wr.t() + torch.from_numpy(bias).float())
    return got.cpu(), want


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('n_img,n', [(24, 56), (6, 27)])
def test_pointwise_64_to_64_streaming_kernel(prec, n_img, n, tmp_path):
    """csrc/conv_igemm.hip conv_pw64_kernel (layer1.0's conv1, model/resnet_cubic.py:88-90: 1x1, 64 -> 64, + bn1 + relu): the
    filter in a wave's registers, 16-pixel blocks streamed from global memory.  Against torch-CPU on identically rounded
    operands in both traversal orders, and bit for bit against the generic 64 x 256-tile kernel on the same launch (a child
    process with CP360_PW64=0: the switch is read once per process); 6 x 27 x 27 pixels: a ragged last block."""
    import subprocess
    import sys
    outs = []
    for order in (0, 1):
        got, want = _pw64_case(prec, n_img, n, order)
        assert rel_err(got.float().numpy(), want.numpy()) <= _TOL[prec]
        outs.append(got)
    assert torch.equal(outs[0], outs[1])                          # the traversal order never changes a result
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = str(tmp_path / 'generic.pt')
    code = ("import torch; from tests import test_gpu_parity as t; "
            "torch.save(t._pw64_case(%r, %d, %d, 0)[0], %r)" % (prec, n_img, n, f))
    e = dict(os.environ, CP360_PW64='0')
    e['PYTHONPATH'] = root + os.pathsep + e.get('PYTHONPATH', '')
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert torch.equal(outs[0], torch.load(f))


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
def test_pointwise_64_to_64_raw_f32_sums_stay_on_the_generic_kernel(prec):
    """cp360_conv_forward2 with out == NULL, partial != NULL, splits == 1 (raw f32 sums) on the shape the streaming 64 -> 64
    kernel takes (M >= 4096): that kernel only stores packed 16-bit results, so the dispatch must leave this mode to the generic
    kernel (round-5 advisor finding: it used to store through a null `out`).  Against torch-CPU on the rounded operands."""
    dt = _TDT[prec]
    n_img, n = 6, 28                                                   # M = 4704 pixels
    w = hashrng.normal(9811, (64, 64, 1, 1), 0, (2.0 / 64) ** 0.5)
    conv = ops.Conv(torch.from_numpy(w), None, torch.zeros(64), 1, 0, False, dt, DEV)
    xt = torch.from_numpy(hashrng.normal(9812, (n_img, n, n, 64))).to(DEV).to(dt)
    part, ns = conv(xt, raw_f32=True, splits=1)
    torch.cuda.synchronize()
    assert ns == 1
    got = part[:n_img * n * n * 64].view(n_img * n * n, 64).cpu()
    want = xt.float().cpu().view(-1, 64) @ torch.from_numpy(w[:, :, 0, 0]).to(dt).float().t()
    assert rel_err(got.numpy(), want.numpy()) <= 1e-4


@pytest.mark.parametrize('prec', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('geom', [(64, 64, 256, 14, 1, 0), (96, 200, 264, 14, 2, 0), (128, 256, 512, 10, 2, 304),
                                  (512, 1024, 2048, 7, 1, 256), (64, 64, 256, 9, 1, 3), (96, 200, 264, 14, 2, 6464),
                                  (512, 1024, 2048, 7, 1, 6464)])
def test_conv_second_source_downsample_fused(prec, geom):
    """cp360_conv_desc.c_in2: out = relu(W3 . mid + b3 + Wd . x[::s, ::s] + bd) in one tile = conv3 + bn3 +
    downsample(conv1x1 stride s + bn) + add + relu of a Bottleneck's first block (resnet_cubic.py:85-106),
    against torch-CPU on identically rounded operands.  Last geometry field: forced tile / 3 = split-K 3."""
    cmid, cin2, cout, n, st, force = geom
    dt = _TDT[prec]
    n_img = 6
    n2 = n * st
    mid = hashrng.normal(9400, (n_img, cmid, n, n))
    x = hashrng.normal(9401, (n_img, cin2, n2, n2))
    w3 = hashrng.normal(9402, (cout, cmid, 1, 1), 0, (2.0 / cmid) ** 0.5)
    wd = hashrng.normal(9403, (cout, cin2, 1, 1), 0, (2.0 / cin2) ** 0.5)
    s3, sd = hashrng.uniform(9404, (cout,), 0.5, 1.5), hashrng.uniform(9405, (cout,), 0.5, 1.5)
    b3, bd = hashrng.normal(9406, (cout,), 0, 0.1), hashrng.normal(9407, (cout,), 0, 0.1)
    rb = (lambda a: torch.from_numpy(a).to(dt).float().numpy()) if prec != 'fp32' else (lambda a: a)
    fold = lambda w, sc: rb((torch.from_numpy(w) * torch.from_numpy(sc)[:, None, None, None]).numpy())
    import torch.nn.functional as Fn
    with torch.no_grad():
        want = Fn.conv2d(torch.from_numpy(rb(mid)), torch.from_numpy(fold(w3, s3)), torch.from_numpy(b3)) + \
            Fn.conv2d(torch.from_numpy(rb(x)), torch.from_numpy(fold(wd, sd)), torch.from_numpy(bd), stride=st)
        want = torch.relu(want).numpy()
    conv = ops.Conv(torch.from_numpy(w3), torch.from_numpy(s3), torch.from_numpy(b3), 1, 0, True, dt, DEV,
                    second=(torch.from_numpy(wd), torch.from_numpy(sd), torch.from_numpy(bd), st))
    mt = ops.nchw_to_nhwc(torch.from_numpy(mid).to(DEV), out_dtype=dt)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    kw = {'splits': 3} if force == 3 else ({'tile_px': force} if force else {})
    got = ops.nhwc_to_nchw(conv(mt, x2=xt, **kw), out_dtype=torch.float32).cpu().numpy()
    assert got.shape == want.shape
    assert rel_err(got, want) <= _TOL[prec], rel_err(got, want)
    with pytest.raises(ValueError):
        conv(mt)                                         # x2 is mandatory for a conv built with second=


def test_resnet_fused_downsample_equals_separate(golden_dir):
    """The whole ResNet with the downsample branches fused into conv3 (default) vs run as separate
    convolutions + residual add: same f32 result up to summation order."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    x = torch.from_numpy(hashrng.normal(4000 + 64, (6, 64, 64, 3), 0.0, 1.0)).to(DEV)
    x4 = ops.cubepad_nhwc(x, 0, c_out=4)
    outs = []
    for fuse in (True, False):
        rc.FUSE_DOWNSAMPLE = fuse
        try:
            m, _ = _load_resnet()
            outs.append(m.features_nhwc(x4).float().cpu().numpy())
        finally:
            rc.FUSE_DOWNSAMPLE = True
    assert rel_err(outs[0], outs[1]) <= 2e-5


@pytest.mark.parametrize('prec', ['fp32', 'fp16'])
def test_launch_order_never_changes_results(prec):
    """cp360_set_launch_order is a traversal hint: the whole static stage (every kernel that honours it: stem + pool,
    generic / ring convolutions, layer1-3 tails) gives the same bits ascending, descending and alternating."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    dt = _TDT[prec]
    m, _ = _load_resnet(prec)
    x = torch.from_numpy(hashrng.normal(4100, (12, 224, 224, 3), 0.0, 1.0)).to(DEV)
    x4 = ops.cubepad_nhwc(x, 0, c_out=4)
    if dt != torch.float32:
        x4 = ops.nchw_to_nhwc(x4.reshape(1, 1, 1, -1), out_dtype=dt).reshape(x4.shape)
    outs = []
    old = rc.LAUNCH_ORDER
    try:
        for mode in (0, 1, 2):
            rc.LAUNCH_ORDER = mode
            outs.append(m.features_nhwc(x4).clone())
    finally:
        rc.LAUNCH_ORDER = old
    assert outs[0].shape == (12, 7, 7, 2048)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
def test_layer1_fused_block_kernel_cube512_faces(prec):
    """K3d at 128x128 faces (cube 512, BASELINE config C5): two output rows per workgroup, two waves per row."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    dt = _TDT[prec]
    m, sd = _load_resnet(prec)
    x = torch.from_numpy(np.abs(hashrng.normal(4700, (6, 128, 128, 64), 0.0, 1.0))).to(DEV).to(dt)
    got = m.layer1_nhwc(x).float().cpu().numpy()
    rc.FUSE_LAYER1 = False
    try:
        sep = m.layer1_nhwc(x).float().cpu().numpy()
    finally:
        rc.FUSE_LAYER1 = True
    assert got.shape == (6, 128, 128, 256)
    assert rel_err(got, sep) <= _TOL[prec], rel_err(got, sep)
    # torch-CPU layer1 (resnet_cubic.py:85-106 on cube-padded input) in f32 from the same 16-bit input
    from oracle import o_resnet
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    xc = x.float().cpu().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for b in range(3):
            xc = o_resnet._bottleneck(xc, sdt, 'layer1.%d' % b, 1, b == 0)
        want = xc.permute(0, 2, 3, 1).numpy()
    assert rel_err(got, want) <= 4 * _TOL[prec], rel_err(got, want)


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('n_img', [6, 12])
def test_layer1_fused_block_kernel(prec, n_img):
    """K3d (csrc/l1block.hip): layer1 as conv1 + one launch per Bottleneck vs (a) the per-convolution path and
    (b) torch-CPU on cube-padded input (reference modules' arithmetic, resnet_cubic.py:85-106) - every
    intermediate is rounded at the same points in all three."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    from oracle import o_resnet
    dt = _TDT[prec]
    m, sd = _load_resnet(prec)
    x = torch.from_numpy(np.abs(hashrng.normal(4500 + n_img, (n_img, 56, 56, 64), 0.0, 1.0))).to(DEV).to(dt)
    got = m.layer1_nhwc(x).float().cpu().numpy()
    rc.FUSE_LAYER1 = False
    try:
        sep = m.layer1_nhwc(x).float().cpu().numpy()
    finally:
        rc.FUSE_LAYER1 = True
    assert got.shape == (n_img, 56, 56, 256)
    assert rel_err(got, sep) <= _TOL[prec], rel_err(got, sep)
    # torch-CPU layer1 in f32 from the same (16-bit) input: loose bound, three blocks of 16-bit roundings
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    xc = x.float().cpu().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for b in range(3):
            xc = o_resnet._bottleneck(xc, sdt, 'layer1.%d' % b, 1, b == 0)
        want = xc.permute(0, 2, 3, 1).numpy()
    assert rel_err(got, want) <= 4 * _TOL[prec], rel_err(got, want)


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('n_img', [6, 30])
def test_layer1_first_block_runs_its_own_conv1_in_the_patch(prec, n_img):
    """csrc/l1block.hip, FIRST (round 6): layer1.0's conv1 (1x1, 64 -> 64, + bn1 + relu, resnet_cubic.py:88-90) runs in place on the band's
    resident patch inside the block's tail kernel (cp360_l1block_forward_first).  The whole layer1 with it == the whole layer1 with conv1 as its
    own launch, bit for bit (the halo pixels are computed again by the neighbouring bands with the same arithmetic), in both launch orders;
    the direct pair (out, next conv1's output) likewise."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    dt = _TDT[prec]
    m, sd = _load_resnet(prec)
    x = torch.from_numpy(np.abs(hashrng.normal(4560 + n_img, (n_img, 56, 56, 64), 0.0, 1.0))).to(DEV).to(dt)
    assert rc.FUSE_L1_FIRST
    fused = m._layer1_fused()
    assert fused[0].w0 is not None
    for order in (0, 1):
        with ops.launch_order(order):
            out_f, mid_f = fused[0].first(x)
            mid0 = list(m.layer1)[0]._plans()['c1'](x)
            out_s, mid_s = fused[0](mid0, x_ds=x)
            whole_f = m.layer1_nhwc(x)
            rc.FUSE_L1_FIRST = False
            try:
                whole_s = m.layer1_nhwc(x)
            finally:
                rc.FUSE_L1_FIRST = True
        torch.cuda.synchronize()
        assert torch.equal(out_f.view(torch.int16), out_s.view(torch.int16)) and torch.equal(mid_f.view(torch.int16), mid_s.view(torch.int16))
        assert torch.equal(whole_f.view(torch.int16), whole_s.view(torch.int16))
    assert float(out_f.float().abs().max()) > 0.1


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('face', [56, 128])
def test_layer1_last_tail_chains_layer2_conv1(prec, face):
    """csrc/l1block.hip, l1block_wide_kernel: layer1's last Bottleneck tail also computes layer2.0's conv1 + bn1 + relu
    (256 -> 128, resnet_cubic.py:88-90) from its output pieces.  out == the unchained tail bit for bit; mid2 vs the
    per-convolution conv1 on the SAME rounded out (same operands, f32 accumulate) and vs torch-CPU; and layer2 fed
    with mid2 == layer2 computing its own conv1."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    dt = _TDT[prec]
    m, sd = _load_resnet(prec)
    x = torch.from_numpy(np.abs(hashrng.normal(4550 + face, (6, face, face, 64), 0.0, 1.0))).to(DEV).to(dt)
    out, mid2 = m.layer1_nhwc(x, want_next=True)
    assert mid2 is not None and tuple(mid2.shape) == (6, face, face, 128)
    rc.CHAIN_L1_L2 = False
    try:
        out_u, none = m.layer1_nhwc(x, want_next=True)
    finally:
        rc.CHAIN_L1_L2 = True
    assert none is None and torch.equal(out, out_u)
    c1 = m.layer2[0]._plans()['c1'](out)
    assert rel_err(mid2.float().cpu().numpy(), c1.float().cpu().numpy()) <= _TOL[prec]
    s1 = sd['layer2.0.bn1.weight'] / np.sqrt(sd['layer2.0.bn1.running_var'] + 1e-5)
    b1 = sd['layer2.0.bn1.bias'] - sd['layer2.0.bn1.running_mean'] * s1
    want = torch.relu(Fn.conv2d(out.float().cpu().permute(0, 3, 1, 2), torch.from_numpy(sd['layer2.0.conv1.weight']))
                      * torch.from_numpy(s1)[None, :, None, None] + torch.from_numpy(b1)[None, :, None, None])
    assert rel_err(mid2.float().cpu().numpy(), want.permute(0, 2, 3, 1).numpy()) <= _TOL[prec]
    y_chain = m.layer2_nhwc(out, mid2).float().cpu().numpy()
    y_own = m.layer2_nhwc(out).float().cpu().numpy()
    assert rel_err(y_chain, y_own) <= _TOL[prec]


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('n_img', [6, 12])
def test_layer2_first_block_fused_kernel(prec, n_img):
    """K3f (csrc/lfirst.hip): layer2.0 after its conv1 as ONE launch - CubePad(1) + conv3x3 stride 2 + bn2 + relu ->
    conv3 + bn3 + downsample(x) + relu (resnet_cubic.py:85-106,145-161) - vs (a) the per-convolution path (generic stride-2
    conv + second-source conv3) on the same rounded operands and (b) torch-CPU ``_bottleneck`` in f32."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    from oracle import o_resnet
    dt = _TDT[prec]
    m, sd = _load_resnet(prec)
    x = torch.from_numpy(np.abs(hashrng.normal(4650 + n_img, (n_img, 56, 56, 256), 0.0, 1.0))).to(DEV).to(dt)
    b0 = m.layer2[0]
    mid = b0._plans()['c1'](x)
    m.layer2_nhwc(x)                                  # builds m._l2f
    got_t = m._l2f(mid, x, chain=False)
    got = got_t.float().cpu().numpy()
    sep = b0.forward_nhwc(x, mid=mid).float().cpu().numpy()
    assert got.shape == (n_img, 28, 28, 512)
    assert rel_err(got, sep) <= _TOL[prec], rel_err(got, sep)
    # the variant that also computes layer2.1's conv1 (512 -> 128) from the output pieces: same out bit for bit, and
    # mid1 == the per-convolution conv1 on that rounded out
    out_c, mid1 = m._l2f(mid, x, chain=True)
    assert torch.equal(out_c, got_t) and tuple(mid1.shape) == (n_img, 28, 28, 128)
    c1 = m.layer2[1]._plans()['c1'](out_c)
    assert rel_err(mid1.float().cpu().numpy(), c1.float().cpu().numpy()) <= _TOL[prec]
    # the whole layer with / without the fused first block
    full = m.layer2_nhwc(x).float().cpu().numpy()
    rc.FUSE_L2_FIRST = False
    try:
        full_sep = m.layer2_nhwc(x).float().cpu().numpy()
    finally:
        rc.FUSE_L2_FIRST = True
    assert rel_err(full, full_sep) <= _TOL[prec], rel_err(full, full_sep)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        want = o_resnet._bottleneck(x.float().cpu().permute(0, 3, 1, 2).contiguous(), sdt, 'layer2.0', 2, True)
    assert rel_err(got, want.permute(0, 2, 3, 1).numpy()) <= 2 * _TOL[prec]


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
def test_layer2_fused_tail_kernel_cube512_faces(prec):
    """K3e at 64x64 faces (cube 512, BASELINE config C5): bands of two output rows (128 pixels = 8 pixel blocks)."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    dt = _TDT[prec]
    m, sd = _load_resnet(prec)
    x = torch.from_numpy(np.abs(hashrng.normal(4800, (6, 128, 128, 256), 0.0, 1.0))).to(DEV).to(dt)
    got = m.layer2_nhwc(x).float().cpu().numpy()
    rc.FUSE_LAYER2 = False
    try:
        sep = m.layer2_nhwc(x).float().cpu().numpy()
    finally:
        rc.FUSE_LAYER2 = True
    assert got.shape == (6, 64, 64, 512)
    assert rel_err(got, sep) <= _TOL[prec], rel_err(got, sep)
    # torch-CPU layer2 (resnet_cubic.py:85-106) in f32 from the same 16-bit input
    from oracle import o_resnet
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    xc = x.float().cpu().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for b in range(4):
            xc = o_resnet._bottleneck(xc, sdt, 'layer2.%d' % b, 2 if b == 0 else 1, b == 0)
        want = xc.permute(0, 2, 3, 1).numpy()
    assert rel_err(got, want) <= 4 * _TOL[prec], rel_err(got, want)


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('n_img', [6, 12])
def test_layer2_fused_tail_kernel(prec, n_img):
    """K3e (csrc/l2block.hip): layer2 with conv2 -> conv3 + residual (-> the next block's conv1) of the identity
    blocks in one launch vs the per-convolution path, vs the tails without the chained conv1, and vs torch-CPU
    (resnet_cubic.py:85-106)."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    from oracle import o_resnet
    dt = _TDT[prec]
    m, sd = _load_resnet(prec)
    x = torch.from_numpy(np.abs(hashrng.normal(4600 + n_img, (n_img, 56, 56, 256), 0.0, 1.0))).to(DEV).to(dt)
    got = m.layer2_nhwc(x).float().cpu().numpy()
    rc.FUSE_LAYER2 = False
    try:
        sep = m.layer2_nhwc(x).float().cpu().numpy()
    finally:
        rc.FUSE_LAYER2 = True
    assert got.shape == (n_img, 28, 28, 512)
    assert rel_err(got, sep) <= _TOL[prec], rel_err(got, sep)
    rc.FUSE_LAYER2_NEXT = False            # the tails without the chained conv1 of the next block
    try:
        unchained = m.layer2_nhwc(x).float().cpu().numpy()
    finally:
        rc.FUSE_LAYER2_NEXT = True
    assert rel_err(got, unchained) <= _TOL[prec], rel_err(got, unchained)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    xc = x.float().cpu().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for b in range(4):
            xc = o_resnet._bottleneck(xc, sdt, 'layer2.%d' % b, 2 if b == 0 else 1, b == 0)
        want = xc.permute(0, 2, 3, 1).numpy()
    assert rel_err(got, want) <= 4 * _TOL[prec], rel_err(got, want)


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('n_img', [6, 12])
def test_layer3_fused_tail_kernel(prec, n_img):
    """K3e at layer3's geometry (csrc/l2block.hip, C = 256, 14x14 faces, bands of 7 rows = 98 pixels in 7 pixel blocks
    with 14 padding slots): layer3 with conv2 -> conv3 + residual of the identity blocks in one launch vs the
    per-convolution path and vs torch-CPU (resnet_cubic.py:85-106)."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    from oracle import o_resnet
    dt = _TDT[prec]
    m, sd = _load_resnet(prec)
    x = torch.from_numpy(np.abs(hashrng.normal(4900 + n_img, (n_img, 28, 28, 512), 0.0, 1.0))).to(DEV).to(dt)
    got = m.layer3_nhwc(x).float().cpu().numpy()
    rc.FUSE_LAYER3 = False
    try:
        sep = m.layer3_nhwc(x).float().cpu().numpy()
    finally:
        rc.FUSE_LAYER3 = True
    assert got.shape == (n_img, 14, 14, 1024)
    assert rel_err(got, sep) <= _TOL[prec], rel_err(got, sep)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    xc = x.float().cpu().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for b in range(6):
            xc = o_resnet._bottleneck(xc, sdt, 'layer3.%d' % b, 2 if b == 0 else 1, b == 0)
        want = xc.permute(0, 2, 3, 1).numpy()
    assert rel_err(got, want) <= 6 * _TOL[prec], rel_err(got, want)


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
def test_layer3_fused_tail_kernel_cube512_faces(prec):
    """K3e at layer3's geometry on 32x32 faces (cube 512, BASELINE config C5): bands of two rows (64 pixels = 4 pixel
    blocks), vs the per-convolution path and vs torch-CPU (resnet_cubic.py:85-106)."""
    from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic as rc
    from oracle import o_resnet
    dt = _TDT[prec]
    m, sd = _load_resnet(prec)
    x = torch.from_numpy(np.abs(hashrng.normal(4950, (6, 64, 64, 512), 0.0, 1.0))).to(DEV).to(dt)
    got = m.layer3_nhwc(x).float().cpu().numpy()
    rc.FUSE_LAYER3 = False
    try:
        sep = m.layer3_nhwc(x).float().cpu().numpy()
    finally:
        rc.FUSE_LAYER3 = True
    assert got.shape == (6, 32, 32, 1024)
    assert rel_err(got, sep) <= _TOL[prec], rel_err(got, sep)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    xc = x.float().cpu().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for b in range(6):
            xc = o_resnet._bottleneck(xc, sdt, 'layer3.%d' % b, 2 if b == 0 else 1, b == 0)
        want = xc.permute(0, 2, 3, 1).numpy()
    assert rel_err(got, want) <= 6 * _TOL[prec], rel_err(got, want)


@pytest.mark.parametrize('tile_px', [128, 256, 304])
@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('splits', [1, 3])
def test_wide_conv_forced_pixel_tiles(tile_px, prec, splits):
    """c_out >= 256 kernels with the pixel tile forced (cp360_conv_desc.tile_px): 256x128 LDS-DMA,
    256x256 ring and the 256x304 ring (19 pixel blocks: waves 0-3 carry 10, waves 4-7 carry 9, three
    waves issue the partial DMA pass).  M = 2*294 = 588 pixels: a full 304 tile + a ragged one; c_out =
    264 gives a ragged channel tile; residual + ReLU exercise the LDS epilogue, splits = 3 the slabs."""
    dt = _TDT[prec]
    n_img, cin, cout, n, k = 12, 96, 264, 7, 3
    x = hashrng.normal(9300, (n_img, cin, n, n))
    w = hashrng.normal(9301, (cout, cin, k, k), 0, (2.0 / (k * k * cin)) ** 0.5)
    bias = hashrng.normal(9303, (cout,), 0, 0.1)
    res = hashrng.normal(9304, (n_img, cout, n, n))
    rb = (lambda a: torch.from_numpy(a).to(dt).float().numpy()) if prec != 'fp32' else (lambda a: a)
    want = _conv_ref(rb(x), rb(w), None, bias, 1, 1, True, rb(res))
    conv = ops.Conv(torch.from_numpy(w), None, torch.from_numpy(bias), 1, 1, True, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    rt = ops.nchw_to_nhwc(torch.from_numpy(res).to(DEV), out_dtype=dt)
    got = ops.nhwc_to_nchw(conv(xt, residual=rt, splits=splits, tile_px=tile_px), out_dtype=torch.float32).cpu().numpy()
    assert rel_err(got, want) <= _TOL[prec], rel_err(got, want)


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('geom', [(12, 96, 264, 7, 3, 1, 1), (12, 96, 264, 7, 3, 1, 3), (6, 512, 512, 14, 1, 1, 1), (6, 128, 256, 14, 3, 2, 1)])
def test_wide_conv_160_pixel_tile(prec, geom):
    """tile_px = 160 (round 6): the 256-channel x 160-pixel ring tile (5 pixel blocks per wave, two waves issue the partial DMA
    pass of 32 rows) - what the planner takes when the 256 / 304-pixel tiles leave CUs without a workgroup (layer4 at 64 frames).
    3x3 CubePad and 1x1 and stride-2 geometries, a ragged last tile (588 / 1176 / 294 pixels), a ragged channel tile (264), residual +
    ReLU through the LDS epilogue, split-K slabs; f32 is refused."""
    dt = _TDT[prec]
    n_img, cin, cout, n, k, st, splits = geom
    x = hashrng.normal(9310, (n_img, cin, n, n))
    w = hashrng.normal(9311, (cout, cin, k, k), 0, (2.0 / (k * k * cin)) ** 0.5)
    bias = hashrng.normal(9313, (cout,), 0, 0.1)
    no = (n + 2 * (k // 2) - k) // st + 1
    res = hashrng.normal(9314, (n_img, cout, no, no))
    rb = lambda a: torch.from_numpy(a).to(dt).float().numpy()
    want = _conv_ref(rb(x), rb(w), None, bias, st, k // 2, True, rb(res))
    conv = ops.Conv(torch.from_numpy(w), None, torch.from_numpy(bias), st, k // 2, True, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    rt = ops.nchw_to_nhwc(torch.from_numpy(res).to(DEV), out_dtype=dt)
    got = ops.nhwc_to_nchw(conv(xt, residual=rt, splits=splits, tile_px=160), out_dtype=torch.float32).cpu().numpy()
    assert got.shape == want.shape and rel_err(got, want) <= _TOL[prec], rel_err(got, want)
    if geom == (12, 96, 264, 7, 3, 1, 1):
        c32 = ops.Conv(torch.from_numpy(w), None, torch.from_numpy(bias), 1, 1, True, torch.float32, DEV)
        with pytest.raises(Exception):
            c32(ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=torch.float32), tile_px=160)


@pytest.mark.parametrize('prec', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('cin', [64, 200])
@pytest.mark.parametrize('use_res', [False, True])
def test_short_k_two_workgroup_kernel(prec, cin, use_res):
    """tile_px = 129: the 256x128 two-stage kernel that runs two workgroups per CU (1x1 convolutions with a
    short K: layer1-3 conv3 / downsample), direct 16-byte epilogue.  M = 6*11*11 = 726 pixels (ragged last
    tile), c_out = 520 (ragged channel tile, multiple of 8), K = 64 (two sub-steps = the two stages) and 200
    (seven sub-steps, K tail)."""
    dt = _TDT[prec]
    n_img, cout, n = 6, 520, 11
    x = hashrng.normal(9500 + cin, (n_img, cin, n, n))
    w = hashrng.normal(9501, (cout, cin, 1, 1), 0, (2.0 / cin) ** 0.5)
    scale = hashrng.uniform(9502, (cout,), 0.5, 1.5)
    bias = hashrng.normal(9503, (cout,), 0, 0.1)
    res = hashrng.normal(9504, (n_img, cout, n, n)) if use_res else None
    rb = (lambda a: torch.from_numpy(a).to(dt).float().numpy()) if prec != 'fp32' else (lambda a: a)
    w_ref = rb(w * scale[:, None, None, None]) if prec != 'fp32' else w * scale[:, None, None, None]
    want = _conv_ref(rb(x), w_ref.astype(np.float32), None, bias, 1, 0, True, None if res is None else rb(res))
    conv = ops.Conv(torch.from_numpy(w), torch.from_numpy(scale), torch.from_numpy(bias), 1, 0, True, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    rt = None if res is None else ops.nchw_to_nhwc(torch.from_numpy(res).to(DEV), out_dtype=dt)
    if prec == 'fp32':          # 16-bit types only: the kernel lives inside a 128-VGPR budget
        with pytest.raises(Exception):
            conv(xt, residual=rt, tile_px=129)
        return
    got = ops.nhwc_to_nchw(conv(xt, residual=rt, tile_px=129), out_dtype=torch.float32).cpu().numpy()
    assert rel_err(got, want) <= _TOL[prec], rel_err(got, want)


@pytest.mark.parametrize('n', [7, 5, 3, 16, 8])
@pytest.mark.parametrize('prec', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('splits', [1, 3, 7])
def test_clip_resident_conv(n, prec, splits):
    """cp360_conv_desc.clip_resident: CubePad(1) + 3x3 on faces <= 7x7 with the clip's activations
    resident in LDS and the taps served as row permutations (channel-major packed weights).  3 clips
    (cubes) of 6 n^2 pixels; c_in = 104: K tail inside a 64-byte channel block (bf16) / exact (fp32);
    c_out = 264: ragged channel tile; splits 3 and 7 cut inside channel blocks; residual + ReLU run the
    LDS epilogue with rows past the clip masked."""
    dt = _TDT[prec]
    # n = 16: the FACE variant (tile = face + ring); n = 8: the HALF variant (tile = three faces, the cube resident)
    n_img, cin, cout, k = (18 if n < 16 else 12), 104, 264, 3
    x = hashrng.normal(9400 + n, (n_img, cin, n, n))
    w = hashrng.normal(9401, (cout, cin, k, k), 0, (2.0 / (k * k * cin)) ** 0.5)
    bias = hashrng.normal(9403, (cout,), 0, 0.1)
    res = hashrng.normal(9404 + n, (n_img, cout, n, n))
    rb = (lambda a: torch.from_numpy(a).to(dt).float().numpy()) if prec != 'fp32' else (lambda a: a)
    want = _conv_ref(rb(x), rb(w), None, bias, 1, 1, True, rb(res))
    conv = ops.Conv(torch.from_numpy(w), None, torch.from_numpy(bias), 1, 1, True, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    rt = ops.nchw_to_nhwc(torch.from_numpy(res).to(DEV), out_dtype=dt)
    got = ops.nhwc_to_nchw(conv(xt, residual=rt, splits=splits, clip_resident=True), out_dtype=torch.float32).cpu().numpy()
    assert rel_err(got, want) <= _TOL[prec], rel_err(got, want)
    # and it equals the generic (tap-major) kernel on the same operands up to summation order
    gen = ops.nhwc_to_nchw(conv(xt, residual=rt, splits=splits, clip_resident=False), out_dtype=torch.float32).cpu().numpy()
    assert rel_err(got, gen) <= _TOL[prec]


@pytest.mark.parametrize('prec', ['fp32', 'bf16', 'fp16'])
def test_stem_conv_and_maxpool(prec):
    dt = _TDT[prec]
    x = hashrng.normal(9100, (6, 3, 32, 32))
    w = hashrng.normal(9101, (64, 3, 7, 7), 0, (2.0 / (49 * 64)) ** 0.5)
    scale = hashrng.uniform(9102, (64,), 0.5, 1.5)
    bias = hashrng.normal(9103, (64,), 0, 0.1)
    if prec != 'fp32':
        rb = lambda a: torch.from_numpy(a).to(dt).float().numpy()
        w_ref = (torch.from_numpy(w) * torch.from_numpy(scale)[:, None, None, None]).to(dt).float().numpy()
        stem_ref = _conv_ref(rb(x), w_ref, None, bias, 2, 3, True)
    else:
        stem_ref = _conv_ref(x, w, scale, bias, 2, 3, True)
    conv = ops.Conv(torch.from_numpy(w), torch.from_numpy(scale), torch.from_numpy(bias), 2, 0, True, dt, DEV, stem=True)
    x3 = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV))                   # [6,32,32,3] f32
    x4 = ops.cubepad_nhwc(x3, 0, c_out=4)
    if prec != 'fp32':
        x4 = ops.nchw_to_nhwc(x4.reshape(1, 1, 1, -1), out_dtype=dt).reshape(x4.shape)
    xp = ops.cubepad_nhwc(x4, 3)
    y = conv(xp)
    got = ops.nhwc_to_nchw(y, out_dtype=torch.float32).cpu().numpy()
    assert got.shape == stem_ref.shape == (6, 64, 16, 16)
    assert rel_err(got, stem_ref) <= _TOL[prec]
    pooled = ops.nhwc_to_nchw(ops.cubepad_maxpool3s2(y), out_dtype=torch.float32).cpu().numpy()
    want_pool = Fn.max_pool2d(o_resnet.cubepad_t(torch.from_numpy(got), 1), 3, 2, 0).numpy()
    assert np.array_equal(pooled, want_pool)     # max of identical values: exact


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
def test_stem_resident_patch_kernel_cube224(prec):
    """K3a at the reference's cube size: CubePad(3) -> 7x7 s2 conv + BN + ReLU on 12 faces of 224x224
    (two cubes: several bands per workgroup are not needed for parity, the persistent loop is exercised by
    168 tiles on <= 256 workgroups... so also run 60 faces to make workgroups loop) against torch-CPU on the
    same rounded operands, and against the generic implicit GEMM (summation order only)."""
    dt = _TDT[prec]
    for n_img, seed, cd in ((12, 9600, 224), (60, 9610, 224), (6, 9620, 512), (18, 9630, 512)):   # 512: BASELINE config C5
        x = hashrng.normal(seed, (n_img, 3, cd, cd))
        w = hashrng.normal(9601, (64, 3, 7, 7), 0, (2.0 / (49 * 64)) ** 0.5)
        scale = hashrng.uniform(9602, (64,), 0.5, 1.5)
        bias = hashrng.normal(9603, (64,), 0, 0.1)
        conv = ops.Conv(torch.from_numpy(w), torch.from_numpy(scale), torch.from_numpy(bias), 2, 0, True, dt, DEV, stem=True)
        x3 = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV))
        x4 = ops.cubepad_nhwc(x3, 0, c_out=4)
        x4 = ops.nchw_to_nhwc(x4.reshape(1, 1, 1, -1), out_dtype=dt).reshape(x4.shape)
        xp = ops.cubepad_nhwc(x4, 3)                                   # [n_img, cd + 6, cd + 6, 4]
        got = conv(xp)                                                 # resident-patch kernel
        assert got.shape == (n_img, cd // 2, cd // 2, 64)
        gen = conv(xp, tile_px=0, splits=1)                            # splits given -> generic path
        g, e = got.float().cpu().numpy(), gen.float().cpu().numpy()
        assert rel_err(g, e) <= _TOL[prec], rel_err(g, e)
        if n_img in (12, 6):
            rb = lambda a: torch.from_numpy(a).to(dt).float().numpy()
            w_ref = rb(w * scale[:, None, None, None])
            want = _conv_ref(rb(x), w_ref, None, bias, 2, 3, True)
            assert rel_err(ops.nhwc_to_nchw(got, out_dtype=torch.float32).cpu().numpy(), want) <= _TOL[prec]


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('order', [0, 1])
def test_stem_pool_fused_kernel_bit_exact(prec, order):
    """csrc/stem.hip stem_pool4_kernel (round 6: two 4-wave workgroups per CU, 4 stem rows per band; + stem_pool_fix_kernel): stem
    conv + BN + ReLU -> CubePad(1) -> max-pool 3x3 s2 in one kernel at cube 224 must equal the two-kernel path (resident stem,
    then cubepad_maxpool3s2) bit for bit - the same MFMA order per output and maxima of identical values; 12 faces (every band of
    a face in its own workgroup) and 300 faces (8400 band tiles on 512 persistent workgroups, several cubes), both launch orders.
    The 8-wave form of rounds 2-5 (CP360_STEM_POOL=8, read once per process): test_stem_pool_8wave_form_bit_exact."""
    dt = _TDT[prec]
    w = hashrng.normal(9641, (64, 3, 7, 7), 0, (2.0 / (49 * 64)) ** 0.5)
    scale = hashrng.uniform(9642, (64,), 0.5, 1.5)
    bias = hashrng.normal(9643, (64,), 0, 0.1)
    conv = ops.Conv(torch.from_numpy(w), torch.from_numpy(scale), torch.from_numpy(bias), 2, 0, True, dt, DEV, stem=True)
    for n_img, seed in ((12, 9640), (300, 9650)):
        x = hashrng.normal(seed, (12, 3, 224, 224))
        x3 = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV))
        x4 = ops.cubepad_nhwc(x3, 0, c_out=4)
        x4 = ops.nchw_to_nhwc(x4.reshape(1, 1, 1, -1), out_dtype=dt).reshape(x4.shape)
        if n_img > 12:                                              # more cubes: the same two, rolled and sign-flipped
            reps = [torch.roll(x4, k, dims=1 + k % 2) * (1.0 if k % 3 else -1.0) for k in range(n_img // 12)]
            x4 = torch.cat(reps, 0).contiguous()
        xp = ops.cubepad_nhwc(x4, 3)
        want = ops.cubepad_maxpool3s2(conv(xp))
        with ops.launch_order(order):
            got = conv.stem_pool(xp)
        assert got is not None and got.shape == want.shape == (n_img, 56, 56, 64)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), \
            int((got.view(torch.int16) != want.view(torch.int16)).sum())


def test_stem_pool_8wave_form_bit_exact():
    """CP360_STEM_POOL=8: stem_pool_kernel (one 8-wave workgroup per CU, double-buffered patch) stays buildable for A/B and must give
    the same bits - the test above in a child process with the switch set."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from tests import test_gpu_parity as t; t.test_stem_pool_fused_kernel_bit_exact('fp16', 0); t.test_stem_pool_fused_kernel_bit_exact('bf16', 1)"
    e = dict(os.environ, CP360_STEM_POOL='8')
    e['PYTHONPATH'] = root + os.pathsep + e.get('PYTHONPATH', '')
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
def test_band3x3_resident_kernel_layer1_conv2(prec):
    """K3c: CubePad(1) + 3x3 conv 64 -> 64 + BN + ReLU on 56x56 faces (12 faces = two cubes: the halo of
    every band comes through cubepad_src from the other faces of ITS cube) against torch-CPU on the same
    rounded operands and against the generic implicit GEMM."""
    dt = _TDT[prec]
    n_img = 12
    x = hashrng.normal(9700, (n_img, 64, 56, 56))
    w = hashrng.normal(9701, (64, 64, 3, 3), 0, (2.0 / (9 * 64)) ** 0.5)
    scale = hashrng.uniform(9702, (64,), 0.5, 1.5)
    bias = hashrng.normal(9703, (64,), 0, 0.1)
    conv = ops.Conv(torch.from_numpy(w), torch.from_numpy(scale), torch.from_numpy(bias), 1, 1, True, dt, DEV)
    xt = ops.nchw_to_nhwc(torch.from_numpy(x).to(DEV), out_dtype=dt)
    got = conv(xt)                                                     # resident-band kernel
    gen = conv(xt, splits=1)                                           # splits given -> generic path
    assert got.shape == gen.shape == (n_img, 56, 56, 64)
    g, e = got.float().cpu().numpy(), gen.float().cpu().numpy()
    assert rel_err(g, e) <= _TOL[prec], rel_err(g, e)
    rb = lambda a: torch.from_numpy(a).to(dt).float().numpy()
    want = _conv_ref(rb(x), rb(w * scale[:, None, None, None]), None, bias, 1, 1, True)
    assert rel_err(ops.nhwc_to_nchw(got, out_dtype=torch.float32).cpu().numpy(), want) <= _TOL[prec]


# ------------------------------------------------------------------ ResNet-50-cubic + CAM
def _load_resnet(prec='fp32'):
    sd = synth.resnet50_state(seed=1)
    m = resnet50(precision=prec)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing)
    return m.to(DEV).eval(), sd


def test_resnet_state_dict_keys_match_reference_names():
    m = resnet50()
    keys = set(m.state_dict().keys())
    for k in synth.resnet50_state(seed=1):
        assert k in keys
    assert len(keys) == 320      # 320 entries incl. num_batches_tracked


def test_resnet_layer4_small_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, 'resnet_cam.npz'))
    m, _ = _load_resnet()
    cubes = mg.synth_cubes(4000 + 64, 64)
    x = torch.from_numpy(np.ascontiguousarray(cubes.transpose(0, 3, 1, 2))).to(DEV)
    seen = []
    h = m._modules.get('layer4').register_forward_hook(lambda mod, i, o: seen.append(o))
    y = m(x)
    h.remove()
    assert len(seen) == 1 and seen[0].shape == (6, 2048, 2, 2)
    got = seen[0].cpu().numpy()
    assert rel_err(got, z['layer4_s']) <= 3e-4
    assert np.array_equal(y.cpu().numpy(), got)


def test_cam_full_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, 'resnet_cam.npz'))
    m, sd = _load_resnet()
    cubes = mg.synth_cubes(4000 + 224, 224)
    score, feat, wsm = CAM(cubes, None, m, 'layer4', 'fc.weight', use_gpu=True)
    assert score.shape == (6, 1000, 7, 7) and feat.shape == (6, 2048, 7, 7) and wsm.shape == (1000, 2048)
    assert np.array_equal(wsm[::97, ::53], z['wsm_pick'])
    assert rel_err(feat[:, ::8], z['layer4_f']) <= 3e-4
    assert rel_err(score, z['cam_f']) <= 3e-4
    # north-star gate: window-normalised CAM within 1e-3 absolute
    mn, mx = z['cam_f'].min(), z['cam_f'].max()
    assert np.max(np.abs((score - mn) / (mx - mn) - (z['cam_f'] - mn) / (mx - mn))) <= 1e-3
    # the model's fc.weight is NOT mutated (the reference's CPU path does, survey a7)
    assert np.array_equal(m.fc.weight.detach().cpu().numpy(), sd['fc.weight'])


def test_cam_bf16_close_to_fp32_oracle(golden_dir):
    z = np.load(os.path.join(golden_dir, 'resnet_cam.npz'))
    m, _ = _load_resnet('bf16')
    score, _, _ = CAM(mg.synth_cubes(4000 + 224, 224), None, m, 'layer4', 'fc.weight')
    # random-init weights amplify bf16 rounding through 16 residual blocks (a few % of
    # the CAM range, mostly a common scale factor); what the temporal stage consumes is
    # the window-normalised map, i.e. the CAM up to an affine map -> check correlation.
    cc = np.corrcoef(score.reshape(-1), z['cam_f'].reshape(-1))[0, 1]
    mn, mx = z['cam_f'].min(), z['cam_f'].max()
    err = np.abs((score - mn) / (mx - mn) - (z['cam_f'] - mn) / (mx - mn))
    assert cc >= 0.995 and np.max(err) <= 0.15, (cc, np.max(err), np.mean(err))


# ------------------------------------------------------------------ ConvLSTM
def test_clstm_small_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, 'clstm.npz'))
    sd = synth.clstm_state(seed=7, input_size=8, hidden_size=8)
    cell = ConvLSTMCell(8, 8)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    cell = cell.to(DEV).eval()
    x = torch.from_numpy(hashrng.uniform(5000, (12, 8, 4, 4))).to(DEV)
    h = torch.from_numpy(hashrng.uniform(5001, (12, 8, 4, 4))).to(DEV)
    c = torch.from_numpy(hashrng.uniform(5002, (12, 8, 4, 4))).to(DEV)
    h1, c1 = cell(x, [h, c])
    h2, c2 = cell(x, [h1, c1])
    for got, key in ((h1, 'small_h1'), (c1, 'small_c1'), (h2, 'small_h2'), (c2, 'small_c2')):
        assert np.max(np.abs(got.cpu().numpy() - z[key])) <= 2e-6, key
    h0, c0 = cell(x)       # prev_state=None -> zeros (clstm.py:47-52)
    zero = torch.zeros(12, 8, 4, 4)
    hw, cw = o_clstm.clstm_step(x.cpu(), zero, zero, {k: torch.from_numpy(v) for k, v in sd.items()})
    assert np.max(np.abs(h0.cpu().numpy() - hw.numpy())) <= 2e-6 and np.max(np.abs(c0.cpu().numpy() - cw.numpy())) <= 2e-6


@pytest.fixture(scope='module')
def full_cell_state():
    return synth.clstm_state(seed=2)


def _full_cell(sd, prec):
    cell = ConvLSTMCell(1000, 1000, precision=prec)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert sorted(cell.state_dict().keys()) == sorted(sd.keys())       # strict names (test_temporal.py:149)
    return cell.to(DEV).eval()


@pytest.mark.parametrize('T', [5, 16])
def test_clstm_full_window_golden(golden_dir, full_cell_state, T):
    z = np.load(os.path.join(golden_dir, 'clstm.npz'))
    cell = _full_cell(full_cell_state, 'fp32')
    frames = synth.cam_clip(6000 + T, T)                                # [T,6,1000,7,7]
    cam = torch.from_numpy(np.ascontiguousarray(frames.transpose(0, 1, 3, 4, 2)).reshape(1, T, 294, 1000)).to(DEV)
    runner = ClipRunner(cell, Cube2Equi(7), 1, T)
    sal, hid = runner.run(cam, return_hidden=True)
    hid_nchw = ops.nhwc_to_nchw(hid).cpu().numpy()
    assert np.max(np.abs(hid_nchw.reshape(-1)[z['pick']] - z['full_hidden_pick_T%d' % T])) <= 1e-3
    assert np.max(np.abs(sal.cpu().numpy()[0] - z['full_map_T%d' % T])) <= 1e-3     # north-star gate


def test_clstm_batched_clips_equal_single(full_cell_state):
    """B clips in one GEMM (M = 294*B, other split-K factor) == each clip alone."""
    cell = _full_cell(full_cell_state, 'fp32')
    T, B = 3, 4
    cams = [synth.cam_clip(6100 + b, T) for b in range(B)]
    pack = lambda f: np.ascontiguousarray(f.transpose(0, 1, 3, 4, 2)).reshape(T, 294, 1000)
    c2e = Cube2Equi(7)
    batched = ClipRunner(cell, c2e, B, T).run(torch.from_numpy(np.stack([pack(f) for f in cams])).to(DEV)).cpu().numpy()
    single = ClipRunner(cell, c2e, 1, T)
    for b in range(B):
        one = single.run(torch.from_numpy(pack(cams[b])[None]).to(DEV)).cpu().numpy()[0]
        assert np.max(np.abs(batched[b] - one)) <= 2e-5
    want = o_clstm.window_saliency(cams[1], {k: torch.from_numpy(v) for k, v in full_cell_state.items()})
    assert np.max(np.abs(batched[1] - want)) <= 1e-3


def test_clstm_bf16_window(golden_dir, full_cell_state):
    z = np.load(os.path.join(golden_dir, 'clstm.npz'))
    cell = _full_cell(full_cell_state, 'bf16')
    frames = synth.cam_clip(6000 + 5, 5)
    cam = torch.from_numpy(np.ascontiguousarray(frames.transpose(0, 1, 3, 4, 2)).reshape(1, 5, 294, 1000)).to(DEV)
    sal = ClipRunner(cell, Cube2Equi(7), 1, 5).run(cam).cpu().numpy()[0]
    err = np.abs(sal - z['full_map_T5'])
    assert np.max(err) <= 2e-2, np.max(err)       # bf16 moves the map by ~1e-3


def test_e2c_padded_layout_equals_project_then_cubepad():
    """Equi2Cube layout 'nhwc4p3' (CubePad(3) fused into K1) == K1 followed by the CubePad kernel, bit for
    bit, for u8 / f32 inputs and every output type; and the fused max-pool's 16-byte path == the 4-channel
    path on the same data."""
    H, W, cd = 256, 512, 64
    e = Equi2Cube(cd, (H, W))
    fr = torch.from_numpy(np.stack([synth.frame_u8(70 + i, H, W) for i in range(3)])).to(DEV)
    for frames in (fr, fr.float() / 255.0):
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            a = e.to_cube_batch(frames, out_dtype=dt, layout='nhwc4p3')
            b = ops.cubepad_nhwc(e.to_cube_batch(frames, out_dtype=dt, layout='nhwc4'), 3)
            assert a.shape == (18, cd + 6, cd + 6, 4) and torch.equal(a.view(torch.uint8), b.view(torch.uint8))
    x = torch.from_numpy(hashrng.normal(72, (6, 10, 10, 64))).to(DEV)
    for dt in (torch.bfloat16, torch.float16):
        xh = x.to(dt)
        full = ops.cubepad_maxpool3s2(xh)                                            # C % 8 == 0: 16-byte path
        part = torch.cat([ops.cubepad_maxpool3s2(xh[..., :60].contiguous()), ops.cubepad_maxpool3s2(xh[..., 60:].contiguous())], -1)
        assert torch.equal(full, part)


def test_window_minmax_and_normalize():
    B, T, P, C = 3, 4, 294, 1000
    x = hashrng.normal(83, (B, T, P, C), 5.0, 100.0)
    xt = torch.from_numpy(x).to(DEV)
    mm = torch.empty((B, 2), device=DEV)
    scratch = torch.empty((B * 512,), device=DEV)
    ops.window_minmax(xt, B, T * P * C, mm, scratch)
    got = mm.cpu().numpy()
    assert np.array_equal(got[:, 0], x.reshape(B, -1).min(1)) and np.array_equal(got[:, 1], x.reshape(B, -1).max(1))
    y = torch.zeros((B, P, 2 * C), device=DEV)
    y2 = torch.empty((B, P, C), device=DEV)
    ops.window_normalize(xt, mm, y, C, y2, B, T, 2, P, C)
    for b in range(B):
        mn, mx = x[b].min(), x[b].max()
        want = (x[b, 2] - mn) / (mx - mn)
        assert np.array_equal(y[b, :, C:].cpu().numpy(), want) and np.array_equal(y2[b].cpu().numpy(), want)
    assert not y[:, :, :C].any()


# ------------------------------------------------------------------ end to end
@pytest.fixture(scope='module')
def e2e_small():
    H, W, cd, T, B = 256, 512, 64, 3, 2
    rs = synth.resnet50_state(seed=1)
    cs = synth.clstm_state(seed=3)
    clips = np.stack([synth.clip_u8(20 + b, T, H, W) for b in range(B)])
    refs = [ph.oracle_pipeline(clips[b], rs, cs, cd, return_all=True) for b in range(B)]
    return dict(H=H, W=W, cd=cd, T=T, B=B, rs=rs, cs=cs, clips=clips, refs=refs)


def test_pipeline_fp32_end_to_end(e2e_small):
    s = e2e_small
    eng = SaliencyEngine(s['rs'], s['cs'], (s['H'], s['W']), s['cd'], clips=s['B'], frames=s['T'], precision='fp32')
    sal = eng(torch.from_numpy(s['clips']).to(DEV)).cpu().numpy()
    cam = eng.cam.cpu().numpy()                       # [B,T,24,1000] NHWC
    for b in range(s['B']):
        ref_sal, ref_cams, _ = s['refs'][b]
        want_cam = ref_cams.transpose(0, 1, 3, 4, 2).reshape(s['T'], -1, 1000)
        mn, mx = want_cam.min(), want_cam.max()
        assert np.max(np.abs((cam[b] - mn) / (mx - mn) - (want_cam - mn) / (mx - mn))) <= 1e-3
        assert sal[b].shape == (4, 8)
        assert np.max(np.abs(sal[b] - ref_sal)) <= 1e-3
    # a second call on the same engine (buffers reused) is deterministic
    sal2 = eng(torch.from_numpy(s['clips']).to(DEV)).cpu().numpy()
    assert np.array_equal(sal, sal2)


def test_pipeline_bf16_end_to_end(e2e_small):
    s = e2e_small
    eng = SaliencyEngine(s['rs'], s['cs'], (s['H'], s['W']), s['cd'], clips=s['B'], frames=s['T'], precision='bf16')
    sal = eng(torch.from_numpy(s['clips']).to(DEV)).cpu().numpy()
    for b in range(s['B']):
        err = np.abs(sal[b] - s['refs'][b][0])
        assert np.max(err) <= 5e-2, np.max(err)


def test_pipeline_graph_replay_equals_eager(e2e_small):
    """hipGraph capture of the whole path: replay == eager bit for bit, and new frame contents written
    into the captured input (or passed as another tensor) give the other clip's result."""
    s = e2e_small
    eng = SaliencyEngine(s['rs'], s['cs'], (s['H'], s['W']), s['cd'], clips=s['B'], frames=s['T'], precision='fp32')
    a = torch.from_numpy(s['clips']).to(DEV)
    b = torch.from_numpy(np.ascontiguousarray(s['clips'][::-1])).to(DEV)          # the two clips swapped
    eager_a = eng(a).clone()
    eager_b = eng(b).clone()
    a0 = a.clone()
    eng.capture(a)                                                                # the graph reads a's memory
    assert torch.equal(eng(a), eager_a)
    assert torch.equal(eng(b).clone(), eager_b)                                   # b is copied into the captured input
    assert torch.equal(eng(a0), eager_a)
    assert torch.equal(eager_a[0], eager_b[1]) and not torch.equal(eager_a[0], eager_a[1])


def test_pipeline_fp16_end_to_end(e2e_small):
    s = e2e_small
    eng = SaliencyEngine(s['rs'], s['cs'], (s['H'], s['W']), s['cd'], clips=s['B'], frames=s['T'], precision='fp16')
    sal = eng(torch.from_numpy(s['clips']).to(DEV)).cpu().numpy()
    for b in range(s['B']):
        err = np.abs(sal[b] - s['refs'][b][0])
        assert np.max(err) <= 1e-2, np.max(err)      # 8x finer mantissa than bf16 (5e-2 bound above)


def test_pipeline_full_size_one_frame_fp32():
    """Config C1/C2 shape: one 960x1920 frame -> 6x224^2 -> CAM, against the oracle."""
    H, W, cd = 960, 1920, 224
    rs = synth.resnet50_state(seed=1)
    frame = synth.frame_u8(31, H, W)
    want = ph.oracle_cam_frames(frame[None], rs, cd)[0]                  # [6,1000,7,7]
    m, _ = _load_resnet()
    e = Equi2Cube(cd, (H, W))
    x4 = e.to_cube_batch(torch.from_numpy(frame[None]).to(DEV))
    from cp_360_weakly_supervised_saliency_amd.static_model.class_activation_model import cam_device
    score, _ = cam_device(x4, m)
    got = ops.nhwc_to_nchw(score).cpu().numpy()
    mn, mx = want.min(), want.max()
    assert np.max(np.abs((got - mn) / (mx - mn) - (want - mn) / (mx - mn))) <= 1e-3


def test_sliding_window_mode_equals_per_window(full_cell_state):
    """The reference's stride-1 sliding window (test_temporal.py:57-65): B windows over ONE
    feature sequence, run in lock step from overlapping memory, equal each window alone
    and the oracle."""
    cell = _full_cell(full_cell_state, 'fp32')
    T, nwin = 3, 4
    seq = synth.cam_clip(6200, T + nwin - 1)                                    # [F,6,1000,7,7]
    pack = lambda f: np.ascontiguousarray(f.transpose(0, 1, 3, 4, 2)).reshape(f.shape[0], 294, 1000)
    c2e = Cube2Equi(7)
    sal = ClipRunner(cell, c2e, nwin, T).run(torch.from_numpy(pack(seq)).to(DEV), sliding=True).cpu().numpy()
    single = ClipRunner(cell, c2e, 1, T)
    sd = {k: torch.from_numpy(v) for k, v in full_cell_state.items()}
    for b in range(nwin):
        one = single.run(torch.from_numpy(pack(seq[b:b + T])[None]).to(DEV)).cpu().numpy()[0]
        assert np.max(np.abs(sal[b] - one)) <= 2e-5
    assert np.max(np.abs(sal[2] - o_clstm.window_saliency(seq[2:2 + T], sd))) <= 1e-3


def test_pipeline_c5_shape_fp32_and_fp16():
    """BASELINE config C5 geometry: 2048x4096 equirectangular, 6x512^2 cube faces, layer4 / ConvLSTM
    at 16x16, saliency 32x64 - one full 16-frame clip (the length bench.py's C5 line runs) against the oracle:
    fp32 within the 1e-3 north-star bound, fp16 (C5's MFMA precision) by the AUC-Judd / CC gate of SURVEY 8(d)."""
    H, W, cd, T = 2048, 4096, 512, 16
    rs = synth.resnet50_state(seed=1)
    cs = synth.clstm_state(seed=2)
    clip = synth.clip_u8(50, T, H, W)
    ref, ref_cams, _ = ph.oracle_pipeline(clip, rs, cs, cd, return_all=True)
    eng = SaliencyEngine(rs, cs, (H, W), cd, clips=1, frames=T, precision='fp32')
    sal = eng(torch.from_numpy(clip[None]).to(DEV)).cpu().numpy()[0]
    assert sal.shape == (32, 64) and ref.shape == (32, 64)
    cam = eng.cam.cpu().numpy()[0]
    want_cam = ref_cams.transpose(0, 1, 3, 4, 2).reshape(T, -1, 1000)
    mn, mx = want_cam.min(), want_cam.max()
    assert np.max(np.abs((cam - mn) / (mx - mn) - (want_cam - mn) / (mx - mn))) <= 1e-3
    assert np.max(np.abs(sal - ref)) <= 1e-3
    del eng
    from oracle import o_metrics
    eng16 = SaliencyEngine(rs, cs, (H, W), cd, clips=1, frames=T, precision='fp16')
    sal16 = eng16(torch.from_numpy(clip[None]).to(DEV)).cpu().numpy()[0]
    fix = synth.fixations_from_map(ref, 150, H // 2, W // 2)     # correlated with the oracle map (not chance level)
    auc_ref = o_metrics.auc_judd(ref, fix, rng=np.random.RandomState(0))
    auc16 = o_metrics.auc_judd(sal16, fix, rng=np.random.RandomState(0))
    cc_ref, cc16 = o_metrics.corr_coeff(ref, fix), o_metrics.corr_coeff(sal16, fix)
    print('C5 fp16: max|d| %.3e dAUC %.3e dCC %.3e' % (np.max(np.abs(sal16 - ref)), auc16 - auc_ref, cc16 - cc_ref))
    assert auc_ref > 0.7 and cc_ref > 0.05
    assert abs(auc16 - auc_ref) <= 1e-3 and abs(cc16 - cc_ref) <= 1e-3
    assert o_metrics.corr_coeff(sal16, ref) >= 0.9999
