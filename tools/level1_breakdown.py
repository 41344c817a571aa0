"""Where one Level-1 static frame (dataset_feat_extractor.py:142-162 on the shims, 1024x2048, cube 224, fp32) spends its
time on the GPU box: host conversions, PCIe copies and device work, piece by piece (each piece synchronised).
``--temporal``: the same for one Level-1 temporal window (test_temporal.py:63-85 on the shims, seq_len 5, fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cp_360_weakly_supervised_saliency_amd import ops
from cp_360_weakly_supervised_saliency_amd.model.resnet_cubic import resnet50
from cp_360_weakly_supervised_saliency_amd.static_model.class_activation_model import CAM, cam_device
from cp_360_weakly_supervised_saliency_amd.utils.equi_to_cube import Equi2Cube
from cp_360_weakly_supervised_saliency_amd.utils.utils import im_norm
from cp_360_weakly_supervised_saliency_amd.utils import synth

dev = 'cuda'
H, W, cd = 1024, 2048, 224


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def temporal():
    """test_temporal.py:63-85 piece by piece: what is the reference's own loop (host numpy, one pageable H2D per tensor), what
    is the shim (layout conversions, buffers) and what is device work."""
    from cp_360_weakly_supervised_saliency_amd.model.clstm import ConvLSTMCell
    from cp_360_weakly_supervised_saliency_amd.utils.cube_to_equi import Cube2Equi
    T = 5
    cs = synth.clstm_state(seed=2)
    cell = ConvLSTMCell(1000, 1000, precision='fp32')
    cell.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in cs.items()})
    cell = cell.to(dev).eval()
    c2e = Cube2Equi(7, device=dev)
    subseq = [f for f in synth.cam_clip(6005, T)]

    def window():
        mx, mn = np.max(subseq), np.min(subseq)
        init = (subseq[0] - mn) / (mx - mn)
        cst = torch.FloatTensor(init).to(dev)
        hidden = torch.FloatTensor(init).to(dev)
        for f in subseq:
            f = torch.FloatTensor((f - mn) / (mx - mn)).to(dev)
            hidden, cst = cell(f, [hidden, cst])
        return torch.squeeze(torch.max(c2e.to_equi_nn(hidden), 1)[0]).cpu().numpy()

    print('threads: torch %d, cpu_count %d' % (torch.get_num_threads(), os.cpu_count()))
    print('one window end to end (reference loop)          %.2f ms' % t(window))
    print('REFERENCE LOOP (host):')
    print('  np.max + np.min over the 5 frames             %.2f ms' % t(lambda: (np.max(subseq), np.min(subseq))))
    mx, mn = np.max(subseq), np.min(subseq)
    print('  (f - mn) / (mx - mn) x 6 (numpy, f32)         %.2f ms' % t(lambda: [(f - mn) / (mx - mn) for f in [subseq[0]] + subseq]))
    nf = [(f - mn) / (mx - mn) for f in subseq]
    print('  torch.FloatTensor(ndarray) x 7 (host copy)    %.2f ms' % t(lambda: [torch.FloatTensor(nf[0]) for _ in range(7)]))
    ft = [torch.FloatTensor(f) for f in nf]
    print('  .to(dev) x 7 (pageable H2D, 1.2 MB each)      %.2f ms' % t(lambda: [ft[0].to(dev) for _ in range(7)]))
    x = ft[0].to(dev)
    h, c = x.clone(), x.clone()
    print('SHIM (ConvLSTMCell.forward, per call; x 5 per window):')
    print('  forward() whole                               %.2f ms' % t(lambda: cell(x, [h, c])))
    n6, cin, w = x.shape[0], x.shape[1], x.shape[2]
    xh = torch.empty((n6, w, w, 2 * cin), dtype=torch.float32, device=dev)
    print('  NCHW -> NHWC of x, h, c (3 launches)          %.2f ms' % t(lambda: (ops.nchw_to_nhwc(x, out=xh, coff=0), ops.nchw_to_nhwc(h, out=xh, coff=cin), ops.nchw_to_nhwc(c))))
    cp = ops.nchw_to_nhwc(c)
    cn, hf = torch.empty_like(cp), torch.empty_like(cp)
    print('  allocations (xh, c_next, h_f32)               %.2f ms' % t(lambda: (torch.empty_like(xh), torch.empty_like(cp), torch.empty_like(cp))))
    with torch.no_grad():
        print('  cell update on the device (step_nhwc)         %.2f ms' % t(lambda: cell.step_nhwc(xh, cp, cn, hf)))
    print('  NHWC -> NCHW of hidden, cell (2 launches)     %.2f ms' % t(lambda: (ops.nhwc_to_nchw(hf), ops.nhwc_to_nchw(cn))))
    hid = ops.nhwc_to_nchw(hf)
    print('OUTPUT: to_equi_nn + torch.max + squeeze + D2H  %.2f ms' % t(lambda: torch.squeeze(torch.max(c2e.to_equi_nn(hid), 1)[0]).cpu().numpy()))


if '--temporal' in sys.argv:
    temporal()
    sys.exit(0)

rs = synth.resnet50_state(seed=1)
model = resnet50(precision='fp32')
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rs.items()}, strict=False)
model = model.to(dev).eval()
frame = synth.frame_u8(31, H, W)
input_img = np.array(frame) / 255.0
e2c = Equi2Cube(cd, input_img, device=dev)


def norm_batch(cubes):
    return np.concatenate([np.expand_dims(im_norm(cubes[i], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]), 0)
                           for i in range(6)], axis=0).astype(np.float32)


cubes = e2c.to_cube(input_img)
batch = norm_batch({i: cubes[i].copy() for i in range(6)})
print('threads: torch %d, cpu_count %d' % (torch.get_num_threads(), os.cpu_count()))
print('to_cube (whole)                      %.2f ms' % t(lambda: e2c.to_cube(input_img)))
f32 = np.ascontiguousarray(input_img, dtype=np.float32)
print('  f64 -> f32 on the host (numpy)     %.2f ms' % t(lambda: np.ascontiguousarray(input_img, dtype=np.float32)))
print('  H2D 25 MB pageable                 %.2f ms' % t(lambda: torch.from_numpy(f32)[None].to(dev)))
pin = torch.empty((1, H, W, 3), dtype=torch.float32).pin_memory()
print('  H2D 25 MB pinned                   %.2f ms' % t(lambda: pin.to(dev, non_blocking=True)))
x = torch.from_numpy(f32)[None].to(dev)
k1 = lambda: ops.equi2cube(x, e2c.grid, cd, torch.float32, 'nchw', scale=1.0, mean=(0., 0., 0.), std=(1., 1., 1.),
                           cv_fixed_point=e2c.cv_fixed_point)
print('  K1 kernel                          %.2f ms' % t(k1))
c = k1()
print('  permute + D2H + astype f64         %.2f ms' % t(lambda: c.permute(0, 2, 3, 1).cpu().numpy().astype(np.float64)))
print('im_norm x 6 + concatenate + astype   %.2f ms' % t(lambda: norm_batch({i: cubes[i].copy() for i in range(6)})))
print('CAM() (whole)                        %.2f ms' % t(lambda: CAM(batch, None, model, 'layer4', 'fc.weight', use_gpu=True)))
img = torch.as_tensor(batch).to(dev)
x4 = ops.cubepad_nhwc(img, 0, c_out=4)
print('  H2D batch + NHWC4                  %.2f ms' % t(lambda: ops.cubepad_nhwc(torch.as_tensor(batch).to(dev), 0, c_out=4)))
with torch.no_grad():
    print('  static stage on the device         %.2f ms' % t(lambda: cam_device(x4, model)))
    score, feat = cam_device(x4, model)
    print('  score -> NCHW -> host              %.2f ms' % t(lambda: ops.nhwc_to_nchw(score).cpu().numpy()))
    print('  feat -> NCHW f32 -> host           %.2f ms' % t(lambda: ops.nhwc_to_nchw(feat, out_dtype=torch.float32).cpu().numpy()))


def frame_once():
    cubes = e2c.to_cube(input_img)
    return CAM(norm_batch(cubes), None, model, 'layer4', 'fc.weight', use_gpu=True)[0]


print('one frame end to end                 %.2f ms' % t(frame_once))
