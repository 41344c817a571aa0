#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  The
reference is imported in memory with the shims listed in SURVEY.md section 8(c);
nothing of its source is written to this repo - the fixtures are inputs (or their
hash-RNG seeds) and the reference's outputs.

    python tests/golden/make_golden.py [--only cubepad,e2c,c2e,resnet,clstm,resize,metrics,overlay]

Shims (all in memory):
  * ``np.int = int``  (alias removed in numpy >= 1.24; cube_pad.py:13,64)
  * stub modules for cv2 / torchvision (imported, unused on this path)
  * ``CubePadding.use_gpu = False`` on every instance (default True would call
    torch.cuda.LongTensor in ``flip``)
  * utils/cube_to_equi.py and static_model/class_activation_model.py contain
    ``.cuda(async=True)`` (a SyntaxError on Python >= 3.7): the source text is
    loaded, ``.cuda(async=True)`` is deleted, ``.cuda()`` -> ``.clone()``, and the
    result is exec'd.
"""
import argparse
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

from cp_360_weakly_supervised_saliency_amd.utils import hashrng, synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def import_reference():
    import torch
    np.int = int  # noqa
    for name in ('cv2', 'torchvision'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    import matplotlib
    matplotlib.use('Agg')
    sys.path.insert(0, REF)
    import model.cube_pad as ref_cp
    import model.resnet_cubic as ref_resnet
    import model.clstm as ref_clstm
    import utils.sph_utils  # noqa
    import utils.equi_to_cube as ref_e2c

    def exec_patched(relpath, modname):
        src = open(os.path.join(REF, relpath)).read()
        src = src.replace('.cuda(async=True)', '').replace('.cuda()', '.clone()')
        mod = types.ModuleType(modname)
        mod.__file__ = os.path.join(REF, relpath)
        exec(compile(src, mod.__file__, 'exec'), mod.__dict__)
        return mod

    ref_c2e = exec_patched('utils/cube_to_equi.py', 'ref_cube_to_equi')
    ref_cam = exec_patched('static_model/class_activation_model.py', 'ref_cam')

    def cpu_pads(module):
        for m in module.modules():
            if isinstance(m, ref_cp.CubePadding):
                m.use_gpu = False
        return module

    return dict(torch=torch, cp=ref_cp, resnet=ref_resnet, clstm=ref_clstm, e2c=ref_e2c,
                c2e=ref_c2e, cam=ref_cam, cpu_pads=cpu_pads)


# --------------------------------------------------------------------------- cubepad
CUBEPAD_SMALL = [(4, 1), (5, 2), (7, 1), (8, 3), (6, [1, 2, 3, 1]), (5, [0, 2, 1, 0]),
                 (5, [2, 0, 0, 3]), (6, [3, 1, 1, 2]), (4, [0, 0, 1, 1]), (4, [1, 1, 0, 0])]
CUBEPAD_HASHED = [(224, 3, 3), (56, 1, 64), (112, 1, 8), (14, 1, 32), (9, 2, 5)]   # (n, p, C): sha only
# the reference's own smoke test (model/cube_pad.py:256-261): CubePad(2) on [12, 64, 256, 256] -> [12, 64, 260, 260]; seeded
# values instead of its zeros so that the hash pins every element
CUBEPAD_SMOKE = (256, 2, 64, 2)                                                        # (n, p, C, cubes)


def cubepad_input(n, C, groups, seed):
    # integer-valued float32: any copy error changes the value, no rounding involved
    return hashrng.integers(seed, (6 * groups, C, n, n), 0, 1 << 20).astype(np.float32)


def gen_cubepad(R, out):
    torch = R['torch']
    small = {}
    for k, (n, p) in enumerate(CUBEPAD_SMALL):
        x = cubepad_input(n, 3, 2, 1000 + k)
        m = R['cpu_pads'](R['cp'].CubePad(p, use_gpu=False))
        y = m(torch.from_numpy(x)).numpy()
        small['x%d' % k] = x
        small['y%d' % k] = y
        small['pad%d' % k] = np.array(p if isinstance(p, list) else [p] * 4, dtype=np.int32)
    np.savez_compressed(os.path.join(out, 'cubepad_small.npz'), **small)
    hashed = {}
    for k, (n, p, C) in enumerate(CUBEPAD_HASHED):
        x = cubepad_input(n, C, 1, 2000 + k)
        m = R['cpu_pads'](R['cp'].CubePad(p, use_gpu=False))
        y = m(torch.from_numpy(x)).numpy()
        hashed['%d_%d_%d' % (n, p, C)] = sha(y)
    n, p, C, groups = CUBEPAD_SMOKE
    y = R['cpu_pads'](R['cp'].CubePad(p, use_gpu=False))(torch.from_numpy(cubepad_input(n, C, groups, 2100))).numpy()
    assert y.shape == (6 * groups, C, n + 2 * p, n + 2 * p)
    hashed['smoke_%d_%d_%d_x%d' % CUBEPAD_SMOKE] = sha(y)
    import json
    json.dump(hashed, open(os.path.join(out, 'cubepad_sha256.json'), 'w'), indent=1, sort_keys=True)
    print('cubepad fixtures written')


# --------------------------------------------------------------------------- e2c grids
E2C_CASES = [(960, 1920, 224), (1024, 2048, 224), (1024, 2048, 256), (2048, 4096, 512), (64, 128, 16)]


def gen_e2c(R, out):
    import json
    meta = {}
    arrs = {}
    for (H, W, cd) in E2C_CASES:
        img = np.zeros((H, W, 3))
        e = R['e2c'].Equi2Cube(cd, img)
        xs = np.stack(e.inXs).reshape(6, cd, cd).astype(np.float32)
        ys = np.stack(e.inYs).reshape(6, cd, cd).astype(np.float32)
        g = np.stack([xs, ys], axis=-1)                       # [6,cd,cd,2] fp32
        key = '%dx%d_%d' % (H, W, cd)
        meta[key] = {'sha256_f32': sha(g),
                     'sha256_f64': sha(np.stack([np.stack(e.inXs), np.stack(e.inYs)]))}
        step = max(1, cd // 8)
        arrs[key + '_rows'] = g[:, ::step, :, :]             # sub-sampled rows, all columns
        if cd <= 16:
            arrs[key + '_full'] = g
    np.savez_compressed(os.path.join(out, 'e2c_grids.npz'), **arrs)
    json.dump(meta, open(os.path.join(out, 'e2c_grids_sha256.json'), 'w'), indent=1, sort_keys=True)
    print('e2c grid fixtures written')


# --------------------------------------------------------------------------- c2e
def gen_c2e(R, out):
    torch = R['torch']
    arrs = {}
    for w in (4, 7, 8, 16):
        c = R['c2e'].Cube2Equi(w)
        arrs['face_map_%d' % w] = c.face_map.astype(np.int8)
        arrs['out_coord_%d' % w] = c.out_coord.astype(np.float64)
        arrs['M_%d' % w] = np.float32(np.max(c.out_coord.astype(np.float32)))
        x = hashrng.normal(3000 + w, (6, 5, w, w))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            y = c.to_equi_nn(torch.from_numpy(x)).detach().numpy()   # as run by torch 2.x: align_corners=False
        arrs['nn_in_%d' % w] = x
        arrs['nn_out_%d' % w] = y
    np.savez_compressed(os.path.join(out, 'c2e.npz'), **arrs)
    print('c2e fixtures written')


# --------------------------------------------------------------------------- resnet / CAM
def ref_resnet50(R, sd_np):
    torch = R['torch']
    model = R['resnet'].resnet50(pretrained=False)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing), (missing, unexpected)
    return R['cpu_pads'](model).eval()


def synth_cubes(seed, cd):
    """[6, cd, cd, 3] float32 HWC 'already normalised' cube batch (CAM's input)."""
    return hashrng.normal(seed, (6, cd, cd, 3), 0.0, 1.0)


def gen_resnet(R, out):
    torch = R['torch']
    sd = synth.resnet50_state(seed=1)
    model = ref_resnet50(R, sd)
    arrs = {}
    # small case (cd = 64 -> layer4 2x2): the reference's own avgpool(7)/fc cannot run
    # at this size (resnet_cubic.py:133,176-178), so its stem and layer modules are
    # called one by one, exactly as ResNet.forward does up to layer4 (:165-175).
    cubes = synth_cubes(4000 + 64, 64)
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(np.transpose(cubes, (0, 3, 1, 2))))
        x = model.relu(model.bn1(model.conv1(model.pad3(x))))
        x = model.maxpool(model.pad1(x))
        x = model.layer4(model.layer3(model.layer2(model.layer1(x))))
    arrs['layer4_s'] = x.numpy().astype(np.float32)
    # full case: the reference's CAM() end to end on the CPU path
    cubes = synth_cubes(4000 + 224, 224)
    score, feat, wsm = R['cam'].CAM(cubes, None, model, 'layer4', 'fc.weight', use_gpu=False)
    arrs['wsm_pick'] = wsm[::97, ::53].astype(np.float32).copy()   # before the restore: wsm aliases fc.weight
    # CAM() mutates fc.weight in place through the numpy view (survey a7); restore
    model.fc.weight.data.copy_(torch.from_numpy(sd['fc.weight']))
    arrs['cam_f'] = score.astype(np.float32)
    arrs['layer4_f'] = feat[:, ::8].astype(np.float32)
    arrs['layer4_sum_f'] = np.float64(np.sum(feat.astype(np.float64)))
    arrs['wsm_min'] = np.float32(np.min(sd['fc.weight']))
    np.savez_compressed(os.path.join(out, 'resnet_cam.npz'), **arrs)
    print('resnet/CAM fixtures written')


# --------------------------------------------------------------------------- clstm
def gen_clstm(R, out):
    torch = R['torch']
    arrs = {}
    # small cell, full tensors, 2 steps
    sd = synth.clstm_state(seed=7, input_size=8, hidden_size=8)
    cell = R['clstm'].ConvLSTMCell(8, 8)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    R['cpu_pads'](cell).eval()
    x = hashrng.uniform(5000, (12, 8, 4, 4))
    h0 = hashrng.uniform(5001, (12, 8, 4, 4))
    c0 = hashrng.uniform(5002, (12, 8, 4, 4))
    with torch.no_grad():
        h1, c1 = cell(torch.from_numpy(x), [torch.from_numpy(h0), torch.from_numpy(c0)])
        h2, c2 = cell(torch.from_numpy(x), [h1, c1])
    arrs.update(small_h1=h1.numpy(), small_c1=c1.numpy(), small_h2=h2.numpy(), small_c2=c2.numpy())
    # full-size cell: window semantics of test_temporal.py:57-85, T = 5 and T = 16
    sd = synth.clstm_state(seed=2, input_size=1000, hidden_size=1000)
    cell = R['clstm'].ConvLSTMCell(1000, 1000)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    R['cpu_pads'](cell).eval()
    c2e = R['c2e'].Cube2Equi(7)
    pick = hashrng.integers(5100, (1024,), 0, 6 * 1000 * 49)
    import warnings
    for T in (5, 16):
        frames = synth.cam_clip(6000 + T, T)
        mx, mn = np.max(frames), np.min(frames)
        init = (frames[0] - mn) / (mx - mn)
        hidden = torch.FloatTensor(init)
        cst = torch.FloatTensor(init)
        with torch.no_grad():
            for t in range(T):
                f = torch.FloatTensor((frames[t] - mn) / (mx - mn))
                hidden, cst = cell(f, [hidden, cst])
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                equi = torch.max(c2e.to_equi_nn(hidden), 1)[0].squeeze().numpy()
        arrs['full_map_T%d' % T] = equi.astype(np.float32)
        arrs['full_hidden_pick_T%d' % T] = hidden.numpy().reshape(-1)[pick]
        print('clstm window T=%d done' % T)
    arrs['pick'] = pick
    np.savez_compressed(os.path.join(out, 'clstm.npz'), **arrs)
    print('clstm fixtures written')


RESIZE_CASES = [((94, 188), (64, 128)), ((54, 96), (48, 96)), ((50, 100), (64, 128)), ((97, 131), (61, 203))]


def resize_input(k):
    (h, w), _ = RESIZE_CASES[k]
    from cp_360_weakly_supervised_saliency_amd.utils import hashrng
    return (hashrng.uniform(7000 + k, (h, w, 3), 0.0, 256.0)).astype(np.uint8)


def gen_resize(out):
    """dataset_feat_extractor.py:131-133 with the REAL Pillow of this image:
    Image.fromarray(frame).convert('RGB').resize((w, h), resample=Image.LANCZOS)."""
    import PIL
    from PIL import Image
    arrs = {'pillow_version': np.array(PIL.__version__)}
    for k, (_, (oh, ow)) in enumerate(RESIZE_CASES):
        a = resize_input(k)
        arrs['y%d' % k] = np.array(Image.fromarray(a).convert('RGB').resize((ow, oh), resample=Image.LANCZOS))
    np.savez_compressed(os.path.join(out, 'resize_lanczos.npz'), **arrs)


# --------------------------------------------------------------------------- metrics (f1)
METRIC_CASES = [((14, 28), (480, 960)), ((32, 64), (512, 1024)), ((120, 240), (120, 240)), ((16, 32), (100, 200))]


def metric_inputs(k):
    """Seeded (saliency, ground-truth) pair of case k: a hash-noise saliency map and a fixation
    map correlated with it (float32; the reference feeds float .npy maps, test_temporal.py:101-110)."""
    (h, w), (H, W) = METRIC_CASES[k]
    sal = hashrng.uniform(8000 + k, (h, w)) ** 2
    gt = synth.fixations_from_map(sal, 8100 + k, H, W)
    return sal.astype(np.float32), gt.astype(np.float32)


def gen_metrics(R, out):
    """utils/eval_saliency.py:14-190 executed from the reference with two in-memory shims: ``cv2.resize``
    = the oracle's restated INTER_LINEAR resize (cv2 is absent; the reference passes INTER_LANCZOS4 in the
    ``dst`` slot, so the default bilinear runs) and a stub for ``utils.cube_to_equi`` (a SyntaxError on
    Python >= 3.7, not used by the metric functions).  Pins the numpy arithmetic of AUC_Judd / CorrCoeff /
    similarity / AUC_Borji; the resize itself stays unpinned (third party)."""
    from oracle import o_metrics
    cv2 = sys.modules['cv2']
    cv2.INTER_LANCZOS4 = 4
    cv2.resize = lambda src, dsize, *a, **k: np.array(o_metrics.resize_linear(src, dsize), copy=True)
    sys.modules.setdefault('utils.cube_to_equi', R['c2e'])
    np.trapz = getattr(np, 'trapz', None) or np.trapezoid          # removed alias in numpy >= 2.? (shim)
    import importlib
    ev = importlib.import_module('utils.eval_saliency')
    arrs = {}
    for k in range(len(METRIC_CASES)):
        sal, gt = metric_inputs(k)
        np.random.seed(0)
        arrs['auc_judd_%d' % k] = np.float64(ev.AUC_Judd(sal.copy(), gt.copy()))
        arrs['cc_%d' % k] = np.float64(ev.CorrCoeff(sal.copy(), gt.copy()))
        arrs['sim_%d' % k] = np.float64(ev.similarity(sal.copy(), gt.copy()))
        np.random.seed(0)
        arrs['auc_borji_%d' % k] = np.float64(ev.AUC_Borji(sal.copy(), gt.copy(), Nsplits=10))
    np.savez_compressed(os.path.join(out, 'metrics.npz'), **arrs)
    print('metrics fixtures written', {k: float(v) for k, v in arrs.items()})


# --------------------------------------------------------------------------- overlay (f4)
OVERLAY_CASES = [((14, 28), (96, 192), 0.5), ((32, 64), (128, 256), 0.5), ((7, 9), (50, 61), 0.3)]


def overlay_inputs(k):
    (h, w), (H, W), alpha = OVERLAY_CASES[k]
    heat = (hashrng.uniform(8500 + k, (h, w)) ** 2).astype(np.float32)          # test_temporal.py:94 squares the map
    img = hashrng.uniform(8600 + k, (H, W, 3), 0.0, 256.0).astype(np.uint8)
    return img, heat, alpha


def gen_overlay(out):
    """utils/utils.py:9-25 executed from the reference (imports only PIL + matplotlib, both in this image).
    In-memory shim: ``Image.CUBIC = Image.BICUBIC`` (the alias was removed in Pillow 10; same filter)."""
    import importlib
    import warnings
    import matplotlib
    from PIL import Image
    matplotlib.use('Agg')
    if not hasattr(Image, 'CUBIC'):
        Image.CUBIC = Image.BICUBIC
    warnings.simplefilter('ignore')
    sys.path.insert(0, REF)
    uu = importlib.import_module('utils.utils')
    arrs = {}
    for k in range(len(OVERLAY_CASES)):
        img, heat, alpha = overlay_inputs(k)
        arrs['y%d' % k] = np.array(uu.overlay(img.copy(), heat.copy(), alpha=alpha))
    np.savez_compressed(os.path.join(out, 'overlay.npz'), **arrs)
    print('overlay fixtures written')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='cubepad,e2c,c2e,resnet,clstm,resize,metrics,overlay')
    args = ap.parse_args()
    parts = args.only.split(',')
    if 'resize' in parts:                      # Pillow only: the reference itself is not needed
        gen_resize(HERE)
        parts.remove('resize')
    if 'overlay' in parts:
        gen_overlay(HERE)
        parts.remove('overlay')
    if not parts:
        return
    R = import_reference()
    R['torch'].set_num_threads(8)
    for part in parts:
        {'cubepad': gen_cubepad, 'e2c': gen_e2c, 'c2e': gen_c2e,
         'resnet': gen_resnet, 'clstm': gen_clstm, 'metrics': gen_metrics}[part](R, HERE)


if __name__ == '__main__':
    main()
