"""Oracle-side compositions shared by the GPU parity tests, smoke() and the
cpu_baseline leg of bench.py.  Imports oracle/ - never imported by the product."""
import numpy as np
import torch

from oracle import o_e2c, o_resnet, o_clstm, o_c2e


def sd_t(sd):
    return {k: (v if torch.is_tensor(v) else torch.from_numpy(v)) for k, v in sd.items()}


def oracle_cubes(frame_u8, cube_dim, grids=None, fixed_point=True):
    """dataset_feat_extractor.py:138-157: u8 frame -> /255 (float64) -> to_cube ->
    im_norm -> float32 [6, 3, cd, cd]."""
    img = np.array(frame_u8) / 255.0
    cubes = o_e2c.to_cube(img, cube_dim, fixed_point=fixed_point, grids=grids)
    return o_e2c.im_norm_batch(cubes)


def oracle_cam_frames(clip_u8, resnet_sd, cube_dim):
    """[T, H, W, 3] u8 -> cube_feat [T, 6, 1000, h, w] float32 (static stage)."""
    sd = sd_t(resnet_sd)
    H, W = clip_u8.shape[1:3]
    grids = o_e2c.grids_f32(cube_dim, H, W)
    out = []
    for t in range(clip_u8.shape[0]):
        score, _ = o_resnet.cam_from_cubes(oracle_cubes(clip_u8[t], cube_dim, grids), sd)
        out.append(score.astype(np.float32))
    return np.stack(out)


def oracle_pipeline(clip_u8, resnet_sd, clstm_sd, cube_dim, align_corners=False, return_all=False):
    """One clip = one window: frames -> saliency [2w, 4w] float32."""
    cams = oracle_cam_frames(clip_u8, resnet_sd, cube_dim)
    hid = o_clstm.window_hidden(cams, sd_t(clstm_sd))
    sal = o_c2e.saliency_from_hidden(hid, align_corners=align_corners)
    return (sal, cams, hid) if return_all else sal


def cubepad_sweep(dev):
    """Seeded geometry sweep of the stand-alone NCHW CubePad against the oracle (see
    tests/test_gpu_parity.py::test_cubepad_nchw_randomised_geometry_sweep)."""
    from oracle import o_cubepad
    from cp_360_weakly_supervised_saliency_amd.model.cube_pad import CubePad
    rng = np.random.RandomState(20260)
    dts = [torch.uint8, torch.int16, torch.int32, torch.int64]
    npd = {torch.uint8: np.uint8, torch.int16: np.int16, torch.int32: np.int32, torch.int64: np.int64}
    cases = []
    for _ in range(60):
        n = int(rng.choice([1, 2, 3, 5, 7, 8, 14, 16, 28, 31, 32, 33, 56, 57, 64, 112, 113, 130]))
        pad = [int(min(n, v)) for v in rng.randint(0, 5, size=4)]
        C = int(rng.randint(1, 24))
        n6 = 6 * int(rng.randint(1, 3))
        dt = dts[int(rng.randint(0, 4))]
        off = int(rng.randint(0, 4))
        cases.append((n, pad, C, n6, dt, off))
    cases += [(56, [1, 1, 1, 1], 64, 12, torch.int16, 0), (57, [2, 0, 1, 3], 50, 12, torch.int16, 1),
              (112, [1, 1, 1, 1], 48, 12, torch.int16, 0), (113, [3, 3, 3, 3], 17, 36, torch.int32, 0),
              (224, [3, 3, 3, 3], 3, 192, torch.int32, 0), (130, [0, 4, 2, 0], 25, 24, torch.uint8, 3),
              (40, [1, 2, 0, 1], 33, 18, torch.int64, 0), (64, [4, 4, 4, 4], 16, 36, torch.int32, 2),
              (256, [1, 1, 1, 1], 16, 36, torch.int16, 0), (28, [1, 1, 1, 1], 128, 24, torch.int16, 0),
              (14, [1, 1, 1, 1], 256, 12, torch.int16, 0), (7, [1, 1, 1, 1], 500, 12, torch.int32, 0),
              (120, [1, 1, 1, 1], 32, 12, torch.int16, 0), (200, [2, 1, 0, 3], 20, 24, torch.uint8, 1)]
    for case, (n, pad, C, n6, dt, off) in enumerate(cases):
        x = rng.randint(0, 120, size=(n6, C, n, n)).astype(npd[dt])
        buf = torch.zeros(x.size + off, dtype=dt, device=dev)
        buf[off:] = torch.from_numpy(x).reshape(-1).to(dev)
        want = o_cubepad.cubepad(x, pad)
        got = CubePad(pad)(buf[off:].view(n6, C, n, n)).cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (case, n, pad, C, n6, dt, off)
