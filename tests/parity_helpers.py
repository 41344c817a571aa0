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
