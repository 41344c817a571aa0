"""On-disk hand-off formats of the reference, so artefacts written by either side
can be consumed by the other.  Host-side numpy only (file I/O is not on the hot path: the fused engine
keeps these tensors in HBM).

* ``<out>/<video>/cube_feat/{:06}.npy`` - float32 [6, 1000, h, w]: the CAM scores of one frame,
  written by static_model/dataset_feat_extractor.py:187-189 (frame counter starts at 2 because of
  its :126-137 bookkeeping) and read back by temporal_model/test_temporal.py:66-78.
* ``<out>/<video>/{:05}.npy`` - float32 [2w, 4w]: the temporal model's equirectangular saliency map of
  the window ending at that frame (test_temporal.py:86-88); ground-truth fixation maps use the same
  naming under ``Wild360_GT/<video>.mp4/`` (:101-102).
"""
import os

import numpy as np


def cube_feat_path(video_dir, frame_no):
    return os.path.join(video_dir, 'cube_feat', '{0:06}.npy'.format(int(frame_no)))


def saliency_path(out_dir, video, frame_no):
    return os.path.join(out_dir, video, '{:05}.npy'.format(int(frame_no)))


def cam_nhwc_to_cube_feat(cam_frame, w):
    """Engine layout [6*w*w, C] (face-major pixels, channels innermost) -> the reference's [6, C, w, w]."""
    a = np.asarray(cam_frame, dtype=np.float32)
    return np.ascontiguousarray(a.reshape(6, w, w, a.shape[-1]).transpose(0, 3, 1, 2))


def cube_feat_to_cam_nhwc(cube_feat):
    """The reference's [6, C, w, w] -> engine layout [6*w*w, C]."""
    a = np.asarray(cube_feat, dtype=np.float32)
    if a.ndim != 4 or a.shape[0] != 6 or a.shape[2] != a.shape[3]:
        raise ValueError("cube_feat must be [6, C, w, w], got %s" % (a.shape,))
    return np.ascontiguousarray(a.transpose(0, 2, 3, 1)).reshape(6 * a.shape[2] * a.shape[3], a.shape[1])


def save_cube_feats(video_dir, cam, w, first_frame_no=2):
    """cam [T, 6*w*w, C] (one clip of the engine's buffer, host or device) -> one .npy per frame."""
    cam = cam.detach().cpu().numpy() if hasattr(cam, 'detach') else np.asarray(cam)
    os.makedirs(os.path.join(video_dir, 'cube_feat'), exist_ok=True)
    for t in range(cam.shape[0]):
        np.save(cube_feat_path(video_dir, first_frame_no + t), cam_nhwc_to_cube_feat(cam[t], w))


def load_cube_feat_window(video_dir, first_frame_no, T):
    """T consecutive cube_feat files -> float32 [T, 6*w*w, C], the input of ``ClipRunner.run``
    (after ``torch.from_numpy(...)[None].cuda()``); window semantics: test_temporal.py:57-65."""
    return np.stack([cube_feat_to_cam_nhwc(np.load(cube_feat_path(video_dir, first_frame_no + t))) for t in range(T)])


def save_saliency(out_dir, video, frame_no, sal):
    sal = sal.detach().cpu().numpy() if hasattr(sal, 'detach') else np.asarray(sal)
    os.makedirs(os.path.join(out_dir, video), exist_ok=True)
    np.save(saliency_path(out_dir, video, frame_no), sal.astype(np.float32))
