"""Hand-off file formats: layouts and names of the reference's .npy artefacts."""
import os

import numpy as np

from cp_360_weakly_supervised_saliency_amd.utils import npy_io, synth


def test_cube_feat_round_trip(tmp_path):
    T, w, C = 3, 7, 1000
    frames = synth.cam_clip(6500, T)                       # [T, 6, C, w, w] as the reference stores them
    cam = np.stack([npy_io.cube_feat_to_cam_nhwc(f) for f in frames])
    assert cam.shape == (T, 6 * w * w, C)
    # engine pixel order = face-major, row-major inside a face; channels innermost
    assert cam[1, 2 * 49 + 3 * 7 + 4, 17] == frames[1, 2, 17, 3, 4]
    npy_io.save_cube_feats(str(tmp_path / 'vid'), cam, w)
    names = sorted(os.listdir(tmp_path / 'vid' / 'cube_feat'))
    assert names == ['000002.npy', '000003.npy', '000004.npy']        # dataset_feat_extractor.py:187-189
    one = np.load(tmp_path / 'vid' / 'cube_feat' / '000003.npy')
    assert one.dtype == np.float32 and one.shape == (6, C, w, w) and np.array_equal(one, frames[1])
    back = npy_io.load_cube_feat_window(str(tmp_path / 'vid'), 2, T)
    assert np.array_equal(back, cam)


def test_saliency_file_name(tmp_path):
    sal = np.arange(14 * 28, dtype=np.float64).reshape(14, 28)
    npy_io.save_saliency(str(tmp_path), 'clip7', 12, sal)
    got = np.load(tmp_path / 'clip7' / '00012.npy')                    # test_temporal.py:86-88
    assert got.dtype == np.float32 and np.array_equal(got, sal.astype(np.float32))
