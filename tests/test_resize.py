"""Frame resize: dataset_feat_extractor.py:131-133 resizes every frame with PIL's
LANCZOS filter.  The oracle restatement is pinned against Pillow itself (golden fixture made with the real
library + a live comparison when Pillow is importable); the HIP kernels are bit-exact against both."""
import os

import numpy as np
import pytest
import torch

from oracle import o_resize
from cp_360_weakly_supervised_saliency_amd.utils import hashrng
from cp_360_weakly_supervised_saliency_amd.utils.resize import LanczosResize, lanczos_tables
from tests.golden import make_golden as mg


def test_oracle_resize_matches_pillow_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, 'resize_lanczos.npz'))
    for k, (_, out_hw) in enumerate(mg.RESIZE_CASES):
        got = o_resize.resize_lanczos_u8(mg.resize_input(k), out_hw)
        assert got.dtype == np.uint8 and np.array_equal(got, z['y%d' % k]), k


def test_oracle_resize_matches_live_pillow():
    Image = pytest.importorskip('PIL.Image')
    for k, ((h, w), (oh, ow)) in enumerate([((120, 250), (64, 128)), ((33, 77), (99, 50)), ((40, 40), (40, 17))]):
        a = hashrng.uniform(7100 + k, (h, w, 3), 0.0, 256.0).astype(np.uint8)
        want = np.array(Image.fromarray(a).convert('RGB').resize((ow, oh), resample=Image.LANCZOS))
        assert np.array_equal(o_resize.resize_lanczos_u8(a, (oh, ow)), want)


@pytest.mark.parametrize('sizes', [(750, 512), (376, 256), (200, 256), (2160, 960), (3840, 1920), (7, 3), (5, 5)])
def test_host_tables_equal_oracle(sizes):
    """cp360_resize_coeffs_host (C, libm) == the oracle's tables (python floats), integer for integer."""
    bo, ko = o_resize.precompute_coeffs(*sizes)
    bl, kl = lanczos_tables(*sizes)
    assert np.array_equal(bo, bl) and np.array_equal(ko, kl)
    assert np.all(ko.sum(1) >= (1 << 22) - 8) and np.all(ko.sum(1) <= (1 << 22) + 8)     # weights sum to 1


@pytest.mark.gpu
def test_gpu_resize_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, 'resize_lanczos.npz'))
    for k, ((h, w), out_hw) in enumerate(mg.RESIZE_CASES):
        a = mg.resize_input(k)
        frames = torch.from_numpy(np.stack([a, a[::-1].copy()])).cuda()          # F = 2
        got = LanczosResize((h, w), out_hw)(frames).cpu().numpy()
        assert np.array_equal(got[0], z['y%d' % k]), k
        assert np.array_equal(got[1], o_resize.resize_lanczos_u8(a[::-1].copy(), out_hw)), k


@pytest.mark.gpu
def test_gpu_resize_full_size_against_pillow_or_oracle():
    """1080x2160 -> 960x1920 (the reference's cfg.equi_w x cfg.equi_h target), one frame."""
    a = hashrng.uniform(7200, (1080, 2160, 3), 0.0, 256.0).astype(np.uint8)
    got = LanczosResize((1080, 2160), (960, 1920))(torch.from_numpy(a[None]).cuda()).cpu().numpy()[0]
    try:
        from PIL import Image
        want = np.array(Image.fromarray(a).convert('RGB').resize((1920, 960), resample=Image.LANCZOS))
    except ImportError:
        want = o_resize.resize_lanczos_u8(a, (960, 1920))
    assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        LanczosResize((1080, 2160), (960, 1920))(torch.zeros((1, 10, 10, 3), dtype=torch.uint8).cuda())


@pytest.mark.gpu
def test_pipeline_with_source_resize_matches_oracle():
    """Decoded 300x600 frames -> K0 resize to 256x512 -> the rest of the path, against the oracle fed with
    the oracle-resized frames (fp32, 1e-3 on the saliency map)."""
    from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
    from cp_360_weakly_supervised_saliency_amd.utils import synth
    from tests.parity_helpers import oracle_pipeline
    Hs, Ws, H, W, cd, T = 300, 600, 256, 512, 64, 2
    rs, cs = synth.resnet50_state(seed=1), synth.clstm_state(seed=3)
    clip = synth.clip_u8(60, T, Hs, Ws)
    small = np.stack([o_resize.resize_lanczos_u8(f, (H, W)) for f in clip])
    ref = oracle_pipeline(small, rs, cs, cd)
    eng = SaliencyEngine(rs, cs, (H, W), cd, clips=1, frames=T, precision='fp32', source_hw=(Hs, Ws))
    sal = eng(torch.from_numpy(clip[None]).cuda()).cpu().numpy()[0]
    assert np.max(np.abs(sal - ref)) <= 1e-3
