"""BASELINE.json configs C2 / C3 / C4 at the resolution they are quoted on (1024x2048 equirectangular,
cube 224), HIP path vs the oracle:

  C2  1 frame, fp32, static path only (equi -> cube -> CubePad ResNet-50 -> CAM):
      window-normalised CAM within 1e-3.
  C3  one 16-frame clip, bf16 (and fp32: saliency within 1e-3 abs).
  C4  the per-GPU shard of config C4: 4 clips x 16 frames batched through one engine, bf16 -
      every clip against the oracle run on that clip alone.

"bf16" is the engine's default 16-bit mode: ConvLSTM (81 % of the flops) in bf16, static stage in fp16 (same
MFMA rate; a bf16 ResNet alone moves the map by 3.4e-3 and CC by 1.2e-3, tests/probe_precision_split.py).
The all-bf16 engine (static_precision='bf16') is run too, against the looser bound it actually meets.

The 16-bit gate follows SURVEY 8(d): AUC-Judd and CC of the build's map against a fixation map
within 1e-3 of the same metrics of the oracle's map - with fixations SAMPLED FROM THE ORACLE MAP
(synth.fixations_from_map), so the oracle scores well above chance and a wrong map cannot pass by
both being at chance level - and additionally CC(build, oracle) >= 0.9999.
Follows /root/reference/temporal_model/test_temporal.py:57-110 (window, c2e, metrics).
"""
import numpy as np
import pytest
import torch

from oracle import o_metrics
from cp_360_weakly_supervised_saliency_amd.utils import synth
from tests import parity_helpers as ph

pytestmark = pytest.mark.gpu

H, W, CD, T, B = 1024, 2048, 224, 16, 4


def _metrics(m, fix):
    return (o_metrics.auc_judd(m, fix, rng=np.random.RandomState(0)), o_metrics.corr_coeff(m, fix))


def gate_16bit(sal, ref, seed, label):
    """|dAUC-Judd| <= 1e-3, |dCC| <= 1e-3 against oracle-correlated fixations, CC(build, oracle) >= 0.9999."""
    fix = synth.fixations_from_map(ref, seed, H // 2, W // 2)
    auc_r, cc_r = _metrics(ref, fix)
    auc, cc = _metrics(sal, fix)
    cc_bo = o_metrics.corr_coeff(sal, ref)
    print('%s: oracle AUC-Judd %.4f CC %.4f | dAUC %+.2e dCC %+.2e CC(build,oracle) %.6f max|d| %.2e'
          % (label, auc_r, cc_r, auc - auc_r, cc - cc_r, cc_bo, float(np.max(np.abs(sal - ref)))))
    assert auc_r > 0.7 and cc_r > 0.1, "fixations must be informative for the oracle map"
    assert abs(auc - auc_r) <= 1e-3, (label, auc, auc_r)
    assert abs(cc - cc_r) <= 1e-3, (label, cc, cc_r)
    assert cc_bo >= 0.9999, (label, cc_bo)


@pytest.fixture(scope='module')
def shard():
    """4 clips x 16 frames of 1024x2048 and the oracle's result for each clip (64 oracle frames)."""
    rs = synth.resnet50_state(seed=1)
    cs = synth.clstm_state(seed=2)
    clips = np.stack([synth.clip_u8(3 + b, T, H, W) for b in range(B)])          # bench.py's clips of rank 0
    refs = [ph.oracle_pipeline(clips[b], rs, cs, CD, return_all=True) for b in range(B)]
    return dict(rs=rs, cs=cs, clips=clips, refs=refs)


def test_c2_one_frame_fp32_static(shard):
    from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
    s = shard
    eng = SaliencyEngine(s['rs'], s['cs'], (H, W), CD, clips=1, frames=1, precision='fp32')
    frame = torch.from_numpy(s['clips'][0, :1]).cuda()
    with torch.no_grad():
        cam = eng.static_stage(frame).cpu().numpy()[0, 0]                        # [294, 1000] NHWC
    want = s['refs'][0][1][0].transpose(0, 2, 3, 1).reshape(294, 1000)          # oracle frame 0 [6,1000,7,7]
    mn, mx = want.min(), want.max()
    err = np.max(np.abs((cam - mn) / (mx - mn) - (want - mn) / (mx - mn)))
    print('C2 fp32 window-normalised CAM max|d| %.2e, relative %.2e' % (err, np.max(np.abs(cam - want)) / np.max(np.abs(want))))
    assert err <= 1e-3
    assert np.max(np.abs(cam - want)) <= 1e-4 * np.max(np.abs(want))


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_c3_one_clip_16_frames(shard, prec):
    from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
    s = shard
    eng = SaliencyEngine(s['rs'], s['cs'], (H, W), CD, clips=1, frames=T, precision=prec)
    sal = eng(torch.from_numpy(s['clips'][:1]).cuda()).cpu().numpy()[0]
    ref = s['refs'][0][0]
    assert sal.shape == ref.shape == (14, 28)
    if prec == 'fp32':
        assert np.max(np.abs(sal - ref)) <= 1e-3                                  # the north-star fp32 bound
    gate_16bit(sal, ref, 200, 'C3 %s' % prec)


def test_c3_all_bf16_static_stage_too(shard):
    """bf16 in BOTH stages (not the default): the bf16 ResNet moves the map ~8x more than fp16 does; it
    stays within 1e-3 on AUC-Judd but not on CC - recorded here with the bounds it meets."""
    from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
    s = shard
    eng = SaliencyEngine(s['rs'], s['cs'], (H, W), CD, clips=1, frames=T, precision='bf16', static_precision='bf16')
    sal = eng(torch.from_numpy(s['clips'][:1]).cuda()).cpu().numpy()[0]
    ref = s['refs'][0][0]
    fix = synth.fixations_from_map(ref, 200, H // 2, W // 2)
    (auc_r, cc_r), (auc, cc) = _metrics(ref, fix), _metrics(sal, fix)
    print('C3 all-bf16: dAUC %+.2e dCC %+.2e CC(build,oracle) %.6f max|d| %.2e'
          % (auc - auc_r, cc - cc_r, o_metrics.corr_coeff(sal, ref), float(np.max(np.abs(sal - ref)))))
    assert abs(auc - auc_r) <= 1e-3 and abs(cc - cc_r) <= 3e-3
    assert o_metrics.corr_coeff(sal, ref) >= 0.9995 and np.max(np.abs(sal - ref)) <= 1e-2


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_c4_shard_4_clips_x_16_frames_batched(shard, prec):
    from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
    s = shard
    eng = SaliencyEngine(s['rs'], s['cs'], (H, W), CD, clips=B, frames=T, precision=prec)
    sal = eng(torch.from_numpy(s['clips']).cuda()).cpu().numpy()
    assert sal.shape == (B, 14, 28)
    for b in range(B):
        ref = s['refs'][b][0]
        if prec == 'fp32':
            assert np.max(np.abs(sal[b] - ref)) <= 1e-3, b
        gate_16bit(sal[b], ref, 210 + b, 'C4 shard %s clip %d' % (prec, b))
    if prec == 'fp32':
        # the static stage of the whole shard (64 frames batched): window-normalised CAM per clip
        cam = eng.cam.cpu().numpy()                                               # [B, T, 294, 1000]
        for b in range(B):
            want = s['refs'][b][1].transpose(0, 1, 3, 4, 2).reshape(T, 294, 1000)
            mn, mx = want.min(), want.max()
            assert np.max(np.abs((cam[b] - mn) / (mx - mn) - (want - mn) / (mx - mn))) <= 1e-3, b


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_c2_variant_cube_256_pipeline(prec):
    """SURVEY 8, "C2 ... cd=256 optional variant" (the reference's only smoke test uses 256-pixel faces, model/cube_pad.py:256-261):
    a 1024x2048 clip through cube 256 -> layer4 8x8 -> ConvLSTM at 8x8 faces -> 16x32 map, against the oracle end to end.
    None of the fused static-stage kernels is specialised to this geometry: the static stage takes the per-convolution path
    (cp360_resnet_plan_describe says so).  The ConvLSTM of ONE clip at 8x8 faces runs the HALF variant of the clip-resident kernel
    (conv_clip_kernel<T, 2>: half a cube per tile, the whole cube resident); four clips would run in the Winograd domain
    (tests/test_wino.py::test_wino_cell_window_matches_oracle[bf16-8-4])."""
    from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
    cd, t = 256, 3
    rs = synth.resnet50_state(seed=1)
    cs = synth.clstm_state(seed=2)
    clip = synth.clip_u8(41, t, H, W)
    ref = ph.oracle_pipeline(clip, rs, cs, cd)
    eng = SaliencyEngine(rs, cs, (H, W), cd, clips=1, frames=t, precision=prec)
    sal = eng(torch.from_numpy(clip[None]).cuda()).cpu().numpy()[0]
    assert sal.shape == ref.shape == (16, 32)
    err = float(np.max(np.abs(sal - ref)))
    print('cube 256 %s: map max|d| %.2e, CC(build, oracle) %.6f' % (prec, err, o_metrics.corr_coeff(sal, ref)))
    if prec == 'fp32':
        assert err <= 1e-3                                                        # the north-star fp32 bound
    else:
        assert err <= 5e-3 and o_metrics.corr_coeff(sal, ref) >= 0.9999
    from cp_360_weakly_supervised_saliency_amd import stage_ctx
    if stage_ctx.USE_CTX:                                                          # (CP360_CTX=0 plans in Python: no context to ask)
        plan = eng.resnet.__dict__['_stage'].describe(6 * t, cd)
        assert 'layer1: GENERIC path' in plan and 'layer3.1-5: GENERIC path' in plan
