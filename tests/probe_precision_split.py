"""Where does the 16-bit error of the saliency map come from?  One 1024x2048 T=16 clip through the
oracle (fp32 CPU) and through the HIP path with the static stage (ResNet + CAM) and the temporal stage
(ConvLSTM) in separately chosen arithmetic types.  Prints max|d|, mean d, CC(build, oracle), dAUC, dCC."""
import os, sys, json
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
from cp_360_weakly_supervised_saliency_amd.utils import synth
from tests.parity_helpers import oracle_pipeline
from oracle import o_metrics

torch.set_num_threads(32)
H, W, cd, T = 1024, 2048, 224, int(os.environ.get('T', 16))
rs, cs = synth.resnet50_state(seed=1), synth.clstm_state(seed=2)
clip = synth.clip_u8(3, T, H, W)
ref = oracle_pipeline(clip, rs, cs, cd)
fix = synth.fixations_from_map(ref, 200, H // 2, W // 2)
m = lambda x: (o_metrics.auc_judd(x, fix, rng=np.random.RandomState(0)), o_metrics.corr_coeff(x, fix))
a0, c0 = m(ref)
print('oracle: AUC %.4f CC %.4f range [%.4f, %.4f]' % (a0, c0, ref.min(), ref.max()))
frames = torch.from_numpy(clip[None]).cuda()
for sp, tp in (('fp32', 'fp32'), ('bf16', 'bf16'), ('fp32', 'bf16'), ('bf16', 'fp32'), ('fp16', 'fp16'), ('fp16', 'bf16'), ('bf16', 'fp16')):
    eng = SaliencyEngine(rs, cs, (H, W), cd, clips=1, frames=T, precision=tp, static_precision=sp)
    sal = eng(frames).cpu().numpy()[0]
    a, c = m(sal)
    d = sal - ref
    print('static %s temporal %s: max|d| %.2e mean d %+.2e std d %.2e CC(b,o) %.6f dAUC %+.2e dCC %+.2e'
          % (sp, tp, np.abs(d).max(), d.mean(), d.std(), o_metrics.corr_coeff(sal, ref), a - a0, c - c0))
    del eng
    torch.cuda.empty_cache()
