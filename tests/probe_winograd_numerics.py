"""CPU gate for a Winograd F(2x2, 3x3) ConvLSTM cell in 16-bit arithmetic (round-5 brief, item 1a).

The oracle's T-frame window (oracle/o_clstm.py) is run three ways on the CAM tensors of the bench clips:
  f32      the oracle itself
  direct   the cell as the HIP path computes it today: x / h / Conv1 / Conv2 outputs rounded to the 16-bit type,
           weights rounded once, f32 accumulation, gates and the cell state in f32
  wino     the same cell with every CubePad(1)+3x3 convolution as F(2x2, 3x3): U = G g G^T from the f32 weights
           rounded ONCE to the 16-bit type, V = B^T d B from the 16-bit activations rounded to the 16-bit type,
           sixteen f32-accumulated products, Y = A^T m A + bias in f32
and the three saliency maps go through tests/test_configs_1024.py's gate (|dAUC-Judd|, |dCC| <= 1e-3 against
fixations sampled from the oracle's map, CC(build, oracle) >= 0.9999).  No GPU.
Follows /root/reference/model/clstm.py:42-82 and /root/reference/temporal_model/test_temporal.py:57-85.

    python tests/probe_winograd_numerics.py [bf16|fp16] [T] [clips]
"""
import os
import sys
import time
import numpy as np
import torch
import torch.nn.functional as Fn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cp_360_weakly_supervised_saliency_amd.utils import synth          # noqa: E402
from tests import parity_helpers as ph                                  # noqa: E402
from oracle import o_metrics, o_c2e                                     # noqa: E402
from oracle.o_resnet import cubepad_t                                   # noqa: E402

_POS = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith('--') and not sys.argv[i].startswith('--')]
DT = {'bf16': torch.bfloat16, 'fp16': torch.float16}[_POS[0] if len(_POS) > 0 else 'bf16']
T = int(_POS[1]) if len(_POS) > 1 else 16
NCLIP = int(_POS[2]) if len(_POS) > 2 else 2
H, W, CD = 1024, 2048, 224

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def rnd(x):
    return x.to(DT).float()


def conv_direct(x, wq, b):
    """x [6,C,w,w] already 16-bit valued; wq rounded weights; f32 accumulate."""
    return Fn.conv2d(cubepad_t(x, 1), wq, b)


def wino_weights(w):
    return rnd(torch.einsum('ij,ocjk,lk->ilco', G, w, G)).contiguous()   # [4,4,Cin,Cout]


def conv_wino(x, U, b):
    n, c, w, _ = x.shape
    xp = cubepad_t(x, 1)                                                  # [n,c,w+2,w+2]
    tw = (w + 1) // 2
    need = 2 * tw + 2
    xp = Fn.pad(xp, (0, need - (w + 2), 0, need - (w + 2)))               # zero rows / columns beyond the cube padding
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                # [n,c,tw,tw,4,4]
    V = rnd(torch.einsum('ij,nctsjk,lk->ilntsc', BT, d, BT))              # [4,4,n,tw,tw,c]
    M = torch.matmul(V.reshape(4, 4, n * tw * tw, c), U)                  # [4,4,tiles,Cout] f32
    Y = torch.einsum('ij,jkto,lk->toil', AT, M, AT)                       # [tiles,Cout,2,2]
    Y = Y.reshape(n, tw, tw, -1, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(n, -1, 2 * tw, 2 * tw)
    return Y[:, :, :w, :w] + b.view(1, -1, 1, 1)


def step(x, hidden, cell, conv, ws, bs, q):
    out = torch.cat((q(x), q(hidden)), 1)
    out = q(Fn.relu(conv(out, ws[0], bs[0])))
    out = q(Fn.relu(conv(out, ws[1], bs[1])))
    gates = conv(out, ws[2], bs[2])
    i_g, f_g, o_g, c_g = gates.chunk(4, 1)
    i_g, f_g, o_g = torch.sigmoid(i_g), torch.sigmoid(f_g), torch.sigmoid(o_g)
    cell = f_g * cell + i_g * torch.tanh(c_g)
    return o_g * torch.tanh(cell), cell


def window(frames, conv, ws, bs, q):
    mx, mn = np.max(frames), np.min(frames)
    nf = [torch.from_numpy(((f - mn) / (mx - mn)).astype(np.float32)) for f in frames]
    hidden, cell = nf[0].clone(), nf[0].clone()
    for f in nf:
        hidden, cell = step(f, hidden, cell, conv, ws, bs, q)
    return o_c2e.saliency_from_hidden(hidden.numpy())


def main():
    """python tests/probe_winograd_numerics.py [bf16|fp16] [T] [clips] [--seeds 2,5] [--threads N] [--out profiles/r06_wino_margin.md]
    --seeds: ConvLSTM weight seeds (synth.clstm_state); every (seed, clip) pair is one sample of the distribution written to --out."""
    argv = sys.argv[1:]
    opt = {}
    for k in ('--seeds', '--threads', '--out'):
        if k in argv:
            opt[k] = argv[argv.index(k) + 1]
    seeds = [int(v) for v in opt.get('--seeds', '2').split(',')]
    torch.set_num_threads(int(opt.get('--threads', os.cpu_count())))
    rs = synth.resnet50_state(seed=1)
    names = ('Conv1', 'Conv2', 'Gates')
    ident = lambda t: t
    rows = []                                   # (seed, clip, label, max|d|, rms, cc_bo, dAUC, dCC)
    cam_cache = {}
    with torch.no_grad():
        for seed in seeds:
            cs = ph.sd_t(synth.clstm_state(seed=seed))
            w32 = [cs[n + '.weight'] for n in names]
            bs = [cs[n + '.bias'] for n in names]
            wq = [rnd(w) for w in w32]
            U = [wino_weights(w) for w in w32]
            for b in range(NCLIP):
                t0 = time.time()
                if b not in cam_cache:
                    cam_cache[b] = ph.oracle_cam_frames(synth.clip_u8(3 + b, T, H, W), rs, CD)
                cams = cam_cache[b]
                ref = window(cams, conv_direct, w32, bs, ident)
                fix = synth.fixations_from_map(ref, 210 + b, H // 2, W // 2)
                m = lambda x: (o_metrics.auc_judd(x, fix, rng=np.random.RandomState(0)), o_metrics.corr_coeff(x, fix))
                a0, c0 = m(ref)
                print('seed %d clip %d: oracle AUC %.4f CC %.4f map range [%.4f, %.4f] (%.0f s)' % (seed, b, a0, c0, ref.min(), ref.max(), time.time() - t0), flush=True)
                for label, conv, ws in (('direct', conv_direct, wq), ('wino  ', conv_wino, U)):
                    sal = window(cams, conv, ws, bs, rnd)
                    a, c = m(sal)
                    d = sal - ref
                    row = (seed, b, label.strip(), float(np.abs(d).max()), float(np.sqrt((d * d).mean())), float(o_metrics.corr_coeff(sal, ref)), a - a0, c - c0)
                    rows.append(row)
                    print('  %s %s: max|d| %.2e rms %.2e CC(b,o) %.6f dAUC %+.2e dCC %+.2e'
                          % (label, str(DT)[6:], row[3], row[4], row[5], row[6], row[7]), flush=True)
    if '--out' in opt:
        write_report(opt['--out'], rows, seeds)


def write_report(path, rows, seeds):
    q = lambda v, p: float(np.percentile(np.abs(v), p))
    with open(path, 'w') as f:
        f.write('# Winograd F(2x2,3x3) ConvLSTM cell in %s against the direct form: distribution of the saliency-map error (CPU emulation)\n\n' % str(DT)[6:])
        f.write('`python tests/probe_winograd_numerics.py %s %d %d --seeds %s`: the oracle\'s T = %d window (oracle/o_clstm.py semantics, '
                'model/clstm.py:42-82, temporal_model/test_temporal.py:57-85) on the oracle CAM tensors of %d synthetic 1024x2048 clips x '
                '%d ConvLSTM weight seeds = %d samples per form; `direct` = the cell as the direct HIP kernels compute it (operands rounded to '
                'the 16-bit type, f32 sums), `wino` = U = G g G^T and V = B^T d B rounded ONCE to the 16-bit type, sixteen f32 products, '
                'Y = A^T M A in f32 (csrc/wino.hip).  Reference of every delta: the f32 oracle window; AUC-Judd / CC against fixations sampled '
                'from the oracle map.  Gate of the GPU tests: |dAUC|, |dCC| <= 1e-3, map max|d| <= 1e-3 only for fp32.\n\n'
                % (str(DT)[6:], T, NCLIP, ','.join(map(str, seeds)), T, NCLIP, len(seeds), NCLIP * len(seeds)))
        f.write('| form | samples | map max\\|d\\|: median / p95 / max | \\|dAUC-Judd\\|: median / p95 / max | \\|dCC\\|: median / p95 / max | min CC(build, oracle) |\n|---|---|---|---|---|---|\n')
        for label in ('direct', 'wino'):
            r = [x for x in rows if x[2] == label]
            md, da, dc = np.array([x[3] for x in r]), np.array([x[6] for x in r]), np.array([x[7] for x in r])
            f.write('| %s | %d | %.2e / %.2e / %.2e | %.2e / %.2e / %.2e | %.2e / %.2e / %.2e | %.6f |\n'
                    % (label, len(r), q(md, 50), q(md, 95), md.max(), q(da, 50), q(da, 95), np.abs(da).max(), q(dc, 50), q(dc, 95), np.abs(dc).max(),
                       min(x[5] for x in r)))
        f.write('\n| seed | clip | form | map max\\|d\\| | rms | CC(build, oracle) | dAUC-Judd | dCC |\n|---|---|---|---|---|---|---|---|\n')
        for x in rows:
            f.write('| %d | %d | %s | %.2e | %.2e | %.6f | %+.2e | %+.2e |\n' % x)


if __name__ == '__main__':
    main()
