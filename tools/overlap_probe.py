"""Would software-pipelining the two stages across batches pay?  The static stage of batch k+1 (HBM-bound layers 1-2, fused
tails) and the ConvLSTM of batch k (MFMA-bound, one 146 KB-LDS workgroup per CU) use different resources - but the clip kernel
leaves no LDS for a second workgroup on its CU.  Measured here: two engines on two streams, one running only its temporal
stage, the other only its static stage, against the same two calls back to back on one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp_360_weakly_supervised_saliency_amd.pipeline import SaliencyEngine
from cp_360_weakly_supervised_saliency_amd.utils import synth

dev = torch.device('cuda')
B, T, H, W = 4, 16, 1024, 2048
rs, cs = synth.resnet50_state(seed=1), synth.clstm_state(seed=2)
e1 = SaliencyEngine(rs, cs, (H, W), 224, clips=B, frames=T, precision='bf16', device=dev)
e2 = SaliencyEngine(rs, cs, (H, W), 224, clips=B, frames=T, precision='bf16', device=dev)
frames = torch.stack([torch.from_numpy(synth.clip_u8(3 + b, T, H, W)) for b in range(B)]).to(dev)
flat = frames.reshape((B * T,) + tuple(frames.shape[2:]))
with torch.no_grad():
    for e in (e1, e2):
        e(frames)
    torch.cuda.synchronize()
    N = 10

    def serial():
        for _ in range(N):
            e2.static_stage(flat)
            e1.temporal_stage()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def overlapped():
        for _ in range(N):
            with torch.cuda.stream(sa):
                e1.temporal_stage()
            with torch.cuda.stream(sb):
                e2.static_stage(flat)
            # one batch = both; the next batch's two halves start when both are done (as a pipelined engine would)
            ea, eb = torch.cuda.Event(), torch.cuda.Event()
            ea.record(sa); eb.record(sb)
            sa.wait_event(eb); sb.wait_event(ea)
    for name, fn in (('serial (one stream)', serial), ('two streams', overlapped), ('serial (one stream)', serial), ('two streams', overlapped)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        print('%-22s %.3f ms per batch  (%.0f frames/s)' % (name, dt * 1e3, B * T / dt))
