"""MI355X-native (gfx950) hot path for 360-degree video saliency.

equi -> cube (K1) -> CubePad (K2) -> ResNet-50-cubic (K3) -> CAM (K4) -> ConvLSTM (K5)
-> cube -> equi saliency (K6), behind the Python module boundary of
hsientzucheng/CP-360-Weakly-Supervised-Saliency.  All compute runs in hand-written HIP
kernels reached through the C ABI of ``csrc/libcp360.so`` (see include/cp360.h);
PyTorch only owns device memory, streams and torch.distributed.  There is no CPU
fallback: every op raises if the HIP library is missing.
"""
__version__ = "0.1.0"
