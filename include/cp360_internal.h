/* libcp360 - INTERNAL entry points: the shape-specific fused kernels that the static-stage context
 * (cp360_resnet_forward, csrc/ctx.hip) is built from.
 *
 * They are exported (tests/ and the CP360_CTX=0 Python planner call them one by one, which is how each is checked against
 * torch-CPU and against the per-convolution path), but they are NOT part of the drop-in boundary: each is specialised to one
 * layer geometry of ResNet-50-cubic at cube size 224 / 512 in a 16-bit type and answers CP360_ERR_UNSUPPORTED for anything
 * else, their packings are private formats, and they may change with any round.  Bind include/cp360.h instead.
 */
#ifndef CP360_INTERNAL_H
#define CP360_INTERNAL_H

#include "cp360.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ K3a: resident-patch stem
 * conv 7x7 stride 2 (3 -> 64) + folded BatchNorm + ReLU of model/resnet_cubic.py:115-128,163-168 on the
 * materialised CubePad(3) input [n_img, cd+6, cd+6, 4] (NHWC4, 16-bit) -> [n_img, cd/2, cd/2, 64], for
 * cube_dim 224 or 512 and CP360_BF16 / CP360_F16 (other sizes / f32: CP360_ERR_UNSUPPORTED - use the generic
 * convolution in its "7 taps of 8 pixels x 4 channels" form).  Same arithmetic as the generic path
 * (16-bit products, f32 accumulate, one rounding); the input rows a band of output rows needs are
 * copied to LDS once and the MFMA fragments are read from the raw rows (no im2col anywhere).
 */
size_t cp360_stem_packed_bytes(int dtype);
int cp360_stem_pack_weights(int dtype, const float* w_oihw /* [64,3,7,7] */, const float* scale /* [64] or NULL */,
                            void* packed, void* stream);
int cp360_stem_forward(int dtype, const void* xp, const void* packed, const float* bias, void* out,
                       int n_img, int cube_dim, int relu, void* stream);
/* The stem AND the max-pool behind it (CubePad(1) + MaxPool2d(3, 2), resnet_cubic.py:169-170) in one kernel: cube size
 * 224, CP360_BF16 / CP360_F16, ReLU on (csrc/stem.hip, stem_pool_kernel): xp as above -> y [n_img, 56, 56, 64]; the
 * 112x112x64 stem output is never written.  `border`: scratch of cp360_stem_pool_border_bytes(n_img) bytes (the four
 * border rows / columns of every face's stem output, from which a second small launch folds the max-pool's cube padding
 * into pooled row 0 / column 0).  Bit-identical to cp360_stem_forward + cp360_cubepad_maxpool3s2.  Other cube sizes:
 * CP360_ERR_UNSUPPORTED. */
size_t cp360_stem_pool_border_bytes(int n_img);
int cp360_stem_pool_forward(int dtype, const void* xp, const void* packed, const float* bias, void* y, void* border,
                            int n_img, int cube_dim, void* stream);

/* ------------------------------------------------------------------ K3c: resident-band 3x3
 * CubePad(1) + conv 3x3 stride 1 (64 -> 64) + folded BatchNorm + ReLU = conv2 / bn2 / relu of layer1's
 * Bottlenecks (model/resnet_cubic.py:85-106) at cube size 224 (56x56 faces), CP360_BF16 / CP360_F16.
 * x [n_img, 56, 56, 64] NHWC -> out [n_img, 56, 56, 64].  Other shapes / f32: CP360_ERR_UNSUPPORTED
 * (cp360_conv_forward handles them).  A band of 4 output rows' padded pixels is gathered to LDS once
 * (through the CubePad map) and all nine taps read it there.
 */
size_t cp360_band3x3_packed_bytes(int dtype);
int cp360_band3x3_pack_weights(int dtype, const float* w_oihw /* [64,64,3,3] */, const float* scale /* or NULL */,
                               void* packed, void* stream);
int cp360_band3x3_forward(int dtype, const void* x, const void* packed, const float* bias, void* out,
                          int n_img, int face, int channels, int relu, void* stream);

/* ------------------------------------------------------------------ K3d: fused Bottleneck tail (layer1)
 * One kernel for  conv2 (CubePad(1) + 3x3, 64 -> 64) + bn2 + relu  ->  conv3 (1x1, 64 -> 256) + bn3
 * + (identity residual | downsample(x) = conv1x1 64 -> 256 + bn) + relu  ->  [optional] the NEXT block's
 * conv1 (1x1, 256 -> 64) + bn1 + relu : model/resnet_cubic.py:85-106 for the three Bottlenecks of layer1 at
 * cube size 224 (56x56 faces) or 512 (128x128 faces: `face`), CP360_BF16 / CP360_F16 only.  Nothing between the
 * convolutions goes through
 * HBM: conv2's accumulators feed conv3 from registers, conv3's rounded output feeds the next conv1.
 *   mid        [n_img, face, face, 64]   this block's conv1 output (NHWC); the other tensors likewise
 *   w2_packed  cp360_l1block_pack_conv2(w2 [64, 64, 3, 3]) (fragment order, cp360_l1block_conv2_bytes), bias2 f32 [64]
 *   w3_frags   cp360_frag_pack_1x1(w3 [256, 64], order 0), bias3 f32 [256] (with the downsample branch: b3 + bd)
 *   residual   [n_img, 56, 56, 256] or NULL;  x_ds [n_img, 56, 56, 64] + wd_frags (order 0, [256, 64]) or NULL:
 *              exactly one of residual / x_ds
 *   out        [n_img, 56, 56, 256]
 *   w1_frags   cp360_frag_pack_1x1(w1_next [64, 256], order 1) + bias1 f32 [64] + out_next [n_img, 56, 56, 64],
 *              or all NULL
 * Other face sizes: CP360_ERR_UNSUPPORTED (the per-convolution path handles them).
 */
/* MFMA A-fragment packing of a 1x1 filter w [n_out, k] (f32, times scale[n_out] or NULL): 1 KiB fragments of
 * 16 rows x 32 k, rows in the kernels' acc_chan order; order 0 = row-block major ([n_out/16][k/32]),
 * order 1 = k-block major ([k/32][n_out/16]).  n_out % 32 == 0, k % 32 == 0. */
size_t cp360_l1block_conv2_bytes(int dtype);
int cp360_l1block_pack_conv2(int dtype, const float* w_oihw, const float* scale, void* packed, void* stream);
size_t cp360_frag_packed_bytes(int dtype, int n_out, int k);
int cp360_frag_pack_1x1(int dtype, const float* w, const float* scale, void* packed, int n_out, int k,
                        int order, void* stream);
int cp360_l1block_forward(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                          const void* w3_frags, const float* bias3, const void* residual, const void* x_ds,
                          const void* wd_frags, void* out, const void* w1_frags, const float* bias1,
                          void* out_next, int n_img, int face, void* stream);

/* layer1.0 with ITS OWN conv1 inside (round 6): x [n_img, 56, 56, 64] is the block input - conv1's input and the downsample source;
 * conv1 (1x1, 64 -> 64, + bn1 + relu, model/resnet_cubic.py:88-90) runs in place on the band's resident patch between the gather and
 * conv2, so neither its launch nor its output tensor exists.  w0_frags = cp360_frag_pack_1x1(w1 [64, 64], order 0), bias0 f32 [64];
 * the rest as cp360_l1block_forward's downsample form (bias3 = b3 + bd).  56x56 faces (cube 224), CP360_BF16 / CP360_F16; bit-identical
 * to the generic conv1 launch + cp360_l1block_forward. */
int cp360_l1block_forward_first(int dtype, const void* x, const void* w0_frags, const float* bias0, const void* w2_packed,
                                const float* bias2, const void* w3_frags, const float* bias3, const void* wd_frags, void* out,
                                const void* w1_frags, const float* bias1, void* out_next, int n_img, int face, void* stream);

/* The LAST Bottleneck of layer1 (identity residual) chained with layer2's first conv1 (1x1, 256 -> 128 + bn1 + relu,
 * model/resnet_cubic.py:88-90 of layer2.0) - the launch that would otherwise re-read the 616 MB (64 frames) this kernel
 * writes.  Arguments as cp360_l1block_forward with residual (no x_ds); w1_frags = cp360_frag_pack_1x1(w1 [128, 256],
 * order 1), bias1 f32 [128] or NULL, out_next [n_img, face, face, 128].  56x56 and 128x128 faces, CP360_BF16 / CP360_F16. */
int cp360_l1block_forward_wide(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                               const void* w3_frags, const float* bias3, const void* residual, void* out,
                               const void* w1_frags, const float* bias1, void* out_next, int n_img, int face,
                               void* stream);

/* ------------------------------------------------------------------ K3e: fused Bottleneck tail (layer2)
 * conv2 (CubePad(1) + 3x3, 128 -> 128) + bn2 + relu -> conv3 (1x1, 128 -> 512) + bn3 + identity residual + relu
 * of layer2's identity Bottlenecks (model/resnet_cubic.py:85-106) at cube size 224 (28x28 faces) or 512 (64x64:
 * `face`), CP360_BF16 / CP360_F16, in one kernel (csrc/l2block.hip).
 *   mid [n_img, 28, 28, 128], w2_packed = cp360_l2block_pack_weights(w2 [128,128,3,3]), bias2 f32 [128],
 *   w3_frags = cp360_frag_pack_1x1(w3 [512, 128], order 0), bias3 f32 [512],
 *   residual / out [n_img, 28, 28, 512].  Other face sizes: CP360_ERR_UNSUPPORTED. */
size_t cp360_l2block_packed_bytes(int dtype);
int cp360_l2block_pack_weights(int dtype, const float* w_oihw, const float* scale, void* packed, void* stream);
int cp360_l2block_forward(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                          const void* w3_frags, const float* bias3, const void* residual, void* out,
                          int n_img, int face, void* stream);
/* The same launch also computing the NEXT identity block's conv1 (1x1, 512 -> 128) + bn1 + relu
 * (model/resnet_cubic.py:88-90) from the output pieces while they are on the chip (28x28 faces only):
 *   w1_frags = cp360_frag_pack_1x1(w1 [128, 512], order 0), bias1 f32 [128] or NULL, out_next [n_img, 28, 28, 128]. */
int cp360_l2block_forward_next(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                               const void* w3_frags, const float* bias3, const void* residual, void* out,
                               const void* w1_frags, const float* bias1, void* out_next, int n_img, int face,
                               void* stream);

/* ------------------------------------------------------------------ K3f: layer2's FIRST Bottleneck after its conv1 (csrc/lfirst.hip)
 * conv2 (CubePad(1) + 3x3 STRIDE 2, 128 -> 128) + bn2 + relu -> conv3 (1x1, 128 -> 512) + bn3 + downsample(x) (1x1 stride 2,
 * 256 -> 512, + bn) + relu of layer2.0 (model/resnet_cubic.py:85-106,145-161) in ONE kernel, cube size 224 (56x56 -> 28x28
 * faces: face_out = 28, anything else CP360_ERR_UNSUPPORTED), CP360_BF16 / CP360_F16.
 *   mid [n_img, 56, 56, 128] (the block's conv1 output), w2_packed = cp360_l2block_pack_weights(w2 [128,128,3,3]), bias2 f32 [128],
 *   w3d_frags = cp360_l2first_pack_w3d(w3 [512,128] x scale3, wd [512,256] x scaled): fragments of the K = 384 filter
 *   [W3 | Wd], bias3d f32 [512] = b3 + bd, x [n_img, 56, 56, 256] (the block input), out [n_img, 28, 28, 512]. */
size_t cp360_l2first_w3d_bytes(int dtype);
int cp360_l2first_pack_w3d(int dtype, const float* w3, const float* scale3, const float* wd, const float* scaled, void* packed,
                           void* stream);
/* w1_frags = cp360_frag_pack_1x1(w1 [128, 512], order 0) + bias1 f32 [128] (or NULL) + out_next [n_img, 28, 28, 128]: the same
 * launch also computes the NEXT block's conv1 (layer2.1: 1x1, 512 -> 128) + bn1 + relu from the output pieces; all NULL: not. */
int cp360_l2first_forward(int dtype, const void* mid, const void* w2_packed, const float* bias2, const void* w3d_frags,
                          const float* bias3d, const void* x, void* out, const void* w1_frags, const float* bias1,
                          void* out_next, int n_img, int face_out, void* stream);

/* The same kernel at layer3's geometry (csrc/l2block.hip, C = 256): conv2 (CubePad(1) + 3x3, 256 -> 256) + bn2 + relu ->
 * conv3 (1x1, 256 -> 1024) + bn3 + identity residual + relu of layer3's identity Bottlenecks at cube size 224
 * (14x14 faces: `face` = 14) or 512 (32x32 faces, bands of two rows: `face` = 32); anything else CP360_ERR_UNSUPPORTED.
 *   mid [n_img, 14, 14, 256], w2_packed = cp360_l3block_pack_weights(w2 [256,256,3,3]), bias2 f32 [256],
 *   w3_frags = cp360_frag_pack_1x1(w3 [1024, 256], order 0), bias3 f32 [1024], residual / out [n_img, 14, 14, 1024]. */
size_t cp360_l3block_packed_bytes(int dtype);
int cp360_l3block_pack_weights(int dtype, const float* w_oihw, const float* scale, void* packed, void* stream);
int cp360_l3block_forward(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                          const void* w3_frags, const float* bias3, const void* residual, void* out,
                          int n_img, int face, void* stream);

/* ------------------------------------------------------------------ launch order (a performance hint)
 * Work-item order of the launches the CALLING THREAD issues from now on: 0 = ascending (default), 1 = descending,
 * 2 = alternating, starting with descending (convolutions, stem, max-pool, Bottleneck tail kernels; the others ignore
 * it and do not count).  Results never depend on it.  Why: the
 * 256 MB Infinity Cache keeps what a kernel wrote LAST; a consumer that walks its input in the producer's order starts
 * with the lines that were evicted first and, for tensors above 256 MB, misses all the way (tools/mall_probe.hip: a
 * 308 MB buffer read back front-to-back 3.8 TB/s, back-to-front 6.3 TB/s).  Alternating the order between a producer
 * and its consumer lets the consumer begin with the resident lines.  Returns the previous setting. */
int cp360_set_launch_order(int mode);

/* ------------------------------------------------------------------ the bare Winograd-domain GEMM (tools/wino_probe.py)
 * m[pos][tile][c] = sum_k v[pos][k / 32][tile][k % 32] * u[pos][c / 256][k / 32][c % 256][k % 32] on caller-made operands:
 * 16 positions x nt channel tiles x mt tile blocks of 384, nsub 64-byte K sub-steps. */
int cp360_wino_gemm_raw(int dtype, const void* u, const void* v, float* m, int nsub, int nt, int mt, int ldm, int c_out,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CP360_INTERNAL_H */
