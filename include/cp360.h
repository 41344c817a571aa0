/* libcp360 - C ABI of the MI355X (gfx950) 360-saliency hot path.
 *
 * The reference (hsientzucheng/CP-360-Weakly-Supervised-Saliency) has no FFI layer:
 * its "operator API" is the Python call surface of model/cube_pad.py,
 * model/resnet_cubic.py, model/clstm.py, utils/equi_to_cube.py,
 * utils/cube_to_equi.py and static_model/class_activation_model.py.  The Python
 * shims in cp_360_weakly_supervised_saliency_amd/ keep those names and bind the
 * entry points below through ctypes.  Each entry point cites
 * the reference code whose arithmetic it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; every buffer is caller-allocated DEVICE memory
 *    unless the parameter name ends in _host;
 *  - every function is asynchronous on the hipStream_t passed as `void* stream`
 *    (0 = the null stream) and never allocates, frees or synchronises;
 *  - return value: 0 = ok, negative = cp360_status; no exceptions, no exit()
 *    (the reference print+exit()s on batch % 6 != 0, cube_pad.py:33-35);
 *  - face order along the batch dimension is [back, down, front, left, right, top]
 *    (cube_pad.py:49), batches are frame-major / face-minor ([6N, ...]);
 *  - "NCHW" is the reference's tensor layout; "NHWC" (channels innermost) is the
 *    layout the fused pipeline keeps in HBM between kernels.
 */
#ifndef CP360_H
#define CP360_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CP360_OK = 0,
    CP360_ERR_BAD_SHAPE = -1,      /* negative / zero / inconsistent sizes        */
    CP360_ERR_BATCH_NOT_6N = -2,   /* cube_pad.py:33-35                           */
    CP360_ERR_NOT_SQUARE = -3,     /* CubePad transposes strips: faces must be n x n */
    CP360_ERR_BAD_DTYPE = -4,
    CP360_ERR_NULL = -5,
    CP360_ERR_ALIGN = -6,          /* channel counts / strides not 16-byte friendly */
    CP360_ERR_HIP = -7,            /* a HIP runtime call failed                   */
    CP360_ERR_UNSUPPORTED = -8
} cp360_status;

typedef enum {
    CP360_F32 = 0,
    CP360_BF16 = 1,
    CP360_F16 = 2,
    CP360_U8 = 3
} cp360_dtype;

const char* cp360_strerror(int status);
/* library / ABI version: major*10000 + minor*100 + patch.  Bumped on EVERY change of a struct, a
 * signature or a packed-weight layout: the binding (_lib.py: ABI_VERSION) refuses a library whose
 * version or sizeof(cp360_conv_desc) differs, so a stale out-of-band .so fails at load time. */
#define CP360_VERSION 305
int cp360_version(void);
/* sizeof(cp360_conv_desc) as the library was compiled. */
size_t cp360_conv_desc_bytes(void);

/* ------------------------------------------------------------------ K2: CubePad
 * Replaces CubePad.forward / CubePadding.forward, model/cube_pad.py:28-42,95-216
 * (33 torch.cat + 8 index_select per call).  Pure permutation-with-replication of
 * input elements -> results are bit-exact for every dtype.
 */

/* Host helper: the (face, i, j) -> flat source index table the kernels implement
 * (f'*n*n + i'*n + j'), int32 [6, n+pt+pd, n+pl+pr] written to host memory.  Same
 * inline function as the device code; used by tests and by callers that want to
 * gather on their own. */
int cp360_cubepad_table_host(int n, int pl, int pr, int pt, int pd, int32_t* table_host);

/* x [n6, C, n, n] -> y [n6, C, n+pt+pd, n+pl+pr], elem_size in {1, 2, 4, 8} bytes. */
int cp360_cubepad_nchw(const void* x, void* y, int n6, int C, int n,
                       int pl, int pr, int pt, int pd, int elem_size, void* stream);

/* x [n6, n, n, C] -> y [n6, n+pt+pd, n+pl+pr, Cy] with Cy >= C (extra channels are
 * written as zero); C*elem_size and Cy*elem_size must be multiples of 4 bytes. */
int cp360_cubepad_nhwc(const void* x, void* y, int n6, int C, int Cy, int n,
                       int pl, int pr, int pt, int pd, int elem_size, void* stream);

/* ------------------------------------------------------------------ layout glue */
/* [N, C, H, W] <-> [N, H, W, C]; dtype CP360_F32 or CP360_BF16 on either side (the
 * fused pipeline's bf16 mode converts here).  The NHWC side may be a channel slice of
 * a wider pixel: ld = elements per NHWC pixel (0 = C, dense), coff = first channel -
 * this is how clstm.py:55's torch.cat((input_, prev_hidden), 1) is laid out. */
int cp360_nchw_to_nhwc(const void* x, void* y, int N, int C, int H, int W,
                       int in_dtype, int out_dtype, int ld_y, int y_coff, void* stream);
int cp360_nhwc_to_nchw(const void* x, void* y, int N, int C, int H, int W,
                       int in_dtype, int out_dtype, int ld_x, int x_coff, void* stream);

/* ------------------------------------------------------------------ K1: equi -> cube
 * Replaces Equi2Cube.to_cube (utils/equi_to_cube.py:112-129, 18 cv2.remap calls per
 * frame) fused with the x/255 of dataset_feat_extractor.py:142, im_norm
 * (utils/utils.py:28-33), the float32 cast and the HWC->CHW permute of
 * class_activation_model.py:55.
 *   equi   [F, H, W, 3]   u8 or f32 (HWC, as decoded frames arrive)
 *   grid   [6, cd, cd, 2] f32 (x, y) - the float32 maps to_cube hands to cv2.remap
 *   out    F frames x 6 faces:
 *            out_layout 0: [6F, 3, cd, cd]           (NCHW, the reference's batch)
 *            out_layout 1: [6F, cd, cd, 4]           (NHWC, 4th channel = 0)
 *   value  = (bilinear(equi * scale) - mean[c]) * istd[c]
 *   cv_fixed_point != 0: OpenCV INTER_LINEAR coordinate quantisation (1/32 px,
 *   round-half-even), taps outside the image contribute 0 (BORDER_CONSTANT).
 */
int cp360_equi2cube(const void* equi, const float* grid, void* out,
                    int F, int H, int W, int cd,
                    const float* mean3_host, const float* istd3_host, float scale,
                    int in_dtype, int out_dtype, int out_layout, int cv_fixed_point,
                    void* stream);

/* ------------------------------------------------------------------ K6: cube -> equi
 * Replaces Cube2Equi.to_equi_nn (utils/cube_to_equi.py:37-66: six full-size
 * grid_samples + masked scatter) and, when out_max != NULL, the channel max of
 * temporal_model/test_temporal.py:83-84.
 *   x        [6B, C, w, w] (layout 0, NCHW) or [6B, w, w, C] (layout 1, NHWC), f32
 *   face_map [2w, 4w] int8, coord [2w, 4w, 2] f32 = pixel-space sampling position
 *            (x, y) already un-normalised on the host (align_corners folded in)
 *   out_full [B, C, 2w, 4w] f32 or NULL;  out_max [B, 2w, 4w] f32 or NULL
 */
int cp360_cube2equi(const float* x, const int8_t* face_map, const float* coord,
                    float* out_full, float* out_max, int B, int C, int w,
                    int layout, void* stream);

/* ------------------------------------------------------------------ K0: PIL-exact Lanczos resize
 * Replaces Image.fromarray(frame).convert('RGB').resize((w, h), resample=Image.LANCZOS) of
 * static_model/dataset_feat_extractor.py:119-142 (the frame resize in front of the cube
 * projection) for uint8 HWC frames, bit-exact with Pillow's 8-bit resampler.
 */
/* Taps per output index for a resize in_size -> out_size (Pillow's ksize); < 0: error. */
int cp360_resize_ksize(int in_size, int out_size);
/* HOST: Pillow's coefficient tables for one axis.  bounds int32 [out_size][2] = (first input
 * index, tap count), kk int32 [out_size][ksize] 22-bit fixed point.  Returns ksize. */
int cp360_resize_coeffs_host(int in_size, int out_size, int* bounds, int* kk);
/* The same for Pillow's other filters: filter 0 = LANCZOS (support 3), 1 = BICUBIC (a = -0.5, support 2 -
 * utils/utils.py:21, the overlay's heatmap.resize(..., resample=Image.CUBIC)). */
int cp360_resize_ksize2(int in_size, int out_size, int filter);
int cp360_resize_coeffs_host2(int in_size, int out_size, int filter, int* bounds, int* kk);
/* in u8 [F, h_in, w_in, 3] -> out u8 [F, h_out, w_out, 3] (device); the filter is whatever the tables hold.  Tables are DEVICE copies of
 * cp360_resize_coeffs_host(w_in, w_out) / (h_in, h_out); tmp u8 [F, h_in, w_out, 3] is needed
 * when both axes change (horizontal pass first, as Pillow). */
int cp360_resize_lanczos_u8(const void* in, void* out, void* tmp, int F, int h_in, int w_in,
                            int h_out, int w_out, const int* hbounds, const int* hkk, int hksize,
                            const int* vbounds, const int* vkk, int vksize, void* stream);

/* ------------------------------------------------------------------ K3/K4/K5: convolution
 * Implicit-GEMM convolution on MFMA.  Replaces nn.Conv2d(+BatchNorm2d eval)(+ReLU)
 * (+residual add) of model/resnet_cubic.py:85-106,163-175, the per-face CAM GEMM of
 * static_model/class_activation_model.py:70-83 (as a 1x1 convolution) and the three
 * 3x3 convolutions of model/clstm.py:56-64.  Zero padding never occurs on this path
 * (every reference conv has padding=0); the CubePad(p) that precedes a 3x3 / 7x7 conv
 * is fused into the tile loader (pad_mode 1) instead of being materialised.
 */
typedef struct {
    int dtype;        /* CP360_F32 (mfma_f32_16x16x4_f32, exact f32), CP360_BF16
                         (mfma_f32_16x16x32_bf16) or CP360_F16 (mfma_f32_16x16x32_f16),
                         f32 accumulate - activations, packed weights and outputs
                         all have this type                                       */
    int n_img;        /* images (6 * frames)                                      */
    int h_in, w_in;   /* spatial size of the (unpadded) input tensor              */
    int c_in;         /* K elements taken per tap (normally = channels)           */
    int pix_stride;   /* elements between neighbouring input pixels (>= c_in for a
                         normal conv; 4 for the stem's 8-pixel x 4-channel trick) */
    int kh, kw;       /* taps                                                     */
    int sy, sx;       /* stride in input pixels                                   */
    int h_out, w_out;
    int c_out;
    int pad_mode;     /* 0: no padding; 1: fused CubePad(pad) - needs h_in == w_in */
    int pad;
    int ld_out;       /* elements between output pixels (>= c_out)                */
    int out_coff;     /* first output channel inside an ld_out-wide pixel         */
    int ld_res;       /* residual pixel stride (0 when residual == NULL)          */
    int relu;
    int splits;       /* split-K factor (>= 1); > 1 needs `partial`               */
    int tile_px;      /* 0: the library's launch planner picks the tile; for c_out >=
                         256 (tuning, tests): 128 / 160 / 256 / 304 force the pixel tile
                         of the 8-wave 256-channel kernels (160: 16-bit types only - the
                         tile the planner takes when the larger ones leave CUs without
                         a workgroup; 129 = the 256x128 short-K
                         kernel that runs two workgroups per CU), 64 forces the
                         4-wave 128x128 kernel; any c_out % 8 == 0: 6464 forces the
                         64 x 64-tile small-M kernel (csrc/conv_small.hip), which the
                         planner picks by itself for f32 launches that cannot fill the
                         chip with the big tiles (one frame = 6 faces: BASELINE C2)  */
    int clip_resident;/* 1: CubePad(1) + 3x3 stride-1 convolution on cube faces small
                         enough that a whole cube (6 n^2 <= 304 pixels, n <= 7: the
                         ConvLSTM of model/clstm.py at cube size 224) is one tile:
                         the packed weights are channel- and tile-major
                         ([n / 256][c / 64 B][tap][n % 256][64 B]) and
                         every tap reads the clip's activations from an LDS-resident
                         tile (cubepad halo never leaves the cube).  Also 16x16 faces
                         (cube size 512): one face per tile, resident with its CubePad
                         ring (18 x 18 rows); and 8x8 faces (cube size 256): half a
                         cube (three faces) per tile with the whole cube resident.
                         Must be the same at pack and forward time; other
                         geometries: UNSUPPORTED                                     */
    int slab_rows;    /* 1: the f32 `partial` sums keep the packed row order inside each
                         32-channel group (the 4-channel group of channel n sits at
                         column (n & ~31) + ((n >> 3) & 3) * 4 + ((n >> 2) & 1) * 16), so
                         a pixel's lanes store 64 contiguous bytes; cp360_conv_finish
                         (same desc) and cp360_lstm_gates(slab_rows = 1) read that order.
                         Needs c_out % 32 == 0.  0: true channel order (raw CAM scores) */
    /* --- second source (0 = none): one more 1x1 filter accumulated into the same output tile, gathered
       from a second tensor in2 [n_img, h_in2, w_in2, pix_stride2] at pixel (oy * sy2, ox * sx2):
       out = act(conv(in) + W2 . in2 + bias (+ residual)).  This is the Bottleneck's downsample branch
       (model/resnet_cubic.py:99-104: conv1x1(stride) + BatchNorm on the block input, added to bn3's output
       before the ReLU) computed inside conv3's tile, so its [M, c_out] result never goes through HBM.
       Needs c_out >= 256, no clip_resident; pack with cp360_conv_pack_weights2, run with
       cp360_conv_forward2. */
    int c_in2, pix_stride2, h_in2, w_in2, sy2, sx2;
} cp360_conv_desc;

/* Bytes of packed weights for a desc (rows padded to the tile, K padded per tap). */
size_t cp360_conv_packed_bytes(const cp360_conv_desc* d);
/* Split-K factor (>= 1) that fills the 256 CUs for this geometry (d->splits ignored). */
int cp360_conv_suggest_splits(const cp360_conv_desc* d);
/* Which kernel and split count the planner picks for this descriptor, as text (e.g. "conv_small 64x64, 228 workgroups, splits 3").
 * Returns the length written (without the terminating 0; truncated to cap - 1) or a negative status. */
int cp360_conv_plan_describe(const cp360_conv_desc* d, char* buf, size_t cap);
/* With BOTH weight layouts at hand (tap-major and clip-resident): 1 = launch this descriptor's geometry on the
 * clip-resident kernel, 0 = on the tap-major path (whose planner then picks the 64 x 64-tile kernel: few cubes and few
 * output channels, e.g. layer4's conv2 of ONE frame in f32, give the clip kernel 2 tiles for 256 CUs).  d->clip_resident
 * and d->splits are ignored. */
int cp360_conv_prefer_clip(const cp360_conv_desc* d);
/* Bytes of split-K workspace (0 when splits == 1). */
size_t cp360_conv_partial_bytes(const cp360_conv_desc* d);
/* Pack OIHW f32 weights [c_out, c_in_w, kh_w, kw_w] times scale[c_out] (BatchNorm
 * folding; NULL = 1) into the kernel's [c_out_pad][tap][c_pad] layout and dtype.
 * stem_mode != 0 packs a [c_out, 3, 7, 7] filter for the kh=7, kw=1, c_in=32 form
 * (k index = kx*4 + c). */
int cp360_conv_pack_weights(const cp360_conv_desc* d, const float* w_oihw, const float* scale,
                            void* packed, int stem_mode, void* stream);
/* Same with a second source (d->c_in2 > 0): w2_oi f32 [c_out, c_in2] times scale2[c_out] is packed behind
 * the taps of the first filter. */
int cp360_conv_pack_weights2(const cp360_conv_desc* d, const float* w_oihw, const float* scale,
                             const float* w2_oi, const float* scale2, void* packed, void* stream);
/* out = act(conv(in) + bias (+ residual)).  bias f32 [c_out] or NULL.
 * d->splits > 1, or out == NULL: the raw f32 sums go to `partial`
 * ([splits, M, c_out], M = n_img*h_out*w_out; bias / residual / relu NOT applied)
 * and cp360_conv_finish or cp360_lstm_gates must follow. */
int cp360_conv_forward(const cp360_conv_desc* d, const void* in, const void* packed_w,
                       const float* bias, const void* residual, void* out,
                       float* partial, void* stream);
/* cp360_conv_forward with the second source in2 (NULL iff d->c_in2 == 0). */
int cp360_conv_forward2(const cp360_conv_desc* d, const void* in, const void* in2, const void* packed_w,
                        const float* bias, const void* residual, void* out,
                        float* partial, void* stream);
/* out = act(sum_s partial[s] + bias (+ residual)) in d->dtype at the desc's ld_out/out_coff. */
int cp360_conv_finish(const cp360_conv_desc* d, const float* partial, const float* bias,
                      const void* residual, void* out, void* stream);
/* The same with one more f32 addend `extra` [M, c_out] in the slabs' column order (d->slab_rows), e.g. the batched
 * x half of the ConvLSTM's first convolution (cp360_clstm_window). */
int cp360_conv_finish_add(const cp360_conv_desc* d, const float* partial, const float* extra, const float* bias,
                          const void* residual, void* out, void* stream);

/* The shape-specific fused kernels (resident-patch stem, stem + max-pool, layer1 / layer2 / layer3 Bottleneck tails, layer2's
 * first block, the launch-order hint) are INTERNAL building blocks of cp360_resnet_forward: include/cp360_internal.h.  A binder
 * of the reference needs the stage contexts (bottom of this file), the per-operator calls K0 / K1 / K2 / K6 / K7 / K8 / K9 and,
 * for its own networks, the generic convolution above. */

/* ------------------------------------------------------------------ K3b: max-pool
 * CubePad(1) + MaxPool2d(3, stride 2, padding 0) (resnet_cubic.py:128,169-170),
 * NHWC, pad fused: x [n6, n, n, C] -> y [n6, (n-1)/2+... , .., C] with
 * h_out = (n + 2 - 3)/2 + 1. */
int cp360_cubepad_maxpool3s2(const void* x, void* y, int n6, int n, int C, int dtype, void* stream);

/* ------------------------------------------------------------------ K5: ConvLSTM gates
 * model/clstm.py:68-80.  gates_partial: f32 [splits, M, 4*Hc] pre-activation sums
 * (split-K slabs of the Gates convolution, channel order in|remember|out|cell),
 * bias f32 [4*Hc]; c_prev/c_next f32 [M, Hc]; h_out (dtype h_dtype) is written at
 * h_out[m*ld_h + h_coff + j]; h_f32 (optional) receives an f32 copy [M, Hc]. */
int cp360_lstm_gates(const float* gates_partial, int splits, const float* bias,
                     const float* c_prev, float* c_next, void* h_out, int h_dtype,
                     int ld_h, int h_coff, float* h_f32, int M, int Hc, int slab_rows, void* stream);
/* The same plus the NEXT step's input: x_next f32 (frame t+1 of clip 0; clip b at + b*clip_stride elements,
 * pixel-major [P, Hc]) is window-normalised with minmax [M/P, 2] (test_temporal.py:77) and written to
 * h_out[m*ld_h + x_coff + j] - the x half of the ConvLSTM's concatenated input - in the same pass (one launch per
 * step less).  Needs input_size == Hc.  x_next == NULL: plain cp360_lstm_gates. */
int cp360_lstm_gates_next(const float* gates_partial, int splits, const float* bias,
                          const float* c_prev, float* c_next, void* h_out, int h_dtype,
                          int ld_h, int h_coff, float* h_f32, int M, int Hc, int slab_rows,
                          const float* x_next, const float* minmax, int x_coff, int P, size_t clip_stride,
                          void* stream);

/* ------------------------------------------------------------------ K5w: CubePad(1) + 3x3 convolution in the Winograd domain
 * The three ConvLSTM convolutions of model/clstm.py:56-64 as F(2x2, 3x3) for the 16-bit types: a w x w face is cut into
 * th x th tiles of 2 x 2 outputs (th = ceil(w / 2)); U = G g G^T is packed once per filter (16 matrices [c_out, c_in]), the
 * input transform V = B^T d B reads the cube-padded 4 x 4 window behind every tile, sixteen f32-accumulated MFMA GEMMs
 * M_p = V_p U_p^T (256 channels x 384 tiles per workgroup, both operands streamed) replace the 9-tap implicit GEMM, and
 * Y = A^T M A (+ bias, ReLU - or the LSTM gate update) writes the activations: 16 / 36 of the direct form's multiplies at even
 * face sizes, 256 / 441 at 7x7.  U and V are rounded once to the 16-bit type, every sum is f32.  f32 convolutions stay on
 * cp360_conv_forward (exact f32 MFMA, the 1e-3 parity path).
 *   in   [n_img, w, w, pix_stride] (the first c_in channels of a pixel), n_img = 6 * cubes
 *   v    workspace of cp360_wino_v_bytes: [16][ceil(c_in / 32)][m_pad][32] (m_pad = tiles rounded up to 384)
 *   m    workspace of cp360_wino_m_bytes: f32 [16][m_pad][c_out].  Odd faces (w = 2 th - 1): a tile of a face's last tile row /
 *        column keeps only the first row / column of its 2 x 2 outputs, which A^T M A computes without transform-domain row 3 /
 *        column 3 - cp360_wino_gemm does NOT write those rows of m (31 of a 7x7 face's 256 (tile, position) pairs), no
 *        cp360_wino_output* reads them, and cp360_wino_input / _output_input write zeros to the matching rows of v.
 *   out  [n_img, w, w, ld_out] at channel out_coff (ld_out 0 = c_out) */
typedef struct {
    int dtype;        /* CP360_BF16 or CP360_F16                                   */
    int n_img;        /* faces (6 * cubes)                                         */
    int face;         /* w: faces are w x w                                        */
    int c_in;         /* % 8 == 0                                                  */
    int pix_stride;   /* elements between neighbouring input pixels (>= c_in)      */
    int c_out;        /* % 8 == 0 (% 16 for the gate epilogue)                     */
    int ld_out, out_coff, relu;
} cp360_wino_desc;
size_t cp360_wino_packed_bytes(const cp360_wino_desc* d);
size_t cp360_wino_v_bytes(const cp360_wino_desc* d);
size_t cp360_wino_m_bytes(const cp360_wino_desc* d);
/* 1: the library's planner runs this geometry in the Winograd domain (16-bit type and at most 0.8 of the direct form's MFMAs
 * after padding the tile count to 384: 4 clips of 7x7 faces, one clip of 16x16 faces, ...), 0: on the direct kernels. */
int cp360_wino_preferred(const cp360_wino_desc* d);
int cp360_wino_pack_weights(const cp360_wino_desc* d, const float* w_oihw /* f32 [c_out, c_in, 3, 3] */, void* packed, void* stream);
int cp360_wino_input(const cp360_wino_desc* d, const void* in, void* v, void* stream);
int cp360_wino_gemm(const cp360_wino_desc* d, const void* v, const void* packed, float* m, void* stream);
int cp360_wino_output(const cp360_wino_desc* d, const float* m, const float* bias, void* out, void* stream);
/* Output transform of the Gates convolution + cp360_lstm_gates_next's update in one pass (gate g of hidden channel j is
 * convolution channel g * Hc + j, Hc = c_out / 4; bias f32 [c_out]); arguments as cp360_lstm_gates_next, h_out in d->dtype. */
int cp360_wino_output_gates(const cp360_wino_desc* d, const float* m, const float* bias, const float* c_prev, float* c_next,
                            void* h_out, int ld_h, int h_coff, float* h_f32, const float* x_next, const float* minmax,
                            int x_coff, size_t clip_stride, void* stream);
/* cp360_wino_output of THIS convolution fused with cp360_wino_input of the NEXT one (same faces, next c_in = this c_out, dense
 * pixels): v_next = what cp360_wino_input would make of this convolution's output, bit for bit, in one launch and without the
 * activation tensor.  Faces up to 9 x 9 (a cube's 32-channel image in LDS); larger: CP360_ERR_UNSUPPORTED, run the two calls. */
int cp360_wino_output_input(const cp360_wino_desc* d, const float* m, const float* bias, void* v_next, void* stream);
/* cp360_wino_input + cp360_wino_gemm + cp360_wino_output. */
int cp360_wino_forward(const cp360_wino_desc* d, const void* in, const void* packed, const float* bias, void* out, void* v,
                       float* m, void* stream);

/* ------------------------------------------------------------------ K7: window normalise
 * temporal_model/test_temporal.py:66-67,70-73,77: per clip min / max over the whole
 * window, then (x - mn) / (mx - mn).
 *   x: B windows of T frames, frame = [P, C] f32 (P = 6*w*w pixels, NHWC); window b starts
 *   at x + b*clip_stride elements (0 = T*P*C, dense clips; P*C = the reference's stride-1
 *   sliding window over one feature sequence, zero-copy) ; minmax [B, 2] f32 (out) */
int cp360_window_minmax(const float* x, float* minmax, float* scratch /* [B*256*2] */,
                        int B, size_t per_clip, size_t clip_stride, void* stream);
/* y[b, p, y_coff + c] = (x[b, t, p, c] - mn_b) / (mx_b - mn_b); y has pixel stride ld_y
 * and dtype y_dtype; optional second destination y2 (f32, [B, P, C]) for the cell. */
int cp360_window_normalize(const float* x, const float* minmax, void* y, int y_dtype,
                           int ld_y, int y_coff, float* y2, int B, int T, int t,
                           int P, int C, size_t clip_stride, void* stream);

/* All T frames of every window in one launch: y [T, B, P, C] (dense, y_dtype) = (x[b, t] - mn_b) / (mx_b - mn_b). */
int cp360_window_normalize_frames(const float* x, const float* minmax, void* y, int y_dtype, int B, int T, int P, int C,
                                  size_t clip_stride, void* stream);

/* ------------------------------------------------------------------ K9: overlay renderer
 * utils/utils.py:9-25 as used by temporal_model/test_temporal.py:90-97.
 * cp360_overlay_colorize: heat f32 [h, w] (optionally squared first, test_temporal.py:94) -> min-max
 * normalised in float32 as numpy does -> 256-entry colormap lookup as matplotlib's Colormap.__call__(bytes=True)
 * -> rgb u8 [h, w, 3].  lut768: device u8 [256, 3] (the shim builds 'jet' on the host).  The bicubic upsample
 * is cp360_resize_lanczos_u8 with cp360_resize_coeffs_host2(filter = 1) tables.
 * cp360_overlay_blend_u8: PIL.Image.blend: out = (u8)(img + alpha * (heat - img)), float arithmetic, 0 <= alpha <= 1. */
int cp360_overlay_colorize(const float* heat, int h, int w, int square, const uint8_t* lut768, uint8_t* rgb,
                           void* stream);
int cp360_overlay_blend_u8(const uint8_t* img, const uint8_t* heat_rgb, uint8_t* out, long long n_bytes,
                           float alpha, void* stream);

/* ------------------------------------------------------------------ K8: saliency metrics
 * utils/eval_saliency.py on the device: AUC_Judd (:90-146), AUC_Borji (:14-87), CorrCoeff (:149-176),
 * similarity (:179-190).  Every reference metric first resizes both maps with
 * cv2.resize(x, (240, 120), cv2.INTER_LANCZOS4) - the flag lands in the `dst` slot, so OpenCV's default
 * INTER_LINEAR runs: cp360_resize_linear_f32.  The AUCs share cp360_metric_auc_prepare (normalised
 * saliency + the values at fixated pixels, F > mean(F) + 2 std(F)) and a workspace of
 * cp360_metric_work_bytes(n) bytes; all pointers are device memory.
 */
/* cv2.resize(src [h, w] f32, (dw, dh)) with INTER_LINEAR (half-pixel centres, edge clamp) -> dst [dh, dw]. */
int cp360_resize_linear_f32(const float* src, int h, int w, float* dst, int dh, int dw, void* stream);
size_t cp360_metric_work_bytes(int n);
/* sal, fix: resized maps, n pixels each.  mode 0 (AUC_Judd): S = sal + jitter (f64 [n] or NULL, the
 * reference's randn / 1e7), min-max normalised in f64.  mode 1 (AUC_Borji): S[S > mean + 2 std] = 1, then
 * min-max normalised in f32.  work[0] (as double) = number of fixated pixels afterwards. */
int cp360_metric_auc_prepare(const float* sal, const float* fix, const double* jitter, int n, int mode,
                             void* work, void* stream);
/* out2[0] = AUC-Judd, out2[1] = number of fixated pixels (doubles). */
int cp360_metric_auc_judd(int n, void* work, double* out2, void* stream);
/* rr int32 [n_fix, n_splits]: the reference's np.random.randint(0, n, (n_fix, n_splits)); aucs f64 [n_splits]
 * (their mean is the score); step = the threshold spacing (0.01). */
int cp360_metric_auc_borji(int n, int n_splits, const int* rr, double step, void* work, double* aucs,
                           void* stream);
/* out2[0] = CorrCoeff(a, b), out2[1] = similarity(a, b) of two resized maps (doubles). */
int cp360_metric_cc_sim(const float* a, const float* b, int n, double* out2, void* stream);

/* ------------------------------------------------------------------ stage contexts: ONE entry point per network stage
 * (csrc/ctx.hip).  A cp360_ctx owns the packed / BatchNorm-folded weights of the two networks and the kernel choice for
 * every layer (fused tail kernels, tile shapes, split-K: everything the per-kernel entry points above leave to the
 * caller), so a binder needs three calls per stage - load, workspace_bytes, forward - instead of re-implementing the
 * launch planning.  The Python shims of this package call these (pipeline.py, static_model/class_activation_model.py:
 * cam_device, temporal_model/test_temporal.py: ClipRunner); the per-kernel entry points stay for tests and tuning.
 *  - a context is bound to one device; one context per GPU; not thread-safe within a context;
 *  - weights are DEVICE f32 tensors in the reference's state-dict layout, read during *_load only (packed copies are
 *    owned by the context, hipMalloc / hipFree inside load / destroy - the only allocations this library makes);
 *  - activations, workspace and outputs are caller-allocated; `workspace` must be 256-byte aligned and at least
 *    *_workspace_bytes(...) for the same shape; forward calls are asynchronous on `stream` and never allocate.
 */
/* Eval-mode BatchNorm2d as y = x * scale + bias (model/resnet_cubic.py:66-70 with eps 1e-5): scale = g / sqrt(var + eps),
 * bias = b - mean * scale, each operation rounded once in f32 (IEEE divide / sqrt, no FMA contraction), all pointers
 * device f32 [n].  The folding cp360_resnet_load applies; public so that a caller planning its own launches folds
 * identically. */
int cp360_fold_bn(const float* bn_weight, const float* bn_bias, const float* bn_mean, const float* bn_var, float eps,
                  float* scale, float* bias, int n, void* stream);

typedef struct cp360_ctx cp360_ctx;
int cp360_create(int device, cp360_ctx** out);
void cp360_destroy(cp360_ctx* ctx);

/* One convolution + its BatchNorm (model/resnet_cubic.py:65-83,115-128): OIHW weight and the four BatchNorm vectors. */
typedef struct {
    const float* weight;
    const float* bn_weight;
    const float* bn_bias;
    const float* bn_mean;
    const float* bn_var;
} cp360_conv_bn;

/* ResNet-50-cubic + CAM.  convs[53] in torchvision's key order: conv1/bn1, then for layerL.B: conv1/bn1, conv2/bn2,
 * conv3/bn3 and, for block 0 of a layer, downsample.0/downsample.1.  fc_weight [num_classes, 2048]; fc_shift = min(fc_weight)
 * when that is negative, else 0 (class_activation_model.py:51-52: the global scalar shift, computed by the caller).
 * dtype: CP360_F32 (exact-f32 MFMA path), CP360_BF16 or CP360_F16. */
int cp360_resnet_load(cp360_ctx* ctx, int dtype, const cp360_conv_bn* convs, int n_convs, const float* fc_weight,
                      int num_classes, float fc_shift, float bn_eps, void* stream);
size_t cp360_resnet_workspace_bytes(cp360_ctx* ctx, int n_img, int cube_dim);
/* faces_p3 [n_img, cube_dim+6, cube_dim+6, 4]: normalised cube faces WITH their CubePad(3) ring, NHWC4, in the loaded
 * dtype (what cp360_equi2cube writes with the CubePad(3)-gathered grid) -> cam_out f32 [n_img, cd/32, cd/32, num_classes]
 * (the per-face CAM scores of class_activation_model.py:70-83, NHWC) and, if feat_out != NULL, the layer4 features
 * [n_img, cd/32, cd/32, 2048] in the loaded dtype.  resnet_cubic.py:163-175 + class_activation_model.py:46-83. */
int cp360_resnet_forward(cp360_ctx* ctx, const void* faces_p3, int n_img, int cube_dim, float* cam_out, void* feat_out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ConvLSTM cell (model/clstm.py:19-82).  w1 [4H, Cin+H, 3, 3], w2 / wg [4H, 4H, 3, 3], biases [4H]; `face` = the cube
 * face size the cell will run at (7 at cube 224, 16 at cube 512): it selects the weight layout of the kernel for that size. */
int cp360_clstm_load(cp360_ctx* ctx, int dtype, const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* wg, const float* bg, int input_size, int hidden_size, int face, void* stream);
/* The 16-bit cell in the Winograd domain ("K5w" above): when a launch shape has enough tiles (cp360_wino_preferred on Conv2's
 * shape - e.g. 4 cubes of 7x7 faces, one cube of 16x16 faces) cp360_clstm_step / cp360_clstm_window run the three convolutions
 * as F(2x2, 3x3).  Its filters (16 / 9 of the direct packing's bytes) are packed on request:
 *   cp360_clstm_wino_state  0: (n_clips, face) runs on the direct kernels; 1: in the Winograd domain; 2: it would, but
 *                           cp360_clstm_load_wino has not been called - the direct kernels run until it has
 *   cp360_clstm_load_wino   the same f32 filters as cp360_clstm_load (call it after that; workspace sizes change with it). */
int cp360_clstm_wino_state(cp360_ctx* ctx, int n_clips, int face);
int cp360_clstm_load_wino(cp360_ctx* ctx, const float* w1, const float* w2, const float* wg, void* stream);
size_t cp360_clstm_workspace_bytes(cp360_ctx* ctx, int n_clips, int face);
/* One cell update for n_clips cubes in lock step on the fused layout: xh [6 n_clips, face, face, Cin + H] (loaded dtype):
 * channels [0, Cin) = the input frame, [Cin, Cin + H) = the previous hidden state - the NEW hidden state is written
 * back there; c_prev / c_next f32 [6 n_clips, face, face, H]; h_f32 (optional) an f32 copy of the new hidden state.
 * x_next != NULL (needs Cin == H): frame t+1 of clip 0 (f32, pixel-major [6 face^2, H]; clip b at + b * clip_stride
 * elements) is window-normalised with minmax [n_clips, 2] (temporal_model/test_temporal.py:77) into the x half of xh in
 * the same pass.  model/clstm.py:42-82. */
int cp360_clstm_step(cp360_ctx* ctx, void* xh, const float* c_prev, float* c_next, float* h_f32, int n_clips, int face,
                     const float* x_next, const float* minmax, size_t clip_stride, void* workspace,
                     size_t workspace_bytes, void* stream);

/* One WINDOW of the temporal stage in one call (temporal_model/test_temporal.py:63-80 for n_clips windows in lock step):
 * min / max over each window, hidden = cell = normalised frame 0, T cell updates (frame 0 fed again first).
 *   cam       f32: window b = T frames of [6 face^2, Cin] (pixel-major) starting at cam + b * clip_stride elements
 *             (clip_stride 0 = dense T * P * Cin; P * Cin = the reference's stride-1 sliding window over one sequence)
 *   xh        [6 n_clips, face, face, Cin + H] scratch in the loaded dtype (the fused [x | h] buffer of cp360_clstm_step)
 *   cell0/1   f32 [6 n_clips, face, face, H] scratch (ping-pong cell state)
 *   h_out     f32 [6 n_clips, face, face, H]: the final hidden state (:80);  h_all (optional) f32 [T, 6 n_clips, face, face, H]:
 *             the hidden state after EVERY step (return_all_steps)
 *   minmax    f32 [n_clips, 2] (out), mm_scratch f32 [n_clips * 512]
 * Needs Cin == H (:70-73).  By default the window issues exactly the launches of T cp360_clstm_step calls (same bits).
 * Opt-in, CP360_XBATCH=1 (measured performance-neutral; it changes Conv1's accumulation order): the x half of Conv1
 * (K = 9 Cin, no recurrence) runs ONCE for all T frames (one M = T * 6 face^2 * n_clips GEMM without split-K; its f32 result
 * joins the h half's split-K slabs in cp360_conv_finish_add), so Conv1's x weights stream once per window instead of T
 * times, at T * M * 4H * 4 bytes more workspace.  workspace: cp360_clstm_window_workspace_bytes. */
size_t cp360_clstm_window_workspace_bytes(cp360_ctx* ctx, int n_clips, int T, int face);
int cp360_clstm_window(cp360_ctx* ctx, const float* cam, size_t clip_stride, int n_clips, int T, int face, void* xh,
                       float* cell0, float* cell1, float* h_out, float* h_all, float* minmax, float* mm_scratch,
                       void* workspace, size_t workspace_bytes, void* stream);

/* The launch plan cp360_resnet_forward would follow for n_img faces of cube_dim^2, one line per layer: which fused kernel or
 * generic path runs (the fused Bottleneck kernels are specialised to 16-bit types and the face sizes of cube 224 / 512; any
 * other geometry - and f32 - takes the per-convolution path, whose tile / split-K choice per launch is listed).  Returns the
 * length written (truncated to cap - 1) or a negative status.  Costs no GPU work. */
int cp360_resnet_plan_describe(cp360_ctx* ctx, int n_img, int cube_dim, char* buf, size_t cap);

/* Diagnostic only (bench.py `held_clock_ghz`): a bare bf16 MFMA loop on pseudo-random operands, n_workgroups x 4 waves, each
 * wave stamped once around `iters` x 16 MFMAs.  stamps: device u64 [n_workgroups * 4][2] = {d s_memtime (shader cycles),
 * d s_memrealtime (100 MHz ticks)} per wave; held clock of a wave = 0.1 GHz * [0] / [1].  A buffer of its own: no output of
 * the library depends on it. */
int cp360_clock_probe(unsigned long long* stamps, int n_workgroups, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CP360_H */
