// One C entry point per network stage (include/cp360.h, "stage contexts"): cp360_ctx owns the packed / BN-folded
// weights of a stage and the kernel choice for every layer - which fused kernel runs where, tile shapes, split-K -
// that the Python shims used to plan (ops.Conv.__call__, model/resnet_cubic.py: layer1_nhwc .. layer4).  A binder of
// another language gets the whole static stage / one ConvLSTM step from ONE call:
//
//   cp360_resnet_forward  = ResNet.forward to layer4 (model/resnet_cubic.py:163-175: CubePad(3) input -> conv7x7 s2 +
//                           bn + relu -> CubePad(1) + maxpool -> 16 Bottlenecks :85-106) + the CAM GEMM of
//                           static_model/class_activation_model.py:46-52,70-83
//   cp360_clstm_step      = ConvLSTMCell.forward (model/clstm.py:42-82) on the fused [x | h] layout
//
// Host-side C++ only: every launch goes through the per-kernel entry points of this library, in exactly the order and
// with exactly the descriptors the Python planning produces, so both paths give the same bits (tests/test_ctx.py).
#include "common.h"
#include <algorithm>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------- small device helpers
__global__ void fold_bn_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ mu,
                               const float* __restrict__ var, float eps, float* __restrict__ scale,
                               float* __restrict__ bias, int n) {
    // eval-mode BatchNorm as y = x * scale + bias, every operation rounded once like torch's g / sqrt(var + eps),
    // b - mu * scale: IEEE divide / sqrt (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt; the __f*_rn
    // intrinsics map to the approximate native forms here) and no contraction into an FMA - the same bits as the
    // host-side folding of model/resnet_cubic.py
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = var[i] + eps;
    const float s = g[i] / sqrtf(v);
    scale[i] = s;
    const float t = mu[i] * s;
    bias[i] = b[i] - t;
}
__global__ void add_vec_kernel(float* __restrict__ a, const float* __restrict__ b, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = a[i] + b[i];
}
__global__ void sub_scalar_kernel(const float* __restrict__ x, float s, float* __restrict__ y, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = x[i] - s;
}

int es_of(int dtype) { return dtype == CP360_F32 ? 4 : 2; }
size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// The context's allocations and launches belong to ITS device, whatever device the calling thread has current
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Owned {                                   // device allocations a context owns
    std::vector<void*> ptrs;
    void* take(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return p;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

// One convolution with resident packed weights (the C++ counterpart of ops.Conv)
struct CConv {
    int c_out = 0, c_in_w = 0, kh_w = 0, kw_w = 0, stride = 1, pad = 0, relu = 0;
    bool stem = false, clip_ok = false;
    int c_in = 0, kh = 0, kw = 0, pix_stride = 0;
    int c_in2 = 0, stride2 = 0;                  // second source (the downsample branch inside conv3)
    void* packed = nullptr;                      // tap-major layout
    void* packed_clip = nullptr;                 // channel-major layout of the clip-resident kernel
    float* bias = nullptr;
};

void fill_desc(const CConv& c, int dtype, int n_img, int h_in, int w_in, int splits, int ld_out, int out_coff,
               int ld_res, int clip_resident, int slab_rows, int h2, int w2, int ps2, cp360_conv_desc* d) {
    *d = cp360_conv_desc{};
    d->dtype = dtype;
    d->n_img = n_img; d->h_in = h_in; d->w_in = w_in;
    d->c_in = c.c_in; d->pix_stride = c.pix_stride;
    d->kh = c.kh; d->kw = c.kw; d->sy = c.stride; d->sx = c.stride;
    const int p2 = 2 * c.pad;
    d->h_out = (h_in + p2 - c.kh) / c.stride + 1;
    d->w_out = c.stem ? (w_in + p2 - c.kw_w) / c.stride + 1 : (w_in + p2 - c.kw) / c.stride + 1;
    d->c_out = c.c_out;
    d->pad_mode = c.pad > 0 ? 1 : 0; d->pad = c.pad;
    d->ld_out = ld_out > 0 ? ld_out : c.c_out;
    d->out_coff = out_coff; d->ld_res = ld_res;
    d->relu = c.relu; d->splits = splits; d->tile_px = 0;
    d->clip_resident = clip_resident; d->slab_rows = slab_rows;
    if (c.c_in2 > 0) {
        d->c_in2 = c.c_in2; d->sy2 = d->sx2 = c.stride2;
        if (h2 > 0) { d->h_in2 = h2; d->w_in2 = w2; d->pix_stride2 = ps2; }
        else { d->h_in2 = (d->h_out - 1) * c.stride2 + 1; d->w_in2 = (d->w_out - 1) * c.stride2 + 1; d->pix_stride2 = c.c_in2; }
    }
}

// pack `c` (weights w_oihw f32 on the device, times scale) in the layout(s) its call sites need
int pack_conv(Owned& own, CConv& c, int dtype, const float* w, const float* scale, const float* w2, const float* scale2,
              bool want_plain, bool want_clip, hipStream_t st) {
    cp360_conv_desc d;
    if (want_plain) {
        fill_desc(c, dtype, 6, c.kh > 8 ? c.kh : 8, c.stem ? 16 : (c.kw > 8 ? c.kw : 8), 1, 0, 0, 0, 0, 0, 0, 0, 0, &d);
        const size_t nb = cp360_conv_packed_bytes(&d);
        if (!nb) return CP360_ERR_UNSUPPORTED;
        if (!(c.packed = own.take(nb))) return CP360_ERR_HIP;
        const int rc = c.c_in2 > 0 ? cp360_conv_pack_weights2(&d, w, scale, w2, scale2, c.packed, st)
                                   : cp360_conv_pack_weights(&d, w, scale, c.packed, c.stem ? 1 : 0, st);
        if (rc) return rc;
    }
    if (want_clip) {
        fill_desc(c, dtype, 6, 7, 7, 1, 0, 0, 0, 1, 0, 0, 0, 0, &d);
        const size_t nb = cp360_conv_packed_bytes(&d);
        if (!nb) return CP360_ERR_UNSUPPORTED;
        if (!(c.packed_clip = own.take(nb))) return CP360_ERR_HIP;
        const int rc = cp360_conv_pack_weights(&d, w, scale, c.packed_clip, 0, st);
        if (rc) return rc;
    }
    return CP360_OK;
}

bool clip_geometry(const CConv& c, int n_img, int h, int w) {
    return c.clip_ok && h == w && (6 * h * w <= 304 || h == 8 || h == 16) && n_img % 6 == 0;
}

void fill_desc(const CConv& c, int dtype, int n_img, int h_in, int w_in, int splits, int ld_out, int out_coff,
               int ld_res, int clip_resident, int slab_rows, int h2, int w2, int ps2, cp360_conv_desc* d);

// clip-resident kernel or tap-major path?  With one packing there is no choice; with both (layer4's conv2) the library's cost
// model decides (cp360_conv_prefer_clip: one frame in f32 runs better on the 64 x 64 tiles of conv_small.hip)
bool use_clip(const CConv& c, int dtype, int n_img, int h, int w) {
    if (!clip_geometry(c, n_img, h, w) || !c.packed_clip) return false;
    if (!c.packed) return true;
    cp360_conv_desc d;
    fill_desc(c, dtype, n_img, h, w, 1, 0, 0, 0, 1, 0, 0, 0, 0, &d);
    return cp360_conv_prefer_clip(&d) != 0;
}

// The launch (or, with `dry`, only the split-K workspace it needs): ops.Conv.__call__.
//   raw: leave the f32 sums in `partial` ([splits, M, c_out]; *splits_out tells how many) for the caller's epilogue.
struct Run {
    bool dry = false;
    size_t partial_need = 0;                     // dry: bytes of split-K workspace
    float* partial = nullptr;
    size_t partial_cap = 0;
    hipStream_t st = nullptr;
    int dtype = 0;

    const float* finish_extra = nullptr;         // one more f32 addend for the NEXT split-K finish (consumed by it)
    std::string* desc = nullptr;                 // dry runs: one line per planned launch (cp360_resnet_plan_describe)
    const char* where = "";
    void note(const char* text) {
        if (desc) { *desc += text; *desc += "\n"; }
    }

    int conv(const CConv& c, const void* in, int n_img, int h, int w, const void* residual, int ld_res, void* out,
             int ld_out, int out_coff, const void* x2, int h2, int w2, int ps2, bool raw, bool raw_slab_rows,
             float* raw_dst, int force_splits, int* splits_out) {
        cp360_conv_desc d;
        const int cr = use_clip(c, dtype, n_img, h, w) ? 1 : 0;
        fill_desc(c, dtype, n_img, h, w, 1, ld_out, out_coff, ld_res, cr, 0, h2, w2, ps2, &d);
        const int splits = force_splits > 0 ? force_splits : cp360_conv_suggest_splits(&d);
        const bool extra = finish_extra != nullptr && !raw;          // the sums go through cp360_conv_finish_add even at splits = 1
        const int sr = (c.c_out % 32 == 0 && (((splits > 1 || extra) && !raw) || (raw && raw_slab_rows))) ? 1 : 0;
        d.splits = splits;
        d.slab_rows = sr;
        if (splits_out) *splits_out = splits;
        const size_t M = (size_t)n_img * d.h_out * d.w_out;
        const bool to_partial = raw || splits > 1 || extra;
        // raw sums with a caller-owned destination AND split-K (the CAM scores at one frame): slabs in the workspace,
        // reduced into raw_dst by an f32 finish
        const bool raw_reduce = raw && raw_dst && splits > 1;
        if (to_partial && (!raw_dst || raw_reduce)) {
            const size_t need = (size_t)splits * M * c.c_out * sizeof(float);
            if (need > partial_need) partial_need = need;
            if (!dry && need > partial_cap) return CP360_ERR_BAD_SHAPE;
        }
        if (dry) {
            if (desc) {
                char plan[200], line[320];
                cp360_conv_desc dd = d;
                dd.splits = 1;
                if (cp360_conv_plan_describe(&dd, plan, sizeof(plan)) < 0) plan[0] = 0;
                snprintf(line, sizeof(line), "  %s conv %dx%d %d -> %d @ %dx%d%s: %s%s", where, c.kh_w, c.kw_w, c.c_in_w, c.c_out, h, w,
                         c.stride > 1 ? " stride 2" : "", plan, force_splits > 0 ? " (split count fixed by the caller)" : "");
                note(line);
            }
            return CP360_OK;
        }
        const void* pk = cr ? c.packed_clip : c.packed;
        if (!pk) return CP360_ERR_UNSUPPORTED;
        if (to_partial) {
            float* dst = raw_dst && !raw_reduce ? raw_dst : partial;
            int rc = cp360_conv_forward2(&d, in, x2, pk, nullptr, nullptr, nullptr, dst, st);
            if (rc) return rc;
            if (raw_reduce) {
                cp360_conv_desc df = d;
                df.dtype = CP360_F32;
                df.relu = 0;
                return cp360_conv_finish_add(&df, dst, nullptr, nullptr, nullptr, raw_dst, st);
            }
            if (raw) return rc;
            const float* ex = finish_extra;
            finish_extra = nullptr;
            return cp360_conv_finish_add(&d, dst, ex, c.bias, residual, out, st);
        }
        return cp360_conv_forward2(&d, in, x2, pk, c.bias, residual, out, nullptr, st);
    }
};

// ---------------------------------------------------------------- ResNet-50-cubic + CAM
struct CBlock {
    CConv c1, c2, c3;
    bool has_ds = false;
    int planes = 0, stride = 1;
    // fused-tail packings (16-bit types): layer1 K3d, layer2 / layer3 K3e
    void *w2 = nullptr, *w3f = nullptr, *wdf = nullptr, *w1f = nullptr;   // w1f: the chained NEXT conv1's fragments
    float *b2 = nullptr, *b3 = nullptr, *b1n = nullptr;
    void* w0f = nullptr;                          // layer1.0: its OWN conv1's fragments (run inside the tail kernel at 56x56 faces)
    float* b0 = nullptr;
    int next_c = 0;
};

struct CResnet {
    bool loaded = false;
    int dtype = 0, num_classes = 0;
    CConv stem, cam;
    void* stem_packed = nullptr;                 // resident-patch stem kernel (16-bit)
    std::vector<CBlock> layer[4];
};

struct CClstm {
    bool loaded = false;
    int dtype = 0, cin = 0, ch = 0;
    CConv c1, c2, g;
    CConv c1x, c1h;                              // Conv1 split into its x / h input halves (cp360_clstm_window, batched x half)
    bool split_ok = false;
    float* gbias = nullptr;
    // Winograd-domain filters U = G g G^T of the three convolutions (cp360_clstm_load_wino; csrc/wino.hip), 16-bit types
    void *u1 = nullptr, *u2 = nullptr, *ug = nullptr;
    bool wino_loaded = false;
};

// the Winograd descriptors of a cell update on n6 faces of face x face
void wino_descs(const CClstm& Cl, int n6, int face, cp360_wino_desc* d1, cp360_wino_desc* d2, cp360_wino_desc* dg) {
    const int c4 = 4 * Cl.ch, cx = Cl.cin + Cl.ch;
    *d1 = cp360_wino_desc{Cl.dtype, n6, face, cx, cx, c4, c4, 0, 1};
    *d2 = cp360_wino_desc{Cl.dtype, n6, face, c4, c4, c4, c4, 0, 1};
    *dg = cp360_wino_desc{Cl.dtype, n6, face, c4, c4, c4, c4, 0, 0};
}
// the library's rule, on Conv2's shape (the same test as ConvLSTMCell.uses_winograd of the CP360_CTX=0 path)
bool wino_wanted(const CClstm& Cl, int n_clips, int face) {
    if (Cl.dtype == CP360_F32) return false;
    cp360_wino_desc d1, d2, dg;
    wino_descs(Cl, 6 * n_clips, face, &d1, &d2, &dg);
    // every one of the three convolutions has to be a shape the Winograd kernels take (Conv1's K = input + hidden channels is
    // not Conv2's): a cell whose Conv1 fails the check stays on the direct kernels instead of failing in cp360_clstm_step
    if (!cp360_wino_v_bytes(&d1) || !cp360_wino_v_bytes(&d2) || !cp360_wino_v_bytes(&dg)) return false;
    return cp360_wino_preferred(&d2) == 1;
}

}  // namespace

struct cp360_ctx {
    int device = 0;
    Owned own_resnet, own_clstm;
    CResnet rn;
    CClstm cl;
};

extern "C" int cp360_fold_bn(const float* bn_weight, const float* bn_bias, const float* bn_mean, const float* bn_var, float eps,
                             float* scale, float* bias, int n, void* stream) {
    if (!bn_weight || !bn_bias || !bn_mean || !bn_var || !scale || !bias) return CP360_ERR_NULL;
    if (n <= 0) return CP360_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(fold_bn_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, bn_weight, bn_bias, bn_mean,
                       bn_var, eps, scale, bias, n);
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_create(int device, cp360_ctx** out) {
    if (!out) return CP360_ERR_NULL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CP360_ERR_HIP;
    cp360_ctx* c = new (std::nothrow) cp360_ctx();
    if (!c) return CP360_ERR_HIP;
    c->device = device;
    *out = c;
    return CP360_OK;
}

extern "C" void cp360_destroy(cp360_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    ctx->own_resnet.release();
    ctx->own_clstm.release();
    delete ctx;
}

namespace {

struct Folded { float* scale; float* bias; };

int fold(Owned& own, const cp360_conv_bn& p, int n, float eps, hipStream_t st, Folded* f) {
    if (!p.weight || !p.bn_weight || !p.bn_bias || !p.bn_mean || !p.bn_var) return CP360_ERR_NULL;
    f->scale = (float*)own.take((size_t)n * 4);
    f->bias = (float*)own.take((size_t)n * 4);
    if (!f->scale || !f->bias) return CP360_ERR_HIP;
    hipLaunchKernelGGL(fold_bn_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.bn_weight, p.bn_bias, p.bn_mean, p.bn_var,
                       eps, f->scale, f->bias, n);
    return CP360_OK;
}

void set_geom(CConv& c, int c_out, int c_in, int k, int stride, int pad, int relu) {
    c.c_out = c_out; c.c_in_w = c_in; c.kh_w = c.kw_w = k; c.stride = stride; c.pad = pad; c.relu = relu;
    c.c_in = c_in; c.kh = c.kw = k; c.pix_stride = c_in;
    c.clip_ok = pad == 1 && stride == 1 && k == 3 && c_out >= 256;
}

}  // namespace

// convs: torchvision order - [0] conv1 / bn1 (stem), then per layer L = 1..4 and block B: conv1, conv2, conv3 and, for
// block 0, the downsample pair behind conv3: 1 + 16 * 3 + 4 = 53 entries.
extern "C" int cp360_resnet_load(cp360_ctx* ctx, int dtype, const cp360_conv_bn* convs, int n_convs,
                                 const float* fc_weight, int num_classes, float fc_shift, float bn_eps, void* stream) {
    if (!ctx || !convs || !fc_weight) return CP360_ERR_NULL;
    if (dtype != CP360_F32 && dtype != CP360_BF16 && dtype != CP360_F16) return CP360_ERR_BAD_DTYPE;
    if (n_convs != 53 || num_classes <= 0 || num_classes % 4 != 0) return CP360_ERR_BAD_SHAPE;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return CP360_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    ctx->own_resnet.release();
    ctx->rn = CResnet();
    CResnet& R = ctx->rn;
    Owned& own = ctx->own_resnet;
    R.dtype = dtype;
    R.num_classes = num_classes;
    const bool h16 = dtype != CP360_F32;
    int rc;
    // ---- stem: conv 7x7 s2 (3 -> 64) in its "7 taps of 8 pixels x 4 channels" form + the resident-patch kernel's packing
    {
        Folded f;
        if ((rc = fold(own, convs[0], 64, bn_eps, st, &f))) return rc;
        CConv& c = R.stem;
        c.c_out = 64; c.c_in_w = 3; c.kh_w = c.kw_w = 7; c.stride = 2; c.pad = 0; c.relu = 1; c.stem = true;
        c.c_in = 32; c.kh = 7; c.kw = 1; c.pix_stride = 4;
        c.bias = f.bias;
        if ((rc = pack_conv(own, c, dtype, convs[0].weight, f.scale, nullptr, nullptr, true, false, st))) return rc;
        if (h16) {
            if (!(R.stem_packed = own.take(cp360_stem_packed_bytes(dtype)))) return CP360_ERR_HIP;
            if ((rc = cp360_stem_pack_weights(dtype, convs[0].weight, f.scale, R.stem_packed, st))) return rc;
        }
    }
    // ---- Bottlenecks
    static const int nblk[4] = {3, 4, 6, 3};
    int idx = 1, inplanes = 64;
    for (int L = 0; L < 4; ++L) {
        const int planes = 64 << L, lstride = L == 0 ? 1 : 2;
        R.layer[L].resize(nblk[L]);
        for (int b = 0; b < nblk[L]; ++b) {
            CBlock& B = R.layer[L][b];
            B.planes = planes;
            B.stride = b == 0 ? lstride : 1;
            B.has_ds = b == 0;
            const cp360_conv_bn &p1 = convs[idx], &p2 = convs[idx + 1], &p3 = convs[idx + 2];
            Folded f1, f2, f3, fd{nullptr, nullptr};
            if ((rc = fold(own, p1, planes, bn_eps, st, &f1))) return rc;
            if ((rc = fold(own, p2, planes, bn_eps, st, &f2))) return rc;
            if ((rc = fold(own, p3, planes * 4, bn_eps, st, &f3))) return rc;
            const cp360_conv_bn* pd = B.has_ds ? &convs[idx + 3] : nullptr;
            if (pd && (rc = fold(own, *pd, planes * 4, bn_eps, st, &fd))) return rc;
            idx += B.has_ds ? 4 : 3;
            set_geom(B.c1, planes, inplanes, 1, 1, 0, 1);
            set_geom(B.c2, planes, planes, 3, B.stride, 1, 1);
            set_geom(B.c3, planes * 4, planes, 1, 1, 0, 1);
            B.c1.bias = f1.bias;
            B.c2.bias = f2.bias;
            B.c3.bias = f3.bias;
            if ((rc = pack_conv(own, B.c1, dtype, p1.weight, f1.scale, nullptr, nullptr, true, false, st))) return rc;
            // conv2: layer4's (7x7 faces at cube 224 / 16x16 at cube 512) may run clip-resident: both layouts
            if ((rc = pack_conv(own, B.c2, dtype, p2.weight, f2.scale, nullptr, nullptr, true, B.c2.clip_ok, st))) return rc;
            if (B.has_ds) {                       // downsample branch as conv3's second source: bias = b3 + bd
                B.c3.c_in2 = inplanes;
                B.c3.stride2 = B.stride;
                B.c3.clip_ok = false;
                hipLaunchKernelGGL(add_vec_kernel, dim3((planes * 4 + 255) / 256), dim3(256), 0, st, f3.bias, fd.bias, planes * 4);
                if ((rc = pack_conv(own, B.c3, dtype, p3.weight, f3.scale, pd->weight, fd.scale, true, false, st))) return rc;
            } else if ((rc = pack_conv(own, B.c3, dtype, p3.weight, f3.scale, nullptr, nullptr, true, false, st))) {
                return rc;
            }
            // fused-tail packings (16-bit)
            if (h16 && L == 0) {
                if (!(B.w2 = own.take(cp360_l1block_conv2_bytes(dtype)))) return CP360_ERR_HIP;
                if ((rc = cp360_l1block_pack_conv2(dtype, p2.weight, f2.scale, B.w2, st))) return rc;
                B.b2 = f2.bias;
                if (!(B.w3f = own.take(cp360_frag_packed_bytes(dtype, 256, 64)))) return CP360_ERR_HIP;
                if ((rc = cp360_frag_pack_1x1(dtype, p3.weight, f3.scale, B.w3f, 256, 64, 0, st))) return rc;
                B.b3 = f3.bias;               // (with the downsample branch: already b3 + bd)
                if (B.has_ds) {
                    if (!(B.wdf = own.take(cp360_frag_packed_bytes(dtype, 256, 64)))) return CP360_ERR_HIP;
                    if ((rc = cp360_frag_pack_1x1(dtype, pd->weight, fd.scale, B.wdf, 256, 64, 0, st))) return rc;
                    if (b == 0 && inplanes == 64 && planes == 64) {       // the block's own conv1 (64 -> 64) for the in-patch form
                        if (!(B.w0f = own.take(cp360_frag_packed_bytes(dtype, 64, 64)))) return CP360_ERR_HIP;
                        if ((rc = cp360_frag_pack_1x1(dtype, p1.weight, f1.scale, B.w0f, 64, 64, 0, st))) return rc;
                        B.b0 = f1.bias;
                    }
                }
            } else if (h16 && L == 1 && B.has_ds) {          // layer2.0 after its conv1 as one launch (K3f, 56x56 -> 28x28 faces)
                if (!(B.w2 = own.take(cp360_l2block_packed_bytes(dtype)))) return CP360_ERR_HIP;
                if ((rc = cp360_l2block_pack_weights(dtype, p2.weight, f2.scale, B.w2, st))) return rc;
                B.b2 = f2.bias;
                if (!(B.w3f = own.take(cp360_l2first_w3d_bytes(dtype)))) return CP360_ERR_HIP;
                if ((rc = cp360_l2first_pack_w3d(dtype, p3.weight, f3.scale, pd->weight, fd.scale, B.w3f, st))) return rc;
                B.b3 = f3.bias;               // b3 + bd
            } else if (h16 && (L == 1 || L == 2) && !B.has_ds) {
                const size_t nb = L == 1 ? cp360_l2block_packed_bytes(dtype) : cp360_l3block_packed_bytes(dtype);
                if (!(B.w2 = own.take(nb))) return CP360_ERR_HIP;
                rc = L == 1 ? cp360_l2block_pack_weights(dtype, p2.weight, f2.scale, B.w2, st)
                            : cp360_l3block_pack_weights(dtype, p2.weight, f2.scale, B.w2, st);
                if (rc) return rc;
                B.b2 = f2.bias;
                if (!(B.w3f = own.take(cp360_frag_packed_bytes(dtype, planes * 4, planes)))) return CP360_ERR_HIP;
                if ((rc = cp360_frag_pack_1x1(dtype, p3.weight, f3.scale, B.w3f, planes * 4, planes, 0, st))) return rc;
                B.b3 = f3.bias;
            }
            // the NEXT conv1 chained onto the PREVIOUS tail kernel: layer1 blocks (256 -> 64, k-major fragments), layer1's
            // last block (layer2.0's conv1, 256 -> 128, k-major), layer2's identity blocks 2.. (512 -> 128, row-major)
            if (h16) {
                CBlock* prev = nullptr;
                int order = 1;
                if (L == 0 && b > 0) prev = &R.layer[0][b - 1];
                else if (L == 1 && b == 0) prev = &R.layer[0][nblk[0] - 1];
                else if (L == 1 && b >= 1) { prev = &R.layer[1][b - 1]; order = 0; }   // (b == 1: onto layer2.0's fused kernel)
                if (prev) {
                    if (!(prev->w1f = own.take(cp360_frag_packed_bytes(dtype, planes, inplanes)))) return CP360_ERR_HIP;
                    if ((rc = cp360_frag_pack_1x1(dtype, p1.weight, f1.scale, prev->w1f, planes, inplanes, order, st))) return rc;
                    prev->b1n = f1.bias;
                    prev->next_c = planes;
                }
            }
            inplanes = planes * 4;
        }
    }
    if (idx != 53) return CP360_ERR_BAD_SHAPE;
    // ---- CAM: 1x1 convolution with the shifted classifier weight (class_activation_model.py:46-52), raw f32 scores
    {
        const long long n = (long long)num_classes * 2048;
        float* w = (float*)own.take((size_t)n * 4);
        if (!w) return CP360_ERR_HIP;
        hipLaunchKernelGGL(sub_scalar_kernel, dim3(1024), dim3(256), 0, st, fc_weight, fc_shift, w, n);
        set_geom(R.cam, num_classes, 2048, 1, 1, 0, 0);
        if ((rc = pack_conv(own, R.cam, dtype, w, nullptr, nullptr, nullptr, true, false, st))) return rc;
    }
    CP360_CHECK_HIP();
    R.loaded = true;
    return CP360_OK;
}

namespace {

struct ResnetWs {
    size_t act = 0, partial = 0, border = 0;
    size_t total() const { return 4 * act + partial + border; }
};

// The static stage as a launch sequence (or, dry, its workspace needs).  faces_p3: [n_img, cd+6, cd+6, 4] NHWC4.
int resnet_run(cp360_ctx* ctx, bool dry, const void* faces_p3, int n_img, int cd, float* cam_out, void* feat_out,
               unsigned char* ws, size_t ws_bytes, hipStream_t st, ResnetWs* need, std::string* desc = nullptr) {
    CResnet& R = ctx->rn;
    if (!R.loaded) return CP360_ERR_NULL;
    if (n_img <= 0 || cd < 32 || cd % 32 != 0) return CP360_ERR_BAD_SHAPE;
    if (n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    const int dtype = R.dtype, es = es_of(dtype);
    const bool h16 = dtype != CP360_F32;
    ResnetWs w;
    w.act = align_up((size_t)n_img * (cd / 2) * (cd / 2) * 64 * es);       // the largest activation: stem output = layer1 output
    w.border = align_up(cp360_stem_pool_border_bytes(n_img));
    Run run;
    run.dry = dry;
    run.st = st;
    run.dtype = dtype;
    run.desc = dry ? desc : nullptr;
    unsigned char* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned char* border = nullptr;
    if (!dry) {
        if (!faces_p3 || !cam_out || !ws) return CP360_ERR_NULL;
        if (((size_t)ws & 255) != 0) return CP360_ERR_ALIGN;
        w.partial = need->partial;
        if (ws_bytes < w.total()) return CP360_ERR_BAD_SHAPE;
        for (int i = 0; i < 4; ++i) buf[i] = ws + i * w.act;
        run.partial = (float*)(ws + 4 * w.act);
        run.partial_cap = w.partial;
        border = ws + 4 * w.act + w.partial;
    }
    int rc = CP360_OK;
    int cur = 0;                                         // buf[cur] = the running activation x; the other three rotate
    auto other = [&](int k) { return buf[(cur + k) & 3]; };
    const int old_order = dry ? 0 : cp360_set_launch_order(2);
#define CK(call) do { rc = (call); if (rc) { if (!dry) cp360_set_launch_order(old_order); return rc; } } while (0)
    // ---- stem + CubePad(1) + max-pool
    int face = cd / 4;
    if (!dry) {
        if (h16 && cd == 224) {
            CK(cp360_stem_pool_forward(dtype, faces_p3, R.stem_packed, R.stem.bias, buf[cur], border, n_img, cd, st));
        } else {
            if (h16 && cd == 512) CK(cp360_stem_forward(dtype, faces_p3, R.stem_packed, R.stem.bias, other(1), n_img, cd, 1, st));
            else CK(run.conv(R.stem, faces_p3, n_img, cd + 6, cd + 6, nullptr, 0, other(1), 0, 0, nullptr, 0, 0, 0, false, false,
                             nullptr, 0, nullptr));
            CK(cp360_cubepad_maxpool3s2(other(1), buf[cur], n_img, cd / 2, 64, dtype, st));
        }
    } else if (!(h16 && (cd == 224 || cd == 512))) {
        run.note("stem: generic convolution (7 taps of 8 pixels x 4 channels) + cubepad_maxpool kernel");
        run.where = "stem";
        CK(run.conv(R.stem, nullptr, n_img, cd + 6, cd + 6, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, 0, false, false, nullptr, 0, nullptr));
    } else {
        run.note(cd == 224 ? "stem: FUSED stem + CubePad(1) + max-pool, one launch (K3a+K3b, cube 224, 16-bit)"
                           : "stem: resident-patch stem kernel (K3a, cube 512, 16-bit) + cubepad_maxpool kernel");
    }
    // per-convolution Bottleneck (Bottleneck.forward_nhwc): x = buf[cur] -> buf[cur] (rotated); mid0: conv1 already done
    char wherebuf[32];
    auto at = [&](int layer, int block) { snprintf(wherebuf, sizeof(wherebuf), "layer%d.%d", layer, block); run.where = wherebuf; };
    auto bottleneck = [&](CBlock& B, int hin, const void* mid0) -> int {
        const int hout = (hin + 2 - 3) / B.stride + 1;
        const void* mid = mid0;
        if (!mid) {
            int r = run.conv(B.c1, buf[cur], n_img, hin, hin, nullptr, 0, other(1), 0, 0, nullptr, 0, 0, 0, false, false, nullptr, 0, nullptr);
            if (r) return r;
            mid = other(1);
        }
        int r = run.conv(B.c2, mid, n_img, hin, hin, nullptr, 0, other(2), 0, 0, nullptr, 0, 0, 0, false, false, nullptr, 0, nullptr);
        if (r) return r;
        if (B.has_ds)
            r = run.conv(B.c3, other(2), n_img, hout, hout, nullptr, 0, other(3), 0, 0, buf[cur], hin, hin, B.c3.c_in2, false, false,
                         nullptr, 0, nullptr);
        else
            r = run.conv(B.c3, other(2), n_img, hout, hout, buf[cur], B.c3.c_out, other(3), 0, 0, nullptr, 0, 0, 0, false, false,
                         nullptr, 0, nullptr);
        cur = (cur + 3) & 3;
        return r;
    };
    // ---- layer1
    const void* mid2 = nullptr;                           // layer2.0's conv1 output when layer1's last tail computed it
    {
        std::vector<CBlock>& Lr = R.layer[0];
        if (h16 && (face == 56 || face == 128)) {
            // CP360_L1_FIRST=0: conv1 of block 0 as its own launch (the form of rounds 2-5; A/B) instead of inside the block's tail kernel
            static const int first_env = []() { const char* e = getenv("CP360_L1_FIRST"); return e ? atoi(e) : 1; }();
            const bool first_in = first_env && face == 56 && Lr[0].w0f != nullptr;
            run.note(first_in ? "layer1: ONE fused launch per Bottleneck (K3d: [block 0: its conv1 on the resident patch +] conv2 + conv3 + residual / downsample + the next conv1)"
                              : "layer1: conv1 of block 0, then ONE fused launch per Bottleneck (K3d: conv2 + conv3 + residual / downsample + the next conv1)");
            at(1, 0);
            // conv1 of the first block (its own launch, or inside the tail kernel), then ONE launch per Bottleneck (K3d); the chained conv1 output ping-pongs
            if (!first_in)
                CK(run.conv(Lr[0].c1, buf[cur], n_img, face, face, nullptr, 0, other(1), 0, 0, nullptr, 0, 0, 0, false, false, nullptr, 0, nullptr));
            if (!dry) {
                // x = buf[cur] (64 ch), mid = other(1); out -> other(2), mid' -> other(3)
                if (first_in)
                    CK(cp360_l1block_forward_first(dtype, buf[cur], Lr[0].w0f, Lr[0].b0, Lr[0].w2, Lr[0].b2, Lr[0].w3f, Lr[0].b3, Lr[0].wdf,
                                                   other(2), Lr[0].w1f, Lr[0].b1n, other(3), n_img, face, st));
                else
                    CK(cp360_l1block_forward(dtype, other(1), Lr[0].w2, Lr[0].b2, Lr[0].w3f, Lr[0].b3, nullptr, buf[cur], Lr[0].wdf,
                                             other(2), Lr[0].w1f, Lr[0].b1n, other(3), n_img, face, st));
                // from here: out = other(2), mid = other(3); free: buf[cur], other(1)
                unsigned char *o = other(2), *m = other(3), *f0 = buf[cur], *f1 = other(1);
                for (size_t b = 1; b < Lr.size(); ++b) {
                    CBlock& B = Lr[b];
                    if (B.next_c == 128)
                        CK(cp360_l1block_forward_wide(dtype, m, B.w2, B.b2, B.w3f, B.b3, o, f0, B.w1f, B.b1n, f1, n_img, face, st));
                    else
                        CK(cp360_l1block_forward(dtype, m, B.w2, B.b2, B.w3f, B.b3, o, nullptr, nullptr, f0, B.w1f, B.b1n,
                                                 B.w1f ? f1 : nullptr, n_img, face, st));
                    unsigned char *no = f0, *nm = f1;
                    f0 = o; f1 = m; o = no; m = nm;
                }
                // re-anchor the rotation: x = o; keep mid2 = m out of the way of the next block's temporaries
                for (int i = 0; i < 4; ++i) if (buf[i] == o) cur = i;
                if (Lr.back().next_c == 128) {
                    mid2 = m;
                    // the first layer2 block writes other(2) / other(3) before it is done with mid2 (= conv2's input):
                    // swap buffer roles so that mid2 sits in the other(1) slot, which that block leaves alone
                    for (int i = 0; i < 4; ++i)
                        if (buf[i] == m && i != ((cur + 1) & 3)) { unsigned char* t = buf[i]; buf[i] = buf[(cur + 1) & 3]; buf[(cur + 1) & 3] = t; break; }
                }
            } else if (Lr.back().next_c == 128) {
                mid2 = (const void*)1;                   // dry: only "conv1 is not run"
            }
        } else {
            run.note("layer1: GENERIC path, one launch per convolution (the fused tail kernels need a 16-bit type and 56x56 / 128x128 faces)");
            for (size_t b = 0; b < Lr.size(); ++b) { at(1, (int)b); CK(bottleneck(Lr[b], face, nullptr)); }
        }
    }
    // ---- layer2 / layer3: first block per convolution, identity blocks as fused tails where the kernels exist
    bool l2_mid1 = false;                                 // layer2.0's kernel also computed layer2.1's conv1
    for (int L = 1; L <= 2; ++L) {
        std::vector<CBlock>& Lr = R.layer[L];
        at(L + 1, 0);
        if (L == 1 && h16 && face == 56) {
            run.note("layer2.0: ONE fused launch after its conv1 (K3f: stride-2 conv2 + conv3 + downsample + layer2.1's conv1)");
            // layer2.0: conv1 (unless layer1's last tail kernel computed it), then ONE launch (K3f)
            const void* mid = mid2;
            if (!mid) {
                CK(run.conv(Lr[0].c1, buf[cur], n_img, face, face, nullptr, 0, other(1), 0, 0, nullptr, 0, 0, 0, false, false, nullptr, 0, nullptr));
                mid = other(1);
            }
            // out -> other(3); the chained conv1 of layer2.1 (when that block runs on the fused tail) -> other(2)
            l2_mid1 = Lr[0].w1f != nullptr;
            if (!dry) CK(cp360_l2first_forward(dtype, mid, Lr[0].w2, Lr[0].b2, Lr[0].w3f, Lr[0].b3, buf[cur], other(3),
                                               l2_mid1 ? Lr[0].w1f : nullptr, l2_mid1 ? Lr[0].b1n : nullptr,
                                               l2_mid1 ? other(2) : nullptr, n_img, face / 2, st));
            cur = (cur + 3) & 3;                          // x = old other(3); old other(2) = the new other(3)
            if (l2_mid1 && !dry) {                        // the identity loop expects its mid in other(1): swap the two free roles
                unsigned char* t = buf[(cur + 3) & 3]; buf[(cur + 3) & 3] = buf[(cur + 1) & 3]; buf[(cur + 1) & 3] = t;
            }
        } else {
            run.note(L == 1 ? "layer2.0: generic path, one launch per convolution (downsample branch inside conv3)"
                            : "layer3.0: generic path, one launch per convolution (downsample branch inside conv3)");
            CK(bottleneck(Lr[0], face, L == 1 ? mid2 : nullptr));
        }
        face /= 2;
        const bool fused = h16 && (L == 1 ? (face == 28 || face == 64) : (face == 14 || face == 32));
        if (!fused) {
            run.note(L == 1 ? "layer2.1-3: GENERIC path, one launch per convolution (the fused tail kernel needs a 16-bit type and 28x28 / 64x64 faces)"
                            : "layer3.1-5: GENERIC path, one launch per convolution (the fused tail kernel needs a 16-bit type and 14x14 / 32x32 faces)");
            for (size_t b = 1; b < Lr.size(); ++b) { at(L + 1, (int)b); CK(bottleneck(Lr[b], face, nullptr)); }
            continue;
        }
        run.note(L == 1 ? "layer2.1-3: conv1 + ONE fused tail launch per Bottleneck (K3e; at 28x28 faces the next block's conv1 rides on the tail)"
                        : "layer3.1-5: conv1 + ONE fused tail launch per Bottleneck (K3e, C = 256)");
        const bool chain = L == 1 && face == 28;          // the next block's conv1 rides on the tail kernel
        bool have_mid = L == 1 && l2_mid1;                // other(1) holds this block's conv1 output
        for (size_t b = 1; b < Lr.size(); ++b) {
            CBlock& B = Lr[b];
            at(L + 1, (int)b);
            if (!have_mid)
                CK(run.conv(B.c1, buf[cur], n_img, face, face, nullptr, 0, other(1), 0, 0, nullptr, 0, 0, 0, false, false, nullptr, 0, nullptr));
            if (dry) { have_mid = chain && B.w1f; continue; }
            if (L == 1 && chain && B.w1f) {
                CK(cp360_l2block_forward_next(dtype, other(1), B.w2, B.b2, B.w3f, B.b3, buf[cur], other(2), B.w1f, B.b1n, other(3),
                                              n_img, face, st));
                // x = other(2), mid = other(3): rotate by 2 so that other(1) is the new mid
                cur = (cur + 2) & 3;
                have_mid = true;
            } else {
                if (L == 1) CK(cp360_l2block_forward(dtype, other(1), B.w2, B.b2, B.w3f, B.b3, buf[cur], other(2), n_img, face, st));
                else CK(cp360_l3block_forward(dtype, other(1), B.w2, B.b2, B.w3f, B.b3, buf[cur], other(2), n_img, face, st));
                cur = (cur + 2) & 3;
                have_mid = false;
            }
        }
    }
    // ---- layer4: per convolution (conv2 on the clip-resident kernel at 7x7 / 16x16 faces)
    run.note("layer4: one launch per convolution (conv2 clip-resident or on small tiles by cp360_conv_prefer_clip; downsample inside conv3)");
    for (size_t b = 0; b < R.layer[3].size(); ++b) {
        at(4, (int)b);
        CK(bottleneck(R.layer[3][b], face, nullptr));
        if (b == 0) face /= 2;
    }
    run.note("CAM: 1x1 convolution with the shifted fc.weight, raw f32 scores (split-K reduced by an f32 finish when the planner splits)");
    run.where = "CAM";
    if (!dry) cp360_set_launch_order(old_order);
    // ---- CAM: raw f32 scores [n_img, face, face, num_classes] straight into the caller's buffer (no bias / activation)
    CK(run.conv(R.cam, buf[cur], n_img, face, face, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, 0, true, false, dry ? (float*)1 : cam_out, 0,
                nullptr));
    if (!dry && feat_out &&
        hipMemcpyAsync(feat_out, buf[cur], (size_t)n_img * face * face * 2048 * es, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return CP360_ERR_HIP;
#undef CK
    w.partial = align_up(run.partial_need);
    if (need && dry) *need = w;
    return CP360_OK;
}

}  // namespace

extern "C" int cp360_resnet_plan_describe(cp360_ctx* ctx, int n_img, int cube_dim, char* buf, size_t cap) {
    if (!ctx || !buf || cap == 0) return CP360_ERR_NULL;
    std::string text;
    char head[160];
    snprintf(head, sizeof(head), "static stage, %d faces of %d^2, %s:", n_img, cube_dim,
             ctx->rn.dtype == CP360_F32 ? "f32" : (ctx->rn.dtype == CP360_F16 ? "f16" : "bf16"));
    text = head;
    text += "\n";
    ResnetWs w;
    const int rc = resnet_run(ctx, true, nullptr, n_img, cube_dim, nullptr, nullptr, nullptr, 0, nullptr, &w, &text);
    if (rc) return rc;
    const size_t n = text.size() < cap - 1 ? text.size() : cap - 1;
    memcpy(buf, text.data(), n);
    buf[n] = 0;
    return (int)n;
}

extern "C" size_t cp360_resnet_workspace_bytes(cp360_ctx* ctx, int n_img, int cube_dim) {
    if (!ctx) return 0;
    ResnetWs w;
    if (resnet_run(ctx, true, nullptr, n_img, cube_dim, nullptr, nullptr, nullptr, 0, nullptr, &w)) return 0;
    return w.total();
}

extern "C" int cp360_resnet_forward(cp360_ctx* ctx, const void* faces_p3, int n_img, int cube_dim, float* cam_out,
                                    void* feat_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx) return CP360_ERR_NULL;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return CP360_ERR_HIP;
    ResnetWs w;
    int rc = resnet_run(ctx, true, nullptr, n_img, cube_dim, nullptr, nullptr, nullptr, 0, nullptr, &w);
    if (rc) return rc;
    return resnet_run(ctx, false, faces_p3, n_img, cube_dim, cam_out, feat_out, (unsigned char*)workspace, workspace_bytes,
                      (hipStream_t)stream, &w);
}

// ---------------------------------------------------------------- ConvLSTM cell
extern "C" int cp360_clstm_load(cp360_ctx* ctx, int dtype, const float* w1, const float* b1, const float* w2, const float* b2,
                                const float* wg, const float* bg, int input_size, int hidden_size, int face, void* stream) {
    if (!ctx || !w1 || !b1 || !w2 || !b2 || !wg || !bg) return CP360_ERR_NULL;
    if (dtype != CP360_F32 && dtype != CP360_BF16 && dtype != CP360_F16) return CP360_ERR_BAD_DTYPE;
    if (input_size <= 0 || hidden_size <= 0 || face <= 0 || hidden_size % 4 != 0) return CP360_ERR_BAD_SHAPE;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return CP360_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    ctx->own_clstm.release();
    ctx->cl = CClstm();
    CClstm& Cl = ctx->cl;
    Owned& own = ctx->own_clstm;
    Cl.dtype = dtype; Cl.cin = input_size; Cl.ch = hidden_size;
    const int c4 = 4 * hidden_size;
    set_geom(Cl.c1, c4, input_size + hidden_size, 3, 1, 1, 1);
    set_geom(Cl.c2, c4, c4, 3, 1, 1, 1);
    set_geom(Cl.g, c4, c4, 3, 1, 1, 0);
    auto own_copy = [&](const float* src, int n) -> float* {
        float* d = (float*)own.take((size_t)n * 4);
        if (d && hipMemcpyAsync(d, src, (size_t)n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return nullptr;
        return d;
    };
    if (!(Cl.c1.bias = own_copy(b1, c4)) || !(Cl.c2.bias = own_copy(b2, c4)) || !(Cl.gbias = own_copy(bg, c4))) return CP360_ERR_HIP;
    // one packing per convolution: the layout the kernel chosen for THIS face size reads (0.72 GB in 16-bit types)
    const bool clip = clip_geometry(Cl.c1, 6, face, face);
    int rc;
    if ((rc = pack_conv(own, Cl.c1, dtype, w1, nullptr, nullptr, nullptr, !clip, clip, st))) return rc;
    if ((rc = pack_conv(own, Cl.c2, dtype, w2, nullptr, nullptr, nullptr, !clip, clip, st))) return rc;
    if ((rc = pack_conv(own, Cl.g, dtype, wg, nullptr, nullptr, nullptr, !clip, clip, st))) return rc;
    // Conv1's x / h halves as convolutions of their own (the batched x half of cp360_clstm_window): contiguous f32 copies
    // of w1[:, :Cin] and w1[:, Cin:], packed, then dropped
    if (clip && input_size == hidden_size && input_size % 8 == 0) {
        set_geom(Cl.c1x, c4, input_size, 3, 1, 1, 0);
        set_geom(Cl.c1h, c4, hidden_size, 3, 1, 1, 1);
        Cl.c1h.pix_stride = input_size + hidden_size;            // reads the h half of the fused [x | h] buffer in place
        Cl.c1h.bias = Cl.c1.bias;
        float* tmp = nullptr;
        const size_t half = (size_t)input_size * 9 * sizeof(float), full = (size_t)(input_size + hidden_size) * 9 * sizeof(float);
        if (hipMalloc((void**)&tmp, (size_t)c4 * half) != hipSuccess) return CP360_ERR_HIP;
        rc = CP360_OK;
        for (int part = 0; part < 2 && !rc; ++part) {
            if (hipMemcpy2DAsync(tmp, half, (const char*)w1 + (part ? half : 0), full, half, c4, hipMemcpyDeviceToDevice, st) != hipSuccess)
                rc = CP360_ERR_HIP;
            if (!rc) rc = pack_conv(own, part ? Cl.c1h : Cl.c1x, dtype, tmp, nullptr, nullptr, nullptr, false, true, st);
        }
        if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = CP360_ERR_HIP;
        (void)hipFree(tmp);
        if (rc) return rc;
        Cl.split_ok = true;
    }
    CP360_CHECK_HIP();
    Cl.loaded = true;
    return CP360_OK;
}

// 0: a cell update on n_clips cubes of face x face runs on the direct kernels; 1: in the Winograd domain (filters loaded); 2: it
// would, once cp360_clstm_load_wino has been called - until then cp360_clstm_step / cp360_clstm_window take the direct kernels
extern "C" int cp360_clstm_wino_state(cp360_ctx* ctx, int n_clips, int face) {
    if (!ctx || !ctx->cl.loaded || n_clips <= 0 || face <= 0) return 0;
    if (!wino_wanted(ctx->cl, n_clips, face)) return 0;
    return ctx->cl.wino_loaded ? 1 : 2;
}

// Pack U = G g G^T of the three filters (f32 OIHW on the device, as cp360_clstm_load's) for the Winograd path: 16 / 9 of the
// direct packing's bytes (1.28 GB at 1000 hidden channels), so it is made only when a launch shape asks for it.
extern "C" int cp360_clstm_load_wino(cp360_ctx* ctx, const float* w1, const float* w2, const float* wg, void* stream) {
    if (!ctx || !w1 || !w2 || !wg) return CP360_ERR_NULL;
    CClstm& Cl = ctx->cl;
    if (!Cl.loaded) return CP360_ERR_NULL;
    if (Cl.dtype == CP360_F32) return CP360_ERR_BAD_DTYPE;
    if (Cl.wino_loaded) return CP360_OK;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return CP360_ERR_HIP;
    cp360_wino_desc d1, d2, dg;
    wino_descs(Cl, 6, 8, &d1, &d2, &dg);
    const cp360_wino_desc* ds[3] = {&d1, &d2, &dg};
    const float* ws[3] = {w1, w2, wg};
    void** us[3] = {&Cl.u1, &Cl.u2, &Cl.ug};
    for (int k = 0; k < 3; ++k) {
        const size_t nb = cp360_wino_packed_bytes(ds[k]);
        if (!nb) return CP360_ERR_UNSUPPORTED;
        if (!(*us[k] = ctx->own_clstm.take(nb))) return CP360_ERR_HIP;
        const int rc = cp360_wino_pack_weights(ds[k], ws[k], *us[k], stream);
        if (rc) return rc;
    }
    Cl.wino_loaded = true;
    return CP360_OK;
}

namespace {
struct ClstmWs {
    size_t act = 0, partial = 0;
    size_t wv = 0, wm = 0;                         // Winograd path: V (transformed input) and M (f32 position sums) instead of slabs
    size_t total() const { return 2 * act + partial + wv + wm; }
};

int clstm_run(cp360_ctx* ctx, bool dry, void* xh, const float* c_prev, float* c_next, float* h_f32, int n_clips, int face,
              const float* x_next, const float* minmax, size_t clip_stride, unsigned char* ws, size_t ws_bytes, hipStream_t st,
              ClstmWs* need) {
    CClstm& Cl = ctx->cl;
    if (!Cl.loaded) return CP360_ERR_NULL;
    if (n_clips <= 0 || face <= 0) return CP360_ERR_BAD_SHAPE;
    const int n6 = 6 * n_clips, c4 = 4 * Cl.ch, es = es_of(Cl.dtype);
    const int M = n6 * face * face;
    ClstmWs w;
    w.act = align_up((size_t)M * c4 * es);
    Run run;
    run.dry = dry;
    run.st = st;
    run.dtype = Cl.dtype;
    unsigned char *a1 = nullptr, *a2 = nullptr;
    if (!dry) {
        if (!xh || !c_prev || !c_next || !ws) return CP360_ERR_NULL;
        if (((size_t)ws & 255) != 0) return CP360_ERR_ALIGN;
        w.partial = need->partial;
        if (ws_bytes < w.total()) return CP360_ERR_BAD_SHAPE;
        a1 = ws;
        a2 = ws + w.act;
        run.partial = (float*)(ws + 2 * w.act);
        run.partial_cap = w.partial;
    }
    int rc, splits = 1;
    if (Cl.wino_loaded && wino_wanted(Cl, n_clips, face)) {
        // the same cell update in the Winograd domain (csrc/wino.hip): per convolution input transform -> 16 GEMMs -> output
        // transform (+ bias + ReLU); the Gates convolution's output transform is the gate kernel
        cp360_wino_desc d1, d2, dg;
        wino_descs(Cl, n6, face, &d1, &d2, &dg);
        // V / M are shared by the three convolutions: sized for the largest (Conv1's K = input + hidden channels exceeds Conv2's
        // 4 * hidden when input > 3 * hidden)
        w.wv = align_up(std::max(cp360_wino_v_bytes(&d1), std::max(cp360_wino_v_bytes(&d2), cp360_wino_v_bytes(&dg))));
        w.wm = align_up(std::max(cp360_wino_m_bytes(&d1), std::max(cp360_wino_m_bytes(&d2), cp360_wino_m_bytes(&dg))));
        if (dry) {
            if (need) *need = w;
            return CP360_OK;
        }
        if (ws_bytes < w.total()) return CP360_ERR_BAD_SHAPE;
        unsigned char* v = ws + 2 * w.act;
        float* m = (float*)(ws + 2 * w.act + w.wv);
        // between two convolutions the output transform of one and the input transform of the next are ONE launch (faces up to
        // 9 x 9: cp360_wino_output_input; larger faces: the two launches through the activation buffers a1 / a2)
        if ((rc = cp360_wino_input(&d1, xh, v, st))) return rc;
        if ((rc = cp360_wino_gemm(&d1, v, Cl.u1, m, st))) return rc;
        rc = cp360_wino_output_input(&d1, m, Cl.c1.bias, v, st);
        if (rc == CP360_ERR_UNSUPPORTED) {
            if ((rc = cp360_wino_output(&d1, m, Cl.c1.bias, a1, st))) return rc;
            rc = cp360_wino_input(&d2, a1, v, st);
        }
        if (rc) return rc;
        if ((rc = cp360_wino_gemm(&d2, v, Cl.u2, m, st))) return rc;
        rc = cp360_wino_output_input(&d2, m, Cl.c2.bias, v, st);
        if (rc == CP360_ERR_UNSUPPORTED) {
            if ((rc = cp360_wino_output(&d2, m, Cl.c2.bias, a2, st))) return rc;
            rc = cp360_wino_input(&dg, a2, v, st);
        }
        if (rc) return rc;
        if ((rc = cp360_wino_gemm(&dg, v, Cl.ug, m, st))) return rc;
        return cp360_wino_output_gates(&dg, m, Cl.gbias, c_prev, c_next, xh, Cl.cin + Cl.ch, Cl.cin, h_f32, x_next, minmax, 0,
                                       clip_stride, st);
    }
    // clstm.py:55-64: cat(x, h) -> [CubePad(1) + conv3x3 + bias + relu] x 2 -> CubePad(1) + conv3x3 (Gates, bias in the gate kernel)
    if ((rc = run.conv(Cl.c1, xh, n6, face, face, nullptr, 0, a1, c4, 0, nullptr, 0, 0, 0, false, false, nullptr, 0, nullptr))) return rc;
    if ((rc = run.conv(Cl.c2, a1, n6, face, face, nullptr, 0, a2, c4, 0, nullptr, 0, 0, 0, false, false, nullptr, 0, nullptr))) return rc;
    const bool sr = c4 % 32 == 0;                            // gate slabs in packed-row order (64-byte stores)
    if ((rc = run.conv(Cl.g, a2, n6, face, face, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, 0, true, sr, nullptr, 0, &splits))) return rc;
    w.partial = align_up(run.partial_need);
    if (dry) {
        if (need) *need = w;
        return CP360_OK;
    }
    // clstm.py:68-80: gates, cell / hidden update; the new hidden goes into the h half of xh; with x_next the next
    // frame's window normalisation (test_temporal.py:77) is written into the x half in the same pass
    return cp360_lstm_gates_next(run.partial, splits, Cl.gbias, c_prev, c_next, xh, Cl.dtype, Cl.cin + Cl.ch, Cl.cin, h_f32, M,
                                 Cl.ch, sr ? 1 : 0, x_next, minmax, 0, 6 * face * face, clip_stride, st);
}
}  // namespace

extern "C" size_t cp360_clstm_workspace_bytes(cp360_ctx* ctx, int n_clips, int face) {
    if (!ctx) return 0;
    ClstmWs w;
    if (clstm_run(ctx, true, nullptr, nullptr, nullptr, nullptr, n_clips, face, nullptr, nullptr, 0, nullptr, 0, nullptr, &w)) return 0;
    return w.total();
}

extern "C" int cp360_clstm_step(cp360_ctx* ctx, void* xh, const float* c_prev, float* c_next, float* h_f32, int n_clips,
                                int face, const float* x_next, const float* minmax, size_t clip_stride, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (!ctx) return CP360_ERR_NULL;
    if (x_next && (!minmax || ctx->cl.cin != ctx->cl.ch)) return CP360_ERR_BAD_SHAPE;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return CP360_ERR_HIP;
    ClstmWs w;
    int rc = clstm_run(ctx, true, nullptr, nullptr, nullptr, nullptr, n_clips, face, nullptr, nullptr, 0, nullptr, 0, nullptr, &w);
    if (rc) return rc;
    return clstm_run(ctx, false, xh, c_prev, c_next, h_f32, n_clips, face, x_next, minmax, clip_stride, (unsigned char*)workspace,
                     workspace_bytes, (hipStream_t)stream, &w);
}

// ---------------------------------------------------------------- one window of the temporal stage
namespace {
struct WindowWs {
    size_t step = 0, xn = 0, px = 0;             // the cell update's own workspace, normalised frames, x-half sums
    size_t total() const { return step + xn + px; }
};

bool window_batches_x(const CClstm& Cl, int n_clips) {
    // Opt-in (CP360_XBATCH=1; measured performance-neutral, docs/rounds/r01-r04_design.md section 3): it changes Conv1's accumulation order, so
    // the default window issues exactly the launches of T cp360_clstm_step calls and gives the same bits
    static const int env = []() { const char* e = getenv("CP360_XBATCH"); return e ? atoi(e) : 0; }();
    (void)n_clips;
    return Cl.split_ok && env != 0;
}

int window_plan(cp360_ctx* ctx, int n_clips, int T, int face, WindowWs* w) {
    CClstm& Cl = ctx->cl;
    if (!Cl.loaded) return CP360_ERR_NULL;
    if (n_clips <= 0 || T <= 0 || face <= 0 || Cl.cin != Cl.ch) return CP360_ERR_BAD_SHAPE;
    ClstmWs s;
    int rc = clstm_run(ctx, true, nullptr, nullptr, nullptr, nullptr, n_clips, face, nullptr, nullptr, 0, nullptr, 0, nullptr, &s);
    if (rc) return rc;
    w->step = s.total();
    if (window_batches_x(Cl, n_clips)) {
        const size_t M = (size_t)6 * n_clips * face * face;
        // the h half's split-K slabs can need more room than the full convolution's (another split count)
        Run run;
        run.dry = true;
        run.dtype = Cl.dtype;
        run.finish_extra = (const float*)16;                   // (dry: only "the finish has an extra addend", never read)
        if ((rc = run.conv(Cl.c1h, nullptr, 6 * n_clips, face, face, nullptr, 0, nullptr, 4 * Cl.ch, 0, nullptr, 0, 0, 0, false, false,
                           nullptr, 0, nullptr)))
            return rc;
        if (align_up(run.partial_need) > s.partial) w->step += align_up(run.partial_need) - s.partial;
        w->xn = align_up((size_t)T * M * Cl.cin * es_of(Cl.dtype));
        w->px = align_up((size_t)T * M * 4 * Cl.ch * sizeof(float));
    }
    return CP360_OK;
}
}  // namespace

extern "C" size_t cp360_clstm_window_workspace_bytes(cp360_ctx* ctx, int n_clips, int T, int face) {
    if (!ctx) return 0;
    WindowWs w;
    return window_plan(ctx, n_clips, T, face, &w) ? 0 : w.total();
}

extern "C" int cp360_clstm_window(cp360_ctx* ctx, const float* cam, size_t clip_stride, int n_clips, int T, int face, void* xh,
                                  float* cell0, float* cell1, float* h_out, float* h_all, float* minmax, float* mm_scratch,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx || !cam || !xh || !cell0 || !cell1 || !h_out || !minmax || !mm_scratch || !workspace) return CP360_ERR_NULL;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return CP360_ERR_HIP;
    CClstm& Cl = ctx->cl;
    WindowWs w;
    int rc = window_plan(ctx, n_clips, T, face, &w);
    if (rc) return rc;
    if (((size_t)workspace & 255) != 0) return CP360_ERR_ALIGN;
    if (workspace_bytes < w.total()) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int P = 6 * face * face, C = Cl.cin, H = Cl.ch, n6 = 6 * n_clips, c4 = 4 * H, es = es_of(Cl.dtype);
    const size_t M = (size_t)n_clips * P;
    const size_t per_clip = (size_t)T * P * C, stride = clip_stride ? clip_stride : per_clip;
    unsigned char* ws = (unsigned char*)workspace;
    // test_temporal.py:66-67: min / max over the whole window;  :70-73: hidden = cell = normalised frame 0
    if ((rc = cp360_window_minmax(cam, minmax, mm_scratch, n_clips, per_clip, stride, st))) return rc;
    if ((rc = cp360_window_normalize(cam, minmax, xh, Cl.dtype, C + H, C, cell0, n_clips, T, 0, P, C, stride, st))) return rc;
    const bool batch = window_batches_x(Cl, n_clips);
    float* px = nullptr;
    Run run;                                                  // for the batched path's convolutions
    run.st = st;
    run.dtype = Cl.dtype;
    if (batch) {
        unsigned char* xn = ws + w.step;
        px = (float*)(ws + w.step + w.xn);
        if ((rc = cp360_window_normalize_frames(cam, minmax, xn, Cl.dtype, n_clips, T, P, C, stride, st))) return rc;
        // x half of Conv1 for all T frames: M = T * 6 face^2 * n_clips, K = 9 Cin, no split-K, raw f32 sums in slab order
        if ((rc = run.conv(Cl.c1x, xn, T * n6, face, face, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, 0, true, c4 % 32 == 0, px, 1, nullptr)))
            return rc;
    } else if ((rc = cp360_window_normalize(cam, minmax, xh, Cl.dtype, C + H, 0, nullptr, n_clips, T, 0, P, C, stride, st))) {
        return rc;                                            // frame 0 is fed again first (:76-79)
    }
    float* c[2] = {cell0, cell1};
    for (int t = 0; t < T; ++t) {
        float* hf = h_all ? h_all + (size_t)t * M * H : (t == T - 1 ? h_out : nullptr);
        if (!batch) {
            const float* xnext = t + 1 < T ? cam + (size_t)(t + 1) * P * C : nullptr;   // frame t+1's normalisation rides on the gates
            if ((rc = cp360_clstm_step(ctx, xh, c[t & 1], c[(t + 1) & 1], hf, n_clips, face, xnext, minmax, stride, ws, w.step, st)))
                return rc;
            continue;
        }
        // Conv1 = finish(h half's split-K slabs + the x half's sums of frame t + bias) -> relu;  Conv2;  Gates;  gate epilogue
        unsigned char* a1 = ws;
        const size_t act = align_up(M * c4 * es);
        unsigned char* a2 = ws + act;
        run.partial = (float*)(ws + 2 * act);
        run.partial_cap = w.step - 2 * act;
        run.finish_extra = px + (size_t)t * M * c4;
        if ((rc = run.conv(Cl.c1h, (const unsigned char*)xh + (size_t)C * es, n6, face, face, nullptr, 0, a1, c4, 0, nullptr, 0, 0, 0, false,
                           false, nullptr, 0, nullptr)))
            return rc;
        if ((rc = run.conv(Cl.c2, a1, n6, face, face, nullptr, 0, a2, c4, 0, nullptr, 0, 0, 0, false, false, nullptr, 0, nullptr))) return rc;
        int splits = 1;
        const bool sr = c4 % 32 == 0;
        if ((rc = run.conv(Cl.g, a2, n6, face, face, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, 0, true, sr, nullptr, 0, &splits))) return rc;
        if ((rc = cp360_lstm_gates_next(run.partial, splits, Cl.gbias, c[t & 1], c[(t + 1) & 1], xh, Cl.dtype, C + H, C, hf, (int)M, H,
                                        sr ? 1 : 0, nullptr, nullptr, 0, P, 0, st)))
            return rc;
    }
    if (h_all && hipMemcpyAsync(h_out, h_all + (size_t)(T - 1) * M * H, M * H * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return CP360_ERR_HIP;
    return CP360_OK;
}
