// K0: PIL-exact Lanczos resize of decoded frames.
//
// dataset_feat_extractor.py:119-142 resizes every frame with
//   Image.fromarray(frame).convert('RGB').resize((1920, 960), resample=Image.LANCZOS)
// before /255 and the cube projection.  Pillow's 8-bit resampler (src/libImaging/Resample.c):
// separable, horizontal pass first, intermediate rounded to uint8, then the vertical pass; per output
// index a window [xmin, xmin + n) of 22-bit fixed-point coefficients, int32 accumulation starting at
// 1 << 21, arithmetic shift, clip to 0..255.  The coefficient tables are built on the host in double
// precision exactly as Pillow does (cp360_resize_coeffs_host); the kernels only gather: HBM-bound
// integer work, bit-exact by construction (tests compare with PIL itself).
#include "common.h"
#include <math.h>
#include <stdlib.h>

#define CP360_RESIZE_PRECISION_BITS (32 - 8 - 2)

static double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

// Pillow's bicubic (Image.CUBIC / BICUBIC): Keys kernel with a = -0.5, support 2 (Resample.c: bicubic_filter) -
// the filter of utils/utils.py:21 (overlay: heatmap.resize(img.size, resample=Image.CUBIC))
static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static double filter_support(int filter) { return filter == 1 ? 2.0 : 3.0; }

extern "C" int cp360_resize_ksize2(int in_size, int out_size, int filter) {
    if (in_size <= 0 || out_size <= 0) return CP360_ERR_BAD_SHAPE;
    if (filter != 0 && filter != 1) return CP360_ERR_UNSUPPORTED;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(filter_support(filter) * filterscale) * 2 + 1;
}
extern "C" int cp360_resize_ksize(int in_size, int out_size) { return cp360_resize_ksize2(in_size, out_size, 0); }

extern "C" int cp360_resize_coeffs_host(int in_size, int out_size, int* bounds, int* kk) {
    return cp360_resize_coeffs_host2(in_size, out_size, 0, bounds, kk);
}

extern "C" int cp360_resize_coeffs_host2(int in_size, int out_size, int filter, int* bounds, int* kk) {
    if (!bounds || !kk) return CP360_ERR_NULL;
    const int ksize = cp360_resize_ksize2(in_size, out_size, filter);
    if (ksize < 0) return ksize;
    double scale = (double)in_size / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = filter_support(filter) * filterscale, ss = 1.0 / filterscale;
    double* w = (double*)malloc(sizeof(double) * ksize);
    if (!w) return CP360_ERR_HIP;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double arg = (x + xmin - center + 0.5) * ss;
            w[x] = filter == 1 ? bicubic_filter(arg) : lanczos_filter(arg);
            ww += w[x];
        }
        int* k = kk + (size_t)xx * ksize;
        for (int x = 0; x < ksize; ++x) {
            double v = 0.0;
            if (x < xmax) v = (ww != 0.0) ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << CP360_RESIZE_PRECISION_BITS))
                         : (int)(0.5 + v * (1 << CP360_RESIZE_PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    free(w);
    return ksize;
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= CP360_RESIZE_PRECISION_BITS;
    return (uint8_t)min(max(v, 0), 255);
}

// One pass along x (HORIZ) or y of u8 [F, H, W, 3] images: a thread produces the 3 channels of one
// output pixel; neighbouring threads are neighbouring output columns (coalesced stores, overlapping
// cached loads).
template <bool HORIZ>
__global__ __launch_bounds__(256) void resize_pass_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          const int* __restrict__ bounds, const int* __restrict__ kk,
                                                          int ksize, int F, int h_in, int w_in, int h_out, int w_out) {
    const long long total = (long long)F * h_out * w_out;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w_out);
        const long long t = idx / w_out;
        const int y = (int)(t % h_out), f = (int)(t / h_out);
        const int o = HORIZ ? x : y;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = kk + (size_t)o * ksize;
        const uint8_t* src = in + ((size_t)f * h_in * w_in + (HORIZ ? (size_t)y * w_in + lo : (size_t)lo * w_in + x)) * 3;
        const size_t step = HORIZ ? 3 : (size_t)w_in * 3;
        int s0 = 1 << (CP360_RESIZE_PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int i = 0; i < n; ++i) {
            const int c = k[i];
            s0 += c * src[0];
            s1 += c * src[1];
            s2 += c * src[2];
            src += step;
        }
        uint8_t* dst = out + (size_t)idx * 3;
        dst[0] = clip8(s0);
        dst[1] = clip8(s1);
        dst[2] = clip8(s2);
    }
}

// ---- faster passes for the common shapes (row bytes a multiple of 4; windows of at most 1024 input pixels per 256
// outputs).  The kernel above issues 3 byte loads per tap and pixel: 0.10 of the HBM peak (profiles/r02_hbm_kernels.md).
// Vertical: a row is W*3 independent bytes with the same coefficients, so a thread owns FOUR consecutive bytes (one
// dword load per tap, coalesced across the row).  Horizontal: a workgroup stages the input window of 256 output pixels
// in LDS with dword loads and the taps read bytes from there.  Both: XCD-contiguous work ranges (neighbouring output rows
// share input rows).  Same integer arithmetic, bit-exact.
__global__ __launch_bounds__(256) void resize_v4_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        const int* __restrict__ bounds, const int* __restrict__ kk,
                                                        int ksize, int F, int h_in, int row_dwords, int h_out) {
    const long long total = (long long)F * h_out * row_dwords;
    const long long chunk = (total + 7) / 8;
    const long long lo_w = (blockIdx.x & 7) * chunk, hi_w = min(total, lo_w + chunk);
    const long long stride = (long long)(gridDim.x >> 3) * 256;
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(in);
    uint32_t* out32 = reinterpret_cast<uint32_t*>(out);
    for (long long idx = lo_w + (long long)(blockIdx.x >> 3) * 256 + threadIdx.x; idx < hi_w; idx += stride) {
        const int xd = (int)(idx % row_dwords);
        const long long t = idx / row_dwords;
        const int y = (int)(t % h_out), f = (int)(t / h_out);
        const int lo = bounds[2 * y], n = bounds[2 * y + 1];
        const int* k = kk + (size_t)y * ksize;
        const uint32_t* src = in32 + ((size_t)f * h_in + lo) * row_dwords + xd;
        int s0 = 1 << (CP360_RESIZE_PRECISION_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
        for (int i = 0; i < n; ++i) {
            const int c = k[i];
            const uint32_t v = src[(size_t)i * row_dwords];
            s0 += c * (int)(v & 0xff);
            s1 += c * (int)((v >> 8) & 0xff);
            s2 += c * (int)((v >> 16) & 0xff);
            s3 += c * (int)(v >> 24);
        }
        out32[idx] = (uint32_t)clip8(s0) | ((uint32_t)clip8(s1) << 8) | ((uint32_t)clip8(s2) << 16) | ((uint32_t)clip8(s3) << 24);
    }
}

constexpr int RH_MAX_PX = 1024;          // input pixels a workgroup's window may span
__global__ __launch_bounds__(256) void resize_h_lds_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           const int* __restrict__ bounds, const int* __restrict__ kk,
                                                           int ksize, int rows, int w_in, int w_out, int xblocks,
                                                           size_t in_bytes) {
    __shared__ uint32_t win[(RH_MAX_PX * 3 + 8) / 4 + 1];
    // XCD-contiguous ranges of (row, x block) items
    const int items = rows * xblocks;
    const int chunk = (items + 7) / 8;
    const int lo_i = (blockIdx.x & 7) * chunk, hi_i = min(items, lo_i + chunk);
    for (int item = lo_i + (int)(blockIdx.x >> 3); item < hi_i; item += (int)(gridDim.x >> 3)) {
        const int row = item / xblocks, xb = item - row * xblocks;
        const int x_first = xb * 256, x_last = min(w_out - 1, x_first + 255);
        const int lo0 = bounds[2 * x_first];
        const int hi0 = bounds[2 * x_last] + bounds[2 * x_last + 1];
        const size_t b0 = ((size_t)row * w_in + lo0) * 3, b1 = ((size_t)row * w_in + hi0) * 3;
        const size_t a0 = b0 & ~(size_t)3;                         // dword-aligned start (the tensor base is 4-byte aligned)
        const int ndw = (int)((b1 - a0 + 3) >> 2);
        __syncthreads();                                           // the previous item's readers are done
        for (int i = threadIdx.x; i < ndw; i += 256) {
            const size_t a = a0 + 4 * (size_t)i;
            uint32_t v;
            if (a + 4 <= in_bytes) v = *reinterpret_cast<const uint32_t*>(in + a);
            else {
                v = 0;
                for (int e = 0; e < 4 && a + e < in_bytes; ++e) v |= (uint32_t)in[a + e] << (8 * e);
            }
            win[i] = v;
        }
        __syncthreads();
        const int x = x_first + threadIdx.x;
        if (x <= x_last) {
            const int lo = bounds[2 * x], n = bounds[2 * x + 1];
            const int* k = kk + (size_t)x * ksize;
            const uint8_t* w8 = reinterpret_cast<const uint8_t*>(win) + (b0 - a0) + (size_t)(lo - lo0) * 3;
            int s0 = 1 << (CP360_RESIZE_PRECISION_BITS - 1), s1 = s0, s2 = s0;
            for (int i = 0; i < n; ++i) {
                const int c = k[i];
                s0 += c * w8[3 * i];
                s1 += c * w8[3 * i + 1];
                s2 += c * w8[3 * i + 2];
            }
            uint8_t* dst = out + ((size_t)row * w_out + x) * 3;
            dst[0] = clip8(s0);
            dst[1] = clip8(s1);
            dst[2] = clip8(s2);
        }
    }
}

// host check for the LDS kernel: the widest input window of any 256-output block (the tables are device memory, so the
// bound comes from the geometry: support * scale on both sides + 256 * scale)
static bool h_window_fits(int w_in, int w_out, int ksize) {
    const double scale = (double)w_in / w_out;
    return 256.0 * scale + ksize + 2 <= RH_MAX_PX;
}

extern "C" int cp360_resize_lanczos_u8(const void* in, void* out, void* tmp, int F, int h_in, int w_in, int h_out,
                                       int w_out, const int* hbounds, const int* hkk, int hksize, const int* vbounds,
                                       const int* vkk, int vksize, void* stream) {
    if (!in || !out) return CP360_ERR_NULL;
    if (F <= 0 || h_in <= 0 || w_in <= 0 || h_out <= 0 || w_out <= 0) return CP360_ERR_BAD_SHAPE;
    const bool need_h = w_out != w_in, need_v = h_out != h_in;
    if (need_h && (!hbounds || !hkk || hksize <= 0)) return CP360_ERR_NULL;
    if (need_v && (!vbounds || !vkk || vksize <= 0)) return CP360_ERR_NULL;
    if (need_h && need_v && !tmp) return CP360_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    auto blocks_of = [](long long total) {
        long long b = (total + 255) / 256;
        return (unsigned)(b > 65535 * 4 ? 65535 * 4 : b);
    };
    if (!need_h && !need_v) {
        if (hipMemcpyAsync(out, in, (size_t)F * h_in * w_in * 3, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return CP360_ERR_HIP;
        return CP360_OK;
    }
    const uint8_t* cur = (const uint8_t*)in;
    static const int slow = []() { const char* e = getenv("CP360_RESIZE_BYTEWISE"); return e ? atoi(e) : 0; }();   // A/B switch
    if (need_h) {       // [F, h_in, w_in] -> [F, h_in, w_out]
        uint8_t* dst = need_v ? (uint8_t*)tmp : (uint8_t*)out;
        // (the window kernel loads aligned dwords relative to the input base: a base that is not 4-byte aligned - a slice
        // of a batch of odd-sized images - takes the bytewise pass)
        if (!slow && h_window_fits(w_in, w_out, hksize) && (long long)F * h_in < (1 << 24) && (reinterpret_cast<size_t>(cur) & 3) == 0) {
            const int xblocks = (w_out + 255) / 256;
            long long items = (long long)F * h_in * xblocks;
            unsigned blocks = (unsigned)(items > 256 * 16 ? 256 * 16 : (items + 7) / 8 * 8);
            hipLaunchKernelGGL(resize_h_lds_kernel, dim3(blocks), dim3(256), 0, st, cur, dst, hbounds, hkk, hksize, F * h_in,
                               w_in, w_out, xblocks, (size_t)F * h_in * w_in * 3);
        } else {
            hipLaunchKernelGGL((resize_pass_kernel<true>), dim3(blocks_of((long long)F * h_in * w_out)), dim3(256), 0, st,
                               cur, dst, hbounds, hkk, hksize, F, h_in, w_in, h_in, w_out);
        }
        cur = dst;
    }
    if (need_v) {       // [F, h_in, w_out] -> [F, h_out, w_out]
        if (!slow && (w_out * 3) % 4 == 0 && (reinterpret_cast<size_t>(cur) & 3) == 0 && (reinterpret_cast<size_t>(out) & 3) == 0) {
            const long long total = (long long)F * h_out * (w_out * 3 / 4);
            long long b = (total + 255) / 256;
            if (b > 256 * 32) b = 256 * 32;
            b = (b + 7) / 8 * 8;
            hipLaunchKernelGGL(resize_v4_kernel, dim3((unsigned)b), dim3(256), 0, st, cur, (uint8_t*)out, vbounds, vkk, vksize,
                               F, h_in, w_out * 3 / 4, h_out);
        } else {
            hipLaunchKernelGGL((resize_pass_kernel<false>), dim3(blocks_of((long long)F * h_out * w_out)), dim3(256), 0, st,
                               cur, (uint8_t*)out, vbounds, vkk, vksize, F, h_in, w_out, h_out, w_out);
        }
    }
    CP360_CHECK_HIP();
    return CP360_OK;
}
