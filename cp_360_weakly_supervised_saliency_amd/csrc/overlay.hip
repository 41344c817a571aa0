// K9: the overlay renderer of utils/utils.py:9-25 (temporal_model/test_temporal.py:90-97) on the device:
//   heatmap = heatmap - min; heatmap /= max; colorize(heatmap, bytes=True)[..., :3]   (matplotlib 'jet', 256 entries)
//   heatmap.resize(img.size, resample=Image.CUBIC)        -> resize.hip with Pillow's bicubic coefficient tables
//   Image.blend(img, heatmap, alpha)                       -> overlay_blend_kernel
// Bit-exact with matplotlib + Pillow: float32 arithmetic in numpy's operation order for the normalisation and the
// colormap index (x * 256 truncated, 1.0 -> 255, NaN -> the colormap's all-zero "bad" colour), Pillow's
// (UINT8)(in1 + alpha * (in2 - in1)) in float for the blend.
#include "common.h"

// numpy / Pillow evaluate a * b + c as two rounded operations: no FMA contraction in this file (HIP's default
// -ffp-contract=fast would fuse them and change the last bit).  Plain operators only: HIP's __fadd_rn / __fmul_rn are header
// functions compiled with contraction ON, and their instructions keep that flag when inlined here.
#pragma clang fp contract(off)

namespace {
__global__ __launch_bounds__(1024) void overlay_colorize_kernel(const float* __restrict__ heat, int n, int square,
                                                                const uint8_t* __restrict__ lut /* [256, 3] */,
                                                                uint8_t* __restrict__ rgb) {
    __shared__ float smn[1024], smx[1024];
    const int tid = threadIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    bool nan = false;
    for (int i = tid; i < n; i += 1024) {
        float v = heat[i];
        if (square) v = ((v) * (v));                      // test_temporal.py:94: equi_output ** 2
        nan |= v != v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    if (nan) mn = mx = NAN;                                   // np.min / np.max propagate NaN
    smn[tid] = mn;
    smx[tid] = mx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
            const float a = smn[tid], b = smn[tid + s], c = smx[tid], d = smx[tid + s];
            smn[tid] = (a != a || b != b) ? NAN : fminf(a, b);
            smx[tid] = (c != c || d != d) ? NAN : fmaxf(c, d);
        }
        __syncthreads();
    }
    const float gmn = smn[0];
    const float gmx = ((smx[0]) - (gmn));                 // max of (heat - min) = max - min (rounding is monotone)
    for (int i = tid; i < n; i += 1024) {
        float v = heat[i];
        if (square) v = ((v) * (v));
        v = (v - gmn) / gmx;
        float xa = v * 256.f;
        if (xa == 256.f) xa = 255.f;
        uint8_t r = 0, g = 0, b = 0;                          // "bad" (NaN): (0, 0, 0, 0)
        if (xa == xa) {
            const int k = xa < 0.f ? 0 : (xa >= 256.f ? 255 : (int)xa);     // under -> lut[0], over -> lut[255]
            r = lut[3 * k]; g = lut[3 * k + 1]; b = lut[3 * k + 2];
        }
        rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
    }
}

__global__ __launch_bounds__(256) void overlay_blend_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                            uint8_t* __restrict__ out, float alpha, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int x = a[i], y = b[i];
        out[i] = (uint8_t)((float)x + alpha * (float)(y - x));
    }
}
}  // namespace

extern "C" int cp360_overlay_colorize(const float* heat, int h, int w, int square, const uint8_t* lut768, uint8_t* rgb,
                                      void* stream) {
    if (!heat || !lut768 || !rgb) return CP360_ERR_NULL;
    if (h <= 0 || w <= 0) return CP360_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(overlay_colorize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, heat, h * w, square, lut768, rgb);
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_overlay_blend_u8(const uint8_t* img, const uint8_t* heat_rgb, uint8_t* out, long long n_bytes,
                                      float alpha, void* stream) {
    if (!img || !heat_rgb || !out) return CP360_ERR_NULL;
    if (n_bytes <= 0) return CP360_ERR_BAD_SHAPE;
    if (!(alpha >= 0.f && alpha <= 1.f)) return CP360_ERR_UNSUPPORTED;      // Pillow's clipping path for other alphas
    long long blocks = (n_bytes + 255) / 256;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    hipLaunchKernelGGL(overlay_blend_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, heat_rgb, out,
                       alpha, n_bytes);
    CP360_CHECK_HIP();
    return CP360_OK;
}
