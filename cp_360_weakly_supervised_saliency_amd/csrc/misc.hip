// HBM-bound glue kernels of the fused pipeline: CubePad+max-pool (K3b), layout / dtype
// conversion at the module boundary, window min/max + normalise (K7).
#include "common.h"

template <typename T> __device__ __forceinline__ float ld_f32(const T* p);
template <> __device__ __forceinline__ float ld_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f32<bf16_raw>(const bf16_raw* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st_f32(T* p, float v);
template <> __device__ __forceinline__ void st_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f32<bf16_raw>(bf16_raw* p, float v) { *p = f32_to_bf16(v); }
template <> __device__ __forceinline__ float ld_f32<f16_raw>(const f16_raw* p) { return (float)*p; }
template <> __device__ __forceinline__ void st_f32<f16_raw>(f16_raw* p, float v) { *p = (f16_raw)v; }

// ------------------------------------------------------------------ K3b
// resnet_cubic.py:169-170: x = pad1(x); x = maxpool(x)  (3x3, stride 2, padding 0) on
// NHWC, CubePad(1) fused through cubepad_src.  Lanes run along channels (contiguous).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void cubepad_maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int n6,
                                                              int n, int C, int ho) {
    const CubePadGeom g{n, 1, 1, 1, 1};
    const int cv = C / VEC;
    const long long total = (long long)n6 * ho * ho * cv;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) * VEC;
        long long t = idx / cv;
        const int ox = (int)(t % ho);
        t /= ho;
        const int oy = (int)(t % ho);
        const int img = (int)(t / ho);
        const int grp = img / 6, f = img - grp * 6;
        float m[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int s = cubepad_src(f, oy * 2 + ky, ox * 2 + kx, g);
                const T* src = x + ((size_t)grp * 6 * n * n + s) * C + c;
#pragma unroll
                for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], ld_f32<T>(src + e));
            }
        T* dst = y + (size_t)(t * ho + ox) * C + c;
#pragma unroll
        for (int e = 0; e < VEC; ++e) st_f32<T>(dst + e, m[e]);
    }
}

// 16-bit types with C % 8 == 0: a thread owns 8 channels = one 16-byte piece (half the index
// arithmetic of the 4-channel form per byte moved, 16-byte loads / stores).
typedef __attribute__((ext_vector_type(4))) unsigned int mp_u32x4;
template <typename T>
__global__ __launch_bounds__(256) void cubepad_maxpool16_kernel(const T* __restrict__ x, T* __restrict__ y, int n6, int n,
                                                                int C, int ho, int reverse) {
    const CubePadGeom g{n, 1, 1, 1, 1};
    const int cv = C / 8;
    const long long total = (long long)n6 * ho * ho * cv;
    // XCD-aware mapping (gridDim.x % 8 == 0): each XCD's L2 serves one contiguous eighth of the outputs - with the
    // round-robin block order the input rows shared by neighbouring output rows crossed the fabric once per XCD
    // (1.12 GB measured against 0.77 GB algorithmic, profiles/r02a_bf16.md)
    const long long chunk = (total + 7) / 8;
    const long long lo = (blockIdx.x & 7) * chunk, hi = min(total, lo + chunk);
    const long long stride = (long long)(gridDim.x >> 3) * blockDim.x;
    for (long long i0 = lo + (long long)(blockIdx.x >> 3) * blockDim.x + threadIdx.x; i0 < hi; i0 += stride) {
        const long long idx = reverse ? total - 1 - i0 : i0;       // descending order (cp360_set_launch_order)
        const int c = (int)(idx % cv) * 8;
        long long t = idx / cv;
        const int ox = (int)(t % ho);
        t /= ho;
        const int oy = (int)(t % ho);
        const int img = (int)(t / ho);
        const int grp = img / 6, f = img - grp * 6;
        mp_u32x4 v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int s = cubepad_src(f, oy * 2 + k / 3, ox * 2 + k % 3, g);
            v[k] = *reinterpret_cast<const mp_u32x4*>(x + ((size_t)grp * 6 * n * n + s) * C + c);
        }
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const T* pv = reinterpret_cast<const T*>(&v[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], ld_f32<T>(pv + e));
        }
        mp_u32x4 o;
        T* po = reinterpret_cast<T*>(&o);
#pragma unroll
        for (int e = 0; e < 8; ++e) st_f32<T>(po + e, m[e]);      // max of representable values: exact
        *reinterpret_cast<mp_u32x4*>(y + (size_t)(t * ho + ox) * C + c) = o;
    }
}

extern "C" int cp360_cubepad_maxpool3s2(const void* x, void* y, int n6, int n, int C, int dtype, void* stream) {
    if (!x || !y) return CP360_ERR_NULL;
    if (n6 <= 0 || n < 2 || C <= 0) return CP360_ERR_BAD_SHAPE;
    if (n6 % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (C % 4 != 0) return CP360_ERR_ALIGN;
    const int ho = (n + 2 - 3) / 2 + 1;
    const long long total = (long long)n6 * ho * ho * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    blocks = (blocks + 7) / 8 * 8;                   // the 16-byte kernel's XCD mapping
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CP360_F32)
        hipLaunchKernelGGL((cubepad_maxpool_kernel<float, 4>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const float*)x, (float*)y, n6, n, C, ho);
    else if (dtype == CP360_BF16 && C % 8 == 0)
        hipLaunchKernelGGL((cubepad_maxpool16_kernel<bf16_raw>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const bf16_raw*)x, (bf16_raw*)y, n6, n, C, ho, cp360_launch_reverse());
    else if (dtype == CP360_F16 && C % 8 == 0)
        hipLaunchKernelGGL((cubepad_maxpool16_kernel<f16_raw>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const f16_raw*)x, (f16_raw*)y, n6, n, C, ho, cp360_launch_reverse());
    else if (dtype == CP360_BF16)
        hipLaunchKernelGGL((cubepad_maxpool_kernel<bf16_raw, 4>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const bf16_raw*)x, (bf16_raw*)y, n6, n, C, ho);
    else if (dtype == CP360_F16)
        hipLaunchKernelGGL((cubepad_maxpool_kernel<f16_raw, 4>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const f16_raw*)x, (f16_raw*)y, n6, n, C, ho);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

// ------------------------------------------------------------------ layout conversion
// 32x32 LDS-tiled transpose x[n][r][c] -> y[n][c][r]: both the global read and the
// global write run along the contiguous dimension of their tensor.  The NHWC side may
// be a channel slice of a wider pixel (ld / coff), e.g. the x- or h-half of the
// ConvLSTM's concatenated input (clstm.py:55 torch.cat((input_, prev_hidden), 1)).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ x, TO* __restrict__ y, int R, int Cc,
                                                        size_t xs_n, size_t xs_r, size_t ys_n, size_t ys_c) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const TI* xin = x + (size_t)n * xs_n;
    TO* yout = y + (size_t)n * ys_n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < R && c < Cc) tile[ty + 8 * k][tx] = ld_f32<TI>(xin + (size_t)r * xs_r + c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < R && c < Cc) st_f32<TO>(yout + (size_t)c * ys_c + r, tile[tx][ty + 8 * k]);
    }
}

template <typename TI, typename TO>
static int launch_transpose(const void* x, void* y, int N, int R, int Cc, size_t xs_n, size_t xs_r, size_t ys_n,
                            size_t ys_c, hipStream_t st) {
    dim3 grid((Cc + 31) / 32, (R + 31) / 32, N);
    hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, dim3(256), 0, st, (const TI*)x, (TO*)y, R, Cc, xs_n, xs_r,
                       ys_n, ys_c);
    CP360_CHECK_HIP();
    return CP360_OK;
}

static int transpose_dispatch(const void* x, void* y, int N, int R, int Cc, size_t xs_n, size_t xs_r, size_t ys_n,
                              size_t ys_c, int in_dtype, int out_dtype, hipStream_t st) {
    if (!x || !y) return CP360_ERR_NULL;
    if (N <= 0 || R <= 0 || Cc <= 0 || N > 65535) return CP360_ERR_BAD_SHAPE;
#define CP360_TR(DI, TI, DO, TO) \
    if (in_dtype == DI && out_dtype == DO) return launch_transpose<TI, TO>(x, y, N, R, Cc, xs_n, xs_r, ys_n, ys_c, st);
    CP360_TR(CP360_F32, float, CP360_F32, float)
    CP360_TR(CP360_F32, float, CP360_BF16, bf16_raw)
    CP360_TR(CP360_F32, float, CP360_F16, f16_raw)
    CP360_TR(CP360_BF16, bf16_raw, CP360_F32, float)
    CP360_TR(CP360_BF16, bf16_raw, CP360_BF16, bf16_raw)
    CP360_TR(CP360_F16, f16_raw, CP360_F32, float)
    CP360_TR(CP360_F16, f16_raw, CP360_F16, f16_raw)
#undef CP360_TR
    return CP360_ERR_BAD_DTYPE;
}

static int elem_sz(int dtype) { return dtype == CP360_F32 ? 4 : ((dtype == CP360_BF16 || dtype == CP360_F16) ? 2 : 0); }

extern "C" int cp360_nchw_to_nhwc(const void* x, void* y, int N, int C, int H, int W, int in_dtype, int out_dtype,
                                  int ld_y, int y_coff, void* stream) {
    if (ld_y == 0) ld_y = C;
    if (ld_y < C + y_coff || y_coff < 0) return CP360_ERR_BAD_SHAPE;
    const size_t hw = (size_t)H * W;
    // x[n][c][hw] -> y[n][hw][ld_y] (+ y_coff)
    void* yb = y ? (void*)((char*)y + (size_t)y_coff * elem_sz(out_dtype)) : y;
    return transpose_dispatch(x, yb, N, C, (int)hw, (size_t)C * hw, hw, hw * ld_y, (size_t)ld_y, in_dtype, out_dtype,
                              (hipStream_t)stream);
}
extern "C" int cp360_nhwc_to_nchw(const void* x, void* y, int N, int C, int H, int W, int in_dtype, int out_dtype,
                                  int ld_x, int x_coff, void* stream) {
    if (ld_x == 0) ld_x = C;
    if (ld_x < C + x_coff || x_coff < 0) return CP360_ERR_BAD_SHAPE;
    const size_t hw = (size_t)H * W;
    // x[n][hw][ld_x] (+ x_coff) -> y[n][c][hw]
    const void* xb = x ? (const void*)((const char*)x + (size_t)x_coff * elem_sz(in_dtype)) : x;
    return transpose_dispatch(xb, y, N, (int)hw, C, hw * ld_x, (size_t)ld_x, (size_t)C * hw, hw, in_dtype, out_dtype,
                              (hipStream_t)stream);
}

// ------------------------------------------------------------------ K7: window min / max
// test_temporal.py:66-67: max / min over every value of the window's T cube_feat arrays.
// Pass 1: 256 workgroups per clip, each reduces a slice to one (min, max) pair in
// `scratch`; pass 2 (one workgroup per clip) folds the 256 pairs.  No float atomics.
// Non-finite inputs (an fp16 static stage whose activations passed 65504: inf, then inf - inf = NaN) POISON the pair:
// fminf / fmaxf drop a NaN, so pass 1 also tests every value's exponent field and a window that holds an inf or a NaN
// gets min = max = NaN - as numpy's max / min over such a window would give NaN (test_temporal.py:66-67) - which the
// normalisation spreads over the whole map and which pipeline.py reads back as its overflow flag (8 bytes per window).
__device__ __forceinline__ unsigned nonfinite_bits(const float4& v) {
    const unsigned e = 0x7f800000u;
    return (unsigned)((__float_as_uint(v.x) & e) == e) | (unsigned)((__float_as_uint(v.y) & e) == e) |
           (unsigned)((__float_as_uint(v.z) & e) == e) | (unsigned)((__float_as_uint(v.w) & e) == e);
}
__device__ __forceinline__ void wave_minmax(float& mn, float& mx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
}

__global__ __launch_bounds__(256) void minmax_pass1(const float* __restrict__ x, float* __restrict__ scratch,
                                                    size_t per_clip, size_t clip_stride) {
    const int b = blockIdx.y;
    const float4* xv = reinterpret_cast<const float4*>(x + (size_t)b * clip_stride);
    const size_t nv = per_clip / 4;
    float mn = INFINITY, mx = -INFINITY;
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        const float4 v = xv[i];
        mn = fminf(fminf(mn, v.x), fminf(fminf(v.y, v.z), v.w));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
        bad |= nonfinite_bits(v);
    }
    wave_minmax(mn, mx);
    const int any_bad = __syncthreads_or((int)bad);
    __shared__ float s[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s[wave * 2] = mn;
        s[wave * 2 + 1] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mn = fminf(mn, s[w * 2]);
            mx = fmaxf(mx, s[w * 2 + 1]);
        }
        if (any_bad) mn = mx = __uint_as_float(0x7fc00000u);
        scratch[((size_t)b * gridDim.x + blockIdx.x) * 2] = mn;
        scratch[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = mx;
    }
}

__global__ __launch_bounds__(256) void minmax_pass2(const float* __restrict__ scratch, float* __restrict__ minmax,
                                                    int nblk) {
    const int b = blockIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    int bad = 0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        const float a = scratch[((size_t)b * nblk + i) * 2], c = scratch[((size_t)b * nblk + i) * 2 + 1];
        bad |= (a != a) || (c != c);                          // a slice that pass 1 poisoned
        mn = fminf(mn, a);
        mx = fmaxf(mx, c);
    }
    wave_minmax(mn, mx);
    const int any_bad = __syncthreads_or(bad);
    __shared__ float s[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s[wave * 2] = mn;
        s[wave * 2 + 1] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mn = fminf(mn, s[w * 2]);
            mx = fmaxf(mx, s[w * 2 + 1]);
        }
        if (any_bad) mn = mx = __uint_as_float(0x7fc00000u);
        minmax[b * 2] = mn;
        minmax[b * 2 + 1] = mx;
    }
}

extern "C" int cp360_window_minmax(const float* x, float* minmax, float* scratch, int B, size_t per_clip,
                                   size_t clip_stride, void* stream) {
    if (!x || !minmax || !scratch) return CP360_ERR_NULL;
    if (B <= 0 || per_clip == 0 || B > 65535) return CP360_ERR_BAD_SHAPE;
    if (clip_stride == 0) clip_stride = per_clip;
    if (per_clip % 4 != 0 || clip_stride % 4 != 0) return CP360_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(minmax_pass1, dim3(256, B), dim3(256), 0, st, x, scratch, per_clip, clip_stride);
    hipLaunchKernelGGL(minmax_pass2, dim3(B), dim3(256), 0, st, scratch, minmax, 256);
    CP360_CHECK_HIP();
    return CP360_OK;
}

// test_temporal.py:70-73,77: (frame - mn) / (mx - mn), written where the ConvLSTM reads it.
template <typename T>
__global__ __launch_bounds__(256) void window_normalize_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ minmax, T* __restrict__ y,
                                                               int ld_y, int y_coff, float* __restrict__ y2,
                                                               size_t clip_stride, int t, int P, int C,
                                                               size_t y_tstride) {
    const int b = blockIdx.y;
    t += blockIdx.z;                                           // cp360_window_normalize_frames: one grid layer per frame
    y += (size_t)blockIdx.z * y_tstride;
    const float mn = minmax[b * 2], mx = minmax[b * 2 + 1];
    const float den = mx - mn;
    const int cq = C / 4;
    const long long total = (long long)P * cq;
    const float* src = x + (size_t)b * clip_stride + (size_t)t * P * C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int pix = (int)(idx / cq), c = (int)(idx - (long long)pix * cq) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)pix * C + c);
        const float o[4] = {(v.x - mn) / den, (v.y - mn) / den, (v.z - mn) / den, (v.w - mn) / den};
        T* dst = y + ((size_t)b * P + pix) * ld_y + y_coff + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) st_f32<T>(dst + e, o[e]);
        if (y2) *reinterpret_cast<float4*>(y2 + ((size_t)b * P + pix) * C + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int cp360_window_normalize(const float* x, const float* minmax, void* y, int y_dtype, int ld_y, int y_coff,
                                      float* y2, int B, int T, int t, int P, int C, size_t clip_stride,
                                      void* stream) {
    if (!x || !minmax || !y) return CP360_ERR_NULL;
    if (B <= 0 || T <= 0 || t < 0 || t >= T || P <= 0 || C <= 0 || B > 65535) return CP360_ERR_BAD_SHAPE;
    if (clip_stride == 0) clip_stride = (size_t)T * P * C;
    if (C % 4 != 0 || ld_y % 4 != 0 || y_coff % 4 != 0 || clip_stride % 4 != 0) return CP360_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)P * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (y_dtype == CP360_F32)
        hipLaunchKernelGGL((window_normalize_kernel<float>), dim3((unsigned)blocks, B), dim3(256), 0, st, x, minmax,
                           (float*)y, ld_y, y_coff, y2, clip_stride, t, P, C, (size_t)0);
    else if (y_dtype == CP360_BF16)
        hipLaunchKernelGGL((window_normalize_kernel<bf16_raw>), dim3((unsigned)blocks, B), dim3(256), 0, st, x, minmax,
                           (bf16_raw*)y, ld_y, y_coff, y2, clip_stride, t, P, C, (size_t)0);
    else if (y_dtype == CP360_F16)
        hipLaunchKernelGGL((window_normalize_kernel<f16_raw>), dim3((unsigned)blocks, B), dim3(256), 0, st, x, minmax,
                           (f16_raw*)y, ld_y, y_coff, y2, clip_stride, t, P, C, (size_t)0);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

// All T frames of every window at once: y [T, B, P, C] (dense, dtype y_dtype) - the input of the batched x half of the
// ConvLSTM's first convolution (csrc/ctx.hip: cp360_clstm_window).
extern "C" int cp360_window_normalize_frames(const float* x, const float* minmax, void* y, int y_dtype, int B, int T, int P,
                                             int C, size_t clip_stride, void* stream) {
    if (!x || !minmax || !y) return CP360_ERR_NULL;
    if (B <= 0 || T <= 0 || P <= 0 || C <= 0 || B > 65535 || T > 65535) return CP360_ERR_BAD_SHAPE;
    if (clip_stride == 0) clip_stride = (size_t)T * P * C;
    if (C % 4 != 0 || clip_stride % 4 != 0) return CP360_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)P * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256) blocks = 256;
    const dim3 grid((unsigned)blocks, B, T);
    const size_t ts = (size_t)B * P * C;
    if (y_dtype == CP360_F32)
        hipLaunchKernelGGL((window_normalize_kernel<float>), grid, dim3(256), 0, st, x, minmax, (float*)y, C, 0, nullptr,
                           clip_stride, 0, P, C, ts);
    else if (y_dtype == CP360_BF16)
        hipLaunchKernelGGL((window_normalize_kernel<bf16_raw>), grid, dim3(256), 0, st, x, minmax, (bf16_raw*)y, C, 0, nullptr,
                           clip_stride, 0, P, C, ts);
    else if (y_dtype == CP360_F16)
        hipLaunchKernelGGL((window_normalize_kernel<f16_raw>), grid, dim3(256), 0, st, x, minmax, (f16_raw*)y, C, 0, nullptr,
                           clip_stride, 0, P, C, ts);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

// ---------------------------------------------------------------- diagnostic: the shader clock held under load
// MI355X lowers its clock under a dense-MFMA load and boxes differ (MI355X_MICROARCH.md, "DVFS give-back" items 5-6), which
// moves the bench line by more than a round's kernel work.  This kernel is a bare bf16 MFMA loop on pseudo-random operands
// (one wave per SIMD on every CU, 16 independent accumulators) stamped once around the loop with s_memtime (shader cycles)
// and s_memrealtime (100 MHz): held clock = d(memtime) / d(memrealtime) x 0.1 GHz.  The stamps go to a buffer of their own
// that nothing else reads; no output of the library is computed from them.  bench.py reports the median over workgroups as
// `held_clock_ghz`, so two driver runs on different boxes can be told apart.
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ stamps, int iters) {
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const unsigned tid = threadIdx.x + blockIdx.x * 256u;
    u32x4_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned h = (tid * 2654435761u) ^ ((unsigned)(i * 4 + e + 1) * 0x9E3779B9u);
            h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
            // two bf16 values in [1, 2) with random signs and mantissas: finite sums, random toggling of the multipliers
            a[i][e] = 0x3F803F80u | (h & 0x807F807Fu);
            b[i][e] = 0x3F803F80u | ((h * 0xC2B2AE35u) & 0x807F807Fu);
        }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]),
                                                                    acc[i][j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    const int wave = tid >> 6;
    if ((threadIdx.x & 63) == 0 || s == 1.2345e-30f) {          // (the second term keeps the loop alive; it never holds)
        stamps[wave * 2] = t1 - t0;
        stamps[wave * 2 + 1] = r1 - r0;
    }
}

extern "C" int cp360_clock_probe(unsigned long long* stamps, int n_workgroups, int iters, void* stream) {
    if (!stamps) return CP360_ERR_NULL;
    if (n_workgroups <= 0 || iters <= 0) return CP360_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(clock_probe_kernel, dim3((unsigned)n_workgroups), dim3(256), 0, (hipStream_t)stream, stamps, iters);
    CP360_CHECK_HIP();
    return CP360_OK;
}
