// K1 equi -> cube (+ /255, ImageNet normalise, layout, dtype) and K6 cube -> equi
// (+ channel max).  Both are gathers bounded by HBM bandwidth.
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------ K1
// utils/equi_to_cube.py:112-129 = cv2.remap(img[:, :, c], inX, inY, INTER_LINEAR) per
// face and channel.  OpenCV's INTER_LINEAR on float maps (imgwarp.cpp: INTER_BITS = 5):
//   sx = cvRound(x * 32) (round half to even), ix = sx >> 5, fx = (sx & 31) / 32
//   out = S[iy][ix]*(1-fx)(1-fy) + S[iy][ix+1]*fx(1-fy) + S[iy+1][ix]*(1-fx)fy + S[iy+1][ix+1]*fx*fy
// with BORDER_CONSTANT 0 for taps outside the image.  One thread per output pixel does
// the 3 channels (the 12 source bytes / floats of a tap pair are adjacent in HWC), then
// applies dataset_feat_extractor.py:142 (/255 -> `scale`), utils/utils.py:28-33 (mean /
// std) and writes NCHW planes (the reference batch, class_activation_model.py:55) or
// NHWC4 pixels (one 16-byte store) for the fused pipeline.
template <typename TI> __device__ __forceinline__ float px_load(const TI* p);
template <> __device__ __forceinline__ float px_load<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float px_load<uint8_t>(const uint8_t* p) { return (float)*p; }

template <typename TO> __device__ __forceinline__ TO out_cvt(float v);
template <> __device__ __forceinline__ float out_cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_raw out_cvt<bf16_raw>(float v) { return f32_to_bf16(v); }
template <> __device__ __forceinline__ f16_raw out_cvt<f16_raw>(float v) { return (f16_raw)v; }

// A thread owns one output pixel g of FU consecutive frames: the grid point, the fixed-point split and
// the bilinear weights are computed once, and the 2 x FU row loads of the frames are in flight together
// (the one-pixel-one-frame form spent its time in two dependent round trips per output: grid, then taps).
template <typename TI, typename TO, int LAYOUT, bool FIXED, int FU>
__global__ __launch_bounds__(256) void equi2cube_kernel(const TI* __restrict__ equi, const float2* __restrict__ grid,
                                                        TO* __restrict__ out, int F, int H, int W, int cd,
                                                        float m0, float m1, float m2, float s0, float s1, float s2,
                                                        float scale) {
    const long long per_frame = 6LL * cd * cd;
    const int nfc = (F + FU - 1) / FU;
    const long long total = per_frame * nfc;
    const size_t frame_elems = (size_t)H * W * 3;
    // XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (private L2s), so consecutive
    // blocks - neighbouring output rows that sample the SAME input lines - would sit on 8 different L2s and every
    // input line would cross the fabric up to 8 times (rocprofv3: 1.87 GB per launch against 0.41 GB algorithmic,
    // profiles/r02a_bf16.md).  Each XCD takes one contiguous eighth of the (frame group, pixel) index space instead
    // (= whole frame groups at the benchmark shape).  gridDim.x is a multiple of 8.
    const long long chunk = (total + 7) / 8;
    const int xcd = blockIdx.x & 7;
    const long long lo = xcd * chunk, hi = min(total, lo + chunk);
    const long long stride = (long long)(gridDim.x >> 3) * blockDim.x;
    for (long long wi = lo + (long long)(blockIdx.x >> 3) * blockDim.x + threadIdx.x; wi < hi; wi += stride) {
        const int fc = (int)(wi / per_frame);
        const int g = (int)(wi - (long long)fc * per_frame);          // face*cd*cd + i*cd + j
        const int fr0 = fc * FU;
        const float2 xy = grid[g];
        int ix, iy;
        float fx, fy;
        if (FIXED) {
            const int sx = __float2int_rn(xy.x * 32.f), sy = __float2int_rn(xy.y * 32.f);
            ix = sx >> 5; iy = sy >> 5;
            fx = (float)(sx & 31) * (1.f / 32.f);
            fy = (float)(sy & 31) * (1.f / 32.f);
        } else {
            const float flx = floorf(xy.x), fly = floorf(xy.y);
            ix = (int)flx; iy = (int)fly;
            fx = xy.x - flx; fy = xy.y - fly;
        }
        const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
        const bool x0ok = ix >= 0 && ix < W, x1ok = ix + 1 >= 0 && ix + 1 < W;
        const bool y0ok = iy >= 0 && iy < H, y1ok = iy + 1 >= 0 && iy + 1 < H;
        auto emit = [&](int fr, float v[3]) {
            v[0] = (v[0] - m0) * s0;
            v[1] = (v[1] - m1) * s1;
            v[2] = (v[2] - m2) * s2;
            if (LAYOUT == 0) {   // [6F, 3, cd, cd]
                const int face = g / (cd * cd), pix = g - face * cd * cd;
                const size_t o = ((size_t)(fr * 6 + face) * 3) * cd * cd + pix;
                out[o] = out_cvt<TO>(v[0]);
                out[o + (size_t)cd * cd] = out_cvt<TO>(v[1]);
                out[o + 2 * (size_t)cd * cd] = out_cvt<TO>(v[2]);
            } else {             // [6F, cd, cd, 4]
                const size_t o = ((size_t)fr * per_frame + g) * 4;
                if constexpr (sizeof(TO) == 4) {
                    *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], 0.f);
                } else {
                    typedef __attribute__((ext_vector_type(4))) TO vec4;
                    const vec4 ov = {out_cvt<TO>(v[0]), out_cvt<TO>(v[1]), out_cvt<TO>(v[2]), out_cvt<TO>(0.f)};
                    *reinterpret_cast<vec4*>(out + o) = ov;
                }
            }
        };
        if (sizeof(TI) == 1 && ix >= 0 && iy >= 0 && ix + 4 < W && iy + 1 < H) {
            // u8 fast path: the two taps of a row are 6 adjacent bytes (HWC).  Instead of six byte
            // loads per row: one 12-byte load from the enclosing 4-byte-aligned address and a funnel
            // shift (ix + 4 < W keeps the 12 bytes inside the row).
            const size_t a0 = ((size_t)iy * W + ix) * 3;
            uint3 d0[FU], d1[FU];
            unsigned sh0[FU], sh1[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int fr = min(fr0 + u, F - 1);                 // frames past F re-read the last one (not stored)
                const size_t ad0 = (size_t)reinterpret_cast<const unsigned char*>(equi) + (size_t)fr * frame_elems + a0;
                const size_t ad1 = ad0 + (size_t)W * 3;
                d0[u] = *reinterpret_cast<const uint3*>(ad0 & ~(size_t)3);
                d1[u] = *reinterpret_cast<const uint3*>(ad1 & ~(size_t)3);
                sh0[u] = (unsigned)(ad0 & 3) * 8;
                sh1[u] = (unsigned)(ad1 & 3) * 8;
            }
            // the 6 bytes of a row's two taps start sh / 8 bytes into the 12 loaded ones: two v_alignbyte_b32 bring them
            // to dwords (bytes 0-3, bytes 4-7), six v_cvt_f32_ubyteN turn them into floats (no 64-bit shifts)
            auto taps = [](const uint3& d, unsigned sh, float lo3[3], float hi3[3]) {
                const unsigned sb = sh >> 3;
                const unsigned wa = __builtin_amdgcn_alignbyte(d.y, d.x, sb), wb = __builtin_amdgcn_alignbyte(d.z, d.y, sb);
                lo3[0] = (float)(wa & 0xffu); lo3[1] = (float)((wa >> 8) & 0xffu); lo3[2] = (float)((wa >> 16) & 0xffu);
                hi3[0] = (float)(wa >> 24); hi3[1] = (float)(wb & 0xffu); hi3[2] = (float)((wb >> 8) & 0xffu);
            };
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                if (fr0 + u >= F) break;
                float t00[3], t01[3], t10[3], t11[3];
                taps(d0[u], sh0[u], t00, t01);
                taps(d1[u], sh1[u], t10, t11);
                float v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = (t00[c] * w00 + t01[c] * w01 + t10[c] * w10 + t11[c] * w11) * scale;
                emit(fr0 + u, v);
            }
        } else {
            for (int u = 0; u < FU && fr0 + u < F; ++u) {
                const TI* base = equi + (size_t)(fr0 + u) * frame_elems;
                float v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float t00 = (x0ok && y0ok) ? px_load<TI>(base + ((size_t)iy * W + ix) * 3 + c) : 0.f;
                    const float t01 = (x1ok && y0ok) ? px_load<TI>(base + ((size_t)iy * W + ix + 1) * 3 + c) : 0.f;
                    const float t10 = (x0ok && y1ok) ? px_load<TI>(base + ((size_t)(iy + 1) * W + ix) * 3 + c) : 0.f;
                    const float t11 = (x1ok && y1ok) ? px_load<TI>(base + ((size_t)(iy + 1) * W + ix + 1) * 3 + c) : 0.f;
                    v[c] = (t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11) * scale;
                }
                emit(fr0 + u, v);
            }
        }
    }
}

template <typename TI, typename TO>
static int launch_e2c(const void* equi, const float* grid, void* out, int F, int H, int W, int cd, const float* mean,
                      const float* istd, float scale, int layout, int fixed, hipStream_t st) {
    static const int fu_env = []() { const char* e = getenv("CP360_E2C_FU"); return e ? atoi(e) : 0; }();
    static const int cap_env = []() { const char* e = getenv("CP360_E2C_CAP"); const int v = e ? atoi(e) : 64; return v < 1 ? 1 : v; }();
    const int fu = fu_env == 8 ? 8 : (fu_env == 2 ? 2 : 4);
    const long long total = (long long)((F + fu - 1) / fu) * 6 * cd * cd;      // one thread per pixel and group of FU frames
    long long blocks = (total + 255) / 256;
    if (blocks > 256LL * cap_env) blocks = 256LL * cap_env;
    blocks = (blocks + 7) / 8 * 8;                                  // XCD mapping in the kernel
#define E2C_LAUNCH(L, FX, FUV)                                                                                  \
    hipLaunchKernelGGL((equi2cube_kernel<TI, TO, L, FX, FUV>), dim3((unsigned)blocks), dim3(256), 0, st,        \
                       (const TI*)equi, (const float2*)grid, (TO*)out, F, H, W, cd, mean[0], mean[1], mean[2],  \
                       istd[0], istd[1], istd[2], scale)
#define E2C_FU(L, FX) { if (fu == 8) E2C_LAUNCH(L, FX, 8); else if (fu == 2) E2C_LAUNCH(L, FX, 2); else E2C_LAUNCH(L, FX, 4); }
    if (layout == 0) { if (fixed) E2C_FU(0, true) else E2C_FU(0, false) }
    else             { if (fixed) E2C_FU(1, true) else E2C_FU(1, false) }
#undef E2C_FU
#undef E2C_LAUNCH
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_equi2cube(const void* equi, const float* grid, void* out, int F, int H, int W, int cd,
                               const float* mean3_host, const float* istd3_host, float scale, int in_dtype,
                               int out_dtype, int out_layout, int cv_fixed_point, void* stream) {
    if (!equi || !grid || !out || !mean3_host || !istd3_host) return CP360_ERR_NULL;
    if (F <= 0 || H <= 0 || W <= 0 || cd <= 0 || (out_layout != 0 && out_layout != 1)) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == CP360_U8 && out_dtype == CP360_F32)
        return launch_e2c<uint8_t, float>(equi, grid, out, F, H, W, cd, mean3_host, istd3_host, scale, out_layout,
                                          cv_fixed_point, st);
    if (in_dtype == CP360_U8 && out_dtype == CP360_BF16)
        return launch_e2c<uint8_t, bf16_raw>(equi, grid, out, F, H, W, cd, mean3_host, istd3_host, scale, out_layout,
                                             cv_fixed_point, st);
    if (in_dtype == CP360_F32 && out_dtype == CP360_F32)
        return launch_e2c<float, float>(equi, grid, out, F, H, W, cd, mean3_host, istd3_host, scale, out_layout,
                                        cv_fixed_point, st);
    if (in_dtype == CP360_F32 && out_dtype == CP360_BF16)
        return launch_e2c<float, bf16_raw>(equi, grid, out, F, H, W, cd, mean3_host, istd3_host, scale, out_layout,
                                           cv_fixed_point, st);
    if (in_dtype == CP360_U8 && out_dtype == CP360_F16)
        return launch_e2c<uint8_t, f16_raw>(equi, grid, out, F, H, W, cd, mean3_host, istd3_host, scale, out_layout,
                                            cv_fixed_point, st);
    if (in_dtype == CP360_F32 && out_dtype == CP360_F16)
        return launch_e2c<float, f16_raw>(equi, grid, out, F, H, W, cd, mean3_host, istd3_host, scale, out_layout,
                                          cv_fixed_point, st);
    return CP360_ERR_BAD_DTYPE;
}

// ------------------------------------------------------------------ K6
// utils/cube_to_equi.py:37-66 + test_temporal.py:83-84.  One wave per output pixel:
// the pixel's face and the 4 bilinear taps are wave-uniform, lanes stride over the
// channels.  NHWC input (layout 1): every tap is one contiguous channel vector ->
// coalesced; NCHW input (layout 0, the reference's tensor): lanes read with stride w*w.
// The channel max is a wave shuffle reduction: the 1000 x 2w x 4w full map never has
// to be written when only the saliency map is wanted.
template <int LAYOUT>
__global__ __launch_bounds__(256) void cube2equi_kernel(const float* __restrict__ x,
                                                        const int8_t* __restrict__ face_map,
                                                        const float2* __restrict__ coord,
                                                        float* __restrict__ out_full, float* __restrict__ out_max,
                                                        int B, int C, int w) {
    const int npix = 8 * w * w;                       // 2w * 4w
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave_global >= B * npix) return;
    const int b = wave_global / npix, pix = wave_global - b * npix;
    const int f = face_map[pix];
    const float2 pc = coord[pix];
    const float flx = floorf(pc.x), fly = floorf(pc.y);
    const int x0 = (int)flx, y0 = (int)fly;
    const float fx = pc.x - flx, fy = pc.y - fly;
    const float wt[4] = {(1.f - fx) * (1.f - fy), fx * (1.f - fy), (1.f - fx) * fy, fx * fy};
    const int xs[4] = {x0, x0 + 1, x0, x0 + 1}, ys[4] = {y0, y0, y0 + 1, y0 + 1};
    bool ok[4];
    size_t off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ok[k] = xs[k] >= 0 && xs[k] < w && ys[k] >= 0 && ys[k] < w;
        const int yy = ok[k] ? ys[k] : 0, xx = ok[k] ? xs[k] : 0;
        if (LAYOUT == 1) off[k] = (((size_t)(b * 6 + f) * w + yy) * w + xx) * C;              // + c
        else             off[k] = ((size_t)(b * 6 + f) * C) * w * w + (size_t)yy * w + xx;    // + c*w*w
    }
    const size_t cstride = LAYOUT == 1 ? 1 : (size_t)w * w;
    float best = -INFINITY;
    for (int c = lane; c < C; c += 64) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = ok[k] ? x[off[k] + (size_t)c * cstride] : 0.f;
            acc += v * wt[k];
        }
        if (out_full) out_full[((size_t)b * C + c) * npix + pix] = acc;
        best = fmaxf(best, acc);
    }
    if (out_max) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
        if (lane == 0) out_max[(size_t)b * npix + pix] = best;
    }
}

extern "C" int cp360_cube2equi(const float* x, const int8_t* face_map, const float* coord, float* out_full,
                               float* out_max, int B, int C, int w, int layout, void* stream) {
    if (!x || !face_map || !coord || (!out_full && !out_max)) return CP360_ERR_NULL;
    if (B <= 0 || C <= 0 || w <= 0 || (layout != 0 && layout != 1)) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int waves = B * 8 * w * w;
    const int blocks = (waves + 3) / 4;
    if (layout == 1)
        hipLaunchKernelGGL((cube2equi_kernel<1>), dim3(blocks), dim3(256), 0, st, x, face_map,
                           (const float2*)coord, out_full, out_max, B, C, w);
    else
        hipLaunchKernelGGL((cube2equi_kernel<0>), dim3(blocks), dim3(256), 0, st, x, face_map,
                           (const float2*)coord, out_full, out_max, B, C, w);
    CP360_CHECK_HIP();
    return CP360_OK;
}
