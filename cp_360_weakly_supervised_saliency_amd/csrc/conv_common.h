// Definitions shared by the convolution kernels of conv_igemm.hip and conv_small.hip: the kernel argument block, the
// MFMA wrappers, the LDS swizzle, the packed-row channel order and the 16-byte pack / unpack helpers.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: stays in VGPRs (HIP's uint4 struct
                                                                   // kept one staging set in scratch memory)

struct ConvK {
    const unsigned char* in;
    const unsigned char* w;
    unsigned char* out;
    const float* bias;
    const unsigned char* res;
    float* partial;
    int n_img, h_in, w_in, c_in, pix_stride, kh, kw, sy, sx, h_out, w_out, c_out;
    int pad_mode, pad, ld_out, out_coff, ld_res, relu, splits;
    int M, c_pad, steps_per_tap, nsteps, steps_per_split, k_total, hw_out;
    int nt, mt, m_fast;
    int clip_rows, nsub, sub_per_split;   // clip-resident kernel: pixels per clip, 64-byte sub-steps in all / per split
    int w_pin;                            // clip-resident kernel: sub-steps at the head of a workgroup's weight stream loaded with the
                                          // default cache policy, the rest non-temporal (conv_igemm.hip, clip_body)
    int reverse;                          // 1: work items in descending order (cp360_set_launch_order)
    int epi_direct;                       // 1: direct 16-byte epilogue, 0: LDS-staged epilogue
    int slab_rows;                        // 1: split-K slabs in packed-row column order (slab_col)
    // second source (cp360_conv_desc.c_in2 > 0): one extra 1x1 "tap" (index kh*kw, packed behind the others)
    // gathered from in2 [n_img, h_in2, w_in2, pix_stride2] at (oy * sy2, ox * sx2) - the Bottleneck's downsample
    // branch accumulated into the conv3 tile (ring kernels only)
    const unsigned char* in2;
    int c_in2, c_pad2, pix_stride2, h_in2, w_in2, sy2, sx2, ntap;
};

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPC = 4; };      // elements per 16-byte chunk
template <> struct Elem<bf16_raw> { static constexpr int EPC = 8; };
template <> struct Elem<f16_raw> { static constexpr int EPC = 8; };

// 16 zero bytes in device memory: invalid tile rows (m >= M) and the K tail (c >= c_in)
// load from here, so the select happens on the ADDRESS before the load and nothing has
// to wait for the loaded data until the ds_write that consumes it.
static __device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// Channel order inside a 32-row group of the packed weights.  MFMA row block i gives a lane the
// four consecutive rows 4*(lane>>4) .. +3; the pack kernel permutes the rows so that blocks 2p
// and 2p+1 TOGETHER give it EIGHT consecutive channels:  packed row 32q + 16*b + 4*g + e  holds
// channel 32q + 8*g + 4*b + e.  A lane then owns a 16-byte (bf16) / 32-byte (f32) piece of a pixel
// and the four lane groups of a pixel 64 / 128 contiguous bytes: outputs, residuals and split-K
// slabs are moved with 16-byte accesses straight from / to global memory.
__device__ __forceinline__ int acc_chan(int i, int lane) { return (i >> 1) * 32 + (lane >> 4) * 8 + (i & 1) * 4; }
// Split-K slabs (cp360_conv_desc.slab_rows) keep the PACKED row order inside each 32-channel group, so the
// four lane groups of a pixel store 64 contiguous bytes per MFMA block (in true channel order a store
// instruction would write 16-byte pieces 32 bytes apart: +37 % HBM write traffic measured).  Column of the
// 4-channel group that starts at channel n (n % 4 == 0):
__host__ __device__ __forceinline__ int slab_col(int n) { return (n & ~31) + ((n >> 3) & 3) * 4 + ((n >> 2) & 1) * 16; }

__device__ __forceinline__ int lds_swz(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename T>
__device__ __forceinline__ void mma_chunk(f32x4& acc, const u32x4& a, const u32x4& b);

template <>
__device__ __forceinline__ void mma_chunk<float>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<bf16_raw>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                  0, 0, 0);
}

template <>
__device__ __forceinline__ void mma_chunk<f16_raw>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0,
                                                 0, 0);
}

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<bf16_raw>(const bf16_raw* p) { return bf16_to_f32(*p); }

// store 4 consecutive channels
__device__ __forceinline__ void store4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_raw* p, const float v[4]) {
    uint2 o;
    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = o;
}
__device__ __forceinline__ void store4(f16_raw* p, const float v[4]) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    const f16x4 o = {(f16_raw)v[0], (f16_raw)v[1], (f16_raw)v[2], (f16_raw)v[3]};
    *reinterpret_cast<f16x4*>(p) = o;
}
__device__ __forceinline__ void load4(const f16_raw* p, float v[4]) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    const f16x4 t = *reinterpret_cast<const f16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
__device__ __forceinline__ void load4(const float* p, float v[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_raw* p, float v[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

// ------------------------------------------------------------------ direct epilogue (16-byte pieces)
// With the acc_chan row order a lane owns 8 consecutive channels of a pixel per block pair: bias,
// residual, ReLU, ONE rounding and the store happen on 16-byte pieces straight against global memory
// (a pixel's four lane groups cover 64 bytes (16-bit types) / 128 bytes (f32) contiguously) - no LDS
// round trip, no barriers, and the residual loads of JB pixel blocks are in flight together.
__device__ __forceinline__ u32x4 pack8(const float v[8], bf16_raw) {
    u32x4 o;
    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
    o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
    return o;
}
__device__ __forceinline__ u32x4 pack8(const float v[8], f16_raw) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8v;
    const f16x8v h = {(f16_raw)v[0], (f16_raw)v[1], (f16_raw)v[2], (f16_raw)v[3],
                      (f16_raw)v[4], (f16_raw)v[5], (f16_raw)v[6], (f16_raw)v[7]};
    return __builtin_bit_cast(u32x4, h);
}
__device__ __forceinline__ void unpack8(const u32x4& r, float v[8], bf16_raw) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack8(const u32x4& r, float v[8], f16_raw) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8v;
    const f16x8v h = __builtin_bit_cast(f16x8v, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
}

// conv_small.hip keeps a [tap][tile row] source-offset table in LDS: kh * kw (+ 1 for a second source) must fit
#define CP360_SMALL_MAX_TAPS 16

// conv_small.hip: the 64 x 64-tile kernel for launches whose pixel count cannot fill the chip with the big tiles
// (k.nt / k.mt / k.m_fast are set inside).  dtype: CP360_F32 / CP360_BF16 / CP360_F16.
void cp360_launch_conv_small(ConvK& k, int dtype, hipStream_t st);
