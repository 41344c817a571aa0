// K3c: CubePad(1) + 3x3 convolution 64 -> 64 (+ folded BN, ReLU) on 56x56 cube faces - layer1's conv2 of
// model/resnet_cubic.py:85-106 at cube size 224 - as a resident-band MFMA kernel for 16-bit types.
//
// The generic implicit GEMM gathers a 128-byte im2col row per output pixel and TAP: nine times the
// activation tensor through the L2 -> LDS path, which bounds it (0.25 ms per conv, 330 TFLOP/s;
// DESIGN.md).  Here a workgroup owns a band of 4 output rows of one face:
//   * the 6 x 58 cube-padded pixels the band needs (348 x 128 B = 44.5 KB) are gathered ONCE by
//     LDS-DMA, each lane fetching through cubepad_src() (the halo comes from the neighbouring faces);
//   * the MFMA B fragment of (output pixel x, tap (ky, kx), k-chunk) is 16 bytes of patch pixel
//     (row + ky) * 58 + x + kx: the nine taps read the same resident pixels, no im2col copy;
//   * the 8 KiB weight slice of a tap streams through a three-slot LDS ring (L2-resident, 74 KB total).
// Patch swizzle: chunk c of pixel p sits at chunk c ^ (2 * ((p >> 1) & 3)); with 128-byte pixels this
// keeps every 16-lane ds_read_b128 group (16 consecutive pixels starting ANYWHERE, two chunk values)
// on 16 distinct 16-byte bank slots (exhaustive search, as for the clip kernel).
// Wave w computes output row w of the band: 56 pixels as 4 MFMA pixel blocks (8 of the 64 columns are
// padding) x 64 channels.  Two workgroups (256 threads, 68 KiB LDS) share a CU, so one band's gather /
// epilogue overlaps the other's MFMAs.  Epilogue: bias + ReLU, one rounding, 16-byte pieces (acc_chan).
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {
constexpr int N = 56, NP = N + 2, C = 64, BAND = 4;
constexpr int PATCH_PX = (BAND + 2) * NP;                    // 348
constexpr int PATCH_INST = (PATCH_PX + 7) / 8;               // 44 DMA instructions of 8 pixels x 128 B
constexpr int PATCH_LDS = PATCH_INST * 1024;                 // 45,056
constexpr int W_TAP = 64 * 128;                              // 8 KiB: one tap's 64 rows x 64 channels
constexpr int W_SLOTS = 3;

__device__ __attribute__((aligned(16))) unsigned int b_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
__device__ __forceinline__ int px_swz(int p) { return ((p >> 1) & 3) << 1; }
__device__ __forceinline__ int w_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T> __device__ __forceinline__ void mma(f32x4& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma<bf16_raw>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<f16_raw>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
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
}  // namespace

// packed weights: [tap][row r][64 channels], row r <- output channel in acc_chan order, BN scale folded
template <typename T>
__global__ __launch_bounds__(256) void band3x3_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                           T* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 9 * 64 * 64) return;
    const int c = idx & 63, r = (idx >> 6) & 63, tap = idx >> 12;
    const int n = (r & ~31) + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);
    const float v = w[(n * 64 + c) * 9 + tap] * (scale ? scale[n] : 1.f);
    if constexpr (__is_same(T, f16_raw)) packed[idx] = (f16_raw)v;
    else packed[idx] = f32_to_bf16(v);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void band3x3_kernel(const T* __restrict__ x, const T* __restrict__ wpk,
                                                         const float* __restrict__ bias, T* __restrict__ out, int relu) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[PATCH_LDS + W_SLOTS * W_TAP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)lds;
    const int tile = blockIdx.x, img = tile / (N / BAND), band = tile - img * (N / BAND);
    const int grp = img / 6, f = img - grp * 6;
    const CubePadGeom geom{N, 1, 1, 1, 1};

    // ---- gather the band's padded pixels: instruction i = patch pixels 8i .. 8i+7, 128 B each
    {
        const T* xg = x + (size_t)grp * 6 * N * N * C;
#pragma unroll 1
        for (int inst = wave; inst < PATCH_INST; inst += 4) {
            const int q = inst * 8 + (lane >> 3);
            const void* src = b_zero16;
            if (q < PATCH_PX) {
                const int pr = q / NP, pc = q - pr * NP;
                const int sp = cubepad_src(f, BAND * band + pr, pc, geom);       // pixel index inside the cube
                src = xg + (size_t)sp * C + (((lane & 7) ^ px_swz(q)) << 3);
            }
            glds16(src, __builtin_amdgcn_readfirstlane(lds_base + inst * 1024));
        }
    }
    // weight slice of tap t -> ring slot s: 8 instructions of 8 rows x 128 B, two per wave
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk);
    auto load_w = [&](int t, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int inst = wave * 2 + q;
            const int row = inst * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            glds16(wb + (size_t)t * W_TAP + row * 128 + chunk * 16,
                   __builtin_amdgcn_readfirstlane(lds_base + PATCH_LDS + s * W_TAP + inst * 1024));
        }
    };
    load_w(0, 0);
    load_w(1, 1);

    const int lrow = lane & 15, lchunk = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        // weights of `tap` landed (the patch DMAs are older); the next tap's two DMAs may still fly
        if (tap < 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // ... for every wave; slot (tap+2)%3 is free
        if (tap + 2 < 9) load_w(tap + 2, (tap + 2) % W_SLOTS);
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned char* Ws = lds + PATCH_LDS + (tap % W_SLOTS) * W_TAP;
        const int pbase = (wave + ky) * NP + kx + lrow;    // patch pixel of this lane in pixel block 0
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const u32x4*>(Ws + w_off(i * 16 + lrow, kk * 4 + lchunk));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = pbase + 16 * j;
                b[j] = *reinterpret_cast<const u32x4*>(lds + p * 128 + (((kk * 4 + lchunk) ^ px_swz(p)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma<T>(acc[i][j], a[i], b[j]);
        }
    }
    // ---- epilogue: output row band*4 + wave of image img
    T* orow = out + ((size_t)img * N + band * BAND + wave) * N * C;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int n = pr * 32 + (lane >> 4) * 8;
        float bb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[e] = bias ? bias[n + e] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xo = j * 16 + lrow;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[2 * pr][j][e] + bb[e];
                v[4 + e] = acc[2 * pr + 1][j][e] + bb[4 + e];
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (xo < N) *reinterpret_cast<u32x4*>(orow + (size_t)xo * C + n) = pack8(v, T());
        }
    }
}

extern "C" size_t cp360_band3x3_packed_bytes(int dtype) {
    return (dtype == CP360_BF16 || dtype == CP360_F16) ? (size_t)9 * W_TAP : 0;
}

extern "C" int cp360_band3x3_pack_weights(int dtype, const float* w_oihw, const float* scale, void* packed, void* stream) {
    if (!w_oihw || !packed) return CP360_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CP360_BF16)
        hipLaunchKernelGGL((band3x3_pack_kernel<bf16_raw>), dim3(144), dim3(256), 0, st, w_oihw, scale, (bf16_raw*)packed);
    else if (dtype == CP360_F16)
        hipLaunchKernelGGL((band3x3_pack_kernel<f16_raw>), dim3(144), dim3(256), 0, st, w_oihw, scale, (f16_raw*)packed);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_band3x3_forward(int dtype, const void* x, const void* packed, const float* bias, void* out,
                                     int n_img, int face, int channels, int relu, void* stream) {
    if (!x || !packed || !out) return CP360_ERR_NULL;
    if (n_img <= 0) return CP360_ERR_BAD_SHAPE;
    if (n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (face != N || channels != C) return CP360_ERR_UNSUPPORTED;      // other shapes: the generic implicit GEMM
    if ((long long)n_img * N * N * C >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(n_img * (N / BAND)));
    if (dtype == CP360_BF16)
        hipLaunchKernelGGL((band3x3_kernel<bf16_raw>), grid, dim3(256), 0, st, (const bf16_raw*)x, (const bf16_raw*)packed, bias,
                           (bf16_raw*)out, relu);
    else if (dtype == CP360_F16)
        hipLaunchKernelGGL((band3x3_kernel<f16_raw>), grid, dim3(256), 0, st, (const f16_raw*)x, (const f16_raw*)packed, bias,
                           (f16_raw*)out, relu);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}
