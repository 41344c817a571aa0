// K3a: the ResNet stem, CubePad(3) -> conv 7x7 stride 2 (3 -> 64) + BN + ReLU
// (model/resnet_cubic.py:115-128,163-168), as a resident-patch MFMA kernel for 16-bit types at cube sizes 224 (the
// reference's: padded faces 230x230, output 112x112) and 512 (BASELINE config C5: 518x518 -> 256x256).
//
// The generic implicit GEMM treats the 7x7 filter as 7 taps of 8 pixels x 4 channels and re-gathers a
// 64-byte im2col row per output pixel and tap: 4.2 GB through the L2 -> LDS path for 64 frames, which
// is what bounds it (0.63 ms).  Here a workgroup owns a band of 8 output rows of one face:
//   * the 21 padded input rows the band needs are ONE contiguous 38,640-byte range of the NHWC4 image
//     and are copied to LDS once by LDS-DMA (double-buffered: the next band loads while this one runs);
//   * the MFMA B fragment of (output pixel ox, filter row ky, k-chunk c) is the 16 bytes at
//     row (2*oy + ky), byte 16*(ox + c) of that patch - pixels 2ox+2c, 2ox+2c+1 x 4 channels - so the
//     fragments are read straight from the raw patch: no im2col copy exists anywhere;
//   * the 64 x (7 x 32) weights (28 KiB) stay in LDS for the life of the workgroup (persistent grid).
// Wave w computes output row w of the band: 112 pixels x 64 channels = 7 x 4 MFMA tiles, 7 k-steps.
// Epilogue: bias + ReLU, one rounding, 16-byte pieces straight to global (acc_chan row order: a lane
// holds 8 consecutive channels), 128 contiguous bytes per pixel.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {
// Geometry per cube size.  BAND output rows per workgroup, WPR waves per output row (8 waves in all): cube 224 ->
// 8 rows x 112 pixels (7 pixel blocks per wave); cube 512 (BASELINE config C5) -> 4 rows x 256 pixels, two waves per
// row (8 pixel blocks each: the accumulators of a whole 256-pixel row would not fit a wave's registers).
template <int CD_> struct StemGeom {
    static constexpr int CD = CD_, WP = CD + 6, WO = CD / 2;      // padded input width, output width
    static constexpr int ROW_BYTES = WP * 8;                      // one padded input row (4 x 16-bit per pixel)
    static constexpr int WPR = CD == 224 ? 1 : 2;
    static constexpr int BAND = 8 / WPR;
    static constexpr int PATCH_ROWS = 2 * (BAND - 1) + 7;         // 21 (13) input rows per band
    static constexpr int PATCH_BYTES = PATCH_ROWS * ROW_BYTES;    // 38,640 (53,872)
    static constexpr int PATCH_INST = (PATCH_BYTES + 1023) / 1024;
    static constexpr int PATCH_LDS = PATCH_INST * 1024;           // DMA granule: 1 KiB
    static constexpr int MJ = WO / 16 / WPR;                      // 7 (8) pixel blocks per wave
    static_assert(WO % (16 * WPR) == 0 && WO % BAND == 0, "cube size");
};
constexpr int W_BYTES = 7 * 64 * 64;                         // [ky][64 rows][32 k] 16-bit

__device__ __attribute__((aligned(16))) unsigned int s_zero16[4] = {0u, 0u, 0u, 0u};

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
__device__ __forceinline__ int w_swz(int row, int chunk) {   // 64-byte weight rows: as the ring kernel's stages
    return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4);
}
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

// packed stem weights: [ky][row r][k] with row r <- channel acc_chan order (row 32q + 16b + 4g + e holds
// channel 32q + 8g + 4b + e), k = kx*4 + ch (kx < 7, ch < 3; the rest zero), BN scale folded in f32
template <typename T>
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                        T* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 7 * 64 * 32) return;
    const int k = idx & 31, r = (idx >> 5) & 63, ky = idx >> 11;
    const int n = (r & ~31) + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);
    const int kx = k >> 2, ch = k & 3;
    float v = 0.f;
    if (kx < 7 && ch < 3) v = w[((n * 3 + ch) * 7 + ky) * 7 + kx] * (scale ? scale[n] : 1.f);
    if constexpr (__is_same(T, f16_raw)) packed[idx] = (f16_raw)v;
    else packed[idx] = f32_to_bf16(v);
}

template <typename T, int CDV>
__global__ __launch_bounds__(512, 2) void stem_kernel(const T* __restrict__ xp, const T* __restrict__ wpk,
                                                      const float* __restrict__ bias, T* __restrict__ out, int n_img,
                                                      int relu, int reverse) {
    typedef StemGeom<CDV> G;
    constexpr int WP = G::WP, WO = G::WO, ROW_BYTES = G::ROW_BYTES, BAND = G::BAND, PATCH_BYTES = G::PATCH_BYTES,
                  PATCH_LDS = G::PATCH_LDS, PATCH_INST = G::PATCH_INST, MJ = G::MJ, WPR = G::WPR;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[W_BYTES + 2 * PATCH_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)lds;
    const int ntiles = n_img * (WO / BAND);
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(xp);
    const size_t img_bytes = (size_t)WP * ROW_BYTES;

    // patch of tile t -> buffer b: PATCH_INST DMA instructions of 1 KiB, wave w issues w, w+8, ...; the bytes
    // past the patch's end come from a zero line (never read by the MFMAs)
    auto load_patch = [&](int t, int b) __attribute__((always_inline)) {
        if (reverse) t = ntiles - 1 - t;                       // descending tile order (cp360_set_launch_order)
        const int img = t / (WO / BAND), band = t - img * (WO / BAND);
        const unsigned char* src0 = xb + (size_t)img * img_bytes + (size_t)(2 * BAND * band) * ROW_BYTES;
#pragma unroll
        for (int q = 0; q < (PATCH_INST + 7) / 8; ++q) {
            const int inst = wave + 8 * q;
            if (inst < PATCH_INST) {
                const int off = inst * 1024 + lane * 16;
                const void* src = off < PATCH_BYTES ? (const void*)(src0 + off) : (const void*)s_zero16;
                glds16(src, __builtin_amdgcn_readfirstlane(lds_base + W_BYTES + b * PATCH_LDS + inst * 1024));
            }
        }
    };
    // weights: 28 KiB = 28 instructions, source-side swizzle (lane l lands in row l>>2, chunk l&3)
    {
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int inst = wave + 8 * q;
            if (inst < 28) {
                const int row = inst * 16 + (lane >> 2);                  // 0..447 = ky*64 + r
                const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
                glds16(wb + row * 64 + chunk * 16, __builtin_amdgcn_readfirstlane(lds_base + inst * 1024));
            }
        }
    }
    int t = blockIdx.x;
    if (t < ntiles) load_patch(t, 0);

    const int lrow = lane & 15, lchunk = lane >> 4;
    constexpr bool BIAS_REGS = MJ <= 7;      // the 8-block variant has no registers to spare: its biases are re-read per band
    float bb[2][8];
    if (BIAS_REGS) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int e = 0; e < 8; ++e) bb[pr][e] = bias ? bias[pr * 32 + (lane >> 4) * 8 + e] : 0.f;
    }

    int buf = 0;
    for (; t < ntiles; t += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of the patch (and weights) landed
        __syncthreads();                                           // everyone's did; the other buffer is free
        if (t + (int)gridDim.x < ntiles) load_patch(t + gridDim.x, buf ^ 1);

        f32x4 acc[4][MJ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int wrow = wave / WPR, col0 = (wave - wrow * WPR) * (MJ * 16);    // this wave's output row and first column
        const unsigned char* P = lds + W_BYTES + buf * PATCH_LDS + (2 * wrow) * ROW_BYTES + 16 * (col0 + lrow + lchunk);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            u32x4 a[4], b[MJ];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const u32x4*>(lds + w_swz(ky * 64 + i * 16 + lrow, lchunk));
#pragma unroll
            for (int j = 0; j < MJ; ++j) b[j] = *reinterpret_cast<const u32x4*>(P + ky * ROW_BYTES + j * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MJ; ++j) mma<T>(acc[i][j], a[i], b[j]);
        }
        // epilogue: output row (band*8 + wave) of image img
        const int tt = reverse ? ntiles - 1 - t : t;
        const int img = tt / (WO / BAND), band = tt - img * (WO / BAND);
        T* orow = out + (((size_t)img * WO + band * BAND + wrow) * WO + col0) * 64;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int n = pr * 32 + (lane >> 4) * 8;
            float bl[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bl[e] = BIAS_REGS ? bb[pr][e] : (bias ? bias[n + e] : 0.f);
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * pr][j][e] + bl[e];
                    v[4 + e] = acc[2 * pr + 1][j][e] + bl[4 + e];
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *reinterpret_cast<u32x4*>(orow + (size_t)(j * 16 + lrow) * 64 + n) = pack8(v, T());
            }
        }
    }
}

// ------------------------------------------------------------------ stem + CubePad(1) + max-pool 3x3 s2 in one kernel
// (resnet_cubic.py:165-170; cube 224, 16-bit types, ReLU on).  Separately the stem writes its 112x112x64 output
// (616 MB for 64 frames) and the max-pool reads it again; here the pooled 56x56x64 map (154 MB) is all that leaves.
//   * a workgroup owns a band of 8 stem rows = 4 pooled rows; pooled row 4b needs stem row 8b-1 as well: that halo
//     row is computed again (9 rows x 7 pixel blocks = 63 units: wave w takes its row's 7 blocks and block w of the
//     halo row), its 23 padded input rows are one contiguous 42 KB range (double-buffered LDS-DMA as above);
//   * a wave pools its own row HORIZONTALLY in registers: after bias + ReLU + rounding a lane holds 8 channels of one
//     pixel as four dwords of two 16-bit values; the left / right neighbour pixels are the neighbouring lanes of
//     its 16-lane row (DPP row_shr / row_shl; lane 0's left neighbour is lane 15 of the previous block: row_ror of
//     that block's register), and the maximum of non-negative bf16 / fp16 values is the maximum of their bit patterns
//     (v_pk_max_u16); even pixels keep the result and store it to LDS (8 rows x 56 x 128 B), the halo row goes there raw;
//   * after a barrier the vertical 3-maximum runs from LDS and the pooled rows leave with 16-byte stores.
// CubePad(1) of the max-pool needs stem pixels of OTHER faces (row -1 / column -1 of the padded map): this kernel
// treats them as absent, also writes the four border rows / columns of every face's stem output to `border`
// ([n_img][top, bottom, left, right][112][64], 57 KB per face) and stem_pool_fix_kernel folds the padding in
// afterwards (pooled row 0 and column 0 of every face: 111 of 3136 pixels).  Bit-identical to the two-kernel path.
namespace {
constexpr int SP_PATCH_ROWS = 23, SP_PATCH_BYTES = SP_PATCH_ROWS * 1840, SP_PATCH_INST = 42, SP_PATCH_LDS = SP_PATCH_INST * 1024;
constexpr int SP_HROW = 56 * 128;                           // one horizontally pooled row
constexpr int SP_EXTRA = 2 * SP_HROW + 112 * 128;           // pooled rows 6, 7 and the raw halo row (28,672 B)
static_assert(6 * SP_HROW == SP_PATCH_LDS, "pooled rows 0-5 take the place of the current patch");
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ u32x4 pkmax4(const u32x4& a, const u32x4& b) {
    return u32x4{pkmax(a.x, b.x), pkmax(a.y, b.y), pkmax(a.z, b.z), pkmax(a.w, b.w)};
}
}  // namespace

template <typename T>
__global__ __launch_bounds__(512) void stem_pool_kernel(const T* __restrict__ xp, const T* __restrict__ wpk,
                                                       const float* __restrict__ bias, T* __restrict__ y,
                                                       T* __restrict__ border, int n_img, int reverse) {
    constexpr int WP = 230, WO = 112, HO = 56, ROW_BYTES = 1840, BAND = 8, NBAND = WO / BAND;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[W_BYTES + 2 * SP_PATCH_LDS + SP_EXTRA];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)lds;
    const int ntiles = n_img * NBAND;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(xp);
    const size_t img_bytes = (size_t)WP * ROW_BYTES;
    unsigned char* extra = lds + W_BYTES + 2 * SP_PATCH_LDS;

    // patch of tile t -> buffer b: padded input rows 16 band - 2 .. 16 band + 20 (band 0 has no halo row: its first two
    // patch rows are never read and come from the zero line)
    auto load_patch = [&](int t, int b) __attribute__((always_inline)) {
        if (reverse) t = ntiles - 1 - t;
        const int img = t / NBAND, band = t - img * NBAND;
        const long long row0 = (long long)16 * band - 2;
        const unsigned char* src0 = xb + (size_t)img * img_bytes + row0 * ROW_BYTES;
#pragma unroll
        for (int q = 0; q < (SP_PATCH_INST + 7) / 8; ++q) {
            const int inst = wave + 8 * q;
            if (inst < SP_PATCH_INST) {
                const int off = inst * 1024 + lane * 16;
                const bool ok = off < SP_PATCH_BYTES && (band > 0 || off >= 2 * ROW_BYTES);
                const void* src = ok ? (const void*)(src0 + off) : (const void*)s_zero16;
                glds16(src, __builtin_amdgcn_readfirstlane(lds_base + W_BYTES + b * SP_PATCH_LDS + inst * 1024));
            }
        }
    };
    {   // weights: as stem_kernel
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int inst = wave + 8 * q;
            if (inst < 28) {
                const int row = inst * 16 + (lane >> 2);
                const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
                glds16(wb + row * 64 + chunk * 16, __builtin_amdgcn_readfirstlane(lds_base + inst * 1024));
            }
        }
    }
    int t = blockIdx.x;
    if (t < ntiles) load_patch(t, 0);

    const int lrow = lane & 15, lchunk = lane >> 4;
    int buf = 0;
    for (; t < ntiles; t += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                           // the patch landed; the other buffer and `extra` are free
        if (t + (int)gridDim.x < ntiles) load_patch(t + gridDim.x, buf ^ 1);
        const int tt = reverse ? ntiles - 1 - t : t;
        const int img = tt / NBAND, band = tt - img * NBAND;
        const bool halo = band > 0 && wave < 7;                    // unit 7: block `wave` of stem row 8 band - 1
        unsigned char* pbuf = lds + W_BYTES + buf * SP_PATCH_LDS;

        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[i][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        // stem row 8 band + w = patch rows 2 (w + 1) ..; the halo row = patch rows 0 ..
        const unsigned char* P = pbuf + (2 * (wave + 1)) * ROW_BYTES + 16 * (lrow + lchunk);
        const unsigned char* PH = pbuf + 16 * (wave * 16 + lrow + lchunk);
        // (not unrolled: with the seven filter rows unrolled the next row's 12 fragment reads are hoisted above this row's
        // MFMAs, 256 VGPRs + 2 spilled; rolled: 204, no scratch, the same time - 4.98 against 4.99 ms per 64-frame stage)
#pragma unroll 1
        for (int ky = 0; ky < 7; ++ky) {
            u32x4 a[4], b[8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const u32x4*>(lds + w_swz(ky * 64 + i * 16 + lrow, lchunk));
#pragma unroll
            for (int u = 0; u < 7; ++u) b[u] = *reinterpret_cast<const u32x4*>(P + ky * ROW_BYTES + u * 256);
            b[7] = *reinterpret_cast<const u32x4*>(PH + ky * ROW_BYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < 8; ++u) mma<T>(acc[i][u], a[i], b[u]);
        }
        __syncthreads();                                           // every wave is done reading the patch
        // ---- bias + ReLU + one rounding; border copies; horizontal 3-maximum; pooled rows / halo row -> LDS
        const int srow = band * BAND + wave;                       // this wave's stem row
        T* bimg = border + (size_t)img * 4 * WO * 64;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int n = pr * 32 + lchunk * 8;
            float bl[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bl[e] = bias ? bias[n + e] : 0.f;
            u32x4 o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[2 * pr][u][e] + bl[e], 0.f);
                    v[4 + e] = fmaxf(acc[2 * pr + 1][u][e] + bl[4 + e], 0.f);
                }
                o[u] = pack8(v, T());
            }
            // the face's border rows / columns, raw (the max-pool's CubePad reads them from other faces)
            if (srow == 0 || srow == WO - 1) {
                T* brow = bimg + (size_t)(srow == 0 ? 0 : 1) * WO * 64;
#pragma unroll
                for (int u = 0; u < 7; ++u) *reinterpret_cast<u32x4*>(brow + (size_t)(u * 16 + lrow) * 64 + n) = o[u];
            }
            if (lrow == 0) *reinterpret_cast<u32x4*>(bimg + ((size_t)2 * WO + srow) * 64 + n) = o[0];
            if (lrow == 15) *reinterpret_cast<u32x4*>(bimg + ((size_t)3 * WO + srow) * 64 + n) = o[6];
            // horizontal 3-maximum at the even pixels: x = 16 u + lrow, neighbours x - 1 / x + 1 (x = 0: no left one)
            unsigned char* hrow = wave < 6 ? pbuf + wave * SP_HROW : extra + (wave - 6) * SP_HROW;
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                u32x4 m = o[u];
                const unsigned* cur = reinterpret_cast<const unsigned*>(&o[u]);
                const unsigned* prv = reinterpret_cast<const unsigned*>(&o[u > 0 ? u - 1 : 0]);
                unsigned* mm = reinterpret_cast<unsigned*>(&m);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned right = (unsigned)__builtin_amdgcn_update_dpp(0, (int)cur[d], 0x101, 0xf, 0xf, true);     // row_shl:1
                    // lane 0 of the 16-lane row: lane 15 of the previous block (row_ror:1 of its register); block 0: nothing
                    const unsigned wrap = u > 0 ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)prv[d], 0x121, 0xf, 0xf, true) : 0u;
                    const unsigned left = (unsigned)__builtin_amdgcn_update_dpp((int)wrap, (int)cur[d], 0x111, 0xf, 0xf, false);   // row_shr:1
                    mm[d] = pkmax(pkmax(cur[d], left), right);
                }
                if ((lrow & 1) == 0)
                    *reinterpret_cast<u32x4*>(hrow + (u * 8 + (lrow >> 1)) * 128 + n * 2) = m;
            }
            if (halo) *reinterpret_cast<u32x4*>(extra + 2 * SP_HROW + (wave * 16 + lrow) * 128 + n * 2) = o[7];
        }
        __syncthreads();                                           // the pooled rows (and the halo row) are complete
        // ---- vertical 3-maximum: pooled row i of the band = stem rows 2i-1 (i = 0: the halo row), 2i, 2i+1
        T* yimg = y + ((size_t)img * HO + band * 4) * HO * 64;
        for (int it = tid; it < 4 * HO * 8; it += 512) {
            const int i = it / (HO * 8), rem = it - i * (HO * 8);
            const int c = rem >> 3, k = rem & 7;
            auto hp = [&](int r) __attribute__((always_inline)) -> u32x4 {
                const unsigned char* base = r < 6 ? pbuf + r * SP_HROW : extra + (r - 6) * SP_HROW;
                return *reinterpret_cast<const u32x4*>(base + c * 128 + k * 16);
            };
            u32x4 m = pkmax4(hp(2 * i), hp(2 * i + 1));
            if (i > 0) {
                m = pkmax4(m, hp(2 * i - 1));
            } else if (band > 0) {
                const unsigned char* hr = extra + 2 * SP_HROW + k * 16;
                m = pkmax4(m, *reinterpret_cast<const u32x4*>(hr + (2 * c) * 128));
                m = pkmax4(m, *reinterpret_cast<const u32x4*>(hr + (2 * c + 1) * 128));
                if (c > 0) m = pkmax4(m, *reinterpret_cast<const u32x4*>(hr + (2 * c - 1) * 128));
            }
            *reinterpret_cast<u32x4*>(yimg + ((size_t)i * HO + c) * 64 + k * 8) = m;
        }
    }
}

// The same stage as TWO workgroups per CU (round 6): a workgroup of 4 waves owns 4 stem rows = 2 pooled rows (+ the halo row above,
// computed again: 35 MFMA units per band, 9 / 9 / 9 / 8 per wave), ONE patch buffer (15 padded input rows = 27.6 KB) and 70 KB of LDS.
// stem_pool_kernel's single 8-wave workgroup per CU runs its phases - MFMAs, bias / ReLU / pack / horizontal maximum (VALU), LDS
// exchange, vertical maximum + stores - strictly one after the other with three workgroup barriers per band (9.8 us per band of
// which 3.6 are MFMA issue time); here the co-resident workgroup's MFMAs run under this one's VALU / store phases instead.
namespace {
constexpr int SQ_PATCH_ROWS = 15, SQ_PATCH_BYTES = SQ_PATCH_ROWS * 1840, SQ_PATCH_INST = 27;
constexpr int SQ_HBUF = 4 * SP_HROW;                         // four horizontally pooled rows (28,672 B) take the patch's place
static_assert(SQ_PATCH_INST * 1024 <= SQ_HBUF, "the pooled rows overlay the patch");
}  // namespace

// (A prefetch form of this kernel - filter fragments from L2 one row ahead instead of from LDS, the 28 KB they free holding a separate
// pooled-row buffer so that the next band's patch lands under the epilogue - measured 215 us against 199: profiles/r06_stem_ab.log.)
template <typename T>
__global__ __launch_bounds__(256, 2) void stem_pool4_kernel(const T* __restrict__ xp, const T* __restrict__ wpk,
                                                            const float* __restrict__ bias, T* __restrict__ y,
                                                            T* __restrict__ border, int n_img, int reverse) {
    constexpr int WP = 230, WO = 112, HO = 56, ROW_BYTES = 1840, BAND = 4, NBAND = WO / BAND;
    // LDS: filter | patch, later the four pooled rows in its place | raw halo row | 64 biases
    constexpr int OFF_PATCH = W_BYTES, OFF_HBUF = W_BYTES, OFF_HRAW = OFF_HBUF + SQ_HBUF, OFF_BIAS = OFF_HRAW + 112 * 128;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[OFF_BIAS + 256];
    static_assert(OFF_BIAS + 256 <= 80 * 1024, "two workgroups per CU");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)lds;
    const int ntiles = n_img * NBAND;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(xp);
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk);
    const size_t img_bytes = (size_t)WP * ROW_BYTES;
    unsigned char* pbuf = lds + OFF_PATCH;
    unsigned char* hbuf = lds + OFF_HBUF;
    unsigned char* hraw = lds + OFF_HRAW;
    float* bias_s = reinterpret_cast<float*>(lds + OFF_BIAS);
    if (tid < 64) bias_s[tid] = bias ? bias[tid] : 0.f;

    // patch of tile t: padded input rows 8 band - 2 .. 8 band + 12 (band 0 has no halo row: its first two patch rows come from the zero line)
    auto load_patch = [&](int t) __attribute__((always_inline)) {
        const int tt = reverse ? ntiles - 1 - t : t;
        const int img = tt / NBAND, band = tt - img * NBAND;
        const long long row0 = (long long)8 * band - 2;
        const unsigned char* src0 = xb + (size_t)img * img_bytes + row0 * ROW_BYTES;
#pragma unroll
        for (int q = 0; q < (SQ_PATCH_INST + 3) / 4; ++q) {
            const int inst = wave + 4 * q;
            if (inst < SQ_PATCH_INST) {
                const int off = inst * 1024 + lane * 16;
                const bool ok = off < SQ_PATCH_BYTES && (band > 0 || off >= 2 * ROW_BYTES);
                const void* src = ok ? (const void*)(src0 + off) : (const void*)s_zero16;
                glds16(src, __builtin_amdgcn_readfirstlane(lds_base + OFF_PATCH + inst * 1024));
            }
        }
    };
    {   // filter: 28 KiB = 28 instructions, source-side swizzle (as stem_kernel), seven per wave
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int inst = wave + 4 * q;
            const int row = inst * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
            glds16(wb + row * 64 + chunk * 16, __builtin_amdgcn_readfirstlane(lds_base + inst * 1024));
        }
    }
    const int lrow = lane & 15, lchunk = lane >> 4;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tt = reverse ? ntiles - 1 - t : t;
        const int img = tt / NBAND, band = tt - img * NBAND;
        __syncthreads();                                           // the previous band's pooled rows (over the patch) have been read
        load_patch(t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                           // the patch (and, first time, the filter) landed
        const bool halo = band > 0;
        // units 0-6: the seven pixel blocks of stem row 4 band + wave (patch rows 2 (wave + 1) ..); units 7, 8: blocks 2 wave,
        // 2 wave + 1 of the halo row 4 band - 1 (patch rows 0 ..; wave 3 has block 6 only - its unit 8 repeats it and is dropped)
        const int hb0 = 2 * wave, hb1 = wave < 3 ? 2 * wave + 1 : 6;
        f32x4 acc[4][9];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 9; ++u) acc[i][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* P = pbuf + (2 * (wave + 1)) * ROW_BYTES + 16 * (lrow + lchunk);
        const unsigned char* PH0 = pbuf + 16 * (hb0 * 16 + lrow + lchunk);
        const unsigned char* PH1 = pbuf + 16 * (hb1 * 16 + lrow + lchunk);
        auto load_w = [&](int ky, u32x4 (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const u32x4*>(lds + w_swz(ky * 64 + i * 16 + lrow, lchunk));
        };
#pragma unroll 1
        for (int ky = 0; ky < 7; ++ky) {
            u32x4 a[4], b[9];
            load_w(ky, a);
#pragma unroll
            for (int u = 0; u < 7; ++u) b[u] = *reinterpret_cast<const u32x4*>(P + ky * ROW_BYTES + u * 256);
            b[7] = *reinterpret_cast<const u32x4*>(PH0 + ky * ROW_BYTES);
            b[8] = *reinterpret_cast<const u32x4*>(PH1 + ky * ROW_BYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < 9; ++u) mma<T>(acc[i][u], a[i], b[u]);
        }
        __syncthreads();                                           // every wave is done reading the patch
        // ---- bias + ReLU + one rounding; border copies; horizontal 3-maximum; pooled rows / halo row -> LDS (as stem_pool_kernel)
        const int srow = band * BAND + wave;                       // this wave's stem row
        T* bimg = border + (size_t)img * 4 * WO * 64;
        unsigned char* hrow = hbuf + wave * SP_HROW;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int n = pr * 32 + lchunk * 8;
            float bl[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bl[e] = bias_s[n + e];
            u32x4 o[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[2 * pr][u][e] + bl[e], 0.f);
                    v[4 + e] = fmaxf(acc[2 * pr + 1][u][e] + bl[4 + e], 0.f);
                }
                o[u] = pack8(v, T());
            }
            if (srow == 0 || srow == WO - 1) {
                T* brow = bimg + (size_t)(srow == 0 ? 0 : 1) * WO * 64;
#pragma unroll
                for (int u = 0; u < 7; ++u) *reinterpret_cast<u32x4*>(brow + (size_t)(u * 16 + lrow) * 64 + n) = o[u];
            }
            if (lrow == 0) *reinterpret_cast<u32x4*>(bimg + ((size_t)2 * WO + srow) * 64 + n) = o[0];
            if (lrow == 15) *reinterpret_cast<u32x4*>(bimg + ((size_t)3 * WO + srow) * 64 + n) = o[6];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                u32x4 m = o[u];
                const unsigned* cur = reinterpret_cast<const unsigned*>(&o[u]);
                const unsigned* prv = reinterpret_cast<const unsigned*>(&o[u > 0 ? u - 1 : 0]);
                unsigned* mm = reinterpret_cast<unsigned*>(&m);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned right = (unsigned)__builtin_amdgcn_update_dpp(0, (int)cur[d], 0x101, 0xf, 0xf, true);     // row_shl:1
                    const unsigned wrap = u > 0 ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)prv[d], 0x121, 0xf, 0xf, true) : 0u;
                    const unsigned left = (unsigned)__builtin_amdgcn_update_dpp((int)wrap, (int)cur[d], 0x111, 0xf, 0xf, false);   // row_shr:1
                    mm[d] = pkmax(pkmax(cur[d], left), right);
                }
                // (chunk ch of pooled pixel px sits at chunk ch ^ (px & 7) of its 128-byte row: the lanes of one store - neighbouring
                //  pixels, the same chunk - would otherwise all hit the same banks, 4-way for these rows and 8-way for the halo row)
                if ((lrow & 1) == 0) {
                    const int px = u * 8 + (lrow >> 1);
                    *reinterpret_cast<u32x4*>(hrow + px * 128 + (((n >> 3) ^ (px & 7)) << 4)) = m;
                }
            }
            if (halo) {
                *reinterpret_cast<u32x4*>(hraw + (hb0 * 16 + lrow) * 128 + (((n >> 3) ^ (lrow & 7)) << 4)) = o[7];
                if (wave < 3) *reinterpret_cast<u32x4*>(hraw + (hb1 * 16 + lrow) * 128 + (((n >> 3) ^ (lrow & 7)) << 4)) = o[8];
            }
        }
        __syncthreads();                                           // the pooled rows (and the halo row) are complete
        // ---- vertical 3-maximum: pooled row i of the band = stem rows 2i-1 (i = 0: the halo row), 2i, 2i+1
        T* yimg = y + ((size_t)img * HO + band * 2) * HO * 64;
        for (int it = tid; it < 2 * HO * 8; it += 256) {
            const int i = it / (HO * 8), rem = it - i * (HO * 8);
            const int c = rem >> 3, k = rem & 7;
            auto hp = [&](int r) __attribute__((always_inline)) -> u32x4 {
                return *reinterpret_cast<const u32x4*>(hbuf + r * SP_HROW + c * 128 + ((k ^ (c & 7)) << 4));
            };
            u32x4 m = pkmax4(hp(2 * i), hp(2 * i + 1));
            if (i > 0) {
                m = pkmax4(m, hp(2 * i - 1));
            } else if (halo) {
                auto hx = [&](int px) __attribute__((always_inline)) -> u32x4 {
                    return *reinterpret_cast<const u32x4*>(hraw + px * 128 + ((k ^ (px & 7)) << 4));
                };
                m = pkmax4(m, hx(2 * c));
                m = pkmax4(m, hx(2 * c + 1));
                if (c > 0) m = pkmax4(m, hx(2 * c - 1));
            }
            *reinterpret_cast<u32x4*>(yimg + ((size_t)i * HO + c) * 64 + k * 8) = m;
        }
    }
}

// CubePad(1) of the max-pool: pooled row 0 and column 0 of every face take the maximum with the padding pixels of their
// windows - padded row -1 / column -1 of the face = border pixels of other faces' stem outputs (cubepad_src).
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_fix_kernel(T* __restrict__ y, const T* __restrict__ border, int n_img) {
    constexpr int WO = 112, HO = 56;
    const CubePadGeom g{WO, 1, 1, 1, 1};
    const int total = n_img * (2 * HO - 1) * 8;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int k = idx & 7;
        int rest = idx >> 3;
        const int e = rest % (2 * HO - 1);                         // 0..55: pooled (0, e); 56..110: pooled (e - 55, 0)
        const int img = rest / (2 * HO - 1);
        const int grp = img / 6, f = img - grp * 6;
        const int oy = e < HO ? 0 : e - (HO - 1), ox = e < HO ? e : 0;
        T* py = y + (((size_t)img * HO + oy) * HO + ox) * 64 + k * 8;
        u32x4 m = *reinterpret_cast<const u32x4*>(py);
        auto pad = [&](int pyy, int pxx) __attribute__((always_inline)) {           // padded coordinates, row or column 0
            const int s = cubepad_src(f, pyy, pxx, g);             // pixel of the group's 6 x 112 x 112 stem output
            const int sf = s / (WO * WO), r = s - sf * (WO * WO);
            const int sy = r / WO, sx = r - sy * WO;
            const int side = sy == 0 ? 0 : sy == WO - 1 ? 1 : sx == 0 ? 2 : 3;
            const int pos = side < 2 ? sx : sy;
            const T* src = border + ((((size_t)grp * 6 + sf) * 4 + side) * WO + pos) * 64 + k * 8;
            m = pkmax4(m, *reinterpret_cast<const u32x4*>(src));
        };
        // the window of pooled (oy, ox) covers padded rows 2 oy .. 2 oy + 2, columns 2 ox .. 2 ox + 2
        if (oy == 0) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) pad(0, 2 * ox + kx);
        }
        if (ox == 0) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
                if (oy > 0 || ky > 0) pad(2 * oy + ky, 0);
        }
        *reinterpret_cast<u32x4*>(py) = m;
    }
}

extern "C" size_t cp360_stem_pool_border_bytes(int n_img) { return n_img > 0 ? (size_t)n_img * 4 * 112 * 64 * 2 : 0; }

extern "C" int cp360_stem_pool_forward(int dtype, const void* xp, const void* packed, const float* bias, void* y,
                                       void* border, int n_img, int cube_dim, void* stream) {
    if (!xp || !packed || !y || !border) return CP360_ERR_NULL;
    if (n_img <= 0) return CP360_ERR_BAD_SHAPE;
    if (n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (cube_dim != 224) return CP360_ERR_UNSUPPORTED;             // other cube sizes: cp360_stem_forward + cp360_cubepad_maxpool3s2
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = n_img * 14;
    const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
    const int rev = cp360_launch_reverse();
    const int fix_blocks = (n_img * 111 * 8 + 255) / 256;
    // CP360_STEM_POOL=8: the round-2 form (one 8-wave workgroup per CU, 8 stem rows per band); default: two 4-wave workgroups per CU
    // (199 against 206 us, profiles/r06_stem_ab.log)
    static const int form = []() { const char* e = getenv("CP360_STEM_POOL"); return e ? atoi(e) : 4; }();
    const int ntiles4 = n_img * 28;
    const dim3 grid4((unsigned)(ntiles4 < 512 ? ntiles4 : 512));
#define CP360_SP(TT)                                                                                                   \
    {                                                                                                                  \
        if (form == 8)                                                                                                 \
            hipLaunchKernelGGL((stem_pool_kernel<TT>), grid, dim3(512), 0, st, (const TT*)xp, (const TT*)packed, bias, (TT*)y, \
                               (TT*)border, n_img, rev);                                                               \
        else                                                                                                           \
            hipLaunchKernelGGL((stem_pool4_kernel<TT>), grid4, dim3(256), 0, st, (const TT*)xp, (const TT*)packed, bias, (TT*)y, \
                               (TT*)border, n_img, rev);                                                               \
        hipLaunchKernelGGL((stem_pool_fix_kernel<TT>), dim3((unsigned)fix_blocks), dim3(256), 0, st, (TT*)y,            \
                           (const TT*)border, n_img);                                                                  \
    }
    if (dtype == CP360_BF16) CP360_SP(bf16_raw)
    else if (dtype == CP360_F16) CP360_SP(f16_raw)
#undef CP360_SP
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" size_t cp360_stem_packed_bytes(int dtype) {
    return (dtype == CP360_BF16 || dtype == CP360_F16) ? (size_t)W_BYTES : 0;
}

extern "C" int cp360_stem_pack_weights(int dtype, const float* w_oihw, const float* scale, void* packed, void* stream) {
    if (!w_oihw || !packed) return CP360_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CP360_BF16)
        hipLaunchKernelGGL((stem_pack_kernel<bf16_raw>), dim3(56), dim3(256), 0, st, w_oihw, scale, (bf16_raw*)packed);
    else if (dtype == CP360_F16)
        hipLaunchKernelGGL((stem_pack_kernel<f16_raw>), dim3(56), dim3(256), 0, st, w_oihw, scale, (f16_raw*)packed);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_stem_forward(int dtype, const void* xp, const void* packed, const float* bias, void* out, int n_img,
                                  int cube_dim, int relu, void* stream) {
    if (!xp || !packed || !out) return CP360_ERR_NULL;
    if (n_img <= 0) return CP360_ERR_BAD_SHAPE;
    if (cube_dim != 224 && cube_dim != 512) return CP360_ERR_UNSUPPORTED;   // other cube sizes: the generic implicit GEMM
    hipStream_t st = (hipStream_t)stream;
#define CP360_STEM(TT, CDV)                                                                                     \
    {                                                                                                           \
        const int ntiles = n_img * (StemGeom<CDV>::WO / StemGeom<CDV>::BAND);                                   \
        const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));                                               \
        hipLaunchKernelGGL((stem_kernel<TT, CDV>), grid, dim3(512), 0, st, (const TT*)xp, (const TT*)packed, bias, \
                           (TT*)out, n_img, relu, cp360_launch_reverse());                                      \
    }
    if (dtype == CP360_BF16) { if (cube_dim == 224) CP360_STEM(bf16_raw, 224) else CP360_STEM(bf16_raw, 512) }
    else if (dtype == CP360_F16) { if (cube_dim == 224) CP360_STEM(f16_raw, 224) else CP360_STEM(f16_raw, 512) }
#undef CP360_STEM
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}
