// K3d: one whole Bottleneck tail of layer1 in ONE kernel (16-bit types; 56x56 faces = cube 224, 128x128 = cube 512):
//
//   mid  --CubePad(1)+conv3x3 64->64 +bn2+relu-->  t  --conv1x1 64->256 +bn3 (+ residual | + downsample(x)) +relu-->  out
//                                                        `--(optional) next block's conv1x1 256->64 +bn1+relu-->  mid'
//
// = conv2 / conv3 / residual add of model/resnet_cubic.py:85-106 and the conv1 of the NEXT block (:88-90).
// Run as separate launches these are HBM-bound (layer1 = 1.9 of the static stage's 7 ms) and every
// tensor between them makes a round trip through HBM: t (154 MB for 64 frames) twice, out (616 MB) written by conv3
// and read again by the next conv1.  Here a workgroup owns a band of 4 output rows of one face, as band3x3.hip:
//   stage 1  conv2 as band3x3_kernel (resident cube-padded band in LDS, nine taps read it there), its weights as MFMA
//            A fragments straight from L2 into registers two taps ahead (no LDS ring, no barrier in the stage);
//   stage 2  the conv2 accumulators ARE the next MFMA's B operand: with the acc_chan row order of the packed
//            weights a lane ends with EIGHT consecutive channels of one pixel per pair of MFMA row blocks - the
//            k-group layout of a 16x16x32 B fragment.  bias + ReLU + one rounding and the packed 16 bytes feed conv3
//            straight from registers (no LDS round trip, no barrier);
//   stage 3  conv3 in 8 passes of 32 output channels (K = 64: 16 MFMAs per pass and wave); its epilogue adds the
//            residual piece (16-byte loads, 64 contiguous bytes per pixel and pass) or, for the first block, the
//            downsample branch as 16 more MFMAs on x's fragments (K = 64 + 64), applies ReLU, rounds once, stores
//            the 16-byte piece - and that packed piece is again a B fragment: k-block `pass` of the next block's
//            conv1, accumulated on the fly (16 MFMAs per pass).
// conv3's (and the next conv1's) weights sit in LDS in fragment order (the conv2 ring and the patch are dead by
// then), the downsample filter's fragments come from L2.  Wave w = output row w of the band, as in stage 1.
// HBM traffic per block (64 frames): mid 154 MB + residual 616 MB + out 616 MB (+ mid' 154 MB) instead of
// 154 + 154 | 154 + 616 + 616 | 616 + 154.
// NEXTC = 128 (round 3): the LAST block of layer1 chains layer2's first conv1 (1x1, 256 -> 128, model/resnet_cubic.py:
// 88-90 of layer2.0), whose own launch re-read the 616 MB this kernel has just written.  128 accumulator channels do not
// fit next to the tail's registers for 64 pixels per wave, so stage 3 runs twice over HALF of the wave's pixel blocks
// (2 x 8 passes; conv3's fragments - 4 KiB per pass - then come from L2 one pass ahead instead of from LDS, which holds
// the 64 KiB of next-conv1 fragments).
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {
constexpr int C = 64, CO = 256;
constexpr int W2_TAP = 8 * 1024;                             // conv2: 8 fragments (4 row blocks x 2 k-blocks) per tap
constexpr int W3_BYTES = CO * C * 2;                         // 32 KiB of conv3 fragments
constexpr int W1_BYTES = C * CO * 2;                         // 32 KiB of next-conv1 fragments
constexpr int STAGE3_LDS = W3_BYTES + W1_BYTES + (CO + C) * 4;
constexpr int C1W = 128;                                     // the wide chained conv1 (layer2.0's: 256 -> 128)
constexpr int W1W_BYTES = C1W * CO * 2;                      // 64 KiB of its fragments
constexpr int STAGE3W_LDS = W1W_BYTES + (CO + C1W) * 4;
// Face size N, BAND output rows per workgroup (4 waves): 56x56 faces (cube 224) -> 4 rows, wave = row (64 pixel slots,
// 8 of them padding); 128x128 faces (cube 512, BASELINE config C5) -> 2 rows, two waves per row (64 pixels each).
template <int N_, int BAND_> struct L1Geom {
    static constexpr int N = N_, BAND = BAND_, NP = N + 2, WPR = 4 / BAND;
    static constexpr int PATCH_PX = (BAND + 2) * NP;                          // 348 / 520
    static constexpr int PATCH_INST = (PATCH_PX + 7) / 8;                     // DMA instructions of 8 pixels x 128 B
    static constexpr int PATCH_LDS = PATCH_INST * 1024;                       // 45,056 / 66,560
    static constexpr int READ_END = ((BAND + 1) * NP + 2 + 64 * WPR) * 128;   // padding columns read past the patch
    static constexpr int STAGE1_LDS = PATCH_LDS > READ_END ? PATCH_LDS : READ_END;
    static constexpr int LDS_BYTES = STAGE1_LDS > STAGE3_LDS ? STAGE1_LDS : STAGE3_LDS;                // 66,816: two per CU
    static constexpr int LDS_BYTES_W = STAGE1_LDS > STAGE3W_LDS ? STAGE1_LDS : STAGE3W_LDS;            // 67,072 (NEXTC = 128)
    static_assert(N % BAND == 0 && 64 * WPR >= N && 4 % BAND == 0, "band geometry");
};

__device__ __attribute__((aligned(16))) unsigned int l_zero16[4] = {0u, 0u, 0u, 0u};

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
// packed row R of a 32-row group <- channel (acc_chan order: MFMA blocks 2q, 2q+1 give a lane 8 consecutive channels)
__host__ __device__ __forceinline__ int row_chan(int R) { return (R & ~31) + ((R >> 2) & 3) * 8 + ((R >> 4) & 1) * 4 + (R & 3); }
}  // namespace

// conv2 weights [64, 64, 3, 3] (times scale) -> MFMA A fragments [tap 9][row block 4][kk 2][lane][8]: lane l holds row
// (l & 15) of the block (rows in acc_chan order), input channels kk*32 + (l>>4)*8 .. +7 of tap (ky, kx)
template <typename T>
__global__ __launch_bounds__(256) void l1_pack_conv2_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                            T* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 9 * C * C) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, kk = (idx >> 9) & 1, rb = (idx >> 10) & 3, tap = idx >> 12;
    const int n = row_chan(rb * 16 + (lane & 15));
    const int c = kk * 32 + (lane >> 4) * 8 + e;
    const float v = w[((size_t)n * C + c) * 9 + tap] * (scale ? scale[n] : 1.f);
    if constexpr (__is_same(T, f16_raw)) packed[idx] = (f16_raw)v;
    else packed[idx] = f32_to_bf16(v);
}

// 1x1 filter w [n_out, k] (times scale[n_out]) -> MFMA A fragments, 1 KiB each ([lane][8 elements]: lane l holds
// row (l & 15), k-group (l >> 4) of the fragment), rows in acc_chan order.
//   order 0 (row-pair major, for conv3 / downsample):  fragment ((p * 2 + rb) * KB + kb), p = 32-row pair
//   order 1 (k major, for the chained next conv1):     fragment (kb * RB + rb), rb = 16-row block
template <typename T>
__global__ __launch_bounds__(256) void frag_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                        T* __restrict__ packed, int n_out, int k, int order) {
    const int KB = k / 32, RB = n_out / 16;
    const int total = n_out * k;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int e = idx & 7, lane = (idx >> 3) & 63, frag = idx >> 9;
        int rb, kb;
        if (order == 0) { kb = frag % KB; rb = frag / KB; }
        else            { rb = frag % RB; kb = frag / RB; }
        const int R = rb * 16 + (lane & 15);
        const int n = row_chan(R);
        const int kk = kb * 32 + (lane >> 4) * 8 + e;
        // (__fmul_rn: the f32 product is rounded BEFORE the 16-bit conversion, as in conv_igemm.hip's pack_weights_kernel - left to the
        //  compiler the multiply and the f16 conversion become one mixed-precision instruction with a single rounding, and 2 of layer1.0's
        //  4096 conv1 weights land on the other side of a tie: the in-patch conv1 must use the bits the separate launch uses)
        const float v = scale ? __fmul_rn(w[(size_t)n * k + kk], scale[n]) : w[(size_t)n * k + kk];
        if constexpr (__is_same(T, f16_raw)) packed[idx] = (f16_raw)v;
        else packed[idx] = f32_to_bf16(v);
    }
}

// Diagnostic build only (-DL1_STAMPS, tools/l1_stamps.sh): s_memtime stamps of wave 0 of every workgroup at the phase boundaries
// (0 start, 1 patch landed + barrier, 2 conv2 done, 3 fragments landed + barrier, 4 + p after pass p of stage 3; 12 end); the product
// build executes no stamp.
#ifdef L1_STAMPS
__device__ unsigned long long g_l1_stamps[8192 * 16];
#define L1_STAMP(k)                                                                                         \
    { __builtin_amdgcn_sched_barrier(0);                                                                    \
      if (wave == 0 && lane == 0 && blockIdx.x < 8192) g_l1_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
      __builtin_amdgcn_sched_barrier(0); }
#else
#define L1_STAMP(k)
#endif

// FIRST (round 6; layer1.0, with DS): `x` is the BLOCK INPUT (the pooled stem output, 64 channels) instead of conv1's output - the block's own
// conv1 (1x1, 64 -> 64, + bn1 + relu, model/resnet_cubic.py:88-90) runs on the resident patch IN PLACE between the gather and conv2 (22 pixel
// blocks x 8 MFMAs per band, halo pixels computed again by the neighbouring bands with the same arithmetic: the same bits as the separate
// launch), so that launch (conv_pw64_kernel: 59 us, 154 MB read + 154 MB written per 64 frames) and its tensor do not exist.
template <typename T, bool DS, bool NEXT, int NV, int BANDV, bool FIRST = false>
__global__ __launch_bounds__(256, 2) void l1block_kernel(const T* __restrict__ x, const T* __restrict__ wpk2,
                                                         const float* __restrict__ bias2, const T* __restrict__ w3f,
                                                         const float* __restrict__ bias3, const T* __restrict__ res,
                                                         const T* __restrict__ xds, const T* __restrict__ wdf,
                                                         T* __restrict__ out, const T* __restrict__ w1f,
                                                         const float* __restrict__ bias1, T* __restrict__ out_next,
                                                         int reverse, const T* __restrict__ w0f = nullptr,
                                                         const float* __restrict__ bias0 = nullptr) {
    static_assert(!FIRST || DS, "the in-patch conv1 exists for the first block (downsample form)");
    typedef L1Geom<NV, BANDV> G;
    constexpr int N = G::N, NP = G::NP, BAND = G::BAND, WPR = G::WPR, PATCH_PX = G::PATCH_PX, PATCH_INST = G::PATCH_INST;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)lds;
    const int tile = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x, img = tile / (N / BAND), band = tile - img * (N / BAND);
    const int grp = img / 6, f = img - grp * 6;
    const CubePadGeom geom{N, 1, 1, 1, 1};
    const int lrow = lane & 15, lchunk = lane >> 4;
    const int wrow = wave / WPR, x0 = (wave - wrow * WPR) * 64;               // this wave's output row and first column
    const size_t row_px = ((size_t)img * N + band * BAND + wrow) * N + x0;    // its first pixel
    L1_STAMP(0)

    // ---- stage 1: conv2 on the resident cube-padded band (band3x3.hip), A fragments from L2 straight into
    // registers two taps ahead: no weight ring in LDS and no barrier inside the stage (l2block.hip)
    {
        const T* xg = x + (size_t)grp * 6 * N * N * C;
#pragma unroll 1
        for (int inst = wave; inst < PATCH_INST; inst += 4) {
            const int q = inst * 8 + (lane >> 3);
            const void* src = l_zero16;
            if (q < PATCH_PX) {
                const int pr = q / NP, pc = q - pr * NP;
                const int sp = cubepad_src(f, BAND * band + pr, pc, geom);
                src = xg + (size_t)sp * C + (((lane & 7) ^ px_swz(q)) << 3);
            }
            glds16(src, __builtin_amdgcn_readfirstlane(lds_base + inst * 1024));
        }
    }
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk2);
    auto load_a = [&](int t, u32x4 (&a)[4][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                a[i][kk] = *reinterpret_cast<const u32x4*>(wb + (size_t)t * W2_TAP + ((i * 2 + kk) * 64 + lane) * 16);
    };
    constexpr int DEPTH = 2;
    u32x4 aq[DEPTH + 1][4][2];
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) load_a(t, aq[t]);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the patch DMAs (and the first fragments)
    __syncthreads();
    if constexpr (FIRST) {
        // conv1 on the patch, in place: pixel block b = patch pixels 16 b .. 16 b + 15 (wave w takes b = w, w + 4, ...); a lane reads the two
        // k-block pieces of its pixel and writes the two 8-channel pieces it ends up holding into the same two slots of that pixel
        u32x4 a0[4][2];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                a0[rb][kb] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(w0f) + ((rb * 2 + kb) * 64 + lane) * 16);
        float b0[2][8];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int e = 0; e < 8; ++e) b0[pr][e] = bias0 ? bias0[pr * 32 + lchunk * 8 + e] : 0.f;
        static_assert(!FIRST || PATCH_INST % 2 == 0, "whole 16-pixel blocks");
#pragma unroll 1
        for (int blk = wave; blk < PATCH_INST / 2; blk += 4) {
            const int q = blk * 16 + lrow;
            unsigned char* px = lds + q * 128;
            u32x4 b[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) b[kb] = *reinterpret_cast<const u32x4*>(px + (((kb * 4 + lchunk) ^ px_swz(q)) << 4));
            f32x4 c[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) c[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) mma<T>(c[rb], a0[rb][kb], b[kb]);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(c[2 * pr][e] + b0[pr][e], 0.f);
                    v[4 + e] = fmaxf(c[2 * pr + 1][e] + b0[pr][4 + e], 0.f);
                }
                *reinterpret_cast<u32x4*>(px + (((pr * 4 + lchunk) ^ px_swz(q)) << 4)) = pack8(v, T());
            }
        }
        __syncthreads();                                       // every pixel of the patch is conv1's output now
    }
    L1_STAMP(1)

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap + DEPTH < 9) load_a(tap + DEPTH, aq[(tap + DEPTH) % (DEPTH + 1)]);
        const int ky = tap / 3, kx = tap - ky * 3;
        const int pbase = (wrow + ky) * NP + kx + x0 + lrow;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = pbase + 16 * j;
                b[j] = *reinterpret_cast<const u32x4*>(lds + p * 128 + (((kk * 4 + lchunk) ^ px_swz(p)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma<T>(acc[i][j], aq[tap % (DEPTH + 1)][i][kk], b[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    L1_STAMP(2)
    // every wave is done with the patch: bring conv3's (and the next conv1's) fragments in
    __syncthreads();
    {
        const unsigned char* s3 = reinterpret_cast<const unsigned char*>(w3f);
#pragma unroll
        for (int q = 0; q < W3_BYTES / 1024 / 4; ++q) {
            const int inst = q * 4 + wave;
            glds16(s3 + inst * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds_base + inst * 1024));
        }
        if (NEXT) {
            const unsigned char* s1 = reinterpret_cast<const unsigned char*>(w1f);
#pragma unroll
            for (int q = 0; q < W1_BYTES / 1024 / 4; ++q) {
                const int inst = q * 4 + wave;
                glds16(s1 + inst * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds_base + W3_BYTES + inst * 1024));
            }
        }
    }
    // ---- stage 2: t = relu(conv2 + b2), rounded once, as B fragments bt[k-block][pixel block]
    u32x4 bt[2][4];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int n = pr * 32 + lchunk * 8;
        float bb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[e] = bias2 ? bias2[n + e] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaxf(acc[2 * pr][j][e] + bb[e], 0.f);
                v[4 + e] = fmaxf(acc[2 * pr + 1][j][e] + bb[4 + e], 0.f);
            }
            bt[pr][j] = pack8(v, T());
        }
    }
    // downsample source: x's fragments of this wave's pixels (k-block kb = channels 32 kb .. +31)
    u32x4 bx[2][4];
    if (DS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xo = x0 + j * 16 + lrow;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const T* src = xo < N ? xds + (row_px + j * 16 + lrow) * C + kb * 32 + lchunk * 8 : reinterpret_cast<const T*>(l_zero16);
                bx[kb][j] = *reinterpret_cast<const u32x4*>(src);
            }
        }
    }
    // acc (stage 1) is dead: the same registers hold the next conv1's accumulators
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_res = [&](int p, u32x4 (&r)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xo = x0 + j * 16 + lrow;
            const T* src = xo < N ? res + (row_px + j * 16 + lrow) * CO + p * 32 + lchunk * 8 : reinterpret_cast<const T*>(l_zero16);
            r[j] = *reinterpret_cast<const u32x4*>(src);
        }
    };
    // residual pieces are requested TWO passes ahead, the downsample filter's fragments one pass ahead (a pass
    // is ~0.3 us of MFMA work, an HBM / L2 round trip several times that)
    u32x4 r[4], r1[4];
    auto load_wd = [&](int p, u32x4 (&a)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                a[rb][kb] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(wdf) +
                                                            (((p * 2 + rb) * 2 + kb) * 64 + lane) * 16);
    };
    u32x4 ad[2][2];
    if (!DS) {
        load_res(0, r);
        load_res(1, r1);
    } else {
        load_wd(0, ad);
    }
    // biases through LDS (the 4 KiB behind the fragments): one ds_read per pass instead of dependent global loads
    float* bias_s = reinterpret_cast<float*>(lds + W3_BYTES + W1_BYTES);
    bias_s[tid] = bias3[tid];
    if (NEXT && tid < C) bias_s[CO + tid] = bias1 ? bias1[tid] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the fragment DMAs (and the first residual pieces)
    __syncthreads();
    L1_STAMP(3)

    // ---- stage 3: conv3 in 8 passes of 32 channels (+ residual | downsample) -> out, chained next conv1
    const unsigned char* W3s = lds;
    const unsigned char* W1s = lds + W3_BYTES;
    // (the first block's form - downsample MFMAs instead of residual loads - runs its eight passes unrolled: 307 -> 286 us; the
    //  identity form measured 362 rolled / 367 unrolled and stays rolled: tools/static_ab.sh)
    constexpr int S3_UNROLL = (DS && NV == 56) ? 8 : 1;         // (128x128 faces: the unrolled form spills in fp16)
#pragma unroll S3_UNROLL
    for (int p = 0; p < 8; ++p) {
        u32x4 r2[4], adn[2][2];
        if (!DS && p < 6) load_res(p + 2, r2);
        if (DS && p < 7) load_wd(p + 1, adn);
        u32x4 a3[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                a3[rb][kb] = *reinterpret_cast<const u32x4*>(W3s + (((p * 2 + rb) * 2 + kb) * 64 + lane) * 16);
        const int n = p * 32 + lchunk * 8;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + n), b1v = *reinterpret_cast<const f32x4*>(bias_s + n + 4);
        f32x4 c3[2][4];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) c3[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma<T>(c3[rb][j], a3[rb][kb], bt[kb][j]);
        if (DS) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma<T>(c3[rb][j], ad[rb][kb], bx[kb][j]);
        }
        u32x4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = c3[0][j][e] + b0[e];
                v[4 + e] = c3[1][j][e] + b1v[e];
            }
            if (!DS) {
                float rv[8];
                unpack8(r[j], rv, T());
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            o[j] = pack8(v, T());
            const int xo = x0 + j * 16 + lrow;
            if (xo < N) *reinterpret_cast<u32x4*>(out + (row_px + j * 16 + lrow) * CO + n) = o[j];
        }
        if (NEXT) {     // out's channels 32p .. 32p+31 = k-block p of the next conv1
            u32x4 a1[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                a1[rb] = *reinterpret_cast<const u32x4*>(W1s + ((p * 4 + rb) * 64 + lane) * 16);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma<T>(acc[rb][j], a1[rb], o[j]);
        }
        if (!DS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[j] = r1[j];
                r1[j] = r2[j];
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) ad[rb][kb] = adn[rb][kb];
        }
        L1_STAMP(4 + p)
    }
    if (NEXT) {
        T* orow = out_next + row_px * C;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int n = pr * 32 + lchunk * 8;
            float bb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bb[e] = bias_s[CO + n + e];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xo = x0 + j * 16 + lrow;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[2 * pr][j][e] + bb[e], 0.f);
                    v[4 + e] = fmaxf(acc[2 * pr + 1][j][e] + bb[4 + e], 0.f);
                }
                if (xo < N) *reinterpret_cast<u32x4*>(orow + (size_t)(j * 16 + lrow) * C + n) = pack8(v, T());
            }
        }
    }
    L1_STAMP(12)
}

// The last block of layer1 (identity residual) with the WIDE chained conv1 (256 -> 128 = layer2.0's conv1): stages 1-2
// as l1block_kernel; stage 3 twice over half of the wave's pixel blocks (see the header).
template <typename T, int NV, int BANDV>
__global__ __launch_bounds__(256, 2) void l1block_wide_kernel(const T* __restrict__ x, const T* __restrict__ wpk2,
                                                              const float* __restrict__ bias2, const T* __restrict__ w3f,
                                                              const float* __restrict__ bias3, const T* __restrict__ res,
                                                              T* __restrict__ out, const T* __restrict__ w1f,
                                                              const float* __restrict__ bias1, T* __restrict__ out_next,
                                                              int reverse) {
    typedef L1Geom<NV, BANDV> G;
    constexpr int N = G::N, NP = G::NP, BAND = G::BAND, WPR = G::WPR, PATCH_PX = G::PATCH_PX, PATCH_INST = G::PATCH_INST;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS_BYTES_W];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)lds;
    const int tile = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x, img = tile / (N / BAND), band = tile - img * (N / BAND);
    const int grp = img / 6, f = img - grp * 6;
    const CubePadGeom geom{N, 1, 1, 1, 1};
    const int lrow = lane & 15, lchunk = lane >> 4;
    const int wrow = wave / WPR, x0 = (wave - wrow * WPR) * 64;
    const size_t row_px = ((size_t)img * N + band * BAND + wrow) * N + x0;

    // ---- stage 1 (as l1block_kernel)
    {
        const T* xg = x + (size_t)grp * 6 * N * N * C;
#pragma unroll 1
        for (int inst = wave; inst < PATCH_INST; inst += 4) {
            const int q = inst * 8 + (lane >> 3);
            const void* src = l_zero16;
            if (q < PATCH_PX) {
                const int pr = q / NP, pc = q - pr * NP;
                const int sp = cubepad_src(f, BAND * band + pr, pc, geom);
                src = xg + (size_t)sp * C + (((lane & 7) ^ px_swz(q)) << 3);
            }
            glds16(src, __builtin_amdgcn_readfirstlane(lds_base + inst * 1024));
        }
    }
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk2);
    auto load_a = [&](int t, u32x4 (&a)[4][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                a[i][kk] = *reinterpret_cast<const u32x4*>(wb + (size_t)t * W2_TAP + ((i * 2 + kk) * 64 + lane) * 16);
    };
    constexpr int DEPTH = 2;
    u32x4 bt[2][4];
    {
        u32x4 aq[DEPTH + 1][4][2];
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) load_a(t, aq[t]);
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + DEPTH < 9) load_a(tap + DEPTH, aq[(tap + DEPTH) % (DEPTH + 1)]);
            const int ky = tap / 3, kx = tap - ky * 3;
            const int pbase = (wrow + ky) * NP + kx + x0 + lrow;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = pbase + 16 * j;
                    b[j] = *reinterpret_cast<const u32x4*>(lds + p * 128 + (((kk * 4 + lchunk) ^ px_swz(p)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma<T>(acc[i][j], aq[tap % (DEPTH + 1)][i][kk], b[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // every wave is done with the patch: bring the chained conv1's fragments in (64 KiB)
        __syncthreads();
        {
            const unsigned char* s1 = reinterpret_cast<const unsigned char*>(w1f);
#pragma unroll
            for (int q = 0; q < W1W_BYTES / 1024 / 4; ++q) {
                const int inst = q * 4 + wave;
                glds16(s1 + inst * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds_base + inst * 1024));
            }
        }
        // ---- stage 2: t = relu(conv2 + b2), rounded once, as B fragments bt[k-block][pixel block]
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int n = pr * 32 + lchunk * 8;
            float bb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bb[e] = bias2 ? bias2[n + e] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[2 * pr][j][e] + bb[e], 0.f);
                    v[4 + e] = fmaxf(acc[2 * pr + 1][j][e] + bb[4 + e], 0.f);
                }
                bt[pr][j] = pack8(v, T());
            }
        }
    }
    float* bias_s = reinterpret_cast<float*>(lds + W1W_BYTES);
    bias_s[tid] = bias3[tid];
    if (tid < C1W) bias_s[CO + tid] = bias1 ? bias1[tid] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the fragment DMAs
    __syncthreads();

    // ---- stage 3, per half h of the wave's pixel blocks (j = 2h, 2h + 1): conv3 in 8 passes of 32 channels + residual
    // -> out, and the chained conv1's 128 channels accumulated on the fly (8 row blocks x 2 pixel blocks)
    const unsigned char* W1s = lds;
    const unsigned char* w3g = reinterpret_cast<const unsigned char*>(w3f);
    auto load_w3 = [&](int p, u32x4 (&a)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                a[rb][kb] = *reinterpret_cast<const u32x4*>(w3g + (((p * 2 + rb) * 2 + kb) * 64 + lane) * 16);
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        auto load_res = [&](int p, u32x4 (&r)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * h + jj;
                const int xo = x0 + j * 16 + lrow;
                const T* src = xo < N ? res + (row_px + j * 16 + lrow) * CO + p * 32 + lchunk * 8 : reinterpret_cast<const T*>(l_zero16);
                r[jj] = *reinterpret_cast<const u32x4*>(src);
            }
        };
        f32x4 acc[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 r[2], r1[2], a3[2][2];
        load_res(0, r);
        load_res(1, r1);
        load_w3(0, a3);
#pragma unroll 1
        for (int p = 0; p < 8; ++p) {
            u32x4 r2[2], a3n[2][2];
            if (p < 6) load_res(p + 2, r2);
            if (p < 7) load_w3(p + 1, a3n);
            const int n = p * 32 + lchunk * 8;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + n), b1v = *reinterpret_cast<const f32x4*>(bias_s + n + 4);
            f32x4 c3[2][2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) c3[rb][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) mma<T>(c3[rb][jj], a3[rb][kb], bt[kb][2 * h + jj]);
            u32x4 o[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * h + jj;
                float v[8], rv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = c3[0][jj][e] + b0[e];
                    v[4 + e] = c3[1][jj][e] + b1v[e];
                }
                unpack8(r[jj], rv, T());
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + rv[e], 0.f);
                o[jj] = pack8(v, T());
                const int xo = x0 + j * 16 + lrow;
                if (xo < N) *reinterpret_cast<u32x4*>(out + (row_px + j * 16 + lrow) * CO + n) = o[jj];
            }
            // out's channels 32p .. 32p+31 = k-block p of the chained conv1 (fragment order 1: (kb * 8 + rb))
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                u32x4 a1[4];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
                    a1[rb] = *reinterpret_cast<const u32x4*>(W1s + ((p * 8 + half * 4 + rb) * 64 + lane) * 16);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) mma<T>(acc[half * 4 + rb][jj], a1[rb], o[jj]);
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                r[jj] = r1[jj];
                r1[jj] = r2[jj];
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) a3[rb][kb] = a3n[rb][kb];
        }
        T* orow = out_next + row_px * C1W;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int n = pr * 32 + lchunk * 8;
            float bb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bb[e] = bias_s[CO + n + e];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * h + jj;
                const int xo = x0 + j * 16 + lrow;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[2 * pr][jj][e] + bb[e], 0.f);
                    v[4 + e] = fmaxf(acc[2 * pr + 1][jj][e] + bb[4 + e], 0.f);
                }
                if (xo < N) *reinterpret_cast<u32x4*>(orow + (size_t)(j * 16 + lrow) * C1W + n) = pack8(v, T());
            }
        }
    }
}

extern "C" size_t cp360_frag_packed_bytes(int dtype, int n_out, int k) {
    if ((dtype != CP360_BF16 && dtype != CP360_F16) || n_out <= 0 || k <= 0 || n_out % 32 != 0 || k % 32 != 0) return 0;
    return (size_t)n_out * k * 2;
}

extern "C" int cp360_frag_pack_1x1(int dtype, const float* w, const float* scale, void* packed, int n_out, int k,
                                   int order, void* stream) {
    if (!w || !packed) return CP360_ERR_NULL;
    if (n_out <= 0 || k <= 0 || n_out % 32 != 0 || k % 32 != 0 || (order != 0 && order != 1)) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((n_out * k + 255) / 256);
    if (dtype == CP360_BF16)
        hipLaunchKernelGGL((frag_pack_kernel<bf16_raw>), dim3(blocks), dim3(256), 0, st, w, scale, (bf16_raw*)packed, n_out, k, order);
    else if (dtype == CP360_F16)
        hipLaunchKernelGGL((frag_pack_kernel<f16_raw>), dim3(blocks), dim3(256), 0, st, w, scale, (f16_raw*)packed, n_out, k, order);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" size_t cp360_l1block_conv2_bytes(int dtype) {
    return (dtype == CP360_BF16 || dtype == CP360_F16) ? (size_t)9 * W2_TAP : 0;
}

extern "C" int cp360_l1block_pack_conv2(int dtype, const float* w_oihw, const float* scale, void* packed, void* stream) {
    if (!w_oihw || !packed) return CP360_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (9 * C * C + 255) / 256;
    if (dtype == CP360_BF16)
        hipLaunchKernelGGL((l1_pack_conv2_kernel<bf16_raw>), dim3(blocks), dim3(256), 0, st, w_oihw, scale, (bf16_raw*)packed);
    else if (dtype == CP360_F16)
        hipLaunchKernelGGL((l1_pack_conv2_kernel<f16_raw>), dim3(blocks), dim3(256), 0, st, w_oihw, scale, (f16_raw*)packed);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_l1block_forward_wide(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                                          const void* w3_frags, const float* bias3, const void* residual, void* out,
                                          const void* w1_frags, const float* bias1, void* out_next, int n_img, int face,
                                          void* stream) {
    if (!mid || !w2_packed || !w3_frags || !bias3 || !out || !residual || !w1_frags || !out_next) return CP360_ERR_NULL;
    if (n_img <= 0) return CP360_ERR_BAD_SHAPE;
    if (n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (face != 56 && face != 128) return CP360_ERR_UNSUPPORTED;
    if ((long long)n_img * face * face * CO >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
#define CP360_L1W(TT, NV, BV)                                                                                       \
    hipLaunchKernelGGL((l1block_wide_kernel<TT, NV, BV>), dim3((unsigned)(n_img * (NV / BV))), dim3(256), 0, st,      \
                       (const TT*)mid, (const TT*)w2_packed, bias2, (const TT*)w3_frags, bias3, (const TT*)residual, \
                       (TT*)out, (const TT*)w1_frags, bias1, (TT*)out_next, cp360_launch_reverse())
#define CP360_L1W_T(TT) { if (face == 56) CP360_L1W(TT, 56, 4); else CP360_L1W(TT, 128, 2); }
    if (dtype == CP360_BF16) CP360_L1W_T(bf16_raw)
    else if (dtype == CP360_F16) CP360_L1W_T(f16_raw)
    else return CP360_ERR_BAD_DTYPE;
#undef CP360_L1W_T
#undef CP360_L1W
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_l1block_forward(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                                     const void* w3_frags, const float* bias3, const void* residual, const void* x_ds,
                                     const void* wd_frags, void* out, const void* w1_frags, const float* bias1,
                                     void* out_next, int n_img, int face, void* stream) {
    if (!mid || !w2_packed || !w3_frags || !bias3 || !out) return CP360_ERR_NULL;
    if ((residual != nullptr) == (x_ds != nullptr)) return CP360_ERR_NULL;        // exactly one of the two
    if (x_ds && !wd_frags) return CP360_ERR_NULL;
    if ((w1_frags != nullptr) != (out_next != nullptr)) return CP360_ERR_NULL;
    if (n_img <= 0) return CP360_ERR_BAD_SHAPE;
    if (n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (face != 56 && face != 128) return CP360_ERR_UNSUPPORTED;                  // other sizes: the per-convolution path
    if ((long long)n_img * face * face * CO >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
#define CP360_L1B(TT, DSV, NX, NV, BV)                                                                          \
    hipLaunchKernelGGL((l1block_kernel<TT, DSV, NX, NV, BV>), dim3((unsigned)(n_img * (NV / BV))), dim3(256), 0, st, \
                       (const TT*)mid, (const TT*)w2_packed, bias2, (const TT*)w3_frags, bias3, (const TT*)residual, \
                       (const TT*)x_ds, (const TT*)wd_frags, (TT*)out, (const TT*)w1_frags, bias1, (TT*)out_next, cp360_launch_reverse())
#define CP360_L1B_F(TT, NV, BV)                                 \
    {                                                           \
        if (x_ds && w1_frags) CP360_L1B(TT, true, true, NV, BV);  \
        else if (x_ds) CP360_L1B(TT, true, false, NV, BV);        \
        else if (w1_frags) CP360_L1B(TT, false, true, NV, BV);    \
        else CP360_L1B(TT, false, false, NV, BV);                 \
    }
#define CP360_L1B_T(TT) { if (face == 56) CP360_L1B_F(TT, 56, 4) else CP360_L1B_F(TT, 128, 2) }
    if (dtype == CP360_BF16) CP360_L1B_T(bf16_raw)
    else if (dtype == CP360_F16) CP360_L1B_T(f16_raw)
    else return CP360_ERR_BAD_DTYPE;
#undef CP360_L1B_T
#undef CP360_L1B_F
#undef CP360_L1B
    CP360_CHECK_HIP();
    return CP360_OK;
}

// layer1.0 with its own conv1 inside (FIRST): x = the block input [n_img, 56, 56, 64] (conv1's input AND the downsample source),
// w0_frags = cp360_frag_pack_1x1(w1 [64, 64], order 0) with bn1 folded, bias0 f32 [64]; the other arguments as cp360_l1block_forward's
// downsample form.  56x56 faces only (cube 224).
extern "C" int cp360_l1block_forward_first(int dtype, const void* x, const void* w0_frags, const float* bias0, const void* w2_packed,
                                           const float* bias2, const void* w3_frags, const float* bias3, const void* wd_frags, void* out,
                                           const void* w1_frags, const float* bias1, void* out_next, int n_img, int face, void* stream) {
    if (!x || !w0_frags || !w2_packed || !w3_frags || !bias3 || !wd_frags || !out) return CP360_ERR_NULL;
    if ((w1_frags != nullptr) != (out_next != nullptr)) return CP360_ERR_NULL;
    if (n_img <= 0) return CP360_ERR_BAD_SHAPE;
    if (n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (face != 56) return CP360_ERR_UNSUPPORTED;                                 // other sizes: conv1 as its own launch + cp360_l1block_forward
    if ((long long)n_img * face * face * CO >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
#define CP360_L1F(TT, NX)                                                                                        \
    hipLaunchKernelGGL((l1block_kernel<TT, true, NX, 56, 4, true>), dim3((unsigned)(n_img * 14)), dim3(256), 0, st,  \
                       (const TT*)x, (const TT*)w2_packed, bias2, (const TT*)w3_frags, bias3, (const TT*)nullptr,   \
                       (const TT*)x, (const TT*)wd_frags, (TT*)out, (const TT*)w1_frags, bias1, (TT*)out_next,      \
                       cp360_launch_reverse(), (const TT*)w0_frags, bias0)
    if (dtype == CP360_BF16) { if (w1_frags) CP360_L1F(bf16_raw, true); else CP360_L1F(bf16_raw, false); }
    else if (dtype == CP360_F16) { if (w1_frags) CP360_L1F(f16_raw, true); else CP360_L1F(f16_raw, false); }
    else return CP360_ERR_BAD_DTYPE;
#undef CP360_L1F
    CP360_CHECK_HIP();
    return CP360_OK;
}

#ifdef L1_STAMPS
extern "C" int cp360_l1_stamps_read(unsigned long long* host) {      // diagnostic build only
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_l1_stamps), sizeof(g_l1_stamps)) == hipSuccess ? 0 : CP360_ERR_HIP;
}
#endif
