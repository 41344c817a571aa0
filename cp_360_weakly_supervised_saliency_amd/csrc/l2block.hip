// K3e: the tail of a layer2 / layer3 identity Bottleneck in ONE kernel (16-bit types):
//
//   mid [.,n,n,C] --CubePad(1)+conv3x3 C->C +bn2+relu--> t --conv1x1 C->4C +bn3 + residual + relu--> out
//
// (conv2 / conv3 / residual add of model/resnet_cubic.py:85-106), for layer2's blocks 1-3 (C = 128; 28x28 faces =
// cube 224, 64x64 = cube 512) and layer3's blocks 1-5 (C = 256; 14x14 faces = cube 224).  The text below describes the
// layer2 geometry; layer3's differences are listed at BtGeom.  Separately these are a
// generic implicit GEMM that re-gathers its im2col rows once per tap (conv2, 0.15 ms per block for 64 frames) and
// an HBM-bound 1x1 (conv3, 0.15 ms) with t making a round trip through HBM in between.  Here, as in l1block.hip:
//   * a workgroup of 4 waves owns a band of 4 output rows (112 pixels = 7 MFMA pixel blocks, no padding columns: a
//     band's pixels are consecutive in memory), two workgroups per CU (NB = 2 - one workgroup of 8 waves with two
//     bands - exists as an A/B variant and measured slower);
//   * stage 1 (conv2): each band's 6 x 30 cube-padded pixels (256 B each, 45 KB) are gathered ONCE by LDS-DMA through
//     cubepad_src(); the nine taps read them there (the 16-byte chunk c of patch pixel (row, col) sits at chunk
//     c ^ ((row * N + col) & 15): the 16 pixels of an MFMA block are consecutive OUTPUT pixels, so this key runs through
//     16 consecutive values for every tap - also where a block wraps from one face row to the next, where the patch
//     index jumps by the two padding columns - and the 16 lanes cover the 16 slots of a bank row).  Wave w computes the 32-channel row pair
//     (w & 3) for all 7 pixel blocks of its band; its A fragments (4 KiB per half tap) come from L2 straight into
//     registers, three half taps ahead - no LDS ring and no barrier inside the stage;
//   * stage 2: t = relu(conv2 + b2), rounded once, goes to LDS ([112 px][128 ch], 272-byte pixel stride) - the
//     K reduction of conv3 needs all four waves' channels - where the band's patch was;
//   * stage 3 (conv3): wave w computes the 32-channel pairs w&3, +4, +8, +12 of the 512 outputs (K = 128: 56 MFMAs
//     per pass), A fragments from L2 (fragment order, prefetched one pass ahead), adds the residual piece (16-byte
//     loads prefetched one pass ahead), ReLU, one rounding, 16-byte stores;
//   * NEXT (layer2, 28x28 faces): the next identity block's conv1 (512 -> 128) from the output pieces, through an LDS
//     slice per pass (see the NEXT branch).
// HBM traffic per block (64 frames): mid 77 MB + residual 308 MB + out 308 MB; t (77 MB x 2) never leaves the CU.
// Work-item order: cp360_set_launch_order (descending band order when `reverse`).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {
constexpr int W_STEP = 16 * 1024;                            // conv2 weights per step: 4 fragments for each of the 4 waves
// Channels C (mid) -> CO = 4 C (out), face size N, BAND output rows per band (4 waves), NB bands per workgroup.
//   layer2 (C = 128): 28x28 faces -> 4 rows = 112 pixels = 7 pixel blocks; 64x64 faces (cube 512, BASELINE config C5) ->
//     2 rows = 128 pixels = 8 blocks.  A wave owns 32 conv2 channels (2 row blocks); a step = half a tap (K = 64).
//   layer3 (C = 256): 32x32 faces (cube 512) -> 2 rows = 64 pixels = 4 blocks; 14x14 faces -> 7 rows = 98 pixels in 7 blocks (the last 14 pixel slots are padding: they read
//     pixel 97 again and store nothing).  A wave owns 2 x 32 conv2 channels, computed one after the other (HC = 2
//     channel halves: the accumulators of 64 channels x 7 blocks do not fit next to the fragment prefetch); the first
//     half's rounded t pieces wait in registers, because the t tile replaces the patch the second half still reads;
//     36 steps per half; conv3's K = 256 runs as two half passes of 4 k-blocks; pixels are 512 bytes.
template <int C_, int N_, int BAND_, int NB> struct BtGeom {
    static constexpr int C = C_, CO = 4 * C_, N = N_, BAND = BAND_, NP = N + 2;
    static constexpr int PXB = C * 2;                        // bytes per pixel
    static constexpr int CH16 = PXB / 16;                    // 16-byte chunks per pixel (16 / 32)
    static constexpr int PPI = 1024 / PXB;                   // pixels per DMA instruction (4 / 2)
    static constexpr int PXV = BAND * N;                     // output pixels per band (consecutive in memory)
    static constexpr int PB = (PXV + 15) / 16, PX = PB * 16;
    static constexpr int PATCH_PX = (BAND + 2) * NP;         // 180 / 264 / 144
    static constexpr int PATCH_INST = PATCH_PX / PPI;
    static constexpr int PATCH_LDS = PATCH_INST * 1024;      // 46,080 / 67,584 / 73,728
    static constexpr int T_STRIDE = PXB + 16;                // pixel stride of the t tile (272 / 528)
    static constexpr int RBW = 2, KK = 2;                    // conv2: row blocks per wave and channel half, k-blocks per step
    static constexpr int HC = C / 128;                       // channel halves (1 / 2)
    static constexpr int SPT = C / 64;                       // steps per tap (2 / 4)
    static constexpr int STEPS = 9 * SPT;                    // per channel half: 18 / 36
    static constexpr int KB3 = C / 32, H3 = KB3 / 4;         // conv3: k-blocks, half passes per pass (1 / 2)
    static constexpr int PASSES = CO / 128;                  // conv3 passes of 32 channels per wave (4 / 8)
    static constexpr int SLICE_LDS = ((PX * T_STRIDE + 1023) / 1024) * 1024;   // one 128-channel slice of `out` (NEXT variant)
    static constexpr int OFF_SLICE = NB * PATCH_LDS;              // behind the patches / t tiles
    static constexpr int OFF_BIAS_NEXT = OFF_SLICE + NB * SLICE_LDS;
    static constexpr int OFF_BIAS = NB * PATCH_LDS;
    static constexpr int LDS_BYTES = OFF_BIAS + (C + CO) * 4;
    static constexpr int LDS_BYTES_NEXT = OFF_BIAS_NEXT + (C + CO + C) * 4;
    static_assert(PATCH_PX % PPI == 0 && N % BAND == 0 && PX * T_STRIDE <= PATCH_LDS && KB3 % 4 == 0,
                  "band geometry");
};

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
__host__ __device__ __forceinline__ int row_chan(int R) { return (R & ~31) + ((R >> 2) & 3) * 8 + ((R >> 4) & 1) * 4 + (R & 3); }
}  // namespace

// conv2 weights [C, C, 3, 3] (times scale) -> MFMA A fragments [channel half hc][step s = tap * SPT + sub][wave 4][i 2][kk 2]
// [lane][8]: row block (hc * 4 + wave) * 2 + i, lane l holds row (l & 15) of it (rows in acc_chan order), channels
// (sub * 2 + kk) * 32 + (l>>4)*8 ..+7 of the tap
template <typename T, int C>
__global__ __launch_bounds__(256) void bt_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                      T* __restrict__ packed) {
    constexpr int SPT = C / 64, STEPS = 9 * SPT;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 9 * C * C) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, kk = (idx >> 9) & 1, i = (idx >> 10) & 1, wv = (idx >> 11) & 3;
    const int rest = idx >> 13;
    const int st = rest % STEPS, hc = rest / STEPS;
    const int tap = st / SPT, sub = st - tap * SPT;
    const int n = row_chan(((hc * 4 + wv) * 2 + i) * 16 + (lane & 15));
    const int c = (sub * 2 + kk) * 32 + (lane >> 4) * 8 + e;
    const float v = w[((size_t)n * C + c) * 9 + tap] * (scale ? scale[n] : 1.f);
    if constexpr (__is_same(T, f16_raw)) packed[idx] = (f16_raw)v;
    else packed[idx] = f32_to_bf16(v);
}

// Diagnostic build only (-DL2_STAMPS, tools/l2_stamps.sh): s_memtime stamps of wave 0 of every workgroup at the phase boundaries (0 start,
// 1 patch landed + barrier, 2 + hc conv2 of channel half hc done, 4 t tile complete + barrier, 5 + q after pass q of stage 3, 15 end); the
// product build executes no stamp.
#ifdef L2_STAMPS
__device__ unsigned long long g_l2_stamps[8192 * 16];
#define L2_STAMP(k)                                                                                         \
    { __builtin_amdgcn_sched_barrier(0);                                                                    \
      if (wave == 0 && lane == 0 && blockIdx.x < 8192) g_l2_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
      __builtin_amdgcn_sched_barrier(0); }
#else
#define L2_STAMP(k)
#endif

template <typename T, int CV, int NB, int NV, int BANDV, bool NEXT>
__global__ __launch_bounds__(256 * NB, 2) void l2block_kernel(const T* __restrict__ x, const T* __restrict__ wpk2,
                                                         const float* __restrict__ bias2, const T* __restrict__ w3f,
                                                         const float* __restrict__ bias3, const T* __restrict__ res,
                                                         T* __restrict__ out, const T* __restrict__ w1f,
                                                         const float* __restrict__ bias1, T* __restrict__ out_next,
                                                         int reverse) {
    typedef BtGeom<CV, NV, BANDV, NB> G;
    constexpr int C = G::C, CO = G::CO, N = G::N, NP = G::NP, BAND = G::BAND, PXV = G::PXV, PB = G::PB,
                  PATCH_INST = G::PATCH_INST, PATCH_LDS = G::PATCH_LDS, T_STRIDE = G::T_STRIDE, PXB = G::PXB, RBW = G::RBW,
                  KK = G::KK, HC = G::HC, SPT = G::SPT, STEPS = G::STEPS, KB3 = G::KB3, H3 = G::H3, PASSES = G::PASSES,
                  OFF_BIAS = NEXT ? G::OFF_BIAS_NEXT : G::OFF_BIAS;
    static_assert(!NEXT || CV == 128, "the chained conv1 exists for layer2");
#ifndef L2_LDS_PAD
#define L2_LDS_PAD 0                                               // (diagnostic builds: > 0 pushes the launch to ONE workgroup per CU)
#endif
    __shared__ __attribute__((aligned(1024))) unsigned char lds[(NEXT ? G::LDS_BYTES_NEXT : G::LDS_BYTES) + L2_LDS_PAD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half_wg = wave >> 2, w4 = wave & 3;              // band A / B of this workgroup, wave inside the band
    const unsigned lds_base = (unsigned)(size_t)lds;
    const int gband = (reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * NB + half_wg;   // global band = img * 7 + band (descending: cp360_set_launch_order)
    const int img = gband / (N / BAND), band = gband - img * (N / BAND);
    const int grp = img / 6, f = img - grp * 6;
    const CubePadGeom geom{N, 1, 1, 1, 1};
    const int lrow = lane & 15, lchunk = lane >> 4;
    const size_t pix0 = (size_t)gband * PXV;                   // the band's first pixel (pixels of a band are consecutive)
    unsigned char* patch = lds + half_wg * PATCH_LDS;
    L2_STAMP(0)

    // ---- stage 1: gather the band's padded pixels: instruction i = patch pixels PPI i .. PPI i + PPI - 1 (4 x 256 B / 2 x 512 B)
    {
        const T* xg = x + (size_t)grp * 6 * N * N * C;
#pragma unroll 1
        for (int inst = w4; inst < PATCH_INST; inst += 4) {
            const int q = inst * G::PPI + lane / G::CH16;
            const int pr = q / NP, pc = q - pr * NP;
            const int sp = cubepad_src(f, BAND * band + pr, pc, geom);
            const T* src = xg + (size_t)sp * C + (((lane & (G::CH16 - 1)) ^ ((pr * N + pc) & 15)) << 3);   // swizzle key: see the B reads
            glds16(src, __builtin_amdgcn_readfirstlane(lds_base + half_wg * PATCH_LDS + inst * 1024));
        }
    }
    // conv2's A fragments come straight from L2 into registers, three steps ahead (a step = half a tap = 4
    // fragments of this wave's row pair): ~96 KiB in flight per CU and no barrier in the whole stage - a ring
    // in LDS (3 x 16 KiB, 2 steps ahead) left the stage waiting on L2 latency at every step (0.25 -> 0.17 ms)
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk2);
    constexpr int DEPTH = 3;
    int pbase[PB];                                             // patch pixel (tap 0, 0) of this lane's pixel in block j
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int pl = min(j * 16 + lrow, PXV - 1), r = pl / N;      // (padding slots of a ragged band read its last pixel)
        pbase[j] = r * NP + (pl - r * N);
    }
    f32x4 acc[RBW][PB];
    u32x4 tkeep[HC > 1 ? PB : 1];                              // first channel half's t pieces (HC = 2)
    float* bias_s = reinterpret_cast<float*>(lds + OFF_BIAS);
    for (int i = tid; i < C + CO; i += 256 * NB) bias_s[i] = i < C ? (bias2 ? bias2[i] : 0.f) : bias3[i - C];
    if (NEXT) for (int i = tid; i < C; i += 256 * NB) bias_s[C + CO + i] = bias1 ? bias1[i] : 0.f;
#pragma unroll
    for (int hc = 0; hc < HC; ++hc) {
        u32x4 aq[DEPTH + 1][RBW][KK];
        const unsigned char* wh = wb + (size_t)hc * STEPS * W_STEP;
        auto load_a = [&](int s, u32x4 (&a)[RBW][KK]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < RBW; ++i)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
                    a[i][kk] = *reinterpret_cast<const u32x4*>(wh + (size_t)s * W_STEP + (((w4 * RBW + i) * KK + kk) * 64 + lane) * 16);
        };
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) load_a(s, aq[s]);
#pragma unroll
        for (int i = 0; i < RBW; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int JG = HC > 1 ? 4 : PB;                    // (two channel halves: the kept t pieces leave room for 4 B fragments at a time)
        constexpr int NG = (PB + JG - 1) / JG;                 // groups per k-block (1 / 2): KK * NG <= 4 groups per step
        static_assert(KK * NG <= 4, "CP360_BT_STEP unrolls four groups");
        if (hc == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the patch DMAs (and the first fragments)
            __syncthreads();
            L2_STAMP(1)
        }
        // B fragments are read one GROUP ahead of the MFMAs that use them (two register sets; a group = up to JG pixel
        // blocks of one k-block): the LDS latency of a group's reads hides under the previous group's MFMAs instead of in
        // front of its own (round 3, measured first on lfirst.hip: -8 %).  The first group of a step is read in the step.
#define CP360_BT_READ(DST, GI, POFF, KOFF, SUB)                                                                \
        {                                                                                                     \
            constexpr int kk_ = (GI) / NG, j0_ = ((GI) % NG) * JG;                                            \
            _Pragma("unroll") for (int u = 0; u < JG; ++u)                                                    \
                if (j0_ + u < PB) {                                                                           \
                    const int p = pbase[j0_ + u] + (POFF);                                                    \
                    DST[u] = *reinterpret_cast<const u32x4*>(patch + p * PXB + (((((SUB) * KK + kk_) * 4 + lchunk) ^ ((lrow + (KOFF)) & 15)) << 4)); \
                }                                                                                             \
        }
#define CP360_BT_GROUP(GI, S, SLOT, POFF, KOFF, SUB)                                                           \
        if constexpr ((GI) < KK * NG) {                                                                       \
            constexpr int kk_ = (GI) / NG, j0_ = ((GI) % NG) * JG;                                            \
            if constexpr ((GI) + 1 < KK * NG) CP360_BT_READ(bq[((GI) + 1) & 1], (GI) + 1, POFF, KOFF, SUB)    \
            _Pragma("unroll") for (int i = 0; i < RBW; ++i)                                                   \
                _Pragma("unroll") for (int u = 0; u < JG; ++u)                                                \
                    if (j0_ + u < PB) mma<T>(acc[i][j0_ + u], aq[(SLOT) % (DEPTH + 1)][i][kk_], bq[(GI) & 1][u]); \
            __builtin_amdgcn_sched_barrier(0);                                                                \
        }
#define CP360_BT_STEP(S, SLOT, POFF, KOFF, SUB)                                                                \
        {                                                                                                     \
            if ((S) + DEPTH < STEPS) load_a((S) + DEPTH, aq[((SLOT) + DEPTH) % (DEPTH + 1)]);                 \
            u32x4 bq[2][JG];                                                                                  \
            CP360_BT_READ(bq[0], 0, POFF, KOFF, SUB)                                                          \
            CP360_BT_GROUP(0, S, SLOT, POFF, KOFF, SUB)                                                       \
            CP360_BT_GROUP(1, S, SLOT, POFF, KOFF, SUB)                                                       \
            CP360_BT_GROUP(2, S, SLOT, POFF, KOFF, SUB)                                                       \
            CP360_BT_GROUP(3, S, SLOT, POFF, KOFF, SUB)                                                       \
        }
        if constexpr (SPT % (DEPTH + 1) == 0) {                // layer3: 4 steps per tap, the tap loop stays rolled
            // (Round 6: carrying the read-ahead of the B fragments across steps and taps - the last group of a step reading the first group of
            // the next - changed nothing: 167 against 165 us, conv2 of a band alone on its CU 29.2 k instead of 30.3 k cycles per channel half for
            // 16.1 k of MFMA issue.  Alone, a wave's stream is ISSUE-bound (per 64-byte step 28 MFMAs leave 224 issue cycles for 14 LDS reads, 4
            // global loads and ~20 vector instructions); with two workgroups per CU the partner's issue slots fill the matrix pipe to ~90 % in this
            // phase: profiles/r06_l2block_l3_phases*.txt.)
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int poff = ky * NP + kx;
#pragma unroll
                for (int sub = 0; sub < SPT; ++sub) CP360_BT_STEP(tap * SPT + sub, sub, poff, ky * N + kx, sub)
            }
        } else {
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const int tap = s / SPT, sub = s - tap * SPT;
                const int ky = tap / 3, kx = tap - ky * 3;
                CP360_BT_STEP(s, s, ky * NP + kx, ky * N + kx, sub)
            }
        }
#undef CP360_BT_STEP
#undef CP360_BT_GROUP
#undef CP360_BT_READ
        L2_STAMP(2 + hc)
        // ---- stage 2: t = relu(conv2 + b2), rounded once -> the band's t tile (in place of its patch, once every wave
        // is done with the patch: after the LAST channel half)
        if (hc == HC - 1) __syncthreads();
        {
            const int n = (hc * 4 + w4) * 32 + lchunk * 8;
            float bb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bb[e] = bias2 ? bias2[n + e] : 0.f;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[0][j][e] + bb[e], 0.f);
                    v[4 + e] = fmaxf(acc[1][j][e] + bb[4 + e], 0.f);
                }
                const u32x4 o = pack8(v, T());
                if (hc < HC - 1) tkeep[j] = o;
                else *reinterpret_cast<u32x4*>(patch + (j * 16 + lrow) * T_STRIDE + n * 2) = o;
            }
        }
    }
    if constexpr (HC > 1) {
        const int n = w4 * 32 + lchunk * 8;
#pragma unroll
        for (int j = 0; j < PB; ++j) *reinterpret_cast<u32x4*>(patch + (j * 16 + lrow) * T_STRIDE + n * 2) = tkeep[j];
    }
    // ---- stage 3: conv3 (+ residual, ReLU): PASSES passes of 32 output channels per wave, each H3 half passes of 4 k-blocks
    auto load_a3 = [&](int u, u32x4 (&a)[2][4]) __attribute__((always_inline)) {     // unit u = pass * H3 + half
        const int p = w4 + 4 * (u / H3), h = u % H3;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                a[rb][kb] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(w3f) +
                                                            ((size_t)((p * 2 + rb) * KB3 + h * 4 + kb) * 64 + lane) * 16);
    };
    // (stage 3's addresses are derived from an opaque copy of the lane's row: computed here, not hoisted above conv2 and
    // spilled across it)
    int lrow3 = lrow;
    if constexpr (HC > 1) asm volatile("" : "+v"(lrow3));
    int pres[PB];                                              // pixel of block j inside the band (padding slots: the last pixel)
#pragma unroll
    for (int j = 0; j < PB; ++j) pres[j] = min(j * 16 + lrow3, PXV - 1);
    auto load_res = [&](int p, u32x4 (&r)[PB]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PB; ++j)
            r[j] = *reinterpret_cast<const u32x4*>(res + (pix0 + pres[j]) * CO + p * 32 + lchunk * 8);
    };
    u32x4 a3[2][4], r[PB];
    load_a3(0, a3);
    load_res(w4, r);                                           // (requested before conv2 instead, next to the patch DMAs: 2-4 % slower; +8 % in l1block)
    __syncthreads();                                           // the t tiles and the biases are complete
    L2_STAMP(4)
    // conv3's accumulators: the first two rows of conv2's (dead by now)
    if constexpr (!NEXT) {
    // (layer3's 14x14 form runs its 16 half passes four at a time: 178 -> 171 us - the rolled loop rotates the prefetched
    //  fragment / residual registers through copies that wait for the loads; the layer2 forms measured the same either way)
    constexpr int S3_UNROLL = (CV == 256 && NV == 14) ? 4 : 1;
#pragma unroll S3_UNROLL
    for (int u = 0; u < PASSES * H3; ++u) {
        const int q = u / H3, h = u - q * H3;
        const int p = w4 + 4 * q;
        u32x4 a3n[2][4], rn[PB];
        if (u + 1 < PASSES * H3) load_a3(u + 1, a3n);
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int j0 = 0; j0 < PB; j0 += 4) {               // pixel blocks in two groups (4 + 3): fewer live fragments
                u32x4 b[4];
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (j0 + v < PB)
                        b[v] = *reinterpret_cast<const u32x4*>(patch + ((j0 + v) * 16 + lrow) * T_STRIDE + ((h * 4 + kb) * 4 + lchunk) * 16);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (j0 + v < PB) mma<T>(acc[rb][j0 + v], a3[rb][kb], b[v]);
                __builtin_amdgcn_sched_barrier(0);             // keep the next group's fragment reads from being hoisted (spills)
            }
        }
        if (h == H3 - 1) {
            if (q + 1 < PASSES) load_res(p + 4, rn);           // lands under the next pass's MFMAs
            const int n = p * 32 + lchunk * 8;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + C + n), b1 = *reinterpret_cast<const f32x4*>(bias_s + C + n + 4);
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                float v[8], rv[8];
                unpack8(r[j], rv, T());
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[0][j][e] + b0[e] + rv[e], 0.f);
                    v[4 + e] = fmaxf(acc[1][j][e] + b1[e] + rv[4 + e], 0.f);
                }
                if (PXV % 16 == 0 || j * 16 + lrow3 < PXV)
                    *reinterpret_cast<u32x4*>(out + (pix0 + j * 16 + lrow3) * CO + n) = pack8(v, T());
            }
            if (q + 1 < PASSES) {
#pragma unroll
                for (int j = 0; j < PB; ++j) r[j] = rn[j];
            }
            L2_STAMP(5 + q)
        }
        if (u + 1 < PASSES * H3) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) a3[rb][kb] = a3n[rb][kb];
        }
    }
    } else {
        // ---- NEXT: the same four passes, and the next block's conv1 (512 -> 128, + bn1 + relu) on the fly.  In pass q the four
        // waves produce `out` channels 128q .. 128q+127 = K slice q of that convolution: the rounded 16-byte pieces also go
        // to an LDS slice ([112 px][128 ch], the t tile's layout), and after a barrier wave w accumulates ITS 32 output
        // channels over the slice (8 A fragments from L2, 28 B-fragment reads, 56 MFMAs).  Registers: no second set of
        // conv3 fragments / residual pieces - both are re-requested right after their last use and land under the
        // 56 MFMAs of the chained convolution.
        unsigned char* slice = lds + G::OFF_SLICE + half_wg * G::SLICE_LDS;
        f32x4 acc1[2][PB];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto load_a1 = [&](int q, u32x4 (&a)[2][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
                    a[rb][kb] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(w1f) +
                                                                ((size_t)((w4 * 2 + rb) * 16 + 4 * q + kb) * 64 + lane) * 16);
        };
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const int p = w4 + 4 * q;
            u32x4 a1[2][4];
            load_a1(q, a1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int j0 = 0; j0 < PB; j0 += 4) {
                    u32x4 b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (j0 + u < PB)
                            b[u] = *reinterpret_cast<const u32x4*>(patch + ((j0 + u) * 16 + lrow) * T_STRIDE + (kb * 4 + lchunk) * 16);
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (j0 + u < PB) mma<T>(acc[rb][j0 + u], a3[rb][kb], b[u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (q < 3) load_a3(q + 1, a3);                        // a3 is dead: the next pass's fragments land under the chained MFMAs
            const int n = p * 32 + lchunk * 8;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + C + n), b1 = *reinterpret_cast<const f32x4*>(bias_s + C + n + 4);
            if (q > 0) __syncthreads();                            // every wave is done reading the previous slice
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                float v[8], rv[8];
                unpack8(r[j], rv, T());
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[0][j][e] + b0[e] + rv[e], 0.f);
                    v[4 + e] = fmaxf(acc[1][j][e] + b1[e] + rv[4 + e], 0.f);
                }
                const u32x4 o = pack8(v, T());
                *reinterpret_cast<u32x4*>(out + (pix0 + j * 16 + lrow3) * CO + n) = o;
                *reinterpret_cast<u32x4*>(slice + (j * 16 + lrow) * T_STRIDE + (w4 * 32 + lchunk * 8) * 2) = o;
            }
            if (q < 3) load_res(p + 4, r);                         // r is dead too
            __syncthreads();                                       // the slice is complete
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int j0 = 0; j0 < PB; j0 += 4) {
                    u32x4 b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (j0 + u < PB)
                            b[u] = *reinterpret_cast<const u32x4*>(slice + ((j0 + u) * 16 + lrow) * T_STRIDE + (kb * 4 + lchunk) * 16);
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (j0 + u < PB) mma<T>(acc1[rb][j0 + u], a1[rb][kb], b[u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // mid' = relu(conv1 + b1), one rounding, 16-byte stores ([px][128 ch])
        const int n1 = w4 * 32 + lchunk * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(bias_s + C + CO + n1), c1 = *reinterpret_cast<const f32x4*>(bias_s + C + CO + n1 + 4);
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaxf(acc1[0][j][e] + c0[e], 0.f);
                v[4 + e] = fmaxf(acc1[1][j][e] + c1[e], 0.f);
            }
            *reinterpret_cast<u32x4*>(out_next + (pix0 + j * 16 + lrow3) * C + n1) = pack8(v, T());
        }
    }
    L2_STAMP(15)
}

#ifdef L2_STAMPS
extern "C" int cp360_l2_stamps_read(unsigned long long* host) {      // diagnostic build only
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_l2_stamps), sizeof(g_l2_stamps)) == hipSuccess ? 0 : CP360_ERR_HIP;
}
#endif

// (Layer3's 14x14 launch with NB = 2 - one 8-wave workgroup per face, the two bands' waves fetching the same conv2 / conv3 fragments so that
// the second request hits the CU's L1 - measured 199.5 us against 165.4: 384 workgroups of 152 KB are 1.5 rounds of one workgroup per CU,
// profiles/r06_l3nb2_ab.log.)
// (Round 6, the 1.5 rounds of layer3's 768 bands on 512 slots: running the last 256 bands as 512 HALF JOBS - two workgroups per band, both gathering the
// band and running conv2, each running half of stage 3's passes, so that every slot gets one whole band and one half job - measured 189 against 170 us:
// the second conv2 costs more than the idle slots of the last round, profiles/r06_l3_halfjobs_ab.log.)
// (A wave-specialised persistent form of this kernel - waves 0-3 gather + conv2 of band i while waves 4-7 run conv3 +
// residual + store of band i-1 from the t tile, two workgroup barriers per band - was built for layer3 and measured
// bit-identical and no faster, 176 vs 175 us: each stage is latency-bound at the waves it has, so halving the waves per
// stage doubles its time.  Stage ablations of the layer3 launch: gather + prologue 38 us, conv2 70 us (69 % of its
// MFMA time), conv3 + residual + store 67 us = 4.6 TB/s; A fragments from hot lines -9 us, conflict-free B reads -5 us.)
static size_t bt_packed_bytes(int dtype, int c) {
    return (dtype == CP360_BF16 || dtype == CP360_F16) ? (size_t)9 * c * c * 2 : 0;
}
template <int C>
static int bt_pack(int dtype, const float* w_oihw, const float* scale, void* packed, void* stream) {
    if (!w_oihw || !packed) return CP360_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (9 * C * C + 255) / 256;
    if (dtype == CP360_BF16)
        hipLaunchKernelGGL((bt_pack_kernel<bf16_raw, C>), dim3(blocks), dim3(256), 0, st, w_oihw, scale, (bf16_raw*)packed);
    else if (dtype == CP360_F16)
        hipLaunchKernelGGL((bt_pack_kernel<f16_raw, C>), dim3(blocks), dim3(256), 0, st, w_oihw, scale, (f16_raw*)packed);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" size_t cp360_l2block_packed_bytes(int dtype) { return bt_packed_bytes(dtype, 128); }
extern "C" size_t cp360_l3block_packed_bytes(int dtype) { return bt_packed_bytes(dtype, 256); }
extern "C" int cp360_l2block_pack_weights(int dtype, const float* w_oihw, const float* scale, void* packed, void* stream) {
    return bt_pack<128>(dtype, w_oihw, scale, packed, stream);
}
extern "C" int cp360_l3block_pack_weights(int dtype, const float* w_oihw, const float* scale, void* packed, void* stream) {
    return bt_pack<256>(dtype, w_oihw, scale, packed, stream);
}

// layer: 2 (C = 128; faces 28 / 64) or 3 (C = 256; faces 14)
static int bt_launch(int layer, int dtype, const void* mid, const void* w2_packed, const float* bias2, const void* w3_frags,
                     const float* bias3, const void* residual, void* out, const void* w1_frags, const float* bias1,
                     void* out_next, int n_img, int face, void* stream) {
    if (!mid || !w2_packed || !w3_frags || !bias3 || !residual || !out) return CP360_ERR_NULL;
    if (n_img <= 0) return CP360_ERR_BAD_SHAPE;
    if (n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (layer == 2 ? (face != 28 && face != 64) : (face != 14 && face != 32)) return CP360_ERR_UNSUPPORTED;
    if (out_next && (layer != 2 || face != 28)) return CP360_ERR_UNSUPPORTED;   // the chained conv1: layer2, 28x28 faces
    const int co = layer == 2 ? 512 : 1024;
    if ((long long)n_img * face * face * co >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    static const int nb_env = []() { const char* e = getenv("CP360_L2_BANDS"); return e ? atoi(e) : 1; }();   // A/B switch
    const int nb = (nb_env == 2 && layer == 2 && face == 28 && !out_next) ? 2 : 1;
#define CP360_L2B(TT, CV, NBV, NV, BV, NX)                                                                           \
    hipLaunchKernelGGL((l2block_kernel<TT, CV, NBV, NV, BV, NX>), dim3((unsigned)(n_img * (NV / BV) / NBV)), dim3(256 * NBV), 0, st, \
                       (const TT*)mid, (const TT*)w2_packed, bias2, (const TT*)w3_frags, bias3, (const TT*)residual, (TT*)out, \
                       (const TT*)w1_frags, bias1, (TT*)out_next, cp360_launch_reverse())
#define CP360_L2B_T(TT)                                                  \
    {                                                                    \
        if (layer == 3 && face == 32) CP360_L2B(TT, 256, 1, 32, 2, false);  \
        else if (layer == 3) CP360_L2B(TT, 256, 1, 14, 7, false);        \
        else if (out_next) CP360_L2B(TT, 128, 1, 28, 4, true);           \
        else if (face == 64) CP360_L2B(TT, 128, 1, 64, 2, false);        \
        else if (nb == 2) CP360_L2B(TT, 128, 2, 28, 4, false);           \
        else CP360_L2B(TT, 128, 1, 28, 4, false);                        \
    }
    if (dtype == CP360_BF16) CP360_L2B_T(bf16_raw)
    else if (dtype == CP360_F16) CP360_L2B_T(f16_raw)
#undef CP360_L2B_T
#undef CP360_L2B
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_l2block_forward(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                                     const void* w3_frags, const float* bias3, const void* residual, void* out,
                                     int n_img, int face, void* stream) {
    return bt_launch(2, dtype, mid, w2_packed, bias2, w3_frags, bias3, residual, out, nullptr, nullptr, nullptr, n_img, face,
                     stream);
}

extern "C" int cp360_l2block_forward_next(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                                          const void* w3_frags, const float* bias3, const void* residual, void* out,
                                          const void* w1_frags, const float* bias1, void* out_next, int n_img, int face,
                                          void* stream) {
    if (!w1_frags || !out_next) return CP360_ERR_NULL;
    return bt_launch(2, dtype, mid, w2_packed, bias2, w3_frags, bias3, residual, out, w1_frags, bias1, out_next, n_img, face,
                     stream);
}

extern "C" int cp360_l3block_forward(int dtype, const void* mid, const void* w2_packed, const float* bias2,
                                     const void* w3_frags, const float* bias3, const void* residual, void* out,
                                     int n_img, int face, void* stream) {
    return bt_launch(3, dtype, mid, w2_packed, bias2, w3_frags, bias3, residual, out, nullptr, nullptr, nullptr, n_img, face,
                     stream);
}
