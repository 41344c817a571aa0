// K3f: the FIRST Bottleneck of layer2 after its conv1, in ONE kernel (16-bit types, cube 224):
//
//   mid [.,56,56,128] --CubePad(1)+conv3x3 STRIDE 2, 128->128 +bn2+relu--> t [.,28,28,128]
//        --conv1x1 128->512 +bn3  +  downsample(x) = conv1x1 stride 2, 256->512 +bn  --add, relu--> out [.,28,28,512]
//
// = conv2 / conv3 / downsample / add of model/resnet_cubic.py:85-106 for layer2.0 (stride on the 3x3, :76-77; the
// downsample reads the UNPADDED block input x [.,56,56,256] at (2 oy, 2 ox), :99-104,145-161).  As separate launches these
// were a generic implicit GEMM that re-gathers its im2col rows per tap (conv2: 171 us for 64 frames, 0.52 PFLOP/s and
// 2.3 TB/s - neither roof) and the second-source ring kernel (conv3 + downsample: 196 us), with t making a round trip
// through HBM.  Here, as l2block.hip:
//   * a workgroup of 4 waves owns a band of 2 output rows (56 pixels = 3.5 MFMA pixel blocks: 8 of the 64 slots are
//     padding), two workgroups per CU;
//   * stage 1 (conv2): the band's 5 x 58 cube-padded INPUT pixels (256 B each, 74 KB) are gathered ONCE by LDS-DMA through
//     cubepad_src(); tap (ky, kx) of output pixel (r, c) reads patch pixel (2r + ky, 2c + kx).  Chunk c of patch pixel
//     (pr, pc) sits at chunk c ^ (((pr >> 1) * 28 + (pc >> 1)) & 15): for a fixed tap that key is the OUTPUT pixel index plus
//     a constant, so the 16 lanes of an MFMA block cover 16 consecutive keys = the 16 slots of a bank row.  A fragments
//     from L2 straight into registers three half taps ahead (the packing of l2block.hip: cp360_l2block_pack_weights);
//   * while conv2 runs, every lane holds its 8 x 16 bytes of the band's downsample input x[:, 2 oy, 2 ox, :] in registers
//     (requested before the first MFMA); after conv2 they go to LDS next to t: one B tile [64 px][128 t + 256 x channels];
//   * stage 3: conv3 and the downsample branch are ONE 1x1 convolution over that tile with K = 384 (fragments of
//     [W3 * s3 | Wd * sd], cp360_l2first_pack_w3d), bias b3 + bd, ReLU, one rounding, 16-byte stores;
//   * NEXT: the next block's conv1 (layer2.1: 1x1, 512 -> 128 + bn1 + relu, resnet_cubic.py:88-90) from the output pieces,
//     as in l2block.hip: in pass q the four waves' rounded pieces (channels 128q .. 128q+127 = K slice q of that
//     convolution) also go to an LDS slice, and after a barrier each wave accumulates its 32 conv1 channels over it (8 A
//     fragments from L2, 32 MFMAs): layer2.1's conv1 launch (96 us, 308 MB re-read) is gone.
// HBM traffic (64 frames): mid 308 MB + x 154 MB (every other pixel and row: whole 512-byte pixels) + out 308 MB.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {
constexpr int C = 128, CX = 256, CO = 512, K3 = C + CX;     // conv2 channels, block-input channels, outputs, stage 3's K
constexpr int N = 28, NI = 56, NPI = NI + 2, BAND = 2;
constexpr int PXB = C * 2;                                   // bytes per patch pixel
constexpr int PATCH_PX = (2 * BAND + 1) * NPI;               // 290
constexpr int PATCH_INST = (PATCH_PX + 3) / 4;               // 73 DMA instructions of 4 pixels
constexpr int PATCH_LDS = PATCH_INST * 1024;                 // 74,752
constexpr int PXV = BAND * N, PB = 4, PX = PB * 16;          // 56 output pixels in 64 slots
constexpr int T_STRIDE = K3 * 2 + 16;                        // 784 B per pixel of the stage-3 tile (49 x 16: conflict-free rows)
constexpr int W_STEP = 16 * 1024;                            // conv2 weights per step (4 fragments for each of the 4 waves)
constexpr int STEPS = 18, KB3 = K3 / 32, H3 = KB3 / 4, PASSES = CO / 128;
constexpr int OFF_BIAS = PATCH_LDS;
constexpr int LDS_BYTES = OFF_BIAS + (C + CO + C) * 4;        // 77,824: two workgroups per CU
constexpr int S_STRIDE = C * 2 + 16;                         // NEXT: pixel stride of the 128-channel slice of `out` (272)
constexpr int OFF_SLICE = PX * T_STRIDE;                     // behind the stage-3 tile, inside the dead patch
static_assert(PX * T_STRIDE + PX * S_STRIDE <= PATCH_LDS, "the stage-3 tile and the slice replace the patch");

__device__ __attribute__((aligned(16))) unsigned int f_zero16[4] = {0u, 0u, 0u, 0u};

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
__host__ __device__ __forceinline__ int row_chan(int R) { return (R & ~31) + ((R >> 2) & 3) * 8 + ((R >> 4) & 1) * 4 + (R & 3); }
}  // namespace

// [W3 * s3 | Wd * sd] (f32 [512, 128] and [512, 256]) -> MFMA A fragments of the K = 384 stage-3 convolution, 1 KiB each
// ([lane][8]: lane l holds row (l & 15), k-group (l >> 4)), rows in acc_chan order, fragment ((p * 2 + rb) * 12 + kb) for the
// 32-row pair p (the order-0 layout of cp360_frag_pack_1x1)
template <typename T>
__global__ __launch_bounds__(256) void lf_pack_w3d_kernel(const float* __restrict__ w3, const float* __restrict__ s3,
                                                          const float* __restrict__ wd, const float* __restrict__ sd,
                                                          T* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= CO * K3) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, frag = idx >> 9;
    const int kb = frag % KB3, rb = frag / KB3;
    const int n = row_chan(rb * 16 + (lane & 15));
    const int kk = kb * 32 + (lane >> 4) * 8 + e;
    const float v = kk < C ? w3[(size_t)n * C + kk] * (s3 ? s3[n] : 1.f) : wd[(size_t)n * CX + (kk - C)] * (sd ? sd[n] : 1.f);
    if constexpr (__is_same(T, f16_raw)) packed[idx] = (f16_raw)v;
    else packed[idx] = f32_to_bf16(v);
}

// (Round 6: a 56-pixel band streams 816 KB of conv2 / conv3 + downsample / next-conv1 fragments from L2 - 4.4 GB per 64-frame launch.  A form with
// TWO bands per 8-wave workgroup, the two wave sets requesting the same fragments at about the same time so that the second request
// could be served by the CU's L1, measured 405 us against 355: the requests do not merge, and one 152 KB workgroup per CU hides less
// than two independent ones.  profiles/r06_l2first_nb_ab.log)
template <typename T, bool NEXT>
__global__ __launch_bounds__(256, 2) void l2first_kernel(const T* __restrict__ mid, const T* __restrict__ wpk2,
                                                         const float* __restrict__ bias2, const T* __restrict__ w3df,
                                                         const float* __restrict__ bias3d, const T* __restrict__ xds,
                                                         T* __restrict__ out, const T* __restrict__ w1f,
                                                         const float* __restrict__ bias1, T* __restrict__ out_next,
                                                         int reverse) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)lds;
    const int gband = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;      // global band = img * 14 + band
    const int img = gband / (N / BAND), band = gband - img * (N / BAND);
    const int grp = img / 6, f = img - grp * 6;
    const CubePadGeom geom{NI, 1, 1, 1, 1};
    const int lrow = lane & 15, lchunk = lane >> 4;
    const size_t pix0 = (size_t)gband * PXV;                                  // the band's first OUTPUT pixel
    unsigned char* patch = lds;

    // ---- stage 1: gather the band's 5 x 58 padded input pixels (instruction i = patch pixels 4i .. 4i+3)
    {
        const T* xg = mid + (size_t)grp * 6 * NI * NI * C;
#pragma unroll 1
        for (int inst = w4; inst < PATCH_INST; inst += 4) {
            const int q = inst * 4 + (lane >> 4);
            const void* src = f_zero16;
            if (q < PATCH_PX) {
                const int pr = q / NPI, pc = q - pr * NPI;
                const int sp = cubepad_src(f, 2 * BAND * band + pr, pc, geom);
                src = xg + (size_t)sp * C + (((lane & 15) ^ (((pr >> 1) * N + (pc >> 1)) & 15)) << 3);
            }
            glds16(src, __builtin_amdgcn_readfirstlane(lds_base + inst * 1024));
        }
    }
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk2);
    constexpr int DEPTH = 3;
    int pbase[PB];                                             // patch pixel of tap (0, 0) of this lane's pixel in block j
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int pl = min(j * 16 + lrow, PXV - 1), r = pl / N;
        pbase[j] = 2 * r * NPI + 2 * (pl - r * N);
    }
    f32x4 acc[2][PB];
    float* bias_s = reinterpret_cast<float*>(lds + OFF_BIAS);
    for (int i = tid; i < C + CO; i += 256) bias_s[i] = i < C ? (bias2 ? bias2[i] : 0.f) : bias3d[i - C];
    if (NEXT && tid < C) bias_s[C + CO + tid] = bias1 ? bias1[tid] : 0.f;
    u32x4 aq[DEPTH + 1][2][2];
    auto load_a = [&](int s, u32x4 (&a)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                a[i][kk] = *reinterpret_cast<const u32x4*>(wb + (size_t)s * W_STEP + (((w4 * 2 + i) * 2 + kk) * 64 + lane) * 16);
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load_a(s, aq[s]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the patch DMAs (and the first fragments)
    __syncthreads();
    // the downsample branch's input: x[img, 2 (2 band + r), 2 c, :] for the band's 56 pixels (64 slots x 512 B = 2048
    // 16-byte pieces, 8 per lane, 32 consecutive lanes = one pixel), requested AFTER the patch has landed (a wait for the patch does not wait for them) and held in registers until conv2 is
    // done with the patch
    u32x4 xr[8];
    {
        const T* xi = xds + (size_t)img * NI * NI * CX;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int u = it * 256 + tid;
            const int pl = min(u >> 5, PXV - 1), r = pl / N, c = pl - r * N;
            xr[it] = *reinterpret_cast<const u32x4*>(xi + ((size_t)(2 * (BAND * band + r)) * NI + 2 * c) * CX + (u & 31) * 8);
        }
    }
    // B fragments are read ONE half step ahead of the MFMAs that use them (two register sets): the LDS latency of a
    // half step's four reads hides under the previous half step's 8 MFMAs instead of in front of its own
    auto read_b = [&](int hs, u32x4 (&b)[PB]) __attribute__((always_inline)) {          // hs = 2 * step + kk
        const int s = hs >> 1, kk = hs & 1, tap = s >> 1, sub = s & 1;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int poff = ky * NPI + kx, koff = (ky >> 1) * N + (kx >> 1);
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int p = pbase[j] + poff;
            b[j] = *reinterpret_cast<const u32x4*>(patch + p * PXB + ((((sub * 2 + kk) * 4 + lchunk) ^ ((lrow + koff) & 15)) << 4));
        }
    };
    u32x4 bq[2][PB];
    read_b(0, bq[0]);
#pragma unroll
    for (int hs = 0; hs < 2 * STEPS; ++hs) {
        const int s = hs >> 1, kk = hs & 1;
        if (kk == 0 && s + DEPTH < STEPS) load_a(s + DEPTH, aq[(s + DEPTH) % (DEPTH + 1)]);
        if (hs + 1 < 2 * STEPS) read_b(hs + 1, bq[(hs + 1) & 1]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) mma<T>(acc[i][j], aq[s % (DEPTH + 1)][i][kk], bq[hs & 1][j]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- stage 2: every wave is done with the patch: the stage-3 tile [64 px][128 t | 256 x] takes its place
    __syncthreads();
    {
        const int n = w4 * 32 + lchunk * 8;
        float bb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[e] = bias_s[n + e];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaxf(acc[0][j][e] + bb[e], 0.f);
                v[4 + e] = fmaxf(acc[1][j][e] + bb[4 + e], 0.f);
            }
            *reinterpret_cast<u32x4*>(patch + (j * 16 + lrow) * T_STRIDE + n * 2) = pack8(v, T());
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int u = it * 256 + tid;
            *reinterpret_cast<u32x4*>(patch + (u >> 5) * T_STRIDE + PXB + (u & 31) * 16) = xr[it];
        }
    }
    // ---- stage 3: out = relu([W3 | Wd] . [t | x] + b3 + bd): 4 passes of 32 output channels per wave, 3 half passes of 4 k-blocks
    auto load_a3 = [&](int u, u32x4 (&a)[2][4]) __attribute__((always_inline)) {      // unit u = pass * H3 + half
        const int p = w4 + 4 * (u / H3), h = u % H3;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                a[rb][kb] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(w3df) +
                                                            ((size_t)((p * 2 + rb) * KB3 + h * 4 + kb) * 64 + lane) * 16);
    };
    u32x4 a3[2][4];
    load_a3(0, a3);
    __syncthreads();                                           // the tile is complete
    // (B fragments one k-block ahead, as in stage 1)
    auto read_b3 = [&](int kbg, u32x4 (&b)[PB]) __attribute__((always_inline)) {          // kbg = k-block 0 .. 11 of the tile
#pragma unroll
        for (int j = 0; j < PB; ++j)
            b[j] = *reinterpret_cast<const u32x4*>(patch + (j * 16 + lrow) * T_STRIDE + (kbg * 4 + lchunk) * 16);
    };
    unsigned char* slice = patch + OFF_SLICE;
    f32x4 acc1[2][PB];
    if (NEXT) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto load_a1 = [&](int q, u32x4 (&a)[2][4]) __attribute__((always_inline)) {       // w1 [128, 512], order 0: K slice q
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                a[rb][kb] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(w1f) +
                                                            ((size_t)((w4 * 2 + rb) * 16 + 4 * q + kb) * 64 + lane) * 16);
    };
#pragma unroll 1
    for (int q = 0; q < PASSES; ++q) {
        const int p = w4 + 4 * q;
        u32x4 a1[2][4];
        if (NEXT) load_a1(q, a1);                              // used behind this pass's barrier
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 b3[2][PB];
        read_b3(0, b3[0]);
#pragma unroll
        for (int h = 0; h < H3; ++h) {
            u32x4 a3n[2][4];
            const int u = q * H3 + h;
            if (u + 1 < PASSES * H3) load_a3(u + 1, a3n);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int kbg = h * 4 + kb;
                if (kbg + 1 < KB3) read_b3(kbg + 1, b3[(kbg + 1) & 1]);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int j = 0; j < PB; ++j) mma<T>(acc[rb][j], a3[rb][kb], b3[kbg & 1][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (u + 1 < PASSES * H3) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) a3[rb][kb] = a3n[rb][kb];
            }
        }
        const int n = p * 32 + lchunk * 8;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + C + n), b1 = *reinterpret_cast<const f32x4*>(bias_s + C + n + 4);
        if (NEXT && q > 0) __syncthreads();                    // every wave is done reading the previous slice
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaxf(acc[0][j][e] + b0[e], 0.f);
                v[4 + e] = fmaxf(acc[1][j][e] + b1[e], 0.f);
            }
            const u32x4 o = pack8(v, T());
            if (j * 16 + lrow < PXV) *reinterpret_cast<u32x4*>(out + (pix0 + j * 16 + lrow) * CO + n) = o;
            if (NEXT) *reinterpret_cast<u32x4*>(slice + (j * 16 + lrow) * S_STRIDE + (w4 * 32 + lchunk * 8) * 2) = o;
        }
        if (NEXT) {
            __syncthreads();                                   // the slice is complete
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                u32x4 b[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    b[j] = *reinterpret_cast<const u32x4*>(slice + (j * 16 + lrow) * S_STRIDE + (kb * 4 + lchunk) * 16);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int j = 0; j < PB; ++j) mma<T>(acc1[rb][j], a1[rb][kb], b[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (NEXT) {     // mid' = relu(conv1 + b1), one rounding, 16-byte stores ([px][128 ch])
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
            if (j * 16 + lrow < PXV) *reinterpret_cast<u32x4*>(out_next + (pix0 + j * 16 + lrow) * C + n1) = pack8(v, T());
        }
    }
}

extern "C" size_t cp360_l2first_w3d_bytes(int dtype) {
    return (dtype == CP360_BF16 || dtype == CP360_F16) ? (size_t)CO * K3 * 2 : 0;
}

extern "C" int cp360_l2first_pack_w3d(int dtype, const float* w3, const float* scale3, const float* wd, const float* scaled,
                                      void* packed, void* stream) {
    if (!w3 || !wd || !packed) return CP360_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (CO * K3 + 255) / 256;
    if (dtype == CP360_BF16)
        hipLaunchKernelGGL((lf_pack_w3d_kernel<bf16_raw>), dim3(blocks), dim3(256), 0, st, w3, scale3, wd, scaled, (bf16_raw*)packed);
    else if (dtype == CP360_F16)
        hipLaunchKernelGGL((lf_pack_w3d_kernel<f16_raw>), dim3(blocks), dim3(256), 0, st, w3, scale3, wd, scaled, (f16_raw*)packed);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_l2first_forward(int dtype, const void* mid, const void* w2_packed, const float* bias2, const void* w3d_frags,
                                     const float* bias3d, const void* x, void* out, const void* w1_frags, const float* bias1,
                                     void* out_next, int n_img, int face_out, void* stream) {
    if (!mid || !w2_packed || !w3d_frags || !bias3d || !x || !out) return CP360_ERR_NULL;
    if ((w1_frags != nullptr) != (out_next != nullptr)) return CP360_ERR_NULL;
    if (n_img <= 0) return CP360_ERR_BAD_SHAPE;
    if (n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (face_out != N) return CP360_ERR_UNSUPPORTED;                          // other sizes: the per-convolution path
    if ((long long)n_img * NI * NI * CX >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(n_img * (N / BAND)));
    const int rev = cp360_launch_reverse();
#define CP360_LF(TT, NX)                                                                                              \
    hipLaunchKernelGGL((l2first_kernel<TT, NX>), grid, dim3(256), 0, st, (const TT*)mid, (const TT*)w2_packed, bias2,      \
                       (const TT*)w3d_frags, bias3d, (const TT*)x, (TT*)out, (const TT*)w1_frags, bias1, (TT*)out_next, rev)
    if (dtype == CP360_BF16) { if (w1_frags) CP360_LF(bf16_raw, true); else CP360_LF(bf16_raw, false); }
    else if (dtype == CP360_F16) { if (w1_frags) CP360_LF(f16_raw, true); else CP360_LF(f16_raw, false); }
#undef CP360_LF
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}
