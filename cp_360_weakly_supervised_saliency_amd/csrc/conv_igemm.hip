// K3 / K4 / K5: implicit-GEMM convolution on MFMA for gfx950.
//
//   out[m, n] = act( sum_{tap, c} W[n, tap, c] * in[src(m, tap), c] + bias[n] (+ res[m, n]) )
//
// m = (image, oy, ox) output pixel, n = output channel, NHWC activations, weights
// pre-packed [c_out_pad][tap][c_pad] (K contiguous).  Replaces nn.Conv2d + folded
// BatchNorm2d + ReLU + residual of model/resnet_cubic.py:85-106,163-175, the CAM GEMM of
// static_model/class_activation_model.py:70-83 and the 3x3 convs of model/clstm.py:56-64.
// The CubePad(p) in front of every 3x3 conv (cube_pad.py:95-216) is NOT materialised:
// src(m, tap) goes through cubepad_src() while the activation tile is gathered.
//
// Kernels in this file (the launch planner plan_of() picks one per call):
//   conv_igemm_kernel<T,WN,WM>   c_out < 256: 128x128 / 64x256 tile, 4 waves, register-staged (described below)
//   conv_igemm_dma_kernel<T,MJ>  c_out >= 256, small M: 256x128 tile, three LDS-DMA stages
//   conv_igemm_ring_kernel<T,BM> c_out >= 256: 256x256 / 256x304 tile, four 64-byte-K stages, staggered waves
//   conv_igemm_ring2_kernel<T>   1x1 with K of 128..512 elements, 16-bit: 256x128, two workgroups per CU
//   conv_clip_kernel<T,MODE>     CubePad(1)+3x3 on 7x7 faces (ConvLSTM, one cube per tile), 8x8 faces (half a cube per
//                                tile, the cube resident) or 16x16 faces (one face + ring per tile): activations
//                                LDS-resident, taps = row permutations
// (the stem and layer1's conv2 have their own resident-tile kernels in stem.hip / band3x3.hip)
//
// Tiling (wave64, MFMA 16x16):
//   workgroup = 4 waves, tile = (WN*64 output channels) x (WM*64 pixels), wave tile 64x64
//   = 4x4 MFMA tiles, f32 accumulators (64 VGPRs).  MFMA "A" operand = weights (rows =
//   output channels), "B" operand = activations (columns = pixels), so a lane ends up
//   with 4 consecutive output channels of one pixel -> 16-byte NHWC stores.
//   K step = 128 bytes per tile row (64 bf16 / 32 f32): every tile row in LDS is one
//   128-byte line of eight 16-byte chunks, chunk c of row r stored at chunk
//   c ^ ((r >> 1) & 7).  A 16-lane ds_read_b128 group (rows r0..r0+15, one logical
//   chunk) then touches 16 distinct 16-byte slots of the 256-byte bank row:
//   conflict-free reads, and the 8 lanes that write one row hit 8 distinct slots.
//   Global -> register -> LDS staging, double-buffered LDS, one barrier per K step;
//   the next step's global loads are in flight while the MFMAs of the current step run.
//   f32 path: v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulate; a lane's
//   16-byte chunk feeds 4 consecutive MFMAs with k = 4*(lane>>4) + e).
//   bf16 path: v_mfma_f32_16x16x32_bf16 (a lane's 16-byte chunk = its 8 k-values).
//   Split-K (gridDim.z) writes f32 partial slabs; conv_finish / lstm_gates reduce them.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"

// (Requesting the residual pieces BEFORE the K loop of the short-K kernel was tried: the 16-32 extra registers spill
// under its 128-VGPR budget and the launch got 40 % slower - layer3 conv3 110 -> 155 us.)
template <typename T, int MJ>
__device__ __forceinline__ void epilogue_direct(const ConvK& p, f32x4 (&acc)[4][MJ], int n0, int m0, int wch0, int wrow0,
                                                int lane, int rows_valid) {
    constexpr int JB = sizeof(T) == 4 ? 2 : 5;             // pixel blocks whose residual pieces are in flight together
    const int ml = lane & 15;
    const T* res = reinterpret_cast<const T*>(p.res);
    T* outp = reinterpret_cast<T*>(p.out);
    const int rlim = min(rows_valid, p.M - m0);            // tile rows that are output pixels
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int n = n0 + wch0 + pr * 32 + (lane >> 4) * 8;
        const bool nok = n < p.c_out;                      // c_out % 8 == 0 on this path
        float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias && nok) {
            const float4 t0 = *reinterpret_cast<const float4*>(p.bias + n);
            const float4 t1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bb[0] = t0.x; bb[1] = t0.y; bb[2] = t0.z; bb[3] = t0.w; bb[4] = t1.x; bb[5] = t1.y; bb[6] = t1.z; bb[7] = t1.w;
        }
#pragma unroll
        for (int j0 = 0; j0 < MJ; j0 += JB) {
            u32x4 rr[JB][sizeof(T) == 4 ? 2 : 1];
            if (res) {
#pragma unroll
                for (int u = 0; u < JB; ++u) {
                    const int row = wrow0 + (j0 + u) * 16 + ml;
                    if (j0 + u < MJ) {
                        const bool ok = nok && row < rlim;
                        const T* src = ok ? res + (size_t)(m0 + row) * p.ld_res + n : reinterpret_cast<const T*>(g_zero16);
                        rr[u][0] = *reinterpret_cast<const u32x4*>(src);
                        if constexpr (sizeof(T) == 4) rr[u][1] = *reinterpret_cast<const u32x4*>(ok ? src + 4 : src);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const int j = j0 + u;
                if (j >= MJ) continue;
                const int row = wrow0 + j * 16 + ml;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * pr][j][e] + bb[e];
                    v[4 + e] = acc[2 * pr + 1][j][e] + bb[4 + e];
                }
                if (res) {
                    float r[8];
                    if constexpr (sizeof(T) == 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            r[e] = __uint_as_float(rr[u][0][e]);
                            r[4 + e] = __uint_as_float(rr[u][1][e]);
                        }
                    } else {
                        unpack8(rr[u][0], r, T());
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (nok && row < rlim) {
                    T* dst = outp + (size_t)(m0 + row) * p.ld_out + p.out_coff + n;
                    if constexpr (sizeof(T) == 4) {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
                        *reinterpret_cast<u32x4*>(dst) = pack8(v, T());
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ epilogue through LDS
// A lane holds 4 consecutive channels of one pixel per MFMA tile: stored directly that is
// 8-byte (bf16) pieces a pixel stride apart - fine for the f32 split-K slabs, wasteful for
// the HBM-bound 1x1 convolutions of ResNet whose traffic is output + residual.  The normal
// path therefore goes through LDS (the pipeline buffers are free after the K loop):
// (A) the residual tile is read with full-line 16-byte accesses into LDS, (B) every lane adds
// bias (+ its residual values from LDS), applies ReLU, rounds ONCE and writes its 4 channels
// back to the same LDS slot, (C) the tile leaves with full-line 16-byte stores.  Tiles that do
// not fit the kernel's LDS are processed in two channel halves.
template <typename T, int BN, int BM, int MJ, int NT, int LDS_BYTES>
__device__ __forceinline__ void epilogue_lds(const ConvK& p, unsigned char* lds, f32x4 (&acc)[4][MJ], int n0, int m0,
                                             int wch0 /* wave's first tile channel */,
                                             int wrow0 /* wave's first tile row (pixel) */, int lane, int tid,
                                             int rows_valid = BM /* tile rows that are output pixels */) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int NH = (BM * (BN * (int)sizeof(T) + 16) <= LDS_BYTES) ? 1                // channel slices
                       : (BM * (BN / 2 * (int)sizeof(T) + 16) <= LDS_BYTES) ? 2 : 4;
    constexpr int HC = BN / NH;                            // channels per half
    constexpr int CPR = HC / EPC;                          // 16-byte chunks per tile row
    constexpr int S = HC * (int)sizeof(T) + 16;            // LDS row stride (bytes), 16-byte aligned
    static_assert(BM * S <= LDS_BYTES, "epilogue tile does not fit the LDS of this kernel");
    constexpr int UB = MJ > 8 ? 4 : 8;                     // 16-byte accesses in flight per thread (register budget)
    const int ml = lane & 15;
    const T* res = reinterpret_cast<const T*>(p.res);
    T* outp = reinterpret_cast<T*>(p.out);
#pragma unroll 1
    for (int h = 0; h < NH; ++h) {
        __syncthreads();                                   // previous users of the LDS are done
        const int nh0 = n0 + h * HC;
        if (res) {
            // batches of UB independent 16-byte loads in flight per thread before the first LDS store:
            // one load per iteration serialised a full HBM latency per 16 bytes (the 1x1 expansion
            // convs of ResNet spent most of their time here)
#pragma unroll 1
            for (int q0 = tid; q0 < BM * CPR; q0 += NT * UB) {
                u32x4 r[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int q = q0 + u * NT;
                    const int row = q / CPR, c = q - row * CPR;
                    const int m = m0 + row, n = nh0 + c * EPC;
                    const bool ok = q < BM * CPR && row < rows_valid && m < p.M && n < p.c_out;
                    r[u] = *reinterpret_cast<const u32x4*>(ok ? res + (size_t)m * p.ld_res + n
                                                              : reinterpret_cast<const T*>(g_zero16));
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int q = q0 + u * NT;
                    const int row = q / CPR, c = q - row * CPR;
                    if (q < BM * CPR) *reinterpret_cast<u32x4*>(lds + row * S + c * 16) = r[u];
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cn = wch0 + (i >> 1) * 32;               // 32-channel group of this MFMA row block
            if (cn / HC != h) continue;                        // wave-uniform
            const int ncol = wch0 + acc_chan(i, lane) - h * HC;   // channel inside this half
            const int n = nh0 + ncol;
            float bb[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.c_out) {
                const float4 t = *reinterpret_cast<const float4*>(p.bias + n);
                bb[0] = t.x; bb[1] = t.y; bb[2] = t.z; bb[3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                const int row = wrow0 + j * 16 + ml;
                T* slot = reinterpret_cast<T*>(lds + row * S) + ncol;
                float v[4] = {acc[i][j][0] + bb[0], acc[i][j][1] + bb[1], acc[i][j][2] + bb[2], acc[i][j][3] + bb[3]};
                if (res) {
                    float r[4];
                    load4(slot, r);
                    v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
                }
                if (p.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                store4(slot, v);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int q0 = tid; q0 < BM * CPR; q0 += NT * UB) {
            u32x4 r[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int q = q0 + u * NT;
                const int row = q / CPR, c = q - row * CPR;
                if (q < BM * CPR) r[u] = *reinterpret_cast<const u32x4*>(lds + row * S + c * 16);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int q = q0 + u * NT;
                const int row = q / CPR, c = q - row * CPR;
                const int m = m0 + row, n = nh0 + c * EPC;
                if (q < BM * CPR && row < rows_valid && m < p.M && n < p.c_out)
                    *reinterpret_cast<u32x4*>(outp + (size_t)m * p.ld_out + p.out_coff + n) = r[u];
            }
        }
    }
}

template <typename T, int WN, int WM>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvK p) {
    constexpr int BN = WN * 64, BM = WM * 64;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int A_PASSES = BN / 32, B_PASSES = BM / 32;
    constexpr int STAGE = (BN + BM) * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    // ---- XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs
    // (each with a private 4 MiB L2): give every XCD a CONTIGUOUS range of work items
    // and order the items so that neighbours share an operand panel.  m_fast: the
    // m-tiles of one (n-tile, split) are adjacent -> the weight panel is fetched from
    // HBM once per XCD and re-read from L2 (ConvLSTM: weights >> activations);
    // otherwise n-tiles are adjacent -> the activation panel is shared (ResNet).
    // Placement only affects speed, never results (bijective map, any placement valid).
    int n0, m0, split;
    {
        const int nwg = p.nt * p.mt * p.splits;
        const int L = blockIdx.x, xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if (p.reverse) w = nwg - 1 - w;
        int nt_i, mt_i;
        if (p.m_fast) {
            mt_i = w % p.mt;
            const int rest = w / p.mt;
            nt_i = rest % p.nt;
            split = rest / p.nt;
        } else {
            nt_i = w % p.nt;
            const int rest = w / p.nt;
            mt_i = rest % p.mt;
            split = rest / p.mt;
        }
        n0 = nt_i * BN;
        m0 = mt_i * BM;
    }
    const int chunk = tid & 7, row0 = tid >> 3;

    // ---- per-thread activation rows.  Only the current element offset of each row is
    // kept in registers; (image, oy, ox) is re-derived from m when the tap changes
    // (kh*kw times per kernel) to keep the K loop's register budget for the pipeline.
    int roff[B_PASSES];
    const CubePadGeom geom{p.h_in, p.pad, p.pad, p.pad, p.pad};
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int pb = 0; pb < B_PASSES; ++pb) {
            const int m = m0 + row0 + 32 * pb;
            int off = -1;
            if (m < p.M) {
                const int img = m / p.hw_out, rem = m - img * p.hw_out;
                const int oy = rem / p.w_out, ox = rem - oy * p.w_out;
                const int py = oy * p.sy + ky, px = ox * p.sx + kx;
                int pix;
                if (p.pad_mode) {
                    const int grp = img / 6, f = img - grp * 6;
                    pix = grp * 6 * p.h_in * p.w_in + cubepad_src(f, py, px, geom);
                } else {
                    pix = (img * p.h_in + py) * p.w_in + px;
                }
                off = pix * p.pix_stride;
            }
            roff[pb] = off;
        }
    };

    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
    int tap = s_begin / p.steps_per_tap;
    int c0 = (s_begin - tap * p.steps_per_tap) * BK;

    const T* in = reinterpret_cast<const T*>(p.in);
    const T* wbase = reinterpret_cast<const T*>(p.w) + (size_t)(n0 + row0) * p.k_total + chunk * EPC;
    const size_t wpass = (size_t)32 * p.k_total;       // 32 tile rows further down

    // two register sets: the loads of K step s+2 are issued while step s is computed and
    // step s+1 (issued one iteration earlier) waits in the other set -> a weight line has
    // more than one full MFMA phase to arrive from HBM before its ds_write needs it
    u32x4 ra0[A_PASSES], rb0[B_PASSES], ra1[A_PASSES], rb1[B_PASSES];
    // (macros, not lambdas taking array references: those kept the sets in scratch memory)
#define CP360_GLOAD(RA, RB)                                                                              \
    {                                                                                                    \
        const int e_ = c0 + chunk * EPC;                                                                 \
        const size_t koff_ = (size_t)tap * p.c_pad + c0;                                                 \
        _Pragma("unroll") for (int pa = 0; pa < A_PASSES; ++pa)                                          \
            RA[pa] = *reinterpret_cast<const u32x4*>(wbase + pa * wpass + koff_);                        \
        const bool kval_ = e_ < p.c_in;                                                                  \
        _Pragma("unroll") for (int pb = 0; pb < B_PASSES; ++pb) {                                        \
            const bool ok_ = kval_ && roff[pb] >= 0;                                                     \
            const T* src_ = ok_ ? in + (size_t)roff[pb] + e_ : reinterpret_cast<const T*>(g_zero16);     \
            RB[pb] = *reinterpret_cast<const u32x4*>(src_);                                              \
        }                                                                                                \
    }
#define CP360_LDS_STORE(BUF, RA, RB)                                                                     \
    {                                                                                                    \
        unsigned char* As_ = lds + (BUF) * STAGE;                                                        \
        unsigned char* Bs_ = As_ + BN * 128;                                                             \
        _Pragma("unroll") for (int pa = 0; pa < A_PASSES; ++pa)                                          \
            *reinterpret_cast<u32x4*>(As_ + lds_swz(row0 + 32 * pa, chunk)) = RA[pa];                    \
        _Pragma("unroll") for (int pb = 0; pb < B_PASSES; ++pb)                                          \
            *reinterpret_cast<u32x4*>(Bs_ + lds_swz(row0 + 32 * pb, chunk)) = RB[pb];                    \
    }
    auto advance = [&]() __attribute__((always_inline)) {
        c0 += BK;
        if (c0 >= p.c_pad) {
            c0 = 0;
            ++tap;
            set_tap(tap);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nloc = s_end - s_begin;
    if (nloc > 0) {
        const int lrow = lane & 15, lchunk = lane >> 4;
        // a K step whose second 64-byte half is all padding (the bf16 / fp16 stem: 32 of 64 elements
        // per tap) skips that half's fragment reads and MFMAs
        const bool half_k = p.steps_per_tap == 1 && 2 * p.c_in <= BK;
        auto compute = [&](int buf) __attribute__((always_inline)) {
            const unsigned char* As = lds + buf * STAGE;
            const unsigned char* Bs = As + BN * 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk == 1 && half_k) break;
                u32x4 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = *reinterpret_cast<const u32x4*>(As + lds_swz(wn * 64 + i * 16 + lrow, kk * 4 + lchunk));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    b[j] = *reinterpret_cast<const u32x4*>(Bs + lds_swz(wm * 64 + j * 16 + lrow, kk * 4 + lchunk));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma_chunk<T>(acc[i][j], a[i], b[j]);
            }
        };
        set_tap(tap);
        CP360_GLOAD(ra0, rb0);                       // step 0
        if (nloc > 1) {
            advance();
            CP360_GLOAD(ra1, rb1);                   // step 1
        }
        CP360_LDS_STORE(0, ra0, rb0);
        __syncthreads();
        int it = 0;
        while (true) {
            // even step: compute LDS buffer 0; set 1 holds step it+1; set 0 is free
            if (it + 2 < nloc) {
                advance();
                CP360_GLOAD(ra0, rb0);
            }
            compute(0);
            if (it + 1 < nloc) CP360_LDS_STORE(1, ra1, rb1);
            __syncthreads();
            if (++it >= nloc) break;
            // odd step: compute LDS buffer 1; set 0 holds step it+1; set 1 is free
            if (it + 2 < nloc) {
                advance();
                CP360_GLOAD(ra1, rb1);
            }
            compute(1);
            if (it + 1 < nloc) CP360_LDS_STORE(0, ra0, rb0);
            __syncthreads();
            if (++it >= nloc) break;
        }
    }

#undef CP360_GLOAD
#undef CP360_LDS_STORE
    // ---- epilogue: through LDS (full-line accesses) unless it is a split-K slab or misaligned
    if (!p.partial && (p.c_out % EPC == 0) && (p.ld_out % EPC == 0) && (p.out_coff % EPC == 0) &&
        (p.ld_res % EPC == 0)) {
        epilogue_lds<T, BN, BM, 4, 256, 2 * STAGE>(p, lds, acc, n0, m0, wn * 64, wm * 64, lane, tid);
        return;
    }
    // ---- epilogue: lane holds channels n..n+3 of pixel m for every (i, j) sub-tile
    const int ml = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + acc_chan(i, lane);
        if (n >= p.c_out) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + ml;
            if (m >= p.M) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.partial) {
                store4(p.partial + ((size_t)split * p.M + m) * p.c_out + (p.slab_rows ? slab_col(n) : n), v);
            } else {
                if (p.bias) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (p.res) {
                    float r[4];
                    load4(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ld_res + n, r);
                    v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
                }
                if (p.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                store4(reinterpret_cast<T*>(p.out) + (size_t)m * p.ld_out + p.out_coff + n, v);
            }
        }
    }
}

// ------------------------------------------------------------------ wide-tile LDS-DMA kernel
// For c_out >= 256 (ConvLSTM, layers 2-4, CAM): 256 (channels) x 128 (pixels) tile, 8 waves
// (4 x 2, each 64 x 64), K step 128 bytes, THREE LDS stages (3 x 48 KiB) filled by LDS-DMA
// (global_load_lds_dwordx4: global -> LDS with no VGPR staging).  A DMA wave-instruction
// writes 1 KiB = 8 tile rows linearly (LDS address = M0 + lane*16), so the XOR swizzle of
// the LDS image is applied on the SOURCE side: lane l, which lands in physical chunk l&7
// of row r, fetches logical chunk (l&7) ^ ((r>>1)&7) - the same involution the ds_read side
// applies.  The per-lane source address also carries the CubePad / im2col gather.
// Pipeline (one barrier per K step): at step `it` a wave waits (counted vmcnt) for its own
// DMA of step `it`, meets the barrier (everyone's step-`it` data has landed and everyone has
// finished reading the buffer of step it-1), issues the DMA of step it+2 into that freed
// buffer and computes step `it`: every HBM/L2 load has two full MFMA phases to arrive.
// The DMA is issued from inline asm (the compiler would otherwise drain vmcnt(0) before
// every ds_read); its completion is counted by hand: DMA_PER_STEP per thread per step.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst /* wave-uniform */) {
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

// the same with the non-temporal hint: a weight stream much larger than the 256 MB Infinity Cache, past the head of a workgroup's
// share (ConvK::w_pin) - it then no longer sweeps the cache of the split-K slabs / activations the next launches read, and the heads
// of the streams, which every workgroup asks for at once when the launch starts, are still there from the previous step (measured on
// the Winograd GEMM first: csrc/wino.hip fill_one, tools/wino_upin_probe.sh)
__device__ __forceinline__ void glds16_nt(const void* gsrc, unsigned lds_dst /* wave-uniform */) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

template <typename T, int MJ>      // MJ = 16-pixel sub-tiles per wave: 4 -> 256x128 tile, 3 LDS stages; 8 -> 256x256, 2 stages
__global__ __launch_bounds__(512, 2) void conv_igemm_dma_kernel(const ConvK p) {
    constexpr int BN = 256, BM = 32 * MJ, NSTAGE = (MJ == 4) ? 3 : 2;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int A_PASSES = BN / 64, B_PASSES = BM / 64;       // 64 tile rows per 512-thread pass
    constexpr int DMA_PER_STEP = A_PASSES + B_PASSES;
    constexpr int STAGE = (BN + BM) * 128;
    constexpr int EPI_BYTES = BM * (BN / (sizeof(T) == 4 ? 2 : 1) * (int)sizeof(T) + 16);
    constexpr int PIPE_BYTES = NSTAGE * STAGE;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[(PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES) ];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    int n0, m0, split;
    {
        const int nwg = p.nt * p.mt * p.splits;
        const int L = blockIdx.x, xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if (p.reverse) w = nwg - 1 - w;
        int nt_i, mt_i;
        if (p.m_fast) {
            mt_i = w % p.mt;
            const int rest = w / p.mt;
            nt_i = rest % p.nt;
            split = rest / p.nt;
        } else {
            nt_i = w % p.nt;
            const int rest = w / p.nt;
            mt_i = rest % p.mt;
            split = rest / p.mt;
        }
        n0 = nt_i * BN;
        m0 = mt_i * BM;
    }
    // DMA role of this lane: tile row (within a 64-row pass) and the logical chunk it fetches
    const int drow = 8 * wave + (lane >> 3);
    const int dchunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);      // = (l&7) ^ ((row>>1)&7)

    int roff[B_PASSES];
    const CubePadGeom geom{p.h_in, p.pad, p.pad, p.pad, p.pad};
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int pb = 0; pb < B_PASSES; ++pb) {
            const int m = m0 + drow + 64 * pb;
            int off = -1;
            if (m < p.M) {
                const int img = m / p.hw_out, rem = m - img * p.hw_out;
                const int oy = rem / p.w_out, ox = rem - oy * p.w_out;
                const int py = oy * p.sy + ky, px = ox * p.sx + kx;
                int pix;
                if (p.pad_mode) {
                    const int grp = img / 6, f = img - grp * 6;
                    pix = grp * 6 * p.h_in * p.w_in + cubepad_src(f, py, px, geom);
                } else {
                    pix = (img * p.h_in + py) * p.w_in + px;
                }
                off = pix * p.pix_stride;
            }
            roff[pb] = off;
        }
    };

    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
    const int nloc = s_end - s_begin;
    int tap = s_begin / p.steps_per_tap;
    int c0 = (s_begin - tap * p.steps_per_tap) * BK;

    const T* in = reinterpret_cast<const T*>(p.in);
    const T* wbase = reinterpret_cast<const T*>(p.w) + (size_t)(n0 + drow) * p.k_total + dchunk * EPC;
    const size_t wpass = (size_t)64 * p.k_total;
    const unsigned lds_base = (unsigned)(size_t)lds;                 // LDS byte offset (low 32 bits of the flat address)
    const unsigned lds_wave = lds_base + (unsigned)(8 * wave) * 128; // this wave's 1 KiB slot inside a 64-row pass

    // DMA of one 64-row pass (q < A_PASSES: weight rows, else activation rows) of the CURRENT (tap, c0)
    auto issue_one = [&](int q, unsigned sbase) __attribute__((always_inline)) {
        if (q < A_PASSES) {
            glds16(wbase + q * wpass + ((size_t)tap * p.c_pad + c0), sbase + q * 64 * 128);
        } else {
            const int pb = q - A_PASSES;
            const int e = c0 + dchunk * EPC;
            const bool ok = e < p.c_in && roff[pb] >= 0;
            const T* src = ok ? in + (size_t)roff[pb] + e : reinterpret_cast<const T*>(g_zero16);
            glds16(src, sbase + BN * 128 + pb * 64 * 128);
        }
    };
    auto issue = [&](int stage) __attribute__((always_inline)) {
        const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)stage * STAGE);
#pragma unroll
        for (int q = 0; q < DMA_PER_STEP; ++q) issue_one(q, sbase);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        c0 += BK;
        if (c0 >= p.c_pad) {
            c0 = 0;
            ++tap;
            set_tap(tap);
        }
    };

    f32x4 acc[4][MJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nloc > 0) {
        const int lrow = lane & 15, lchunk = lane >> 4;
        set_tap(tap);
        issue(0);                                   // step 0 -> stage 0
        if (NSTAGE == 3 && nloc > 1) {
            advance();
            issue(1);                               // step 1 -> stage 1
        }
        int stage = 0;                              // stage holding step `it`
        // One K step = 8 groups of MJ MFMAs (2 k-blocks x 4 channel blocks).  REFILL: the DMA
        // instructions that refill the stage freed by the barrier are spread over the groups, so
        // their address arithmetic / M0 traffic issues in the shadow of MFMAs; the first goes
        // out under the latency of the first fragment reads.
#define CP360_DMA_STEP(REFILL)                                                                             \
        {                                                                                                  \
            const unsigned char* As = lds + stage * STAGE;                                                 \
            const unsigned char* Bs = As + BN * 128;                                                       \
            unsigned sbase = 0;                                                                            \
            u32x4 a[4], b[MJ];                                                                             \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                             \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
                    a[i] = *reinterpret_cast<const u32x4*>(As + lds_swz(wn * 64 + i * 16 + lrow, kk * 4 + lchunk)); \
                _Pragma("unroll") for (int j = 0; j < MJ; ++j)                                             \
                    b[j] = *reinterpret_cast<const u32x4*>(Bs + lds_swz(wm * (16 * MJ) + j * 16 + lrow, kk * 4 + lchunk)); \
                if (REFILL && kk == 0) {                                                                   \
                    advance();                                                                             \
                    sbase = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)(stage == 0 ? NSTAGE - 1 : stage - 1) * STAGE); \
                }                                                                                          \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
                    if (REFILL && kk * 4 + i < DMA_PER_STEP) issue_one(kk * 4 + i, sbase);                 \
                    _Pragma("unroll") for (int j = 0; j < MJ; ++j) mma_chunk<T>(acc[i][j], a[i], b[j]);    \
                }                                                                                          \
            }                                                                                              \
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;                                                   \
        }
        int it = 0;
        for (; it + NSTAGE - 1 < nloc; ++it) {      // steady state: NSTAGE-2 younger DMA groups in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * DMA_PER_STEP) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            CP360_DMA_STEP(true)
        }
        for (; it < nloc; ++it) {                   // drain: no refill
            if (NSTAGE == 3 && it + 1 < nloc) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STEP) : "memory");
            else                              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            CP360_DMA_STEP(false)
        }
#undef CP360_DMA_STEP
    }

    // ---- epilogue: through LDS (full-line accesses) unless it is a split-K slab or misaligned
    const int ml = lane & 15;
    const bool lds_epi = !p.partial && (p.c_out % EPC == 0) && (p.ld_out % EPC == 0) && (p.out_coff % EPC == 0) &&
                         (p.ld_res % EPC == 0);
    if (lds_epi) {
        epilogue_lds<T, BN, BM, MJ, 512, (PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES)>(
            p, lds, acc, n0, m0, wn * 64, wm * (16 * MJ), lane, tid);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + acc_chan(i, lane);
        if (n >= p.c_out) continue;
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int m = m0 + wm * (16 * MJ) + j * 16 + ml;
            if (m >= p.M) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.partial) {
                store4(p.partial + ((size_t)split * p.M + m) * p.c_out + (p.slab_rows ? slab_col(n) : n), v);
            } else {
                if (p.bias) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (p.res) {
                    float r[4];
                    load4(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ld_res + n, r);
                    v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
                }
                if (p.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                store4(reinterpret_cast<T*>(p.out) + (size_t)m * p.ld_out + p.out_coff + n, v);
            }
        }
    }
}

// ------------------------------------------------------------------ 256 x {256, 304} tile, 4-stage ring
// Same 8-wave / 256-channel tile as conv_igemm_dma_kernel<T, 8>, but the K step is HALF a line
// (64 bytes per tile row = one MFMA k-block) and the LDS holds a ring of FOUR stages:
// the DMA of sub-step s+3 is issued at the start of sub-step s, so a load has three MFMA
// phases (1.5 of the 2-stage kernel's steps) to arrive - the 2-stage kernel spent ~30 % of its
// wave cycles in s_waitcnt/barrier (SQ_WAIT_ANY, profiles/).  A DMA wave-instruction now
// covers 16 rows x 64 B; LDS rows are 64 B = 4 chunks, chunk c of row r stored at chunk
// c ^ ((-(r >> 2)) & 3), which keeps every 16-lane ds_read_b128 group on 16 distinct 16-byte
// slots of the 256-byte bank row (4 rows per bank row).
//
// Pixel-tile width BM:
//   256: waves = 4 channel quarters x 2 pixel halves, 4 x 8 MFMA tiles each.
//   304: for the ConvLSTM at 7x7 faces, where a clip is 6*49 = 294 pixels: 19 sixteen-pixel
//        blocks per tile, so B clips are exactly B tiles (M = 1176 pads to 1216 instead of 1280,
//        and 16 channel tiles x 4 pixel tiles x 4 K-splits = 256 workgroups = one per CU).
//        Waves 0-3 (channel quarter = wave) take pixel blocks 0-9, waves 4-7 blocks 10-18: the
//        two waves that share a SIMD (w and w+4) carry 10 + 9 blocks, so every SIMD's matrix
//        pipe has the same work.  The 48 extra activation rows are one more DMA instruction
//        for waves 0-2 (their vmcnt budget is counted separately).
__device__ __forceinline__ int lds_swz64(int row, int chunk) {
    return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt_upto(int n);   // s_waitcnt vmcnt(min(n, N)), n wave-uniform
template <> __device__ __forceinline__ void wait_vmcnt_upto<0>(int) { wait_vmcnt<0>(); }
template <int N> __device__ __forceinline__ void wait_vmcnt_upto(int n) {
    if (n >= N) wait_vmcnt<N>();
    else wait_vmcnt_upto<N - 1>(n);
}

template <int BM, int NS = 4> struct RingGeom {
    static constexpr int BN = 256, NSTAGE = NS;
    static constexpr int A_PASSES = BN / 128;                    // 128 tile rows per 512-thread pass
    static constexpr int B_FULL = BM / 128;                      // full activation passes
    static constexpr int B_REM = BM % 128;                       // rows of the partial pass (0 or 48)
    static constexpr int B_PASSES = B_FULL + (B_REM ? 1 : 0);
    static constexpr int XWAVES = B_REM / 16;                    // waves that issue the partial pass
    static constexpr int DMA_BASE = A_PASSES + B_FULL;           // DMA instructions per sub-step (every wave)
    static constexpr int STAGE = (BN + BM) * 64;
    template <typename T> static constexpr int epi_bytes() {
        return BM * (BN / (sizeof(T) == 4 ? 2 : 1) * (int)sizeof(T) + 16);
    }
    template <typename T> static constexpr int lds_bytes() {
        return NSTAGE * STAGE > epi_bytes<T>() ? NSTAGE * STAGE : epi_bytes<T>();
    }
};

// K loop + epilogue of one wave: channels [wch0, wch0+64) x MJ pixel blocks from tile row wrow0.
// JH = pixel blocks of the first half of a sub-step (their activation fragments are held at once;
// the other MJ - JH reuse the registers).  LAG: this wave runs half a sub-step behind (waves 4-7).
template <typename T, int BM, int MJ, int JH, bool LAG, int NS = 4>
__device__ __forceinline__ void ring_body(const ConvK& p, unsigned char* lds, const int n0, const int m0, const int split,
                                          const int wave, const int lane, const int tid, const int wch0,
                                          const int wrow0) {
    typedef RingGeom<BM, NS> G;
    constexpr int BN = G::BN, STAGE = G::STAGE, A_PASSES = G::A_PASSES, B_PASSES = G::B_PASSES;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BKS = 4 * EPC;                               // K elements per sub-step (64 bytes)
    constexpr int D0 = G::DMA_BASE, D1 = G::DMA_BASE + 1;
    const bool xw = G::XWAVES > 0 && wave < G::XWAVES;         // wave-uniform: this wave issues the partial pass

    // DMA role: 16 rows x 4 chunks per wave-instruction; lane l lands in row l>>2, physical
    // chunk l&3 and therefore fetches logical chunk (l&3) ^ f(row), f = (-(row>>2)) & 3
    const int drow = 16 * wave + (lane >> 2);                                  // row inside a 128-row pass
    const int dchunk = (lane & 3) ^ ((0 - ((4 * wave + (lane >> 4)) & 3)) & 3); // ((drow>>2)&3) = (4*wave + (lane>>4)) & 3

    int roff[B_PASSES];
    const CubePadGeom geom{p.h_in, p.pad, p.pad, p.pad, p.pad};
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        const bool sec = tap >= p.ntap;                        // the second source's tap (wave-uniform)
#pragma unroll
        for (int pb = 0; pb < B_PASSES; ++pb) {
            const int m = m0 + drow + 128 * pb;
            int off = -1;
            if (m < p.M && (pb < G::B_FULL || xw)) {
                const int img = m / p.hw_out, rem = m - img * p.hw_out;
                const int oy = rem / p.w_out, ox = rem - oy * p.w_out;
                const int py = oy * p.sy + ky, px = ox * p.sx + kx;
                int pix;
                if (sec) {
                    off = ((img * p.h_in2 + oy * p.sy2) * p.w_in2 + ox * p.sx2) * p.pix_stride2;
                    roff[pb] = off;
                    continue;
                }
                if (p.pad_mode) {
                    const int grp = img / 6, f = img - grp * 6;
                    pix = grp * 6 * p.h_in * p.w_in + cubepad_src(f, py, px, geom);
                } else {
                    pix = (img * p.h_in + py) * p.w_in + px;
                }
                off = pix * p.pix_stride;
            }
            roff[pb] = off;
        }
    };

    // sub-step range of this split (two sub-steps per 128-byte K step of the packed layout)
    const int s_begin = 2 * split * p.steps_per_split;
    const int s_end = 2 * min(p.nsteps, (split + 1) * p.steps_per_split);
    const int nloc = s_end - s_begin;
    const int sub_per_tap = 2 * p.steps_per_tap;
    int tap = min(s_begin / sub_per_tap, p.ntap);              // ntap = the second source's tap (if any)
    int c0 = (s_begin - tap * sub_per_tap) * BKS;

    const T* in = reinterpret_cast<const T*>(p.in);
    const T* in2 = reinterpret_cast<const T*>(p.in2);
    const T* wbase = reinterpret_cast<const T*>(p.w) + (size_t)(n0 + drow) * p.k_total + dchunk * EPC;
    const size_t wpass = (size_t)128 * p.k_total;
    const unsigned lds_base = (unsigned)(size_t)lds;
    const unsigned lds_wave = lds_base + (unsigned)(16 * wave) * 64;          // this wave's 1 KiB inside a pass

    // DMA of one pass (q < A_PASSES: weight rows, else activation rows) of the CURRENT (tap, c0)
    auto issue_one = [&](int q, unsigned sbase) __attribute__((always_inline)) {
        if (q < A_PASSES) {
            glds16(wbase + q * wpass + ((size_t)tap * p.c_pad + c0), sbase + q * 128 * 64);
        } else {
            const int pb = q - A_PASSES;
            const int e = c0 + dchunk * EPC;
            const bool sec = tap >= p.ntap;
            const bool ok = e < (sec ? p.c_in2 : p.c_in) && roff[pb] >= 0;
            const T* src = ok ? (sec ? in2 : in) + (size_t)roff[pb] + e : reinterpret_cast<const T*>(g_zero16);
            glds16(src, sbase + BN * 64 + pb * 128 * 64);
        }
    };
    auto issue = [&](int stage) __attribute__((always_inline)) {
        const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)stage * STAGE);
#pragma unroll
        for (int q = 0; q < D0; ++q) issue_one(q, sbase);
        if (G::XWAVES > 0 && xw) issue_one(D0, sbase);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        c0 += BKS;
        if (c0 >= p.c_pad && tap < p.ntap) {                   // (the second source's tap is the last one)
            c0 = 0;
            ++tap;
            set_tap(tap);
        }
    };

    f32x4 acc[4][MJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nloc > 0) {
        const int lrow = lane & 15, lchunk = lane >> 4;
        set_tap(tap);
        issue(0);
#pragma unroll
        for (int k = 1; k < NS - 1; ++k)
            if (nloc > k) { advance(); issue(k); }
        int stage = 0;
        // A sub-step is two PHASES separated by a second barrier, and the two waves that share a
        // SIMD (w and w+4) run half a sub-step apart (MI355X_MICROARCH.md, "Two waves per SIMD",
        // item 9).  HEAD = fragment reads of the sub-step's stage (a[4] + the first JH pixel
        // blocks), the refill DMA and the first JH*4 MFMAs, column by column: as soon as the four
        // MFMAs of column j are issued its registers are re-loaded with pixel block JH + j, so the
        // second half's operands arrive under the first half's MFMAs; TAIL = the remaining MFMAs,
        // all operands in registers.
        //   waves 0-3 (LAG = false):  B1  HEAD(s)    B2  TAIL(s)
        //   waves 4-7 (LAG = true ):  B1  TAIL(s-1)  B2  HEAD(s)
        // so while one wave of a SIMD waits for its LDS reads the other feeds the matrix pipe from
        // registers.  Stage s is read between B1(s) and B1(s+1) by both groups and refilled (with
        // sub-step s+4) after B1(s+1), exactly as without the stagger; a lagging wave drains its
        // LDS reads (lgkmcnt) before B1 because the tail operands it read last are first used after it.
        // ALLUP (the 8-wave big-tile variants): the load half issues every fragment read and the refill DMA and no
        // MFMA, the compute half all of them from registers (as the clip kernel; -8 % there); the short-K two-workgroup
        // variant (128-VGPR budget) keeps the JH / MJ - JH split with register reuse
        constexpr bool ALLUP = NS > 2 && MJ <= 8;               // (the 10-block 304-pixel tile would spill: 256 VGPRs)
        constexpr int NBR = ALLUP ? MJ : JH;
        u32x4 a[4], b[NBR];
        if (LAG) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < NBR; ++j) b[j] = u32x4{0u, 0u, 0u, 0u};
        }
#define CP360_RING_HEAD(REFILL)                                                                            \
        {                                                                                                  \
            const unsigned char* As = lds + stage * STAGE;                                                 \
            const unsigned char* Bs = As + BN * 64;                                                        \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                  \
                a[i] = *reinterpret_cast<const u32x4*>(As + lds_swz64(wch0 + i * 16 + lrow, lchunk));      \
            _Pragma("unroll") for (int j = 0; j < NBR; ++j)                                                \
                b[j] = *reinterpret_cast<const u32x4*>(Bs + lds_swz64(wrow0 + j * 16 + lrow, lchunk));     \
            unsigned sbase = 0;                                                                            \
            if (REFILL) {                                                                                  \
                advance();                                                                                 \
                sbase = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)(stage == 0 ? NS - 1 : stage - 1) * STAGE); \
            }                                                                                              \
            if (ALLUP) {                                                                                   \
                _Pragma("unroll") for (int q = 0; q < D0; ++q)                                             \
                    if (REFILL) issue_one(q, sbase);                                                       \
            } else {                                                                                       \
                _Pragma("unroll") for (int j = 0; j < JH; ++j) {                                           \
                    if (REFILL && j < D0) issue_one(j, sbase);                                             \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) mma_chunk<T>(acc[i][j], a[i], b[j]);     \
                    if (JH + j < MJ)   /* column j is done: its registers take pixel block JH + j */       \
                        b[j] = *reinterpret_cast<const u32x4*>(Bs + lds_swz64(wrow0 + (JH + j) * 16 + lrow, lchunk)); \
                }                                                                                          \
                _Pragma("unroll") for (int q = JH; q < D0; ++q)   /* more DMA passes than MFMA columns (BM = 128) */ \
                    if (REFILL) issue_one(q, sbase);                                                       \
            }                                                                                              \
            if (REFILL && G::XWAVES > 0 && xw) issue_one(D0, sbase);                                       \
            stage = stage == NS - 1 ? 0 : stage + 1;                                                       \
        }
#define CP360_RING_TAIL()                                                                                  \
        {                                                                                                  \
            _Pragma("unroll") for (int j = ALLUP ? 0 : JH; j < MJ; ++j)                                    \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) mma_chunk<T>(acc[i][j], a[i], b[ALLUP ? j : j - JH]); \
        }
#define CP360_RING_STEP(REFILL)                                                                            \
        {                                                                                                  \
            if (LAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                    \
            __builtin_amdgcn_s_barrier();                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (!LAG) CP360_RING_HEAD(REFILL) else CP360_RING_TAIL()                                       \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            __builtin_amdgcn_s_barrier();                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (!LAG) CP360_RING_TAIL() else CP360_RING_HEAD(REFILL)                                       \
        }
        int it = 0;
        for (; it + NS - 1 < nloc; ++it) {          // steady state: NS-2 younger DMA groups in flight
            if (xw) wait_vmcnt<(NS - 2) * D1>(); else wait_vmcnt<(NS - 2) * D0>();
            CP360_RING_STEP(true)
        }
        for (; it < nloc; ++it) {                   // drain: no refill
            const int young = min(NS - 2, nloc - 1 - it);
            wait_vmcnt_upto<(NS - 2) * D1>(young * (xw ? D1 : D0));
            CP360_RING_STEP(false)
        }
        if (LAG) CP360_RING_TAIL()
#undef CP360_RING_STEP
#undef CP360_RING_HEAD
#undef CP360_RING_TAIL
    }

    const int ml = lane & 15;
    if (!p.partial && p.epi_direct) {
        epilogue_direct<T, MJ>(p, acc, n0, m0, wch0, wrow0, lane, BM);
        return;
    }
    if (!p.partial && (p.c_out % EPC == 0) && (p.ld_out % EPC == 0) && (p.out_coff % EPC == 0) &&
        (p.ld_res % EPC == 0)) {
        epilogue_lds<T, BN, BM, MJ, 512, G::template lds_bytes<T>()>(p, lds, acc, n0, m0, wch0, wrow0, lane, tid);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wch0 + acc_chan(i, lane);
        if (n >= p.c_out) continue;
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int m = m0 + wrow0 + j * 16 + ml;
            if (m >= p.M) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.partial) {
                store4(p.partial + ((size_t)split * p.M + m) * p.c_out + (p.slab_rows ? slab_col(n) : n), v);
            } else {
                if (p.bias) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (p.res) {
                    float r[4];
                    load4(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ld_res + n, r);
                    v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
                }
                if (p.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                store4(reinterpret_cast<T*>(p.out) + (size_t)m * p.ld_out + p.out_coff + n, v);
            }
        }
    }
}

template <typename T, int BM>
__global__ __launch_bounds__(512, 2) void conv_igemm_ring_kernel(const ConvK p) {
    typedef RingGeom<BM> G;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G::template lds_bytes<T>()];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int n0, m0, split;
    {
        const int nwg = p.nt * p.mt * p.splits;
        const int L = blockIdx.x, xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if (p.reverse) w = nwg - 1 - w;
        int nt_i, mt_i;
        if (p.m_fast) {
            mt_i = w % p.mt;
            const int rest = w / p.mt;
            nt_i = rest % p.nt;
            split = rest / p.nt;
        } else {
            nt_i = w % p.nt;
            const int rest = w / p.nt;
            mt_i = rest % p.mt;
            split = rest / p.mt;
        }
        n0 = nt_i * G::BN;
        m0 = mt_i * BM;
    }
    if constexpr (BM == 256) {
        if (wave < 4) ring_body<T, 256, 8, 4, false>(p, lds, n0, m0, split, wave, lane, tid, (wave >> 1) * 64, (wave & 1) * 128);
        else          ring_body<T, 256, 8, 4, true>(p, lds, n0, m0, split, wave, lane, tid, (wave >> 1) * 64, (wave & 1) * 128);
    } else if constexpr (BM == 160) {
        // (round 6) 160-pixel tile: 5 pixel blocks per wave.  For launches whose 256 / 304-pixel tiles leave a third of the CUs without a
        // workgroup (layer4's 1x1 and 3x3 convolutions at 64 frames: M = 18816 pixels x 512 channels = 148 tiles of 256 pixels) - 236
        // tiles of 160 give (nearly) every CU one, without a split-K slab round trip
        if (wave < 4) ring_body<T, 160, 5, 3, false>(p, lds, n0, m0, split, wave, lane, tid, (wave >> 1) * 64, (wave & 1) * 80);
        else          ring_body<T, 160, 5, 3, true>(p, lds, n0, m0, split, wave, lane, tid, (wave >> 1) * 64, (wave & 1) * 80);
    } else {
        static_assert(BM == 304, "pixel tile is 160, 256 or 304");
        if (wave < 4) ring_body<T, 304, 10, 5, false>(p, lds, n0, m0, split, wave, lane, tid, wave * 64, 0);
        else          ring_body<T, 304, 9, 5, true>(p, lds, n0, m0, split, wave, lane, tid, (wave - 4) * 64, 160);
    }
}

// Short-K variant (the HBM-bound 1x1 convolutions of ResNet, K <= 512): 256 x 128 tile, two 24 KiB
// stages, 64 accumulator registers per lane - two workgroups fit a CU (launch bounds 4 waves per
// SIMD), so one tile's epilogue traffic overlaps the other's operand loads and MFMAs.  A single
// 160 KiB-LDS workgroup per CU runs load, compute, residual and store phases strictly one after the
// other (ablations in DESIGN.md).  Epilogue: direct 16-byte pieces (no LDS).
// (Three 24 KiB stages instead of two - 72 KiB, still two workgroups per CU - measured the same on the K <= 512 B launches and slower than
// the 256x304 ring at K = 1 - 2 KiB: 99 / 117 us against 86 / 76 per launch, profiles/r06_ring2_ab.log.)
template <typename T>
__global__ __launch_bounds__(512, 4) void conv_igemm_ring2_kernel(const ConvK p) {
    typedef RingGeom<128, 2> G;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G::template lds_bytes<T>()];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int n0, m0, split;
    {
        const int nwg = p.nt * p.mt * p.splits;
        const int L = blockIdx.x, xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if (p.reverse) w = nwg - 1 - w;
        int nt_i, mt_i;
        if (p.m_fast) {
            mt_i = w % p.mt;
            const int rest = w / p.mt;
            nt_i = rest % p.nt;
            split = rest / p.nt;
        } else {
            nt_i = w % p.nt;
            const int rest = w / p.nt;
            mt_i = rest % p.mt;
            split = rest / p.mt;
        }
        n0 = nt_i * G::BN;
        m0 = mt_i * 128;
    }
    if (wave < 4) ring_body<T, 128, 4, 2, false, 2>(p, lds, n0, m0, split, wave, lane, tid, (wave >> 1) * 64, (wave & 1) * 64);
    else          ring_body<T, 128, 4, 2, true, 2>(p, lds, n0, m0, split, wave, lane, tid, (wave >> 1) * 64, (wave & 1) * 64);
}

// ------------------------------------------------------------------ clip-resident ConvLSTM convolution
// CubePad(1) + 3x3 convolution on 7x7 cube faces (model/clstm.py:56-64 at the 224-pixel cube size).
// CubePad only ever copies from the other faces of the SAME cube (cube_pad.py:114-216), so the nine
// taps of an output pixel gather from the 6*49 = 294 pixels of its own clip - exactly one 304-row
// tile.  The generic kernels re-fetch the activation rows of every tap from L2 (LDS-DMA loads never
// hit L1: TCP_TCC_READ_REQ == TCP_TOTAL_CACHE_ACCESSES in profiles/), and the L2 -> LDS fill, not the
// MFMA pipe, bounds them (profiles/r01_pmc_korder.txt).  Here
//   * the packed weights are channel-major and tile-major ([n / 256][c / 64B][tap][n % 256][64B],
//     cp360_conv_desc.clip_resident): the nine taps of a 64-byte channel block are consecutive sub-steps
//     and the 16 KiB of weights a workgroup needs per sub-step are one contiguous block (full cache
//     lines, one DRAM page instead of 256 rows 72 KB apart: +3..4 % measured);
//   * the 64-byte channel block of ALL 294 pixels of the clip is brought into LDS ONCE (19 KiB, three
//     buffers) and every tap's B fragments are read from it through a per-(tap, row) source-row table:
//     a tap is a row permutation of the resident tile;
//   * only the weight stream (16 KiB per sub-step, six-stage ring = five sub-steps of prefetch) is
//     DMA'd per sub-step: the fill per sub-step drops from 35 KiB to 16 + 19/9 = 18 KiB.
// The resident tile has its own swizzle: chunk c of row r sits at chunk c ^ (2 * ((r >> 2) & 1)).  A
// tap reads 16 CONSECUTIVE source rows starting anywhere (interior pixels: row + (ky-1)*7 + (kx-1)), and
// with this function every 16-lane ds_read_b128 group still covers 16 distinct 16-byte slots of the
// bank row for any start (the ring's (-(r >> 2)) & 3 is conflict-free only for starts that are
// multiples of 16: 9.6e7 conflict cycles per launch measured with it, profiles/).
// Tile / wave layout and the half-sub-step stagger are those of the 256x304 ring kernel.
__device__ __forceinline__ int clip_swz(int row) { return ((row >> 2) & 1) << 1; }

// FACE variant (16x16 faces, the ConvLSTM at cube size 512 - BASELINE config C5): a tile is ONE face (256
// pixels, 8 + 8 pixel blocks) and the resident tile is that face WITH its CubePad(1) ring, 18 x 18 = 324
// rows gathered through cubepad_src() at DMA time; a tap then reads row (y + ky) * 18 + x + kx.
// HALF variant (8x8 faces, cube size 256 - the reference's own smoke-test size, model/cube_pad.py:256-261): a cube is 384
// pixels, more than one 304-row tile and more accumulators than a wave has, so a tile is HALF a cube (three faces = 192
// pixels, 6 + 6 pixel blocks) while the resident tile is the WHOLE cube (384 rows: the taps of a face read its
// neighbours); the source-row table is the clip variant's, built for the tile's half.
// MODE: 0 = one cube per tile (faces up to 7x7), 1 = FACE, 2 = HALF.
template <int MODE> struct ClipGeom {
    static constexpr bool FACE = MODE == 1, HALF = MODE == 2;
    static constexpr int BN = 256, BM = FACE ? 256 : HALF ? 192 : 304, NW = 6, NA = 2;
    static constexpr int RROWS = FACE ? 336 : HALF ? 384 : 304;  // resident rows (324 used in the FACE variant)
    static constexpr int GROUP_ROWS = FACE ? 128 : HALF ? 96 : 160;   // tile rows of wave group 0 (waves 0-3)
    static constexpr int XW = (RROWS - 256) / 16;                // waves that carry the resident rows past 256
    static constexpr int WSTAGE = BN * 64;                       // 16 KiB of weights per sub-step
    static constexpr int ATILE = RROWS * 64;                     // 19 / 21 / 24 KiB: one channel block of the clip / face / cube
    static constexpr int TAB_ROW = 32;                           // bytes per (tap, wave group, lane row): 10 u16 + pad
    static constexpr int TAB_BYTES = 9 * 2 * 16 * TAB_ROW;
    static constexpr int OFF_ACT = NW * WSTAGE;
    static constexpr int OFF_TAB = OFF_ACT + NA * ATILE;
    static constexpr int LDS_BYTES = OFF_TAB + TAB_BYTES;        // 146,432 B (150,528 B FACE, 156,672 B HALF)
};

template <typename T, int MJ, int JH, bool LAG, int MODE>
__device__ __forceinline__ void clip_body(const ConvK& p, unsigned char* lds, const int n0, const int clip, const int split,
                                          const int wave, const int lane, const int tid, const int wch0, const int grp) {
    typedef ClipGeom<MODE> G;
    constexpr bool FACE = G::FACE, HALF = G::HALF;
    constexpr int BN = G::BN, BM = G::BM;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BKS = 4 * EPC;                               // K elements per sub-step (64 bytes)
    constexpr int TAPS = 9;
    const int wrow0 = grp * G::GROUP_ROWS;
    const bool xw = wave < G::XW;                              // the waves that carry the resident rows past 256
    const int rows_valid = FACE ? 256 : HALF ? 192 : p.clip_rows;   // tile rows that are output pixels
    const int m0 = clip * rows_valid;                          // `clip` = cube (clip mode), face image (FACE) or half cube (HALF)

    // DMA role (as in the ring kernel): 16 rows x 4 chunks per wave-instruction, source-side swizzle
    const int drow = 16 * wave + (lane >> 2);
    const int dchunk = (lane & 3) ^ ((0 - ((4 * wave + (lane >> 4)) & 3)) & 3);   // weight stages: ring swizzle
    const int dchunk_a = (lane & 3) ^ clip_swz(drow);                            // activation tile (drow + 128 q: same)
    const T* in = reinterpret_cast<const T*>(p.in);
    // packed weights [n / 256][sub-step][n % 256][BKS]: this tile's sub-step s is the 16 KiB at s * 256 * BKS
    const T* wbase = reinterpret_cast<const T*>(p.w) + (size_t)(n0 / 256) * 256 * p.k_total + drow * BKS + dchunk * EPC;
    constexpr int wpass = 128 * BKS;
    constexpr int WSTEP = 256 * BKS;
    const unsigned lds_base = (unsigned)(size_t)lds;
    const unsigned lds_wave = lds_base + (unsigned)(16 * wave) * 64;
    int aoff[3];                                               // element offset of this lane's resident rows (fixed)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int r = drow + 128 * q;
        int off = -1;
        if (FACE) {
            if (r < 18 * 18 && (q < 2 || xw)) {                // resident row r = padded pixel (r / 18, r % 18) of the face
                const int cube = clip / 6, f = clip - cube * 6;
                const CubePadGeom geom{16, 1, 1, 1, 1};
                off = (cube * 6 * 256 + cubepad_src(f, r / 18, r - (r / 18) * 18, geom)) * p.pix_stride;
            }
        } else if (HALF) {
            off = ((clip >> 1) * 384 + r) * p.pix_stride;     // the whole cube is resident (r < 384 = three passes)
        } else if (r < p.clip_rows && (q < 2 || xw)) {
            off = (m0 + r) * p.pix_stride;
        }
        aoff[q] = off;
    }

    const int s_begin = split * p.sub_per_split;
    const int s_end = min(p.nsub, s_begin + p.sub_per_split);
    const int nloc = s_end - s_begin;
    const int cb0 = s_begin / TAPS;
    int tap = s_begin - cb0 * TAPS;                            // compute position: tap, activation buffer
    int abuf = cb0 % G::NA;                                    // NA = 2: the tile of block cb+1 lands while block cb is computed
    int next_act = cb0;                                        // next channel block to bring in
    int woff = s_begin * WSTEP;                                // DMA position inside the tile's packed weights
    const int wnt_from = p.w_pin >= nloc ? 0x7fffffff : (s_begin + p.w_pin) * WSTEP;   // from here on the weight DMAs are non-temporal

    auto issue_w = [&](int q, unsigned sbase) __attribute__((always_inline)) {
        if (woff >= wnt_from) glds16_nt(wbase + q * wpass + woff, sbase + q * 128 * 64);      // (uniform)
        else glds16(wbase + q * wpass + woff, sbase + q * 128 * 64);
    };
    auto issue_a = [&](int q) __attribute__((always_inline)) {   // pass q of channel block next_act
        const int e = next_act * BKS + dchunk_a * EPC;
        const bool ok = e < p.c_in && aoff[q] >= 0;
        const T* src = ok ? in + (size_t)aoff[q] + e : reinterpret_cast<const T*>(g_zero16);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)(G::OFF_ACT + (next_act % G::NA) * G::ATILE));
        glds16(src, dst + q * 128 * 64);
    };
    auto issue_act_tile = [&]() __attribute__((always_inline)) {
        issue_a(0);
        issue_a(1);
        if (xw) issue_a(2);
        ++next_act;
    };

    f32x4 acc[4][MJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nloc > 0) {
        const int lrow = lane & 15, lchunk = lane >> 4;
        const unsigned cx = (unsigned)lchunk << 4;             // entry ^ cx = swizzled chunk address
        const unsigned char* tabp = lds + G::OFF_TAB + (grp * 16 + lrow) * G::TAB_ROW;
        // source-row entries of the NEXT head (10 u16: bytes 0-15 = blocks 0-7, 16-19 = blocks 8-9)
        u32x4 e03;
        unsigned e4;
        auto load_ent = [&](int t) __attribute__((always_inline)) {
            e03 = *reinterpret_cast<const u32x4*>(tabp + t * (2 * 16 * G::TAB_ROW));
            e4 = *reinterpret_cast<const unsigned*>(tabp + t * (2 * 16 * G::TAB_ROW) + 16);
        };
        auto ent = [&](int j) __attribute__((always_inline)) -> unsigned {
            const unsigned w = j < 2 ? e03.x : j < 4 ? e03.y : j < 6 ? e03.z : j < 8 ? e03.w : e4;
            return ((j & 1) ? (w >> 16) : (w & 0xffffu)) ^ cx;
        };
        // byte addresses (inside lds) of the NEXT head's B fragments, decoded from the entries at the end of a TAIL -
        // under its MFMAs and before a barrier - so that a HEAD opens with its ds_reads instead of ~40 VALU
        // instructions in the window right behind the barrier (ablation: the kernel without the per-sub-step entry
        // handling ran 8 % faster; MI355X_MICROARCH.md, two waves per SIMD, item 6)
        unsigned ba[MJ];
        auto decode = [&]() __attribute__((always_inline)) {
            const unsigned abase = (unsigned)(G::OFF_ACT + abuf * G::ATILE);
#pragma unroll
            for (int j = 0; j < MJ; ++j) ba[j] = abase + ent(j);
        };
        // prologue: the first activation tile - and the second one when the range starts inside a
        // channel block, because the tap-0 request of the next block would come too late - (older than
        // every weight DMA), then NW-1 weight stages
        issue_act_tile();
        if (tap != 0) issue_act_tile();
        {
            const unsigned sb = __builtin_amdgcn_readfirstlane(lds_wave);
            issue_w(0, sb); issue_w(1, sb);
#pragma unroll
            for (int k = 1; k < G::NW - 1; ++k)
                if (nloc > k) { woff += WSTEP; issue_w(0, sb + k * G::WSTAGE); issue_w(1, sb + k * G::WSTAGE); }
        }
        // ---- source-row table (built while the prologue's DMAs are in flight: ~2-3 us of address arithmetic under their
        // HBM latency instead of in front of it): entry(tap, group, lane row, block j) = LDS byte offset (row * 64 +
        // swizzle bits) inside the resident tile of the pixel that tap `tap` of tile row
        // group*160 + j*16 + lrow reads (through CubePad); rows past the clip read row 0 (discarded).
        {
            unsigned short* tab = reinterpret_cast<unsigned short*>(lds + G::OFF_TAB);
            const int n = p.h_in, nn = n * n;
            const CubePadGeom geom{n, 1, 1, 1, 1};
            for (int idx = tid; idx < 9 * 2 * 16 * 10; idx += 512) {
                const int j = idx % 10, lr = (idx / 10) % 16, g = (idx / 160) % 2, t = idx / 320;
                const int row = g * G::GROUP_ROWS + j * 16 + lr;
                int src = 0;
                if (FACE) {
                    if (j < 8) src = ((row >> 4) + t / 3) * 18 + (row & 15) + t % 3;   // row of the padded 18x18 face
                } else if (HALF) {
                    if (j < 6) {                                               // pixel (clip & 1) * 192 + row of the cube
                        const int pix = (clip & 1) * 192 + row;
                        src = cubepad_src(pix >> 6, ((pix >> 3) & 7) + t / 3, (pix & 7) + t % 3, geom);
                    }
                } else if (row < p.clip_rows && (g == 0 || j < 9)) {
                    const int f = row / nn, rem = row - f * nn;
                    const int y = rem / n, x = rem - y * n;
                    src = cubepad_src(f, y + t / 3, x + t % 3, geom);          // pixel index inside the clip
                }
                tab[((t * 2 + g) * 16 + lr) * (G::TAB_ROW / 2) + j] =
                    (unsigned short)(src * 64 + (clip_swz(src) << 4));
            }
            __syncthreads();
        }
        load_ent(tap);
        decode();
        int stage = 0;
        u32x4 a[4], b[MJ];
        if (LAG) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < MJ; ++j) b[j] = u32x4{0u, 0u, 0u, 0u};
        }
        // HEAD / TAIL / stagger: see ring_body.  New here: B fragments come from the resident
        // activation tile through the entries; at tap 0 the tile of the NEXT channel block is
        // requested (into the other buffer, whose block ended before this sub-step's first barrier).
#define CP360_CLIP_HEAD(REFILL)                                                                            \
        {                                                                                                  \
            const unsigned char* As = lds + stage * G::WSTAGE;                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                  \
                a[i] = *reinterpret_cast<const u32x4*>(As + lds_swz64(wch0 + i * 16 + lrow, lchunk));      \
            _Pragma("unroll") for (int j = 0; j < MJ; ++j)   /* every column's fragment: all reads of the sub-step up front */ \
                b[j] = *reinterpret_cast<const u32x4*>(lds + ba[j]);                                       \
            unsigned sbase = 0;                                                                            \
            if (REFILL) {                                                                                  \
                woff += WSTEP;                                                                             \
                sbase = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)(stage == 0 ? G::NW - 1 : stage - 1) * G::WSTAGE); \
            }                                                                                              \
            if (tap == 0) issue_act_tile();                                                                \
            if (REFILL) { issue_w(0, sbase); issue_w(1, sbase); }                                          \
            _Pragma("unroll") for (int j = 0; j < JH; ++j)                                                 \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) mma_chunk<T>(acc[i][j], a[i], b[j]);         \
            stage = stage == G::NW - 1 ? 0 : stage + 1;                                                    \
            ++tap;                                                                                         \
            if (tap == TAPS) {                                                                             \
                tap = 0;                                                                                   \
                abuf = abuf == G::NA - 1 ? 0 : abuf + 1;                                                   \
            }                                                                                              \
            load_ent(tap);                                                                                 \
        }
#define CP360_CLIP_TAIL()                                                                                  \
        {                                                                                                  \
            _Pragma("unroll") for (int j = JH; j < MJ; ++j)                                                \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) mma_chunk<T>(acc[i][j], a[i], b[j]);         \
            decode();                                                                                      \
        }
#define CP360_CLIP_STEP(REFILL)                                                                            \
        {                                                                                                  \
            if (LAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                    \
            __builtin_amdgcn_s_barrier();                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (!LAG) CP360_CLIP_HEAD(REFILL) else CP360_CLIP_TAIL()                                       \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            __builtin_amdgcn_s_barrier();                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (!LAG) CP360_CLIP_TAIL() else CP360_CLIP_HEAD(REFILL)                                       \
        }
        // vmcnt: DMAs complete in issue order.  Before sub-step `it` its weight stage (issued NW-1
        // heads earlier) must have landed; younger than it are the weight stages of the next NW-2
        // sub-steps (2 DMAs each) and an activation tile (2 or 3 DMAs, issued in front of its head's
        // weight DMAs) if one was requested in one of the last NW-2 heads, i.e. iff the compute tap is
        // in 1 .. NW-2 and that head belongs to this range.
        constexpr int YW = 2 * (G::NW - 2);
        const int na = xw ? 3 : 2;
        int it = 0;
        for (; it + G::NW - 1 < nloc; ++it) {
            // steady state: always the conservative count.  (When an activation tile is among the younger DMAs the
            // exact count would be YW + 2 or + 3; the tile was requested up to NW-2 sub-steps ago and has long landed,
            // and the branch-free loop top measured 1.5-3 % faster.)
            wait_vmcnt<YW>();
            CP360_CLIP_STEP(true)
        }
        for (; it < nloc; ++it) {                   // drain: no weight refill
            const bool act_young = tap >= 1 && tap <= G::NW - 2 && it >= tap;
            const int young = 2 * min(G::NW - 2, nloc - 1 - it);
            // an activation tile requested during the drain may be younger than fewer weight stages
            // than the count assumes: wait for everything then (at most NW-2 sub-steps, once)
            if (act_young && young < YW) wait_vmcnt<0>();
            else wait_vmcnt_upto<YW + 3>(young + (act_young ? na : 0));
            CP360_CLIP_STEP(false)
        }
        if (LAG) CP360_CLIP_TAIL()
#undef CP360_CLIP_STEP
#undef CP360_CLIP_HEAD
#undef CP360_CLIP_TAIL
    }

    const int ml = lane & 15;
    if (!p.partial && (p.c_out % EPC == 0) && (p.ld_out % EPC == 0) && (p.out_coff % EPC == 0) &&
        (p.ld_res % EPC == 0)) {
        epilogue_lds<T, BN, BM, MJ, 512, G::LDS_BYTES>(p, lds, acc, n0, m0, wch0, wrow0, lane, tid, rows_valid);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wch0 + acc_chan(i, lane);
        if (n >= p.c_out) continue;
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int row = wrow0 + j * 16 + ml;
            if (row >= rows_valid) continue;
            const int m = m0 + row;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.partial) {
                store4(p.partial + ((size_t)split * p.M + m) * p.c_out + (p.slab_rows ? slab_col(n) : n), v);
            } else {
                if (p.bias) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (p.res) {
                    float r[4];
                    load4(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ld_res + n, r);
                    v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
                }
                if (p.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                store4(reinterpret_cast<T*>(p.out) + (size_t)m * p.ld_out + p.out_coff + n, v);
            }
        }
    }
}

// ------------------------------------------------------------------ pointwise 64 -> 64 (16-bit): layer1.0's conv1
// A 1x1 convolution of 64 channels into 64 is a stream: 128 bytes in and 128 out per pixel for 8 KFLOP.  The generic 64x256-tile
// kernel above stages it through LDS at 256 registers (two workgroups per CU): 91 us for the 1.2 M pixels of 64 frames = 3.4 TB/s.
// Here the whole filter (8 MFMA A fragments = 32 registers) stays in a wave's registers, a wave takes blocks of 16 pixels in a
// grid-stride loop - B fragments straight from global memory (a block's 2 KiB are contiguous), 8 MFMAs, bias + ReLU + one rounding,
// 16-byte stores - PWU blocks at a time so that 4 KiB of loads are in flight per wave at ~100 registers (4-5 waves per SIMD).
// Same products and the same order of the two K = 32 halves as conv_igemm_kernel: the same bits.  The traversal honours the launch
// order (ascending / descending pixel blocks over time): the next launch starts where this one's freshest lines are.
template <typename T>
__global__ __launch_bounds__(256) void conv_pw64_kernel(const ConvK p) {
    constexpr int PWU = 2;
    const int lane = threadIdx.x & 63, lrow = lane & 15, lchunk = lane >> 4;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const T* w = reinterpret_cast<const T*>(p.w);
    const T* in = reinterpret_cast<const T*>(p.in);
    T* out = reinterpret_cast<T*>(p.out);
    u32x4 a[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            a[i][kk] = *reinterpret_cast<const u32x4*>(w + (size_t)(i * 16 + lrow) * p.k_total + kk * 32 + lchunk * 8);
    float bb[2][8];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[pr][e] = p.bias ? p.bias[pr * 32 + lchunk * 8 + e] : 0.f;
    const int nblk = (p.M + 15) / 16;
    for (int pb0 = wave_g * PWU; pb0 < nblk; pb0 += nwaves * PWU) {
        u32x4 b[PWU][2];
#pragma unroll
        for (int u = 0; u < PWU; ++u) {
            const int pb = p.reverse ? nblk - 1 - (pb0 + u) : pb0 + u;  // (descending traversal under cp360_set_launch_order)
            const int m = min(max(pb, 0) * 16 + lrow, p.M - 1);         // (blocks past the end re-read a valid pixel; not stored)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                b[u][kk] = *reinterpret_cast<const u32x4*>(in + (size_t)m * p.pix_stride + kk * 32 + lchunk * 8);
        }
#pragma unroll
        for (int u = 0; u < PWU; ++u) {
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i) mma_chunk<T>(acc[i], a[i][kk], b[u][kk]);
            const int pb = p.reverse ? nblk - 1 - (pb0 + u) : pb0 + u;
            const int m = pb < 0 ? p.M : pb * 16 + lrow;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * pr][e] + bb[pr][e];
                    v[4 + e] = acc[2 * pr + 1][e] + bb[pr][4 + e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (m < p.M) *reinterpret_cast<u32x4*>(out + (size_t)m * p.ld_out + p.out_coff + pr * 32 + lchunk * 8) = pack8(v, T());
            }
        }
    }
}

constexpr int CLIP_JH = 0;
template <typename T, int MODE>
__global__ __launch_bounds__(512, 2) void conv_clip_kernel(const ConvK p) {
    typedef ClipGeom<MODE> G;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int n0, clip, split;
    {   // XCD-aware mapping, clips fastest: the workgroups of one (channel tile, split) share the weight stream
        const int nwg = p.nt * p.mt * p.splits;
        const int L = blockIdx.x, xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if (p.reverse) w = nwg - 1 - w;
        clip = w % p.mt;
        const int rest = w / p.mt;
        n0 = (rest % p.nt) * G::BN;
        split = rest / p.nt;
    }
    constexpr int JHC = CLIP_JH;                       // MFMA columns issued in the load half of a sub-step (see clip_body)
    if constexpr (MODE == 1) {
        if (wave < 4) clip_body<T, 8, JHC, false, 1>(p, lds, n0, clip, split, wave, lane, tid, wave * 64, 0);
        else          clip_body<T, 8, JHC, true, 1>(p, lds, n0, clip, split, wave, lane, tid, (wave - 4) * 64, 1);
    } else if constexpr (MODE == 2) {
        if (wave < 4) clip_body<T, 6, JHC, false, 2>(p, lds, n0, clip, split, wave, lane, tid, wave * 64, 0);
        else          clip_body<T, 6, JHC, true, 2>(p, lds, n0, clip, split, wave, lane, tid, (wave - 4) * 64, 1);
    } else {
        if (wave < 4) clip_body<T, 10, JHC, false, 0>(p, lds, n0, clip, split, wave, lane, tid, wave * 64, 0);
        else          clip_body<T, 9, JHC, true, 0>(p, lds, n0, clip, split, wave, lane, tid, (wave - 4) * 64, 1);
    }
}

// ------------------------------------------------------------------ split-K finish
template <typename T>
__global__ __launch_bounds__(256) void conv_finish_kernel(const float* __restrict__ partial, int splits,
                                                          const float* __restrict__ bias, const T* __restrict__ res,
                                                          int ld_res, T* __restrict__ out, int ld_out, int out_coff,
                                                          int M, int c_out, int relu, int slab_rows,
                                                          const float* __restrict__ extra) {
    const int q = c_out >> 2;
    const long long total = (long long)M * q;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / q), n = (int)(idx - (long long)m * q) * 4;
        const int nc = slab_rows ? slab_col(n) : n;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) {
            const float4 t = *reinterpret_cast<const float4*>(partial + ((size_t)s * M + m) * c_out + nc);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (extra) {     // one more f32 addend in the slabs' column order (cp360_conv_finish_add)
            const float4 t = *reinterpret_cast<const float4*>(extra + (size_t)m * c_out + nc);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (bias) {
            const float4 bb = *reinterpret_cast<const float4*>(bias + n);
            v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
        if (res) {
            float r[4];
            load4(res + (size_t)m * ld_res + n, r);
            v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
        }
        if (relu) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        store4(out + (size_t)m * ld_out + out_coff + n, v);
    }
}

// ------------------------------------------------------------------ ConvLSTM gate epilogue
// model/clstm.py:68-80: gates = [in | remember | out | cell] chunks of Hc channels.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ __launch_bounds__(256) void lstm_gates_kernel(const float* __restrict__ gp, int splits,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ c_prev, float* __restrict__ c_next,
                                                         T* __restrict__ h_out, int ld_h, int h_coff,
                                                         float* __restrict__ h_f32, int M, int Hc, int slab_rows,
                                                         const float* __restrict__ x_next,
                                                         const float* __restrict__ minmax, int x_coff, int P,
                                                         size_t clip_stride) {
    const int q = Hc >> 2, G = 4 * Hc;
    const long long total = (long long)M * q;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / q), j = (int)(idx - (long long)m * q) * 4;
        float g[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 bb = *reinterpret_cast<const float4*>(bias + k * Hc + j);
            g[k][0] = bb.x; g[k][1] = bb.y; g[k][2] = bb.z; g[k][3] = bb.w;
        }
        for (int s = 0; s < splits; ++s) {
            const float* base = gp + ((size_t)s * M + m) * G;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = k * Hc + j;
                const float4 t = *reinterpret_cast<const float4*>(base + (slab_rows ? slab_col(c) : c));
                g[k][0] += t.x; g[k][1] += t.y; g[k][2] += t.z; g[k][3] += t.w;
            }
        }
        const float4 cp = *reinterpret_cast<const float4*>(c_prev + (size_t)m * Hc + j);
        const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
        float cn[4], hn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ig = sigmoidf_(g[0][e]), fg = sigmoidf_(g[1][e]), og = sigmoidf_(g[2][e]);
            const float cg = tanhf(g[3][e]);
            cn[e] = fg * cpv[e] + ig * cg;
            hn[e] = og * tanhf(cn[e]);
        }
        *reinterpret_cast<float4*>(c_next + (size_t)m * Hc + j) = make_float4(cn[0], cn[1], cn[2], cn[3]);
        store4(h_out + (size_t)m * ld_h + h_coff + j, hn);
        if (h_f32) *reinterpret_cast<float4*>(h_f32 + (size_t)m * Hc + j) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        if (x_next) {   // the NEXT step's input frame, window-normalised (test_temporal.py:77), into the x half of the same pixel
            const int b = m / P, pix = m - b * P;
            const float mn = minmax[2 * b], den = minmax[2 * b + 1] - mn;
            const float4 v = *reinterpret_cast<const float4*>(x_next + (size_t)b * clip_stride + (size_t)pix * Hc + j);
            const float xn[4] = {(v.x - mn) / den, (v.y - mn) / den, (v.z - mn) / den, (v.w - mn) / den};
            store4(h_out + (size_t)m * ld_h + x_coff + j, xn);
        }
    }
}

// ------------------------------------------------------------------ weight packing
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                           T* __restrict__ packed, int c_out, int c_out_pad, int c_in,
                                                           int c_pad, int kh, int kw, int stem_mode, int chan_major,
                                                           const float* __restrict__ w2, const float* __restrict__ scale2,
                                                           int c_in2, int c_pad2) {
    const int taps = kh * kw;
    constexpr int BKS = 64 / (int)sizeof(T);          // elements per 64-byte sub-step
    const int ktot = taps * c_pad + c_pad2;           // c_pad2 > 0: the second source's 1x1 filter behind the taps
    const long long total = (long long)c_out_pad * ktot;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        int c, tap, n;
        if (c_pad2 > 0) {   // [n][tap][c_pad] ... [c_pad2]  (tap-major only)
            const int k = (int)(idx % ktot);
            n = (int)(idx / ktot);
            tap = min(k / c_pad, taps);            // taps = the second source's region (c_pad2 wide)
            c = k - tap * c_pad;
        } else if (chan_major) {   // [n / 256][c / BKS][tap][n % 256][BKS]   (c_pad is a multiple of BKS here): the
            // 256 rows x 64 bytes one workgroup needs per sub-step are ONE contiguous 16 KiB block
            const int e = (int)(idx % BKS);
            long long t = idx / BKS;
            const int r = (int)(t % 256);
            t /= 256;
            tap = (int)(t % taps);
            t /= taps;
            const int cpb = c_pad / BKS;
            c = (int)(t % cpb) * BKS + e;
            n = (int)(t / cpb) * 256 + r;
        } else {            // [n][tap][c_pad]
            c = (int)(idx % c_pad);
            const long long t = idx / c_pad;
            tap = (int)(t % taps);
            n = (int)(t / taps);
        }
        // n is the packed row; its channel (acc_chan): row 32q + 16b + 4g + e <- channel 32q + 8g + 4b + e
        n = (n & ~31) + ((n >> 2) & 3) * 8 + ((n >> 4) & 1) * 4 + (n & 3);
        float v = 0.f;
        if (n < c_out && tap >= taps) {                // second source: [c_out, c_in2] filter
            if (c < c_in2) v = w2[(size_t)n * c_in2 + c] * (scale2 ? scale2[n] : 1.f);
        } else if (n < c_out) {
            if (stem_mode) {   // desc kh=7, kw=1, c_in=32: k = kx*4 + ch of a [c_out,3,7,7] filter
                const int kx = c >> 2, ch = c & 3, ky = tap;
                if (c < c_in && kx < 7 && ch < 3) v = w[(((size_t)n * 3 + ch) * 7 + ky) * 7 + kx];
            } else if (c < c_in) {
                const int ky = tap / kw, kx = tap - ky * kw;
                v = w[(((size_t)n * c_in + c) * kh + ky) * kw + kx];
            }
            if (scale) v *= scale[n];
        }
        if constexpr (sizeof(T) == 4)
            packed[idx] = v;
        else if constexpr (__is_same(T, f16_raw))
            packed[idx] = (f16_raw)v;
        else
            packed[idx] = f32_to_bf16(v);
    }
}

// ------------------------------------------------------------------ host side
static int elem_bytes(int dtype) { return dtype == CP360_F32 ? 4 : ((dtype == CP360_BF16 || dtype == CP360_F16) ? 2 : 0); }
static int bk_of(int dtype) { return 128 / elem_bytes(dtype); }
// K padding per tap: a 128-byte step for the tap-major layout, a 64-byte sub-step for the channel-major one
static int c_pad_of(const cp360_conv_desc* d) {
    const int unit = d->clip_resident ? bk_of(d->dtype) / 2 : bk_of(d->dtype);
    return (d->c_in + unit - 1) / unit * unit;
}
static int round_up(int a, int b) { return (a + b - 1) / b * b; }
static int c_pad2_of(const cp360_conv_desc* d) { return d->c_in2 > 0 ? round_up(d->c_in2, bk_of(d->dtype)) : 0; }

// What conv_small_kernel (csrc/conv_small.hip) requires of a descriptor - ONE predicate for the planner (small_eligible) and
// for a caller that forces the tile (tile_px == 6464, check_desc): shape limits -> UNSUPPORTED, alignment of its 16-byte
// output / residual accesses in the 16-bit types -> ALIGN.
static int small_shape_status(const cp360_conv_desc* d) {
    if (d->clip_resident || d->c_out % 8 != 0) return CP360_ERR_UNSUPPORTED;
    if (d->kh * d->kw + (d->c_in2 > 0 ? 1 : 0) > CP360_SMALL_MAX_TAPS) return CP360_ERR_UNSUPPORTED;
    // 32-bit byte offsets inside the tensors and inside a tile's 64 weight rows
    const long long es = elem_bytes(d->dtype);
    if ((long long)d->n_img * d->h_in * d->w_in * d->pix_stride * es >= (1LL << 32)) return CP360_ERR_UNSUPPORTED;
    if (d->c_in2 > 0 && (long long)d->n_img * d->h_in2 * d->w_in2 * d->pix_stride2 * es >= (1LL << 32)) return CP360_ERR_UNSUPPORTED;
    if (64LL * ((long long)d->kh * d->kw * c_pad_of(d) + c_pad2_of(d)) * es >= (1LL << 32)) return CP360_ERR_UNSUPPORTED;
    if (d->dtype != CP360_F32 && (d->ld_out % 8 != 0 || d->out_coff % 8 != 0 || d->ld_res % 8 != 0)) return CP360_ERR_ALIGN;
    return CP360_OK;
}

static int check_desc(const cp360_conv_desc* d) {
    if (!d) return CP360_ERR_NULL;
    if (d->dtype != CP360_F32 && d->dtype != CP360_BF16 && d->dtype != CP360_F16) return CP360_ERR_BAD_DTYPE;
    if (d->n_img <= 0 || d->h_in <= 0 || d->w_in <= 0 || d->c_in <= 0 || d->kh <= 0 || d->kw <= 0 || d->sy <= 0 ||
        d->sx <= 0 || d->h_out <= 0 || d->w_out <= 0 || d->c_out <= 0 || d->splits < 1 || d->pix_stride <= 0)
        return CP360_ERR_BAD_SHAPE;
    const int epc = 16 / elem_bytes(d->dtype);
    if (d->tile_px != 0 && d->tile_px != 64 && d->tile_px != 128 && d->tile_px != 129 && d->tile_px != 160 && d->tile_px != 256 &&
        d->tile_px != 304 && d->tile_px != 6464)
        return CP360_ERR_BAD_SHAPE;
    if (d->tile_px == 160 && d->dtype == CP360_F32) return CP360_ERR_UNSUPPORTED;      // the 160-pixel ring tile: 16-bit types
    if (d->tile_px == 6464) {                                  // forced small tile: exactly what the planner requires of it
        const int rc = small_shape_status(d);
        if (rc) return rc;
    }
    if (d->tile_px == 129 && d->dtype == CP360_F32) return CP360_ERR_UNSUPPORTED;   // 16-bit types only (128-VGPR budget)
    if (d->slab_rows != 0 && d->slab_rows != 1) return CP360_ERR_BAD_SHAPE;
    if (d->slab_rows && d->c_out % 32 != 0) return CP360_ERR_ALIGN;
    if (d->clip_resident != 0 && d->clip_resident != 1) return CP360_ERR_BAD_SHAPE;
    // clip-resident kernel: CubePad(1) + 3x3 stride 1 on faces whose cube (6 n^2 pixels) fits one 304-row tile, on 8x8 faces
    // (half a cube per tile, the cube resident) and on 16x16 faces (one face + its ring per tile)
    if (d->clip_resident && !(d->pad_mode == 1 && d->pad == 1 && d->kh == 3 && d->kw == 3 && d->sy == 1 && d->sx == 1 &&
                              d->h_in == d->w_in && (6 * d->h_in * d->w_in <= 304 || d->h_in == 16 || d->h_in == 8) && d->c_out >= 256 &&
                              d->pix_stride >= d->c_in && d->tile_px == 0))
        return CP360_ERR_UNSUPPORTED;
    if (d->c_in2 < 0) return CP360_ERR_BAD_SHAPE;
    if (d->c_in2 > 0) {    // second source: ring kernels only (c_out >= 256), no clip-resident / stem forms
        if (d->clip_resident || d->c_out < 256 || d->pix_stride < d->c_in) return CP360_ERR_UNSUPPORTED;
        if (d->c_in2 % epc != 0) return CP360_ERR_ALIGN;
        if (d->pix_stride2 < d->c_in2 || d->h_in2 <= 0 || d->w_in2 <= 0 || d->sy2 <= 0 || d->sx2 <= 0 ||
            (d->h_out - 1) * d->sy2 >= d->h_in2 || (d->w_out - 1) * d->sx2 >= d->w_in2)
            return CP360_ERR_BAD_SHAPE;
        if ((long long)d->n_img * d->h_in2 * d->w_in2 * d->pix_stride2 >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    }
    if (d->c_in % epc != 0 || d->c_out % 4 != 0 || d->ld_out % 4 != 0 || d->out_coff % 4 != 0 || d->ld_res % 4 != 0)
        return CP360_ERR_ALIGN;
    if (d->ld_out < d->c_out + d->out_coff) return CP360_ERR_BAD_SHAPE;
    if (d->pad_mode) {
        if (d->n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
        if (d->h_in != d->w_in) return CP360_ERR_NOT_SQUARE;
        if (d->pad < 0 || d->pad > d->h_in) return CP360_ERR_BAD_SHAPE;
    }
    const int pad2 = d->pad_mode ? 2 * d->pad : 0;
    if ((d->h_out - 1) * d->sy + d->kh > d->h_in + pad2) return CP360_ERR_BAD_SHAPE;
    // pix_stride < c_in is the stem form: one tap reads c_in/pix_stride neighbouring pixels
    const int span = d->pix_stride >= d->c_in ? 1 : (d->c_in + d->pix_stride - 1) / d->pix_stride;
    if (span > 1 && d->pad_mode) return CP360_ERR_UNSUPPORTED;
    if ((d->w_out - 1) * d->sx + (d->kw - 1) + span > d->w_in + pad2) return CP360_ERR_BAD_SHAPE;
    if ((long long)d->n_img * d->h_in * d->w_in * d->pix_stride >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    return CP360_OK;
}

extern "C" size_t cp360_conv_packed_bytes(const cp360_conv_desc* d) {
    if (check_desc(d)) return 0;
    return (size_t)round_up(d->c_out, 256) * ((size_t)d->kh * d->kw * c_pad_of(d) + c_pad2_of(d)) * elem_bytes(d->dtype);
}

extern "C" size_t cp360_conv_partial_bytes(const cp360_conv_desc* d) {
    if (check_desc(d) || d->splits <= 1) return 0;
    return (size_t)d->splits * d->n_img * d->h_out * d->w_out * d->c_out * sizeof(float);
}

// ---- launch planning: tile shape + split-K from a small cost model (microseconds).
// Calibrated on MI355X with tools/bench_conv.py: one K step (128 bytes of K per tile row)
// of a resident workgroup costs about 2.2 us for the 256x256 LDS-DMA tile, 1.25 us for
// 256x128 (one 8-wave workgroup per CU) and 1.5 us for the 4-wave tiles (two per CU); f32
// steps are MFMA-bound and about 3x longer.  A launch runs ceil(workgroups / slots) rounds
// of (fixed cost + steps x step cost); split-K adds the f32 slab round trip through HBM.
struct ConvPlan {
    int bn, bm, slots, splits;
    double cost;
    long long wgs = 0;           // workgroups of the launch (tiles x splits)
};

static ConvPlan plan_candidate(const cp360_conv_desc* d, int bn, int bm, int slots, double t_step) {
    const long long M = (long long)d->n_img * d->h_out * d->w_out;
    const long long wgs = ((d->c_out + bn - 1) / bn) * ((M + bm - 1) / bm);
    const int bk = bk_of(d->dtype);
    const int nsteps = d->kh * d->kw * (round_up(d->c_in, bk) / bk) + c_pad2_of(d) / bk;
    if (d->dtype == CP360_F32) t_step *= 3.0;
    const double t_fixed = 4.0;
    ConvPlan best{bn, bm, slots, 1, 0.0};
    for (int s = 1; s <= 32; ++s) {
        if (s > 1 && nsteps / s < 8) break;
        const long long rounds = (wgs * s + slots - 1) / slots;
        const int steps = (nsteps + s - 1) / s;
        double cost = (double)rounds * (t_fixed + steps * t_step);
        if (s > 1) cost += 4.0 + (double)s * (double)M * d->c_out * 8.0 / 4.0e6;   // slab write + read at ~4 TB/s
        if (s == 1 || cost < best.cost) {
            best.splits = s;
            best.cost = cost;
            best.wgs = wgs * s;
        }
    }
    return best;
}

// conv_small.hip: 64 x 64 tiles, 8 waves, two workgroups per CU.  Measured on MI355X (tools/exp_small.sh, one frame): an f32
// K step (128 bytes per tile row: 16 KB into the CU, 1024 MFMA cycles per SIMD) takes 0.70 us per workgroup on a CU - the
// CU's ~10 bytes / clock of global-load throughput, not the matrix pipe (0.43 us), is what a step waits for - so a launch
// costs about (workgroups per CU) x steps x 0.7 us + a fixed 3 us per wave of 1024 workgroups.
static bool small_eligible(const cp360_conv_desc* d) {
    static const int small = []() {
        const char* e = getenv("CP360_SMALL");               // A/B switch: 0 = never, 1 = f32 only (default), 2 = every dtype
        return e ? atoi(e) : 1;
    }();
    if (!small || (d->dtype != CP360_F32 && small < 2)) return false;
    return small_shape_status(d) == CP360_OK;
}

static ConvPlan plan_small(const cp360_conv_desc* d) {
    const long long M = (long long)d->n_img * d->h_out * d->w_out;
    const long long tiles = ((d->c_out + 63) / 64) * ((M + 63) / 64);
    const int bk = bk_of(d->dtype);
    const int nsteps = d->kh * d->kw * (round_up(d->c_in, bk) / bk) + c_pad2_of(d) / bk;
    const double t_step = d->dtype == CP360_F32 ? 0.7 : 0.35;
    ConvPlan best{64, 64, 1024, 1, 0.0};
    for (int s = 1; s <= 32; ++s) {
        if (s > 1 && nsteps / s < 4) break;
        const long long W = tiles * s;
        const int steps = (nsteps + s - 1) / s;
        const long long per_cu = d->dtype == CP360_F32 ? (W + 255) / 256 : (W + 1023) / 1024;
        double cost = (double)((W + 1023) / 1024) * 3.0 + (double)per_cu * steps * t_step;
        if (s > 1) cost += 4.0 + (double)s * (double)M * d->c_out * 8.0 / 4.0e6;
        if (s == 1 || cost < best.cost) {
            best.splits = s;
            best.cost = cost;
            best.wgs = W;
        }
    }
    return best;
}

static ConvPlan plan_big(const cp360_conv_desc* d);
// pixel tiles of the clip-resident kernel: a face at 16x16, half a cube at 8x8, otherwise a cube
static int clip_tiles(const cp360_conv_desc* d) { return d->h_in == 16 ? d->n_img : d->h_in == 8 ? d->n_img / 3 : d->n_img / 6; }

static ConvPlan plan_of(const cp360_conv_desc* d) {
    ConvPlan best = plan_big(d);
    if (d->tile_px == 6464) return plan_small(d);
    if (d->tile_px == 0 && small_eligible(d)) {
        // the small tiles when the model says so - or when the best big-tile launch cannot even give every CU a workgroup
        // (the two cost models are calibrated separately, and the big-tile one is optimistic exactly there: layer1's conv3
        // + downsample of ONE frame: modelled 19 us on 147 workgroups, measured 47; the small tiles: 21)
        const ConvPlan sm = plan_small(d);
        if (sm.cost < best.cost || best.wgs < 256) best = sm;
    }
    return best;
}

static ConvPlan plan_big(const cp360_conv_desc* d) {
    if (d->clip_resident) {
        // one 256-channel x clip tile per workgroup, 64-byte sub-steps; about 0.9 us per sub-step
        const int wgs = ((d->c_out + 255) / 256) * clip_tiles(d);
        const int nsub = 9 * (c_pad_of(d) / (bk_of(d->dtype) / 2));
        const double t_sub = (d->dtype == CP360_F32 ? 2.7 : 0.9) * (d->h_in == 8 ? 0.7 : 1.0);   // (6 + 6 instead of 10 + 9 pixel blocks)
        const long long M = (long long)d->n_img * d->h_out * d->w_out;
        ConvPlan best{256, 304, 256, 1, 0.0};
        for (int s = 1; s <= 32; ++s) {
            if (s > 1 && nsub / s < 16) break;
            const int rounds = (wgs * s + 255) / 256;
            double cost = (double)rounds * (4.0 + (double)((nsub + s - 1) / s) * t_sub);
            if (s > 1) cost += 4.0 + (double)s * (double)M * d->c_out * 8.0 / 4.0e6;
            if (s == 1 || cost < best.cost) {
                best.splits = s;
                best.cost = cost;
                best.wgs = (long long)wgs * s;
            }
        }
        return best;
    }
    if (d->c_out <= 64) return plan_candidate(d, 64, 256, 512, 1.5);
    if (d->c_out < 256) return plan_candidate(d, 128, 128, 512, 1.5);
    ConvPlan best = plan_candidate(d, 256, 256, 256, 2.2);
    const ConvPlan b = plan_candidate(d, 256, 128, 256, 1.25);
    if (b.cost < best.cost) best = b;
    static const int no304 = []() {
        const char* e = getenv("CP360_NO304");             // A/B switch for tools/bench_conv.py
        return e ? atoi(e) : 0;
    }();
    const ConvPlan c = plan_candidate(d, 256, 304, 256, 2.2 * 304.0 / 256.0);   // 19 pixel blocks: 7x7 cube faces
    if (!no304 && c.cost < best.cost) best = c;
    // 160-pixel tile (16-bit types): wins where the larger tiles cannot give every CU a workgroup
    static const int use160 = []() {
        const char* e = getenv("CP360_TILE160");           // A/B switch
        return e ? atoi(e) : 1;
    }();
    if (use160 && d->dtype != CP360_F32 && d->c_in2 == 0) {
        const ConvPlan t160 = plan_candidate(d, 256, 160, 256, 2.2 * 176.0 / 256.0);
        if (t160.cost < best.cost) best = t160;
    }
    // short-K 1x1 convolutions (HBM-bound): 256x128 tile with two workgroups per CU (pixel tile id 129)
    static const int ring2 = []() {
        const char* e = getenv("CP360_RING2");             // A/B switch for tools/bench_conv.py
        return e ? atoi(e) : 1;
    }();
    // (K of at least two 128-byte steps: at K = 64 elements the one-workgroup 256x304 tile measured faster)
    const int kbytes = (d->c_in + d->c_in2) * elem_bytes(d->dtype);
    static const int ring2_kmax = []() {
        // K of at most four 128-byte steps.  (Up to 1 KiB it used to take the HBM-bound 1x1 convolutions of layers 2-4;
        // measured per launch, 64 frames: l2.0 conv3 + downsample (K = 768 B) 232 -> 199 us on the 256x304 ring, l3.0
        // conv1 (1 KiB) 153 -> 124, l4 conv3 (1 KiB) 84 -> 71; at 512 B - l3 conv3 - the two tie.)  A/B switch.
        const char* e = getenv("CP360_RING2_KMAX");
        return e ? atoi(e) : 512;
    }();
    if (ring2 && d->dtype != CP360_F32 && d->kh * d->kw == 1 && kbytes >= 256 && kbytes <= ring2_kmax) {
        ConvPlan r2 = plan_candidate(d, 256, 128, 512, 1.25);
        r2.bm = 129;
        if (r2.cost < best.cost) best = r2;
    }
    // the 3-stage 256x128 DMA kernel has no second-source loader: its ring equivalents take over
    if (d->c_in2 > 0 && best.bm == 128) best.bm = d->dtype == CP360_F32 ? 256 : 129;
    return best;
}

// Tile geometry for a given (caller-chosen) split count: same candidates, splits fixed.
static void tile_of(const cp360_conv_desc* d, int* bn, int* bm, int* slots) {
    ConvPlan pl = plan_of(d);      // a caller-chosen split count keeps the tile the model prefers at ITS best split
    *bn = pl.bn;
    *bm = pl.bm;
    *slots = pl.slots;
    if (pl.bm == 64) return;                         // conv_small.hip (chosen by the model or forced by tile_px 6464)
    if (d->c_out >= 256 && d->tile_px == 64) {       // forced: the 4-wave 128x128 kernel
        *bn = 128;
        *bm = 128;
        *slots = 512;
    } else if (d->c_out >= 256 && d->tile_px) {
        *bn = 256;
        *bm = d->tile_px;
        *slots = 256;
    }
}

extern "C" int cp360_conv_suggest_splits(const cp360_conv_desc* d) {
    cp360_conv_desc t = *d;
    t.splits = 1;
    if (check_desc(&t)) return 1;
    return plan_of(&t).splits;
}

// For a caller that holds BOTH weight layouts of a CubePad(1) + 3x3 convolution on small faces (layer4's conv2): should this
// launch take the clip-resident kernel (1) or the tap-major path (0)?  The clip-resident tile is 256 channels x a whole cube:
// with few cubes and few channels (one frame, 512 channels: 2 tiles) it cannot fill the chip even with split-K, and the
// 64 x 64 tiles of conv_small.hip win.  Only that comparison is made: where the small-tile kernel is not eligible (16-bit
// types by default) the answer is 1, as before round 4.
extern "C" int cp360_conv_prefer_clip(const cp360_conv_desc* d) {
    if (!d) return 0;
    cp360_conv_desc c = *d;
    c.splits = 1;
    c.tile_px = 0;
    c.clip_resident = 1;
    if (check_desc(&c)) return 0;
    cp360_conv_desc t = c;
    t.clip_resident = 0;
    static const int force = []() { const char* e = getenv("CP360_PREFER_CLIP"); return e ? atoi(e) : -1; }();   // A/B switch
    if (force >= 0 && !check_desc(&t)) return force ? 1 : 0;
    if (check_desc(&t) || !small_eligible(&t)) return 1;
    return plan_big(&c).cost <= plan_small(&t).cost ? 1 : 0;
}

extern "C" int cp360_conv_plan_describe(const cp360_conv_desc* d, char* buf, size_t cap) {
    if (!d || !buf || cap == 0) return CP360_ERR_NULL;
    cp360_conv_desc t = *d;
    t.splits = 1;
    int rc = check_desc(&t);
    if (rc) return rc;
    const ConvPlan pl = plan_of(&t);
    int bn = 0, bm = 0, slots = 0;
    tile_of(&t, &bn, &bm, &slots);
    const long long M = (long long)t.n_img * t.h_out * t.w_out;
    const char* name;
    long long wgs;
    if (t.clip_resident) {
        name = t.h_in == 16 ? "conv_clip 256 ch x one 16x16 face (activations LDS-resident)"
               : t.h_in == 8 ? "conv_clip 256 ch x half a cube of 8x8 faces (the cube LDS-resident)"
                             : "conv_clip 256 ch x one cube (activations LDS-resident)";
        wgs = (long long)((t.c_out + 255) / 256) * clip_tiles(&t);
    } else if (bm == 64) {
        name = "conv_small 64 ch x 64 px (8 waves)";
        wgs = (long long)((t.c_out + 63) / 64) * ((M + 63) / 64);
    } else if (t.c_out < 256 || bn != 256) {
        const bool narrow = t.c_out <= 64;
        name = narrow ? "conv_igemm 64 ch x 256 px (4 waves)" : "conv_igemm 128 ch x 128 px (4 waves)";
        wgs = narrow ? (long long)((t.c_out + 63) / 64) * ((M + 255) / 256) : (long long)((t.c_out + 127) / 128) * ((M + 127) / 128);
    } else {
        const int px = bm == 129 ? 128 : bm;
        name = bm == 129 ? "conv_igemm_ring2 256 ch x 128 px (two workgroups per CU)"
               : bm == 304 ? "conv_igemm_ring 256 ch x 304 px" : bm == 160 ? "conv_igemm_ring 256 ch x 160 px" : bm >= 256 ? "conv_igemm_ring 256 ch x 256 px" : "conv_igemm_dma 256 ch x 128 px";
        wgs = (long long)((t.c_out + 255) / 256) * ((M + px - 1) / px);
    }
    const int n = snprintf(buf, cap, "%s, %lld workgroups x split-K %d%s, model %.0f us", name, wgs * pl.splits, pl.splits,
                           t.c_in2 > 0 ? " (+ second source)" : "", pl.cost);
    return n < 0 ? CP360_ERR_BAD_SHAPE : (n >= (int)cap ? (int)cap - 1 : n);
}

static int pack_weights_impl(const cp360_conv_desc* d, const float* w_oihw, const float* scale, const float* w2,
                             const float* scale2, void* packed, int stem_mode, void* stream);

extern "C" int cp360_conv_pack_weights(const cp360_conv_desc* d, const float* w_oihw, const float* scale, void* packed,
                                       int stem_mode, void* stream) {
    if (d && d->c_in2 > 0) return CP360_ERR_NULL;              // a second source needs its filter: cp360_conv_pack_weights2
    return pack_weights_impl(d, w_oihw, scale, nullptr, nullptr, packed, stem_mode, stream);
}

extern "C" int cp360_conv_pack_weights2(const cp360_conv_desc* d, const float* w_oihw, const float* scale,
                                        const float* w2_oi, const float* scale2, void* packed, void* stream) {
    if (!d || d->c_in2 <= 0) return CP360_ERR_BAD_SHAPE;
    if (!w2_oi) return CP360_ERR_NULL;
    return pack_weights_impl(d, w_oihw, scale, w2_oi, scale2, packed, 0, stream);
}

static int pack_weights_impl(const cp360_conv_desc* d, const float* w_oihw, const float* scale, const float* w2,
                             const float* scale2, void* packed, int stem_mode, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w_oihw || !packed) return CP360_ERR_NULL;
    if (stem_mode && !(d->kh == 7 && d->kw == 1 && d->c_in == 32)) return CP360_ERR_UNSUPPORTED;
    if (stem_mode && d->clip_resident) return CP360_ERR_UNSUPPORTED;
    const int c_pad = c_pad_of(d);
    const int c_out_pad = round_up(d->c_out, 256);
    const int c_pad2 = c_pad2_of(d);
    const long long total = (long long)c_out_pad * ((long long)d->kh * d->kw * c_pad + c_pad2);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == CP360_F32)
        hipLaunchKernelGGL((pack_weights_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, w_oihw, scale,
                           (float*)packed, d->c_out, c_out_pad, d->c_in, c_pad, d->kh, d->kw, stem_mode, d->clip_resident,
                           w2, scale2, d->c_in2, c_pad2);
    else if (d->dtype == CP360_F16)
        hipLaunchKernelGGL((pack_weights_kernel<f16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, w_oihw, scale,
                           (f16_raw*)packed, d->c_out, c_out_pad, d->c_in, c_pad, d->kh, d->kw, stem_mode, d->clip_resident,
                           w2, scale2, d->c_in2, c_pad2);
    else
        hipLaunchKernelGGL((pack_weights_kernel<bf16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, w_oihw, scale,
                           (bf16_raw*)packed, d->c_out, c_out_pad, d->c_in, c_pad, d->kh, d->kw, stem_mode, d->clip_resident,
                           w2, scale2, d->c_in2, c_pad2);
    CP360_CHECK_HIP();
    return CP360_OK;
}

template <typename T, int WN, int WM>
static void launch_conv(ConvK& k, hipStream_t st) {
    k.nt = (k.c_out + WN * 64 - 1) / (WN * 64);
    k.mt = (k.M + WM * 64 - 1) / (WM * 64);
    // share whichever operand panel is larger through the XCD's L2
    k.m_fast = ((long long)k.c_out * k.k_total > (long long)k.M * k.kh * k.kw * k.c_in) ? 1 : 0;
    dim3 grid((unsigned)(k.nt * k.mt * k.splits), 1, 1);
    hipLaunchKernelGGL((conv_igemm_kernel<T, WN, WM>), grid, dim3(256), 0, st, k);
}

extern "C" int cp360_conv_forward(const cp360_conv_desc* d, const void* in, const void* packed_w, const float* bias,
                                  const void* residual, void* out, float* partial, void* stream) {
    return cp360_conv_forward2(d, in, nullptr, packed_w, bias, residual, out, partial, stream);
}

extern "C" int cp360_conv_forward2(const cp360_conv_desc* d, const void* in, const void* in2, const void* packed_w,
                                   const float* bias, const void* residual, void* out, float* partial, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!in || !packed_w) return CP360_ERR_NULL;
    if ((d->c_in2 > 0) != (in2 != nullptr)) return CP360_ERR_NULL;
    if (d->splits > 1 && !partial) return CP360_ERR_NULL;
    if (d->splits == 1 && !out && !partial) return CP360_ERR_NULL;
    if (residual && d->ld_res < d->c_out) return CP360_ERR_BAD_SHAPE;
    ConvK k;
    k.in = (const unsigned char*)in;
    k.w = (const unsigned char*)packed_w;
    k.out = (unsigned char*)out;
    k.bias = bias;
    k.res = (const unsigned char*)residual;
    k.partial = (d->splits > 1 || !out) ? partial : nullptr;   // out == NULL: raw f32 sums [splits, M, c_out]
    k.n_img = d->n_img; k.h_in = d->h_in; k.w_in = d->w_in; k.c_in = d->c_in; k.pix_stride = d->pix_stride;
    k.kh = d->kh; k.kw = d->kw; k.sy = d->sy; k.sx = d->sx; k.h_out = d->h_out; k.w_out = d->w_out;
    k.c_out = d->c_out; k.pad_mode = d->pad_mode; k.pad = d->pad; k.ld_out = d->ld_out; k.out_coff = d->out_coff;
    k.ld_res = d->ld_res; k.relu = d->relu; k.splits = d->splits;
    k.hw_out = d->h_out * d->w_out;
    k.M = d->n_img * k.hw_out;
    const int bk = bk_of(d->dtype);
    k.c_pad = c_pad_of(d);
    k.steps_per_tap = k.c_pad / bk;                              // 128-byte steps (tap-major layout only)
    k.in2 = (const unsigned char*)in2;
    k.c_in2 = d->c_in2; k.c_pad2 = c_pad2_of(d); k.pix_stride2 = d->pix_stride2; k.h_in2 = d->h_in2; k.w_in2 = d->w_in2;
    k.sy2 = d->sy2; k.sx2 = d->sx2;
    k.ntap = d->kh * d->kw;
    k.nsteps = d->kh * d->kw * k.steps_per_tap + k.c_pad2 / bk;
    k.steps_per_split = (k.nsteps + d->splits - 1) / d->splits;
    k.k_total = d->kh * d->kw * k.c_pad + k.c_pad2;
    k.slab_rows = d->slab_rows;
    k.reverse = cp360_launch_reverse();
    k.clip_rows = 6 * d->h_out * d->w_out;
    k.nsub = k.k_total / (bk / 2);
    k.sub_per_split = (k.nsub + d->splits - 1) / d->splits;
    {   // weight streams that cannot stay in the 256 MB Infinity Cache (the ConvLSTM's 148 - 590 MB per launch): non-temporal past
        // the first CP360_CLIP_WPIN (default 8) sub-steps of every workgroup; small filters (layer4: 4.7 MB) keep the default policy
        static const int wpin = []() { const char* e = getenv("CP360_CLIP_WPIN"); return e ? atoi(e) : 8; }();
        const size_t wbytes = (size_t)((d->c_out + 255) / 256) * 256 * k.k_total * (d->dtype == CP360_F32 ? 4 : 2);
        k.w_pin = wbytes >= ((size_t)96 << 20) ? wpin : 0x7fffffff;
    }
    // Epilogue of the ring kernels: LDS-staged full-line stores by default; the direct 16-byte-piece
    // epilogue measured the same on the big tiles (within 1 %) and is what the two-workgroups-per-CU
    // short-K kernel uses (no LDS left for a staged tile there).  CP360_EPI=1 selects it everywhere.
    static const int epi_mode = []() {
        const char* e = getenv("CP360_EPI");
        return e ? atoi(e) : 0;
    }();
    const bool epi_ok = d->c_out % 8 == 0 && d->ld_out % 8 == 0 && d->out_coff % 8 == 0 && d->ld_res % 8 == 0;
    k.epi_direct = (epi_mode && epi_ok) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (d->clip_resident) {
        const int mode = d->h_in == 16 ? 1 : d->h_in == 8 ? 2 : 0;   // tile = a face + its ring / half a cube / a cube
        k.nt = (k.c_out + 255) / 256;
        k.mt = clip_tiles(d);
        k.m_fast = 1;
        dim3 grid((unsigned)(k.nt * k.mt * k.splits), 1, 1);
#define CP360_CLIP(TT)                                                                                    \
        {                                                                                                     \
            if (mode == 1)      hipLaunchKernelGGL((conv_clip_kernel<TT, 1>), grid, dim3(512), 0, st, k);     \
            else if (mode == 2) hipLaunchKernelGGL((conv_clip_kernel<TT, 2>), grid, dim3(512), 0, st, k);     \
            else                hipLaunchKernelGGL((conv_clip_kernel<TT, 0>), grid, dim3(512), 0, st, k);     \
        }
        if (d->dtype == CP360_F32) CP360_CLIP(float)
        else if (d->dtype == CP360_F16) CP360_CLIP(f16_raw)
        else CP360_CLIP(bf16_raw)
#undef CP360_CLIP
        CP360_CHECK_HIP();
        return CP360_OK;
    }
    const bool narrow = d->c_out <= 64;
    int bn_ = 0, bm_ = 0, slots_ = 0;
    tile_of(d, &bn_, &bm_, &slots_);
    if (bm_ == 64) {                                           // small-M launches: 64 x 64 tiles (conv_small.hip)
        cp360_launch_conv_small(k, d->dtype, st);
        CP360_CHECK_HIP();
        return CP360_OK;
    }
    // pointwise 64 -> 64 in a 16-bit type (layer1.0's conv1): the streaming kernel; CP360_PW64=0 keeps the generic tile kernel (A/B)
    static const int use_pw64 = []() { const char* e = getenv("CP360_PW64"); return e ? atoi(e) : 1; }();
    if (use_pw64 && d->dtype != CP360_F32 && d->kh == 1 && d->kw == 1 && d->sy == 1 && d->sx == 1 && d->pad == 0 && d->c_in == 64 &&
        d->c_out == 64 && d->c_in2 == 0 && d->splits <= 1 && !residual && d->tile_px == 0 && d->pix_stride % 8 == 0 &&
        d->ld_out % 8 == 0 && d->out_coff % 8 == 0 && k.M >= 4096 && k.out != nullptr && k.partial == nullptr) {
        long long blocks = ((long long)(k.M + 15) / 16 + 7) / 8;                 // two passes of PWU blocks per wave at most ...
        if (blocks > 256 * 8) blocks = 256 * 8;                                    // ... and a persistent grid beyond 8 workgroups per CU
        if (d->dtype == CP360_F16) hipLaunchKernelGGL((conv_pw64_kernel<f16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((conv_pw64_kernel<bf16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, k);
        CP360_CHECK_HIP();
        return CP360_OK;
    }
    if (d->c_out < 256) bn_ = bm_ = slots_ = 0;
    const bool wide = d->c_out >= 256 && bn_ == 256;           // else: the 4-wave 128x128 kernel, two workgroups per CU
    if (d->c_in2 > 0) {                                        // second source: ring kernels only
        if (!wide) return CP360_ERR_UNSUPPORTED;
        if (bm_ == 128) bm_ = d->dtype == CP360_F32 ? 256 : 129;
    }
    if (wide) {
        // 256x256 tiles carry 1.5x the flops per byte brought into the CU; use them unless the
        // pixel count pads badly (small-M launches) - then 256x128
        const bool big = bm_ >= 256;
        const int bm = bm_ == 129 ? 128 : bm_;
        if (bm_ == 129 && epi_ok) k.epi_direct = 1;
        k.nt = (k.c_out + 255) / 256;
        k.mt = (k.M + bm - 1) / bm;
        k.m_fast = ((long long)k.c_out * k.k_total > (long long)k.M * k.kh * k.kw * k.c_in) ? 1 : 0;
        dim3 grid((unsigned)(k.nt * k.mt * k.splits), 1, 1);
        static const int use_ring = []() {
            const char* e = getenv("CP360_RING");           // A/B switch for tools/bench_conv.py
            return e ? atoi(e) : 1;
        }();
        // the DMA kernels have no second-source loader (their K loop would run the extra tap on the first tensor):
        // a second-source descriptor always takes a ring kernel, whatever CP360_RING says
        const bool ring = use_ring || d->c_in2 > 0;
        if (d->c_in2 > 0 && !big && bm_ != 129) return CP360_ERR_UNSUPPORTED;
#define CP360_WIDE(TT)                                                                                     \
        {                                                                                                      \
            if (bm == 304) hipLaunchKernelGGL((conv_igemm_ring_kernel<TT, 304>), grid, dim3(512), 0, st, k);   \
            else if (bm == 160) { if constexpr (sizeof(TT) == 2) hipLaunchKernelGGL((conv_igemm_ring_kernel<TT, 160>), grid, dim3(512), 0, st, k); } \
            else if (big && ring) hipLaunchKernelGGL((conv_igemm_ring_kernel<TT, 256>), grid, dim3(512), 0, st, k); \
            else if (big) hipLaunchKernelGGL((conv_igemm_dma_kernel<TT, 8>), grid, dim3(512), 0, st, k);       \
            else          hipLaunchKernelGGL((conv_igemm_dma_kernel<TT, 4>), grid, dim3(512), 0, st, k);       \
        }
        if (bm_ == 129 && d->dtype == CP360_F16) hipLaunchKernelGGL((conv_igemm_ring2_kernel<f16_raw>), grid, dim3(512), 0, st, k);
        else if (bm_ == 129) hipLaunchKernelGGL((conv_igemm_ring2_kernel<bf16_raw>), grid, dim3(512), 0, st, k);
        else if (d->dtype == CP360_F32) CP360_WIDE(float)
        else if (d->dtype == CP360_F16) CP360_WIDE(f16_raw)
        else CP360_WIDE(bf16_raw)
#undef CP360_WIDE
    } else if (d->dtype == CP360_F32) {
        if (narrow) launch_conv<float, 1, 4>(k, st);
        else launch_conv<float, 2, 2>(k, st);
    } else if (d->dtype == CP360_F16) {
        if (narrow) launch_conv<f16_raw, 1, 4>(k, st);
        else launch_conv<f16_raw, 2, 2>(k, st);
    } else {
        if (narrow) launch_conv<bf16_raw, 1, 4>(k, st);
        else launch_conv<bf16_raw, 2, 2>(k, st);
    }
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_conv_finish(const cp360_conv_desc* d, const float* partial, const float* bias,
                                 const void* residual, void* out, void* stream) {
    return cp360_conv_finish_add(d, partial, nullptr, bias, residual, out, stream);
}

extern "C" int cp360_conv_finish_add(const cp360_conv_desc* d, const float* partial, const float* extra, const float* bias,
                                     const void* residual, void* out, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!partial || !out) return CP360_ERR_NULL;
    const int M = d->n_img * d->h_out * d->w_out;
    const long long total = (long long)M * (d->c_out / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == CP360_F32)
        hipLaunchKernelGGL((conv_finish_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, partial, d->splits,
                           bias, (const float*)residual, d->ld_res, (float*)out, d->ld_out, d->out_coff, M, d->c_out,
                           d->relu, d->slab_rows, extra);
    else if (d->dtype == CP360_F16)
        hipLaunchKernelGGL((conv_finish_kernel<f16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, partial,
                           d->splits, bias, (const f16_raw*)residual, d->ld_res, (f16_raw*)out, d->ld_out,
                           d->out_coff, M, d->c_out, d->relu, d->slab_rows, extra);
    else
        hipLaunchKernelGGL((conv_finish_kernel<bf16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, partial,
                           d->splits, bias, (const bf16_raw*)residual, d->ld_res, (bf16_raw*)out, d->ld_out,
                           d->out_coff, M, d->c_out, d->relu, d->slab_rows, extra);
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_lstm_gates(const float* gates_partial, int splits, const float* bias, const float* c_prev,
                                float* c_next, void* h_out, int h_dtype, int ld_h, int h_coff, float* h_f32, int M,
                                int Hc, int slab_rows, void* stream) {
    return cp360_lstm_gates_next(gates_partial, splits, bias, c_prev, c_next, h_out, h_dtype, ld_h, h_coff, h_f32, M, Hc,
                                 slab_rows, nullptr, nullptr, 0, 0, 0, stream);
}

extern "C" int cp360_lstm_gates_next(const float* gates_partial, int splits, const float* bias, const float* c_prev,
                                     float* c_next, void* h_out, int h_dtype, int ld_h, int h_coff, float* h_f32, int M,
                                     int Hc, int slab_rows, const float* x_next, const float* minmax, int x_coff,
                                     int P, size_t clip_stride, void* stream) {
    if (!gates_partial || !bias || !c_prev || !c_next || !h_out) return CP360_ERR_NULL;
    if (x_next && (!minmax || P <= 0 || M % P != 0 || x_coff % 4 != 0 || clip_stride % 4 != 0 || x_coff + Hc > ld_h ||
                   (x_coff < h_coff + Hc && h_coff < x_coff + Hc)))
        return CP360_ERR_BAD_SHAPE;
    if (splits < 1 || M <= 0 || Hc <= 0) return CP360_ERR_BAD_SHAPE;
    if (Hc % 4 != 0 || ld_h % 4 != 0 || h_coff % 4 != 0) return CP360_ERR_ALIGN;
    if (slab_rows && (4 * Hc) % 32 != 0) return CP360_ERR_ALIGN;
    const long long total = (long long)M * (Hc / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = (hipStream_t)stream;
    if (h_dtype == CP360_F32)
        hipLaunchKernelGGL((lstm_gates_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, gates_partial, splits,
                           bias, c_prev, c_next, (float*)h_out, ld_h, h_coff, h_f32, M, Hc, slab_rows, x_next, minmax, x_coff, P, clip_stride);
    else if (h_dtype == CP360_BF16)
        hipLaunchKernelGGL((lstm_gates_kernel<bf16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, gates_partial,
                           splits, bias, c_prev, c_next, (bf16_raw*)h_out, ld_h, h_coff, h_f32, M, Hc, slab_rows, x_next, minmax, x_coff, P,
                           clip_stride);
    else if (h_dtype == CP360_F16)
        hipLaunchKernelGGL((lstm_gates_kernel<f16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, gates_partial,
                           splits, bias, c_prev, c_next, (f16_raw*)h_out, ld_h, h_coff, h_f32, M, Hc, slab_rows, x_next, minmax, x_coff, P,
                           clip_stride);
    else
        return CP360_ERR_BAD_DTYPE;
    CP360_CHECK_HIP();
    return CP360_OK;
}
