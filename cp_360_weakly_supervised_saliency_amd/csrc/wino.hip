// Winograd F(2x2, 3x3) form of CubePad(1) + 3x3 convolution (the three ConvLSTM convolutions of model/clstm.py:56-64) for the
// 16-bit types.  A face of w x w pixels is cut into th x th tiles of 2 x 2 outputs (th = ceil(w / 2)); with
//     U_p = (G g G^T)_p      [c_out, c_in]   one matrix per position p of the 4 x 4 transform domain (packed once)
//     V_p = (B^T d B)_p      [tiles, c_in]   d = the 4 x 4 window of the cube-padded input behind a tile
//     M_p = V_p . U_p^T      [tiles, c_out]  sixteen plain GEMMs, f32 sums
//     Y   = A^T M A + bias   the tile's 2 x 2 outputs
// the convolution needs 16 multiplies per tile and channel pair instead of 9 per pixel: 16 / 36 of the direct form's MFMAs at
// even face sizes (8x8, 16x16), 256 / 441 at 7x7 (16 tiles cover 8 x 8 of which 49 outputs are used).  Transforms are f32 on
// exact 16-bit values, U and V are rounded ONCE to the 16-bit type (tests/probe_winograd_numerics.py: the T = 16 window keeps
// |dAUC-Judd|, |dCC| < 1e-4 against the oracle).  f32 stays on the direct kernels.
//
//   wino_in_kernel     NHWC activations (through cubepad_src) -> V   [pos][c / 32][tile][32]   (64-byte rows, tile-major)
//   wino_gemm_kernel   256 channels x 384 tiles x one position per workgroup, both operands streamed by LDS-DMA
//   wino_out_kernel    M [pos][tile][c_out] f32 -> A^T M A + bias (+ ReLU) -> NHWC activations, or -> the LSTM gate update
#include "conv_common.h"

namespace {

constexpr int WG_BN = 256, WG_BM = 384, WG_NS = 4;
constexpr int WG_STAGE = (WG_BN + WG_BM) * 64;             // 40 KiB per 64-byte sub-step; 4 stages = all 160 KiB of LDS

struct WinoK {
    const unsigned char* u;      // [16][nt][nsub][256][64 B]
    const unsigned char* v;      // [16][nsub][m_pad][64 B]
    float* m;                    // [16][m_pad][ldm]
    int nsub, nt, mt, m_pad, ldm, c_out, reverse;
    int tpf, th, odd;            // tiles per face, per face side; odd = the faces are odd-sized (0: every row of M is written)
    int u_pin;                   // sub-steps of a workgroup's U stream loaded with the default cache policy (the rest: non-temporal)
};

// Odd faces (w = 2 th - 1): a tile of the last tile row / column keeps only the first row / column of its 2 x 2 outputs, and
// A^T M A computes those without transform-domain row 3 / column 3 - 31 of a 7x7 face's 256 (tile, position) pairs are never read.
// The GEMM does not store them and the output transforms do not load them (12 % of M's bytes in both directions).
__device__ __forceinline__ bool wino_dead_row(int pos, int ty, int th) { return (pos >> 2) == 3 && ty == th - 1; }
__device__ __forceinline__ bool wino_dead_col(int pos, int tx, int th) { return (pos & 3) == 3 && tx == th - 1; }

__device__ __forceinline__ int swz64(int row, int chunk) { return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4); }

template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_upto(int n);
template <> __device__ __forceinline__ void vm_wait_upto<0>(int) { vm_wait<0>(); }
template <int N> __device__ __forceinline__ void vm_wait_upto(int n) {
    if (n >= N) vm_wait<N>();
    else vm_wait_upto<N - 1>(n);
}

#ifdef WINO_LAB
// the A/B laboratory's form of this section (every knob, the timing ablations, the in-kernel stamps): tools/lab/wino_lab.h, used
// only by tools/wino_variants.sh / wino_stamps.sh / wino_cell_ab.sh - never by the Makefile
#include "../../tools/lab/wino_lab.h"
#else
// Settings of the transform kernels below (each measured on tools/wino_cell_probe.py, docs/rounds/r05.md section 5; the
// laboratory build can override them): how a kernel avoids the never-read positions of odd faces (load_positions MODE), zeros in
// the matching V rows, non-temporal loads of M where M is read once.
constexpr bool kZeroDeadV = true;
constexpr int kDlOutIn = 2, kDlGates = 1, kDlOut = 1;
constexpr bool kTabLate = false, kNtOutIn = true, kNtGates = false;

// One LDS-DMA instruction = 1 KiB (16 tile rows x 64 B): "scalar base + 32-bit lane offset" addressing, M0 = the LDS destination.
// Both operand blocks of a sub-step are contiguous (16 KiB of U, 24 KiB of V), so a sub-step's fill is 40 such instructions.
// nt: this piece is non-temporal.  The GEMM loads only the first WinoK::u_pin sub-steps of every workgroup's U block with the
// default policy (V always): the 1.3 GB of weights a cell update streams no longer sweep the 256 MB Infinity Cache, so the 86 MB
// of M and the 49 MB of V between the launches stay in it, and the HEAD of every workgroup's U stream - what all 256 workgroups
// ask for at once when a launch starts - is still there from the previous cell update (cell update 480 us with everything
// default, 474 all non-temporal, 454 with the first 4-8 sub-steps default, 460+ from 16 up).
__device__ __forceinline__ void fill_one(const unsigned char* base, unsigned off, unsigned dst, bool is_u, bool u_nt) {
    unsigned keep;
    if (is_u && u_nt)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
    else if (is_u)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
}

// K loop + slab stores of one wave: channels [wch0, +64) x tile rows [wrow0, +192) = 4 x 12 MFMA blocks (192 accumulator
// registers).  A sub-step is a HEAD (fragment reads of its LDS stage, the first 6 columns; a column's registers are re-loaded
// with column 6 + j as soon as its MFMAs are issued) and a TAIL (the other 6 columns, from registers); the two waves of a
// SIMD (w and w + 4) run half a sub-step apart (waves 4-7 LAG: their TAIL of sub-step s-1 comes before their HEAD of s).
// Structure (in-kernel stamps and same-box A/B, docs/rounds/r05.md sections 1 and 5): ONE barrier per sub-step; every wave issues
// its 5 of the sub-step's 40 refill DMAs in front of its HEAD's MFMAs; a lagging wave runs its HEAD's MFMAs at priority 1, so that
// the two waves of a SIMD reach the barrier together.
template <typename T, bool LAG>
__device__ __forceinline__ void gemm_body(const WinoK& p, unsigned char* lds, const int pos, const int nt_i, const int mt_i,
                                          const int wave, const int lane, const int wch0, const int wrow0) {
    constexpr int NS = WG_NS, MJ = 12, JH = MJ / 2, NU = 5;      // NU: DMAs per wave and sub-step
    const int nsub = p.nsub;
    const unsigned char* ub = p.u + ((size_t)(pos * p.nt + nt_i) * nsub) * (WG_BN * 64);
    const size_t vstep = (size_t)p.m_pad * 64;
    const unsigned char* vb = p.v + (size_t)pos * nsub * vstep + (size_t)mt_i * (WG_BM * 64);
    // DMA role: lane l lands in row 16 * wave + (l >> 2) (+ 128 per pass), physical chunk l & 3, and fetches the logical chunk of
    // that row (source-side swizzle)
    const unsigned o0 = (unsigned)((16 * wave + (lane >> 2)) * 64 + ((((lane & 3) ^ ((0 - ((4 * wave + (lane >> 4)) & 3)) & 3))) << 4));
    const unsigned lds_wave = (unsigned)(size_t)lds + (unsigned)wave * 1024;

    f32x4 acc[4][MJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, lchunk = lane >> 4;
    unsigned rdst = 0;                                            // LDS destination (this wave's slot) of the refill in progress
    int fills = 0;                                                // sub-steps requested so far (uniform)
    // pass q of a sub-step's fill: 0, 1 = the U block's two 128-row passes; 2, 3, 4 = the V block's three
    auto dma = [&](int u) __attribute__((always_inline)) {
        const int q = u % 5, half = u / 5;
        const unsigned src = o0 + (unsigned)((q < 2 ? q : q - 2) * 0x2000 + half * 0x1000);
        fill_one(q < 2 ? ub : vb, src, rdst + (unsigned)(q * 0x2000 + half * 0x1000), q < 2, fills >= p.u_pin);
    };
    auto refill_begin = [&](int stage) __attribute__((always_inline)) {
        rdst = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)stage * WG_STAGE);
    };
    auto refill_end = [&]() __attribute__((always_inline)) {
        ub += WG_BN * 64;
        vb += vstep;
        ++fills;
    };
#pragma unroll
    for (int k = 0; k < NS - 1; ++k)
        if (k < nsub) {
            refill_begin(k);
#pragma unroll
            for (int u = 0; u < NU; ++u) dma(u);
            refill_end();
        }
    int stage = 0;
    u32x4 a[4], b[JH];
    if (LAG) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < JH; ++j) b[j] = u32x4{0u, 0u, 0u, 0u};
    }
    // (HEAD / TAIL / STEP are macros, not lambdas: as lambdas the same statements compile to a main loop that measured 2 - 4 %
    // slower, tools/wino_cell_ab.sh product against laboratory build.)
    // REFILL: the stage read in the PREVIOUS sub-step (every wave finished with it before this sub-step's barrier) takes sub-step
    // it + NS - 1.  A lagging wave's TAIL runs in front of its HEAD: the freed stage is the one behind `stage` there too.
#define WINO_HEAD(REFILL)                                                                                  \
    {                                                                                                      \
        const unsigned char* As = lds + stage * WG_STAGE;                                                  \
        const unsigned char* Bs = As + WG_BN * 64;                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
            a[i] = *reinterpret_cast<const u32x4*>(As + swz64(wch0 + i * 16 + lrow, lchunk));              \
        _Pragma("unroll") for (int j = 0; j < JH; ++j)                                                     \
            b[j] = *reinterpret_cast<const u32x4*>(Bs + swz64(wrow0 + j * 16 + lrow, lchunk));             \
        if (REFILL) {                                                                                      \
            refill_begin(stage == 0 ? NS - 1 : stage - 1);                                                 \
            _Pragma("unroll") for (int u = 0; u < NU; ++u) dma(u);                                         \
            refill_end();                                                                                  \
        }                                                                                                  \
        if (LAG) __builtin_amdgcn_s_setprio(1);                                                            \
        _Pragma("unroll") for (int j = 0; j < JH; ++j) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) mma_chunk<T>(acc[i][j], a[i], b[j]);             \
            b[j] = *reinterpret_cast<const u32x4*>(Bs + swz64(wrow0 + (JH + j) * 16 + lrow, lchunk));      \
        }                                                                                                  \
        if (LAG) __builtin_amdgcn_s_setprio(0);                                                            \
        stage = stage == NS - 1 ? 0 : stage + 1;                                                           \
    }
#define WINO_TAIL()                                                                                        \
    {                                                                                                      \
        _Pragma("unroll") for (int j = JH; j < MJ; ++j) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) mma_chunk<T>(acc[i][j], a[i], b[j - JH]);        \
        }                                                                                                  \
    }
#define WINO_STEP(REFILL)                                                                                  \
    {                                                                                                      \
        if (LAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
        __builtin_amdgcn_s_barrier();                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (!LAG) WINO_HEAD(REFILL) else WINO_TAIL()                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (!LAG) WINO_TAIL() else WINO_HEAD(REFILL)                                                       \
    }
    // vmcnt (DMAs complete in issue order): before sub-step `it` its stage must have landed; younger are the NS - 2 stages behind it
    int it = 0;
    for (; it + NS - 1 < nsub; ++it) {
        vm_wait<(NS - 2) * NU>();
        WINO_STEP(true)
    }
    for (; it < nsub; ++it) {
        vm_wait_upto<(NS - 2) * NU>(min(NS - 2, nsub - 1 - it) * NU);
        WINO_STEP(false)
    }
    if (LAG) WINO_TAIL()
#undef WINO_STEP
#undef WINO_HEAD
#undef WINO_TAIL

    // slabs: M[pos][tile][channel], a lane owns 4 consecutive channels of one tile per block (16-byte stores).  Tile rows outermost:
    // the four 64-byte pieces of a row's 256 bytes leave back to back and merge into full lines in L2.  Odd faces: the never-read
    // (tile, position) pairs (wino_dead_row / wino_dead_col) are not stored.
    float* mp = p.m + ((size_t)pos * p.m_pad + (size_t)mt_i * WG_BM + wrow0 + lrow) * p.ldm + nt_i * WG_BN + wch0 + lchunk * 4;
    const bool edge_pos = p.odd && ((pos >> 2) == 3 || (pos & 3) == 3);        // (uniform) this position has never-read tiles
    int t0 = 0;                                                                 // this lane's tile inside its face, block 0
    if (edge_pos) t0 = (mt_i * WG_BM + wrow0 + lrow) % p.tpf;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        bool dead = false;
        if (edge_pos) {
            const int t = p.tpf == 16 ? t0 : (t0 + j * 16) % p.tpf, ty = t / p.th, tx = t - ty * p.th;
            dead = wino_dead_row(pos, ty, p.th) || wino_dead_col(pos, tx, p.th);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!dead && nt_i * WG_BN + wch0 + i * 16 + lchunk * 4 < p.c_out)
                *reinterpret_cast<f32x4*>(mp + (size_t)j * 16 * p.ldm + i * 16) = acc[i][j];
    }
}

template <typename T>
__global__ __launch_bounds__(512, 2) void wino_gemm_kernel(const WinoK p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[WG_NS * WG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pos, nt_i, mt_i;
    {   // XCD-aware mapping: the channel tiles of one (position, tile block) are neighbours on one XCD and share V_p through its L2
        const int nwg = 16 * p.nt * p.mt;
        const int L = blockIdx.x, xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if (p.reverse) w = nwg - 1 - w;
        nt_i = w % p.nt;
        const int rest = w / p.nt;
        mt_i = rest % p.mt;
        pos = rest / p.mt;
    }
    const int wch0 = (wave >> 1) * 64, wrow0 = (wave & 1) * 192;
    if (wave < 4) gemm_body<T, false>(p, lds, pos, nt_i, mt_i, wave, lane, wch0, wrow0);
    else          gemm_body<T, true>(p, lds, pos, nt_i, mt_i, wave, lane, wch0, wrow0);
}
#endif  // WINO_LAB


// ------------------------------------------------------------------ U = G g G^T, packed [pos][n / 256][c / 32][n % 256][32]
// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; f32 arithmetic on the f32 filter, ONE rounding to T.  Rows past c_out and
// channels past c_in are zero (they meet V's zero padding / are never stored).
template <typename T>
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, T* __restrict__ u, int c_out, int c_in,
                                                        int nt, int nsub) {
    const int ngrp = nsub * 4;                                    // 8-channel groups per row
    const long long total = (long long)nt * 256 * ngrp;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx / ngrp), cg = (int)(idx - (long long)co * ngrp);
        float uo[16][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = cg * 8 + e;
            float g[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) g[k] = (co < c_out && ci < c_in) ? w[((size_t)co * c_in + ci) * 9 + k] : 0.f;
            float t[4][3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                t[0][kx] = g[kx];
                t[1][kx] = 0.5f * (g[kx] + g[3 + kx] + g[6 + kx]);
                t[2][kx] = 0.5f * (g[kx] - g[3 + kx] + g[6 + kx]);
                t[3][kx] = g[6 + kx];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uo[r * 4 + 0][e] = t[r][0];
                uo[r * 4 + 1][e] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
                uo[r * 4 + 2][e] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
                uo[r * 4 + 3][e] = t[r][2];
            }
        }
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            T* dst = u + ((((size_t)pos * nt + (co >> 8)) * nsub + (cg >> 2)) * 256 + (co & 255)) * 32 + (cg & 3) * 8;
            *reinterpret_cast<u32x4*>(dst) = pack8(uo[pos], T());
        }
    }
}

// ------------------------------------------------------------------ V = B^T d B
// One workgroup = one face x 64 channels (two 64-byte sub-steps): the face's padded window of (2 th + 2)^2 pixels - CubePad(1)
// through cubepad_src(), zeros past it where w is odd - is gathered to LDS once (128 B per pixel), then every (tile, 8-channel
// chunk) item reads its 4 x 4 window from there and stores its 16 positions.  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].
// TH (here and in the output transforms): th as a compile-time constant (4 / 8: the 7x7, 8x8 and 16x16 faces of the ConvLSTM at cube 224 / 256 / 512 -
// the index arithmetic of an item becomes shifts instead of 32-bit divisions, which sat in front of every item's loads), 0: generic.
template <typename T, int TH>
__global__ __launch_bounds__(256) void wino_in_kernel(const T* __restrict__ in, T* __restrict__ v, int w, int th_, int c_in,
                                                      int pix_stride, int nsub, int m_pad, int ncb) {
    const int th = TH ? TH : th_;
    extern __shared__ __attribute__((aligned(16))) unsigned char patch[];
    const int img = blockIdx.x / ncb, cb = blockIdx.x - img * ncb;
    const int pw = 2 * th + 2, wp = w + 2;
    const int cube = img / 6, f = img - cube * 6;
    const CubePadGeom geom{w, 1, 1, 1, 1};
    for (int i = threadIdx.x; i < pw * pw * 8; i += 256) {
        const int pix = i >> 3, ch = i & 7;
        const int py = pix / pw, px = pix - py * pw;
        const int c = cb * 64 + ch * 8;
        u32x4 val = u32x4{0u, 0u, 0u, 0u};
        if (py < wp && px < wp && c < c_in)
            val = *reinterpret_cast<const u32x4*>(in + ((size_t)cube * 6 * w * w + cubepad_src(f, py, px, geom)) * pix_stride + c);
        *reinterpret_cast<u32x4*>(patch + pix * 128 + ch * 16) = val;
    }
    __syncthreads();
    // items = (row i of V, tile, 16-byte chunk): 4 x th^2 x 8 of them, i outermost so that the lanes of a wave store neighbouring tiles
    // of one position (8 tiles x 64 B contiguous per sub-step and store instruction)
    const int tpf = th * th, per_row = tpf * 8;
    const size_t pstride = (size_t)nsub * m_pad * 32;
    for (int it = threadIdx.x; it < 4 * per_row; it += 256) {
        const int i = it / per_row, r = it - i * per_row;
        const int tile = r >> 3, ch = r & 7;
        const int sub = cb * 2 + (ch >> 2);
        if (sub >= nsub) continue;
        const int ty = tile / th, tx = tile - ty * th;
        const unsigned char* base = patch + ((2 * ty) * pw + 2 * tx) * 128 + ch * 16;
        T* dst = v + ((size_t)sub * m_pad + (size_t)img * tpf + tile) * 32 + (ch & 3) * 8 + (size_t)(i * 4) * pstride;
        // e[c] = (B^T d)[i][c]: rows (0 - 2), (1 + 2), (2 - 1), (1 - 3) of the window
        const int ra = i == 0 ? 0 : i == 2 ? 2 : 1, rb = i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : 3;
        float e[4][8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float da[8], db[8];
            unpack8(*reinterpret_cast<const u32x4*>(base + (ra * pw + c) * 128), da, T());
            unpack8(*reinterpret_cast<const u32x4*>(base + (rb * pw + c) * 128), db, T());
#pragma unroll
            for (int k = 0; k < 8; ++k) e[c][k] = i == 1 ? da[k] + db[k] : da[k] - db[k];
        }
        float o[4][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o[0][k] = e[0][k] - e[2][k];
            o[1][k] = e[1][k] + e[2][k];
            o[2][k] = e[2][k] - e[1][k];
            o[3][k] = e[1][k] - e[3][k];
        }
        // (never-read positions of an odd face's last tile row / column: zeros - the GEMM's multiplies on them toggle nothing)
        // (as a bit mask, not a select: the four stores stay unconditional and back to back)
        const unsigned keep_r = (kZeroDeadV && (w & 1) && i == 3 && ty == th - 1) ? 0u : ~0u, keep_c = (kZeroDeadV && (w & 1) && tx == th - 1) ? 0u : ~0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 val = pack8(o[j], T());
            const unsigned k = j == 3 ? (keep_r & keep_c) : keep_r;
            *reinterpret_cast<u32x4*>(dst + (size_t)j * pstride) = u32x4{val.x & k, val.y & k, val.z & k, val.w & k};
        }
    }
}

// ------------------------------------------------------------------ Y = A^T M A + bias, A^T = [[1,1,1,0],[0,1,-1,-1]]
// s[a][j] = sum_i A^T[a][i] M[i][j] column by column, then Y[a][0] = s[a][0] + s[a][1] + s[a][2], Y[a][1] = s[a][1] - s[a][2] - s[a][3].
// The 16 position values of a (tile, 4-channel group) are requested together (16 independent 16-byte loads in flight per call;
// callers issue two or four calls before the first use: these kernels have few threads per CU and live on memory parallelism).
// Address = (uniform plane base) + (32-bit per-thread byte offset): the "scalar base + vector offset" load form, one offset register
// for all 16 loads instead of 16 64-bit pointers.
// dead_r / dead_c: this tile is in the last tile row / column of an odd face - the GEMM does not store transform-domain row 3 /
// column 3 there and nothing it could hold reaches an output that is kept (they only enter the 2 x 2 outputs past the face).
// MODE 0: all sixteen loads as they are (every row of M must then have been written); 1: the dead positions are not loaded
// (predicated loads, zeros stand in); 2: a dead position is loaded from the SAME position of the tile one column / row back
// (byte offsets back_c / back_r) - a live value another lane of the workgroup fetches anyway, so the sixteen loads stay
// unconditional in the "scalar base + vector offset" form and the dead lines are still never touched.
// Which MODE each kernel uses: the constants kDlOut / kDlOutIn / kDlGates at the top of this file.
template <int MODE, bool NT = false>
__device__ __forceinline__ void load_positions(const float* __restrict__ m, size_t pstride, unsigned voff, f32x4 (&mm)[16],
                                               bool dead_r, bool dead_c, unsigned back_r, unsigned back_c) {
    auto ld = [&](int p, unsigned off) __attribute__((always_inline)) {
        if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(m + (size_t)p * pstride) + off));
        else return *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(m + (size_t)p * pstride) + off);
    };
    if constexpr (MODE == 1) {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 12; ++p)
            if ((p & 3) != 3) mm[p] = ld(p, voff);
        mm[3] = mm[7] = mm[11] = mm[12] = mm[13] = mm[14] = mm[15] = z;
        if (!dead_c) { mm[3] = ld(3, voff); mm[7] = ld(7, voff); mm[11] = ld(11, voff); }
        if (!dead_r) { mm[12] = ld(12, voff); mm[13] = ld(13, voff); mm[14] = ld(14, voff); }
        if (!dead_r && !dead_c) mm[15] = ld(15, voff);
    } else {
        const unsigned vc = MODE == 2 && dead_c ? voff - back_c : voff, vr = MODE == 2 && dead_r ? voff - back_r : voff;
        const unsigned vrc = MODE == 2 ? vc - (voff - vr) : voff;
#pragma unroll
        for (int p = 0; p < 16; ++p) mm[p] = ld(p, p == 15 ? vrc : (p >= 12 ? vr : ((p & 3) == 3 ? vc : voff)));
    }
}
__device__ __forceinline__ void out_transform(const f32x4 (&mm)[16], f32x4 (&y)[4]) {
    f32x4 s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s0[j] = mm[j] + mm[4 + j] + mm[8 + j];
        s1[j] = mm[4 + j] - mm[8 + j] - mm[12 + j];
    }
    y[0] = s0[0] + s0[1] + s0[2];
    y[1] = s0[1] - s0[2] - s0[3];
    y[2] = s1[0] + s1[1] + s1[2];
    y[3] = s1[1] - s1[2] - s1[3];
}

// One thread per (tile, 4-channel group): 16 independent 16-byte loads, 8-byte (16-bit types) stores; 384 k threads at 4 clips of 7x7
// faces.  (8 channels per thread - 32 loads up front, 152 registers - measured 22 us against 17 for the same bytes.)
template <typename T, int TH>
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ m, const float* __restrict__ bias,
                                                       T* __restrict__ out, int tiles, int w, int th_, int c_out, int ldm,
                                                       int m_pad, int ld_out, int out_coff, int relu) {
    const int th = TH ? TH : th_;
    const int ng = c_out >> 2;
    const long long total = (long long)tiles * ng;
    const size_t pstride = (size_t)m_pad * ldm;
    const int tpf = th * th;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int tg = (int)(idx / ng), c = (int)(idx - (long long)tg * ng) * 4;
        f32x4 mm[16], y[4];
        const int img = tg / tpf, t = tg - img * tpf, ty = t / th, tx = t - ty * th;
        load_positions<kDlOut>(m, pstride, (unsigned)((tg * ldm + c) * 4), mm, (w & 1) && ty == th - 1, (w & 1) && tx == th - 1,
                                    (unsigned)(th * ldm * 4), (unsigned)(ldm * 4));   // < 2^32: checked at launch
        const f32x4 bb = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        out_transform(mm, y);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = 2 * ty + (q >> 1), ox = 2 * tx + (q & 1);
            if (oy >= w || ox >= w) continue;
            float v[4] = {y[q][0] + bb[0], y[q][1] + bb[1], y[q][2] + bb[2], y[q][3] + bb[3]};
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            store4(out + ((size_t)(img * w + oy) * w + ox) * ld_out + out_coff + c, v);
        }
    }
}

// Output transform of one convolution AND the input transform of the next in one launch (faces small enough that a cube's
// activations of a 32-channel block fit in LDS: w <= 9): a workgroup = (cube, 32 channels = one K sub-step of the next GEMM).
// Phase 1: every (tile, 4-channel group) item reads its 16 positions of M, transforms, adds bias, applies ReLU, rounds to T and puts
// the tile's pixels into the LDS image of the cube; phase 2: every (V row i, tile, 8-channel chunk) item gathers its 4 x 4 window from
// that image through a CubePad(1) source table and stores four positions of the next convolution's V.  The values are exactly those
// of wino_out_kernel + wino_in_kernel (same roundings) - the 16-bit activation tensor just never goes through memory, and the
// 49 MB of V writes overlap the 98 MB of M reads instead of following them in a second launch.
// CB = channels per workgroup: 32 (one K sub-step of the next GEMM; faces up to 9 x 9).  (CB = 16 - half a sub-step, for 16 x 16 faces,
// where a cube's image of 32 channels would be 98 KB of LDS - works and is slower than the two kernels: see the launcher.)
template <typename T, int CB, int TH>
__global__ __launch_bounds__(768, TH == 4 ? 6 : 5) void wino_out_in_kernel(const float* __restrict__ m, const float* __restrict__ bias,
                                                          T* __restrict__ v, int w, int th_, int c_out, int ldm, int m_pad, int relu,
                                                          int nsub, int nblk) {
    const int th = TH ? TH : th_;
    constexpr int ROW = CB * 2, NG = CB / 4, NCH = CB / 8;                // bytes per pixel of the image, 4-channel groups, 16-byte chunks
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int cube = blockIdx.x / nblk, cb = blockIdx.x - cube * nblk;
    const int ww = w * w, P = 6 * ww, tpf = th * th, pw = 2 * th + 2, wp = w + 2, pp = pw * pw;
    unsigned char* act = sm;                                               // [P][ROW]
    unsigned short* tab = reinterpret_cast<unsigned short*>(sm + ((P * ROW + 15) & ~15));   // [6][pw][pw]: pixel of the cube, 0xffff = zero
    const CubePadGeom geom{w, 1, 1, 1, 1};
    // (the source table is only read in phase 2: it is built behind the first item's loads of M, not in front of them)
    auto build_tab = [&]() __attribute__((always_inline)) {
        for (int i = threadIdx.x; i < 6 * pp; i += blockDim.x) {
            const int f = i / pp, r = i - f * pp, py = r / pw, px = r - py * pw;
            tab[i] = (py < wp && px < wp) ? (unsigned short)cubepad_src(f, py, px, geom) : (unsigned short)0xffff;
        }
    };
    bool tab_done = !kTabLate;
    if (!kTabLate) build_tab();
    const size_t pstride = (size_t)m_pad * ldm;
    for (int it = threadIdx.x; it < 6 * tpf * NG; it += blockDim.x) {
        const int tl = it / NG, g = it - tl * NG, c = cb * CB + g * 4;
        f32x4 y[4];
        f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
        const int f = tl / tpf, t = tl - f * tpf, ty = t / th, tx = t - ty * th;
        if (c < c_out) {
            f32x4 mm[16];
            load_positions<kDlOutIn, kNtOutIn>(m, pstride, (unsigned)(((cube * 6 * tpf + tl) * ldm + c) * 4), mm, (w & 1) && ty == th - 1,
                                          (w & 1) && tx == th - 1, (unsigned)(th * ldm * 4), (unsigned)(ldm * 4));
            if (bias) bb = *reinterpret_cast<const f32x4*>(bias + c);
            if (!tab_done) { build_tab(); tab_done = true; }
            out_transform(mm, y);
        } else {                                                           // channels past c_out: the next V's zero padding
            y[0] = y[1] = y[2] = y[3] = bb;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = 2 * ty + (q >> 1), ox = 2 * tx + (q & 1);
            if (oy >= w || ox >= w) continue;
            float o[4] = {y[q][0] + bb[0], y[q][1] + bb[1], y[q][2] + bb[2], y[q][3] + bb[3]};
            if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
            store4(reinterpret_cast<T*>(act + (f * ww + oy * w + ox) * ROW) + g * 4, o);
        }
    }
    if (!tab_done) build_tab();
    __syncthreads();
    const int per_row = 6 * tpf * NCH;
    const size_t vps = (size_t)nsub * m_pad * 32;                          // elements between two positions of V
    const int sub = (cb * CB) >> 5, coff = (cb * CB) & 31;                 // this block's place inside the 32-channel rows of V
    for (int it = threadIdx.x; it < 4 * per_row; it += blockDim.x) {
        const int i = it / per_row, r = it - i * per_row;
        const int tl = r / NCH, ch = r - tl * NCH;
        const int f = tl / tpf, t = tl - f * tpf, ty = t / th, tx = t - ty * th;
        const unsigned short* tb = tab + f * pp + (2 * ty) * pw + 2 * tx;
        const int ra = i == 0 ? 0 : i == 2 ? 2 : 1, rb = i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : 3;
        float e[4][8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned pa = tb[ra * pw + c], pb = tb[rb * pw + c];
            const u32x4 z = u32x4{0u, 0u, 0u, 0u};
            const u32x4 xa = pa == 0xffffu ? z : *reinterpret_cast<const u32x4*>(act + pa * ROW + ch * 16);
            const u32x4 xb = pb == 0xffffu ? z : *reinterpret_cast<const u32x4*>(act + pb * ROW + ch * 16);
            float da[8], db[8];
            unpack8(xa, da, T());
            unpack8(xb, db, T());
#pragma unroll
            for (int k = 0; k < 8; ++k) e[c][k] = i == 1 ? da[k] + db[k] : da[k] - db[k];
        }
        float o[4][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o[0][k] = e[0][k] - e[2][k];
            o[1][k] = e[1][k] + e[2][k];
            o[2][k] = e[2][k] - e[1][k];
            o[3][k] = e[1][k] - e[3][k];
        }
        T* dst = v + (size_t)(i * 4) * vps + ((size_t)sub * m_pad + (size_t)cube * 6 * tpf + tl) * 32 + coff + ch * 8;
        const unsigned keep_r = (kZeroDeadV && (w & 1) && i == 3 && ty == th - 1) ? 0u : ~0u, keep_c = (kZeroDeadV && (w & 1) && tx == th - 1) ? 0u : ~0u;   // as wino_in_kernel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 val = pack8(o[j], T());
            const unsigned k = j == 3 ? (keep_r & keep_c) : keep_r;
            *reinterpret_cast<u32x4*>(dst + (size_t)j * vps) = u32x4{val.x & k, val.y & k, val.z & k, val.w & k};
        }
    }
}

// The Gates convolution's output transform + the cell update of model/clstm.py:68-80 (gate order in, remember, out, cell),
// with the optional window normalisation of the NEXT frame into the x half (as lstm_gates_kernel, conv_igemm.hip).
__device__ __forceinline__ float wsigmoid(float x) { return 1.f / (1.f + __expf(-x)); }

// A block = 64 (tile, 4-channel group) items x 4 waves.  Wave k brings in gate k's 16 positions of every item (1 KiB contiguous
// per position and wave: 16 independent loads in flight per lane - with one thread per item doing all four gates the launch
// had 96 k threads of 64 dependent-ish loads each and ran at a third of the memory system's rate), transforms them and leaves the
// item's four pixels in LDS; after the barrier wave q finishes pixel q of every item from the four gates.
template <typename T, int TH>
__global__ __launch_bounds__(256, 6) void wino_gates_kernel(const float* __restrict__ m, const float* __restrict__ bias,
                                                         const float* __restrict__ c_prev, float* __restrict__ c_next,
                                                         T* __restrict__ h_out, int ld_h, int h_coff, float* __restrict__ h_f32,
                                                         int tiles, int w, int th_, int Hc, int ldm, int m_pad,
                                                         const float* __restrict__ x_next, const float* __restrict__ minmax,
                                                         int x_coff, size_t clip_stride) {
    __shared__ f32x4 ex[4][4][64];                                 // [gate][pixel][item]: lanes of a wave touch consecutive 16-byte slots
    const int th = TH ? TH : th_;
    const int ng = Hc >> 2;
    const size_t pstride = (size_t)m_pad * ldm;
    const int tpf = th * th, P = 6 * w * w;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned total = (unsigned)tiles * (unsigned)ng, idx = blockIdx.x * 64u + lane;       // < 2^31: checked at launch
    const bool live = idx < total;
    const int tg = live ? (int)(idx / (unsigned)ng) : 0, j = live ? (int)(idx - (unsigned)tg * ng) * 4 : 0;
    const int img = tg / tpf, t = tg - img * tpf, ty = t / th, tx = t - ty * th;
    {
        f32x4 mm[16], y[4];
        load_positions<kDlGates, kNtGates>(m, pstride, (unsigned)((tg * ldm + wv * Hc + j) * 4), mm, (w & 1) && ty == th - 1, (w & 1) && tx == th - 1,
                                      (unsigned)(th * ldm * 4), (unsigned)(ldm * 4));   // < 2^32: checked at launch
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + wv * Hc + j);
        out_transform(mm, y);
#pragma unroll
        for (int q = 0; q < 4; ++q) ex[wv][q][lane] = y[q] + bb;
    }
    __syncthreads();
    const int oy = 2 * ty + (wv >> 1), ox = 2 * tx + (wv & 1);
    if (!live || oy >= w || ox >= w) return;
    const size_t mpx = (size_t)(img * w + oy) * w + ox;
    const f32x4 cp = *reinterpret_cast<const f32x4*>(c_prev + mpx * Hc + j);
    const f32x4 gi = ex[0][wv][lane], gf = ex[1][wv][lane], go = ex[2][wv][lane], gc = ex[3][wv][lane];
    float cn[4], hn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ig = wsigmoid(gi[e]), fg = wsigmoid(gf[e]), og = wsigmoid(go[e]);
        const float cg = tanhf(gc[e]);
        cn[e] = fg * cp[e] + ig * cg;
        hn[e] = og * tanhf(cn[e]);
    }
    *reinterpret_cast<f32x4*>(c_next + mpx * Hc + j) = f32x4{cn[0], cn[1], cn[2], cn[3]};
    store4(h_out + mpx * ld_h + h_coff + j, hn);
    if (h_f32) *reinterpret_cast<f32x4*>(h_f32 + mpx * Hc + j) = f32x4{hn[0], hn[1], hn[2], hn[3]};
    if (x_next) {
        const int b = (int)(mpx / P), pix = (int)(mpx - (size_t)b * P);
        const float mn = minmax[2 * b], den = minmax[2 * b + 1] - mn;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x_next + (size_t)b * clip_stride + (size_t)pix * Hc + j);
        const float xn[4] = {(xv[0] - mn) / den, (xv[1] - mn) / den, (xv[2] - mn) / den, (xv[3] - mn) / den};
        store4(h_out + mpx * ld_h + x_coff + j, xn);
    }
}

}  // namespace

// ------------------------------------------------------------------ C ABI
namespace {
struct WinoGeom { int th, tpf, tiles, mt, m_pad, nsub, nt, ldm; };

int wino_check(const cp360_wino_desc* d, WinoGeom* g) {
    if (!d) return CP360_ERR_NULL;
    if (d->dtype != CP360_BF16 && d->dtype != CP360_F16) return CP360_ERR_BAD_DTYPE;
    if (d->n_img <= 0 || d->face < 2 || d->c_in <= 0 || d->c_out <= 0 || d->pix_stride < d->c_in) return CP360_ERR_BAD_SHAPE;
    if (d->n_img % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (d->c_in % 8 != 0 || d->c_out % 8 != 0 || d->pix_stride % 8 != 0 || d->ld_out % 8 != 0 || d->out_coff % 8 != 0) return CP360_ERR_ALIGN;
    if (d->ld_out != 0 && d->ld_out < d->c_out + d->out_coff) return CP360_ERR_BAD_SHAPE;
    if ((long long)d->n_img * d->face * d->face * d->pix_stride >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    g->th = (d->face + 1) / 2;
    g->tpf = g->th * g->th;
    g->tiles = d->n_img * g->tpf;
    g->mt = (g->tiles + WG_BM - 1) / WG_BM;
    g->m_pad = g->mt * WG_BM;
    g->nsub = (d->c_in + 31) / 32;
    g->nt = (d->c_out + WG_BN - 1) / WG_BN;
    g->ldm = d->c_out;
    if ((long long)g->m_pad * g->ldm * 4 >= (1LL << 32)) return CP360_ERR_BAD_SHAPE;      // 32-bit byte offsets inside one position's plane of M
    return CP360_OK;
}
}  // namespace

// Sub-steps (64 bytes of K) at the head of every workgroup's weight stream that keep the default cache policy (see fill_one):
// 4 x 256 workgroups x 16 KiB = 16 MB per convolution (2 - 8 measured alike, 12 and more slower).  CP360_WINO_UPIN overrides it (A/B; a value >= the K loop = all default).
static int wino_u_pin() {
    static const int v = []() { const char* e = getenv("CP360_WINO_UPIN"); return e ? atoi(e) : 4; }();
    return v;
}

extern "C" size_t cp360_wino_packed_bytes(const cp360_wino_desc* d) {
    WinoGeom g;
    return wino_check(d, &g) ? 0 : (size_t)16 * g.nt * g.nsub * WG_BN * 64;
}
extern "C" size_t cp360_wino_v_bytes(const cp360_wino_desc* d) {
    WinoGeom g;
    return wino_check(d, &g) ? 0 : (size_t)16 * g.nsub * g.m_pad * 64;
}
extern "C" size_t cp360_wino_m_bytes(const cp360_wino_desc* d) {
    WinoGeom g;
    return wino_check(d, &g) ? 0 : (size_t)16 * g.m_pad * g.ldm * sizeof(float);
}

// 1: the planner would run this CubePad(1) + 3x3 convolution in the Winograd domain: 16-bit type, and its 16 GEMMs over the padded
// tile count need at most 0.8 of the direct form's MFMAs (4 clips of 7x7 faces: 0.58; one clip of 16x16 faces: 0.44; one clip of
// 7x7 faces: 1.8 - the direct clip-resident kernel stays).  CP360_WINO=0 / 1 forces never / whenever supported.
extern "C" int cp360_wino_preferred(const cp360_wino_desc* d) {
    WinoGeom g;
    if (wino_check(d, &g)) return 0;
    static const int env = []() { const char* e = getenv("CP360_WINO"); return e ? atoi(e) : -1; }();
    if (env == 0) return 0;
    if (env == 1) return 1;
    if (d->c_out < 1024 || d->c_in < 256) return 0;
    return 16.0 * g.m_pad <= 0.8 * 9.0 * d->n_img * d->face * d->face ? 1 : 0;
}

extern "C" int cp360_wino_pack_weights(const cp360_wino_desc* d, const float* w_oihw, void* packed, void* stream) {
    WinoGeom g;
    int rc = wino_check(d, &g);
    if (rc) return rc;
    if (!w_oihw || !packed) return CP360_ERR_NULL;
    const long long total = (long long)g.nt * 256 * g.nsub * 4;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == CP360_F16)
        hipLaunchKernelGGL((wino_pack_kernel<f16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, w_oihw, (f16_raw*)packed, d->c_out, d->c_in, g.nt, g.nsub);
    else
        hipLaunchKernelGGL((wino_pack_kernel<bf16_raw>), dim3((unsigned)blocks), dim3(256), 0, st, w_oihw, (bf16_raw*)packed, d->c_out, d->c_in, g.nt, g.nsub);
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_wino_input(const cp360_wino_desc* d, const void* in, void* v, void* stream) {
    WinoGeom g;
    int rc = wino_check(d, &g);
    if (rc) return rc;
    if (!in || !v) return CP360_ERR_NULL;
    const int ncb = (g.nsub + 1) / 2, pw = 2 * g.th + 2;
    const size_t lds = (size_t)pw * pw * 128;
    if (lds > 64 * 1024) return CP360_ERR_UNSUPPORTED;                 // faces up to 42 x 42
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(d->n_img * ncb));
#define CP360_WIN(TT, THV)                                                                                                  \
    hipLaunchKernelGGL((wino_in_kernel<TT, THV>), grid, dim3(256), lds, st, (const TT*)in, (TT*)v, d->face, g.th, d->c_in, \
                       d->pix_stride, g.nsub, g.m_pad, ncb)
    if (d->dtype == CP360_F16) { if (g.th == 4) CP360_WIN(f16_raw, 4); else if (g.th == 8) CP360_WIN(f16_raw, 8); else CP360_WIN(f16_raw, 0); }
    else { if (g.th == 4) CP360_WIN(bf16_raw, 4); else if (g.th == 8) CP360_WIN(bf16_raw, 8); else CP360_WIN(bf16_raw, 0); }
#undef CP360_WIN
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_wino_gemm(const cp360_wino_desc* d, const void* v, const void* packed, float* m, void* stream) {
    WinoGeom g;
    int rc = wino_check(d, &g);
    if (rc) return rc;
    if (!v || !packed || !m) return CP360_ERR_NULL;
    WinoK k;
    k.u = (const unsigned char*)packed; k.v = (const unsigned char*)v; k.m = m;
    k.nsub = g.nsub; k.nt = g.nt; k.mt = g.mt; k.m_pad = g.m_pad; k.ldm = g.ldm; k.c_out = d->c_out;
    k.tpf = g.tpf; k.th = g.th; k.odd = d->face & 1;
    k.u_pin = wino_u_pin();
    k.reverse = cp360_launch_reverse();
    dim3 grid((unsigned)(16 * g.nt * g.mt));
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == CP360_F16) hipLaunchKernelGGL((wino_gemm_kernel<f16_raw>), grid, dim3(512), 0, st, k);
    else hipLaunchKernelGGL((wino_gemm_kernel<bf16_raw>), grid, dim3(512), 0, st, k);
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_wino_output(const cp360_wino_desc* d, const float* m, const float* bias, void* out, void* stream) {
    WinoGeom g;
    int rc = wino_check(d, &g);
    if (rc) return rc;
    if (!m || !out) return CP360_ERR_NULL;
    const int ld_out = d->ld_out ? d->ld_out : d->c_out;
    const long long total = (long long)g.tiles * (d->c_out / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
#define CP360_WO(TT, THV)                                                                                                      \
    hipLaunchKernelGGL((wino_out_kernel<TT, THV>), dim3((unsigned)blocks), dim3(256), 0, st, m, bias, (TT*)out, g.tiles, d->face, \
                       g.th, d->c_out, g.ldm, g.m_pad, ld_out, d->out_coff, d->relu)
#define CP360_WO_T(TT) { if (g.th == 4) CP360_WO(TT, 4); else if (g.th == 8) CP360_WO(TT, 8); else CP360_WO(TT, 0); }
    if (d->dtype == CP360_F16) CP360_WO_T(f16_raw)
    else CP360_WO_T(bf16_raw)
#undef CP360_WO_T
#undef CP360_WO
    CP360_CHECK_HIP();
    return CP360_OK;
}

// cp360_wino_output of THIS convolution fused with cp360_wino_input of the NEXT one (same faces, next c_in = this c_out, dense
// pixels): v_next receives what cp360_wino_input would make of this convolution's output - same bits, no activation tensor.
// CP360_ERR_UNSUPPORTED for faces above 9 x 9 (the cube's 32-channel image would exceed 40 KB of LDS): the caller then runs the two
// kernels.
extern "C" int cp360_wino_output_input(const cp360_wino_desc* d, const float* m, const float* bias, void* v_next, void* stream) {
    WinoGeom g;
    int rc = wino_check(d, &g);
    if (rc) return rc;
    if (!m || !v_next) return CP360_ERR_NULL;
    const int pw = 2 * g.th + 2, P = 6 * d->face * d->face;
    // 32 channels per workgroup: a cube's image must stay under 40 KB (faces up to 9 x 9).  The 16-channel form for 16 x 16 faces (49 KB,
    // 250 workgroups for one cube, 64-byte pieces of M) was built and measured at 57 us against 18 + 18 for the two kernels: not used.
    constexpr int cb = 32;
    const size_t lds = (size_t)((P * cb * 2 + 15) & ~15) + (size_t)6 * pw * pw * 2;
    if (lds > 40 * 1024) return CP360_ERR_UNSUPPORTED;
    const int nsub = (d->c_out + 31) / 32, nblk = nsub * (32 / cb);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((d->n_img / 6) * nblk));
#define CP360_WOI(TT, THV)                                                                                                      \
    hipLaunchKernelGGL((wino_out_in_kernel<TT, 32, THV>), grid, dim3(768), lds, st, m, bias, (TT*)v_next, d->face, g.th, d->c_out, \
                       g.ldm, g.m_pad, d->relu, nsub, nblk)
    if (d->dtype == CP360_F16) { if (g.th == 4) CP360_WOI(f16_raw, 4); else CP360_WOI(f16_raw, 0); }
    else { if (g.th == 4) CP360_WOI(bf16_raw, 4); else CP360_WOI(bf16_raw, 0); }
#undef CP360_WOI
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_wino_output_gates(const cp360_wino_desc* d, const float* m, const float* bias, const float* c_prev, float* c_next,
                                       void* h_out, int ld_h, int h_coff, float* h_f32, const float* x_next, const float* minmax,
                                       int x_coff, size_t clip_stride, void* stream) {
    WinoGeom g;
    int rc = wino_check(d, &g);
    if (rc) return rc;
    if (!m || !bias || !c_prev || !c_next || !h_out) return CP360_ERR_NULL;
    if (d->c_out % 16 != 0) return CP360_ERR_ALIGN;
    const int Hc = d->c_out / 4;
    if (ld_h % 4 != 0 || h_coff % 4 != 0 || h_coff + Hc > ld_h) return CP360_ERR_BAD_SHAPE;
    if (x_next && (!minmax || x_coff % 4 != 0 || clip_stride % 4 != 0 || x_coff + Hc > ld_h || (x_coff < h_coff + Hc && h_coff < x_coff + Hc)))
        return CP360_ERR_BAD_SHAPE;
    const long long total = (long long)g.tiles * (Hc / 4);
    if (total >= (1LL << 31)) return CP360_ERR_BAD_SHAPE;
    const long long blocks = (total + 63) / 64;                 // one 256-thread block per 64 items
    hipStream_t st = (hipStream_t)stream;
#define CP360_WG(TT, THV)                                                                                                           \
    hipLaunchKernelGGL((wino_gates_kernel<TT, THV>), dim3((unsigned)blocks), dim3(256), 0, st, m, bias, c_prev, c_next, (TT*)h_out, ld_h, \
                       h_coff, h_f32, g.tiles, d->face, g.th, Hc, g.ldm, g.m_pad, x_next, minmax, x_coff, clip_stride)
    if (d->dtype == CP360_F16) { if (g.th == 4) CP360_WG(f16_raw, 4); else if (g.th == 8) CP360_WG(f16_raw, 8); else CP360_WG(f16_raw, 0); }
    else { if (g.th == 4) CP360_WG(bf16_raw, 4); else if (g.th == 8) CP360_WG(bf16_raw, 8); else CP360_WG(bf16_raw, 0); }
#undef CP360_WG
    CP360_CHECK_HIP();
    return CP360_OK;
}

// in -> V -> M -> out: the whole convolution (v / m: workspaces of cp360_wino_v_bytes / cp360_wino_m_bytes)
extern "C" int cp360_wino_forward(const cp360_wino_desc* d, const void* in, const void* packed, const float* bias, void* out, void* v,
                                  float* m, void* stream) {
    int rc;
    if ((rc = cp360_wino_input(d, in, v, stream))) return rc;
    if ((rc = cp360_wino_gemm(d, v, packed, m, stream))) return rc;
    return cp360_wino_output(d, m, bias, out, stream);
}

#if defined(WINO_LAB) && defined(WINO_STAMPS)
extern "C" int cp360_wino_stamps_read(unsigned long long* host) {      // diagnostic build only
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_stamps), sizeof(g_wino_stamps)) == hipSuccess ? 0 : CP360_ERR_HIP;
}
#endif

// the bare GEMM on caller-made operands (tools/wino_probe.py)
extern "C" int cp360_wino_gemm_raw(int dtype, const void* u, const void* v, float* m, int nsub, int nt, int mt, int ldm,
                                   int c_out, void* stream) {
    if (!u || !v || !m) return CP360_ERR_NULL;
    if (dtype != CP360_BF16 && dtype != CP360_F16) return CP360_ERR_BAD_DTYPE;
    if (nsub < 1 || nt < 1 || mt < 1 || ldm % 4 != 0 || c_out % 4 != 0 || c_out > nt * WG_BN || ldm < c_out) return CP360_ERR_BAD_SHAPE;
    WinoK k;
    k.u = (const unsigned char*)u; k.v = (const unsigned char*)v; k.m = m;
    k.nsub = nsub; k.nt = nt; k.mt = mt; k.m_pad = mt * WG_BM; k.ldm = ldm; k.c_out = c_out;
    k.tpf = 16; k.th = 4; k.odd = 0;
    k.u_pin = wino_u_pin();
    k.reverse = cp360_launch_reverse();
    dim3 grid((unsigned)(16 * nt * mt));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CP360_F16) hipLaunchKernelGGL((wino_gemm_kernel<f16_raw>), grid, dim3(512), 0, st, k);
    else hipLaunchKernelGGL((wino_gemm_kernel<bf16_raw>), grid, dim3(512), 0, st, k);
    CP360_CHECK_HIP();
    return CP360_OK;
}
