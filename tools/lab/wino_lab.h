// LABORATORY ONLY - never part of libcp360.so.  Included by csrc/wino.hip in place of its product GEMM when that file is compiled
// with -DWINO_LAB (tools/wino_variants.sh, tools/wino_stamps.sh, tools/wino_cell_ab.sh): the round-5 form of the Winograd GEMM with
// every A/B knob (cache-policy IDs, DMA placements, barrier count, wave priorities, dead-row handling), the timing ablations
// (WINO_ABL) and the in-kernel s_memtime stamps (WINO_STAMPS).  With no -D beyond WINO_LAB it computes what the product build
// computes (the defaults below ARE the shipped settings); docs/rounds/r05.md has the tables these knobs produced.
// One LDS-DMA instruction = 1 KiB (16 tile rows x 64 B): "scalar base + 32-bit lane offset" addressing, M0 = the LDS destination.
// Both operand blocks of a sub-step are contiguous (16 KiB of U, 24 KiB of V), so a sub-step's fill is 40 such instructions.
// Cache policy of the U (weight) / V stream loads: 0 default, 1 nt, 2 sc1, 3 sc0 sc1 (tools/wino_variants.sh A/B: no difference)
#ifndef WINO_UPOL_ID
#define WINO_UPOL_ID 0
#endif
#ifndef WINO_VPOL_ID
#define WINO_VPOL_ID 0
#endif
#if WINO_UPOL_ID == 1
#define WINO_UPOL " nt"
#elif WINO_UPOL_ID == 2
#define WINO_UPOL " sc1"
#elif WINO_UPOL_ID == 3
#define WINO_UPOL " sc0 sc1"
#else
#define WINO_UPOL ""
#endif
#if WINO_VPOL_ID == 1
#define WINO_VPOL " nt"
#elif WINO_VPOL_ID == 2
#define WINO_VPOL " sc1"
#elif WINO_VPOL_ID == 3
#define WINO_VPOL " sc0 sc1"
#else
#define WINO_VPOL ""
#endif
// u_nt: this piece of the weight stream is non-temporal.  The GEMM loads only the first WinoK::u_pin sub-steps of every workgroup's
// U block with the default policy: the 1.3 GB of weights a cell update streams no longer sweep the 256 MB Infinity Cache, so the
// 86 MB of M and the 49 MB of V between the launches stay in it, and the HEAD of every workgroup's U stream - what all 256
// workgroups ask for at once when a launch starts - is still there from the previous cell update (tools/wino_upin_probe.sh:
// cell update 480 us with everything default, 474 all non-temporal, 454 with the first 4-8 sub-steps default, 460+ from 16 up).
__device__ __forceinline__ void fill_one(const unsigned char* base, unsigned off, unsigned dst, bool is_u, bool u_nt = false) {
    unsigned keep;
    if (is_u && u_nt)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
    else if (is_u)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" WINO_UPOL "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" WINO_VPOL "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
}

// ---- structure knobs (defaults = what tools/wino_stamps_ab.sh measured fastest, in cycles per sub-step by in-kernel stamps; the others
// stay buildable for A/B.  docs/rounds/r05.md has the table: 2620 cycles for the first version, 2348 for the defaults)
// WINO_NO_B2 1 (default): one barrier per sub-step - the second only paced the two wave groups, no LDS hazard depends on it (2620 -> 2476).
// WINO_DMA_PLACE 0 (default since the weight stream became non-temporal): the refill's DMAs in front of the HEAD's MFMAs; 2: spread
//   behind the TAIL's MFMA columns (the default while the stream swept the Infinity Cache: 2476 -> 2388 cycles then; now 469 against 461 us
//   per cell update, five alternations on one box); 3: behind the TAIL's last MFMA (499 us).
// WINO_HEAD_PRIO 1 (default): a lagging wave raises its priority for its HEAD's MFMAs: both waves of a SIMD then reach the barrier
//   together (-> 2348; a STATIC priority for either group only swaps who waits).
// WINO_LEAD_DMA 0 (default); 1: the four leading waves (which win the matrix-pipe arbitration and wait ~600 cycles at the barrier)
//   issue the whole fill, 10 DMAs each, the lagging waves none - measured SLOWER (2660): the lagging HEAD got longer.
#ifndef WINO_LEAD_DMA
#define WINO_LEAD_DMA 0
#endif
#ifndef WINO_DMA_PLACE
#define WINO_DMA_PLACE 0
#endif
// WINO_HEAD_PRIO p > 0: a LAGGING wave raises its priority to p for the MFMAs of its HEAD (A/B)
#ifndef WINO_HEAD_PRIO
#define WINO_HEAD_PRIO 1
#endif
#ifndef WINO_NO_B2
#define WINO_NO_B2 1
#endif
#ifndef WINO_SKIP_DEAD
#define WINO_SKIP_DEAD 3    // bit 0: the GEMM does not store the never-read rows of M at odd faces; bit 1: the input transforms zero their V rows (A/B)
#endif
#ifndef WINO_MSTORE_NT
#define WINO_MSTORE_NT 0    // cache policy of the slab stores (A/B): 1 nt (+25 us per cell update: M must stay in the Infinity Cache), 2 sc1, 3 sc0 sc1
#endif
#ifndef WINO_MLOAD_NT
#define WINO_MLOAD_NT 1     // bit 0: wino_out_in's loads of M are non-temporal (measured: -1.2 us per launch), bit 1: wino_gates' (+0.9 us: off)
#endif
#ifndef WINO_ABL
#define WINO_ABL 0          // timing ablations (tools/wino_variants.sh): 1 = every U block aliases the first, 2 = every V block, 4 = no MFMA
#endif
#if WINO_NO_B2
#define WINO_B2()
#else
#define WINO_B2() __builtin_amdgcn_s_barrier();
#endif

// Diagnostic build only (-DWINO_STAMPS, tools/wino_stamps.sh): s_memtime stamps of waves 0 and 4 around the phases of sub-steps
// WINO_STAMP_S0 .. +7 go to a buffer of their own; the product build executes no stamp.
#ifdef WINO_STAMPS
#ifndef WINO_STAMP_S0
#define WINO_STAMP_S0 40
#endif
__device__ unsigned long long g_wino_stamps[256 * 2 * 8 * 6];
#define WINO_STAMP(k)                                                                                      \
    if ((wave & 3) == 0 && lane == 0 && it >= WINO_STAMP_S0 && it < WINO_STAMP_S0 + 8 && blockIdx.x < 256)    \
        g_wino_stamps[((blockIdx.x * 2 + (wave >> 2)) * 8 + (it - WINO_STAMP_S0)) * 6 + (k)] = __builtin_amdgcn_s_memtime();
#define WINO_STAMP_END() __builtin_amdgcn_sched_barrier(0); WINO_STAMP(5)
#else
#define WINO_STAMP(k)
#define WINO_STAMP_END()
#endif

// K loop + slab stores of one wave: channels [wch0, +64) x tile rows [wrow0, +192) = 4 x 12 MFMA blocks (192 accumulator
// registers).  A sub-step is a HEAD (fragment reads of its LDS stage, the first 6 columns; a column's registers are re-loaded
// with column 6 + j as soon as its MFMAs are issued) and a TAIL (the other 6 columns, from registers); the two waves of a
// SIMD (w and w + 4) run half a sub-step apart (waves 4-7 LAG: their TAIL of sub-step s-1 comes before their HEAD of s).
template <typename T, bool LAG, int MJ_ = 12>
__device__ __forceinline__ void gemm_body(const WinoK& p, unsigned char* lds, const int pos, const int nt_i, const int mt_i,
                                          const int wave, const int lane, const int wch0, const int wrow0) {
    constexpr int NS = WG_NS, MJ = MJ_, JH = MJ_ / 2;
    constexpr bool LOADER = !WINO_LEAD_DMA || !LAG;              // this wave issues DMAs
    constexpr int NU = WINO_LEAD_DMA ? 10 : 5;                   // ... this many per sub-step
    const int nsub = p.nsub;
    const unsigned char* ub = p.u + ((size_t)(pos * p.nt + nt_i) * nsub) * (WG_BN * 64);
    const size_t vstep = (size_t)p.m_pad * 64;
    const unsigned char* vb = p.v + (size_t)pos * nsub * vstep + (size_t)mt_i * (WG_BM * 64);
    // DMA role: lane l lands in row 16 * slot + (l >> 2) (+ 128 per pass), physical chunk l & 3, and fetches the logical chunk of
    // that row (source-side swizzle); slot = the wave, and for a leading wave that loads for its partner also wave + 4 (+ 4 KiB)
    const unsigned o0 = (unsigned)((16 * wave + (lane >> 2)) * 64 + ((((lane & 3) ^ ((0 - ((4 * wave + (lane >> 4)) & 3)) & 3))) << 4));
    const unsigned lds_wave = (unsigned)(size_t)lds + (unsigned)wave * 1024;

    constexpr int MJS = (WINO_ABL & 16) ? 12 : MJ;               // (ablation 16: the skipped column blocks are still stored)
    f32x4 acc[4][MJS];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MJS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, lchunk = lane >> 4;
    unsigned rdst = 0;                                            // LDS destination (this wave's slot) of the refill in progress
    int fills = 0;                                                // sub-steps requested so far (uniform)
    // unit u of a sub-step's fill: pass q = u % 5 (0, 1: the U block's two 128-row passes; 2, 3, 4: the V block's three), slot half
    // u / 5 (the partner's rows: + 64 rows = + 4 KiB on both sides)
    auto dma = [&](int u) __attribute__((always_inline)) {
        const int q = u % 5, half = u / 5;
        const unsigned src = o0 + (unsigned)((q < 2 ? q : q - 2) * 0x2000 + half * 0x1000);
        fill_one(q < 2 ? ub : vb, src, rdst + (unsigned)(q * 0x2000 + half * 0x1000), q < 2, fills >= p.u_pin);
    };
    auto refill_begin = [&](int stage) __attribute__((always_inline)) {
        rdst = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)stage * WG_STAGE);
    };
    auto refill_end = [&]() __attribute__((always_inline)) {
        if (!(WINO_ABL & 1)) ub += WG_BN * 64;
        if (!(WINO_ABL & 2)) vb += vstep;
        ++fills;
    };
    if (LOADER) {
#pragma unroll
        for (int k = 0; k < NS - 1; ++k)
            if (k < nsub) {
                refill_begin(k);
#pragma unroll
                for (int u = 0; u < NU; ++u) dma(u);
                refill_end();
            }
    }
    int stage = 0;
    u32x4 a[4], b[JH];
    if (LAG) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < JH; ++j) b[j] = u32x4{0u, 0u, 0u, 0u};
    }
#if WINO_ABL & 4
#define WINO_MMA(cc, aa, bb) cc[0] = __uint_as_float(__float_as_uint(cc[0]) ^ aa[0] ^ bb[0])
#else
#define WINO_MMA(c, x, y) mma_chunk<T>(c, x, y)
#endif
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
        if (LOADER && REFILL && (WINO_DMA_PLACE == 0 || !LAG)) refill_begin(stage == 0 ? NS - 1 : stage - 1); \
        if (LOADER && REFILL && WINO_DMA_PLACE == 0) {                                                     \
            _Pragma("unroll") for (int u = 0; u < NU; ++u) dma(u);                                         \
            refill_end();                                                                                  \
        }                                                                                                  \
        if (WINO_HEAD_PRIO > 0 && LAG) __builtin_amdgcn_s_setprio(WINO_HEAD_PRIO);                         \
        _Pragma("unroll") for (int j = 0; j < JH; ++j) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) WINO_MMA(acc[i][j], a[i], b[j]);                 \
            b[j] = *reinterpret_cast<const u32x4*>(Bs + swz64(wrow0 + (JH + j) * 16 + lrow, lchunk));      \
        }                                                                                                  \
        if (WINO_HEAD_PRIO > 0 && LAG) __builtin_amdgcn_s_setprio(0);                                      \
        stage = stage == NS - 1 ? 0 : stage + 1;                                                           \
    }
#define WINO_TAIL(REFILL)                                                                                  \
    {                                                                                                      \
        if (LOADER && REFILL && WINO_DMA_PLACE != 0 && LAG) refill_begin(stage == 0 ? NS - 1 : stage - 1); \
        _Pragma("unroll") for (int j = JH; j < MJ; ++j) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) WINO_MMA(acc[i][j], a[i], b[j - JH]);            \
            if (LOADER && REFILL && WINO_DMA_PLACE == 2)                                                   \
                _Pragma("unroll") for (int u = (j - JH) * (NU / 5); u < (j == MJ - 1 ? NU : (j - JH + 1) * (NU / 5)) && u < NU; ++u) dma(u); \
        }                                                                                                  \
        if (LOADER && REFILL && WINO_DMA_PLACE == 3)                                                       \
            _Pragma("unroll") for (int u = 0; u < NU; ++u) dma(u);                                         \
        if (LOADER && REFILL && WINO_DMA_PLACE != 0) refill_end();                                         \
    }
#define WINO_STEP(REFILL)                                                                                  \
    {                                                                                                      \
        if (LAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
        WINO_STAMP(1)                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        WINO_STAMP(2)                                                                                      \
        if (!LAG) WINO_HEAD(REFILL) else WINO_TAIL(REFILL)                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        WINO_STAMP(3)                                                                                      \
        WINO_B2()                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        WINO_STAMP(4)                                                                                      \
        if (!LAG) WINO_TAIL(REFILL) else WINO_HEAD(REFILL)                                                 \
        WINO_STAMP_END()                                                                                   \
    }
    // vmcnt (DMAs complete in issue order): before sub-step `it` its stage must have landed; younger are the NS - 2 stages behind it
    int it = 0;
    for (; it + NS - 1 < nsub; ++it) {
        WINO_STAMP(0)
        if (LOADER) vm_wait<(NS - 2) * NU>();
        WINO_STEP(true)
    }
    for (; it < nsub; ++it) {
        if (LOADER) vm_wait_upto<(NS - 2) * NU>(min(NS - 2, nsub - 1 - it) * NU);
        WINO_STEP(false)
    }
    if (LAG) WINO_TAIL(false)
#undef WINO_STEP
#undef WINO_HEAD
#undef WINO_TAIL

    // slabs: M[pos][tile][channel], a lane owns 4 consecutive channels of one tile per block (16-byte stores).  Tile rows outermost:
    // the four 64-byte pieces of a row's 256 bytes leave back to back and merge into full lines in L2
    float* mp = p.m + ((size_t)pos * p.m_pad + (size_t)mt_i * WG_BM + wrow0 + lrow) * p.ldm + nt_i * WG_BN + wch0 + lchunk * 4;
    const bool edge_pos = (WINO_SKIP_DEAD & 1) && p.odd && ((pos >> 2) == 3 || (pos & 3) == 3);        // (uniform) this position has never-read tiles
    int t0 = 0;                                                                 // this lane's tile inside its face, block 0
    if (edge_pos) t0 = (mt_i * WG_BM + wrow0 + lrow) % p.tpf;
#pragma unroll
    for (int j = 0; j < MJS; ++j) {
        bool dead = false;
        if (edge_pos) {
            const int t = p.tpf == 16 ? t0 : (t0 + j * 16) % p.tpf, ty = t / p.th, tx = t - ty * p.th;
            dead = wino_dead_row(pos, ty, p.th) || wino_dead_col(pos, tx, p.th);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!(WINO_ABL & 32) && !dead && nt_i * WG_BN + wch0 + i * 16 + lchunk * 4 < p.c_out) {   // (ablation 32: no slab stores)
#if WINO_MSTORE_NT == 1
                __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(mp + (size_t)j * 16 * p.ldm + i * 16));
#elif WINO_MSTORE_NT == 2
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(mp + (size_t)j * 16 * p.ldm + i * 16), "v"(acc[i][j]) : "memory");
#elif WINO_MSTORE_NT == 3
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(mp + (size_t)j * 16 * p.ldm + i * 16), "v"(acc[i][j]) : "memory");
#else
                *reinterpret_cast<f32x4*>(mp + (size_t)j * 16 * p.ldm + i * 16) = acc[i][j];
#endif
            }
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
#if WINO_ABL & 8                                                    // timing ablation: positions whose last tile row / column is unused skip column blocks
    const int cls = ((pos >> 2) == 3) + ((pos & 3) == 3);
    if (cls == 1) {
        if (wave < 4) gemm_body<T, false, 10>(p, lds, pos, nt_i, mt_i, wave, lane, wch0, wrow0);
        else          gemm_body<T, true, 10>(p, lds, pos, nt_i, mt_i, wave, lane, wch0, wrow0);
        return;
    }
    if (cls == 2) {
        if (wave < 4) gemm_body<T, false, 8>(p, lds, pos, nt_i, mt_i, wave, lane, wch0, wrow0);
        else          gemm_body<T, true, 8>(p, lds, pos, nt_i, mt_i, wave, lane, wch0, wrow0);
        return;
    }
#endif
    if (wave < 4) gemm_body<T, false>(p, lds, pos, nt_i, mt_i, wave, lane, wch0, wrow0);
    else          gemm_body<T, true>(p, lds, pos, nt_i, mt_i, wave, lane, wch0, wrow0);
}


// the transform kernels' knobs (product: constants in wino.hip)
#ifndef WINO_DL_OUTIN
#define WINO_DL_OUTIN 2
#endif
#ifndef WINO_TAB_LATE
#define WINO_TAB_LATE 0     // wino_out_in: 1 = the CubePad source table is built behind the first loads of M (measured: more registers, slower)
#endif
#ifndef WINO_DL_GATES
#define WINO_DL_GATES 1
#endif
#ifndef WINO_DL_OUT
#define WINO_DL_OUT 1
#endif
constexpr bool kZeroDeadV = (WINO_SKIP_DEAD & 2) != 0;
constexpr int kDlOutIn = WINO_DL_OUTIN, kDlGates = WINO_DL_GATES, kDlOut = WINO_DL_OUT;
constexpr bool kTabLate = WINO_TAB_LATE != 0, kNtOutIn = (WINO_MLOAD_NT & 1) != 0, kNtGates = (WINO_MLOAD_NT & 2) != 0;
