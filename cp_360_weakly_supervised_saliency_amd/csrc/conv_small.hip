// K3 small-M form: implicit-GEMM convolution with a 64 (output channels) x 64 (pixels) tile per 8-wave workgroup.
//
// Why it exists (profiles/r04a_c2_fp32_static_timeline.md): the reference's static driver processes ONE frame = 6 cube
// faces per iteration (static_model/dataset_feat_extractor.py:119-162, class_activation_model.py:55-83), so layers 2-4 of
// ResNet-50-cubic (model/resnet_cubic.py:85-106) are GEMMs of M = 4704 / 1176 / 294 pixels.  The big tiles of
// conv_igemm.hip (256 x 128 ... 256 x 304, 8 waves) cut such a launch into 12-150 workgroups for 256 CUs, and in f32
// (v_mfma_f32_16x16x4_f32: 32 cycles per SIMD for 1024 MACs) a 256 x 128 tile is 3.4 us PER 128-byte K step: the
// launches were MFMA-bound on a sixth of the chip.  Here a tile is 1/8 of that, so the same launch gives 8x the
// workgroups before any split-K, and a K step is 1024 MFMA cycles per SIMD.
//
// Structure.  Register-staged global loads (two steps ahead) -> swizzled LDS (two buffers) -> fragments -> MFMA, one
// barrier per K step.  The 64 x 64 tile is four 32 x 32 wave tiles; EACH is computed by TWO waves that sit on the same
// SIMD and split every 128-byte K step between them (waves 0-3 take its first 64 bytes, waves 4-7 the second; the two
// accumulator sets are added through LDS once, after the K loop).  A first 4-wave version of this kernel spent 40 % of a
// step outside the matrix pipe (barrier, fragment reads, LDS stores, address arithmetic, then 32 MFMAs, strictly in that
// order: 0.86 us per f32 step against 0.46 of MFMA time) - with two waves per SIMD one wave's overhead runs under the
// other's MFMAs, PROVIDED they are not in lockstep (MI355X_MICROARCH.md, "Two waves per SIMD", item 9): waves 4-7
// therefore trail by one step - after the barrier they first multiply the fragments they fetched in the previous
// iteration and do their loads afterwards, while waves 0-3 do their loads first and multiply afterwards.
//
// With the acc_chan row order of the packed weights the two row blocks of a wave give a lane EIGHT consecutive channels
// of one pixel: bias, residual, ReLU, one rounding and the store (or the split-K slab store) are 16-byte pieces
// straight against global memory.  Supports everything the ring kernels do for the ResNet / CAM call sites: CubePad
// gather (cubepad_src) in the loader, stride, the stem's pixel-run taps, split-K slabs (both column orders), raw f32 sums,
// and the second source (the Bottleneck's downsample branch as one more 1x1 "tap" of conv3's K loop,
// resnet_cubic.py:99-100).
#include "conv_common.h"
#include <stdlib.h>

namespace {

// One wave's share of a K step: one 64-byte half (16 f32 / 32 16-bit k values) of its 32 x 32 tile.  f32: a lane's 16-byte
// chunk holds its four k values of four CONSECUTIVE v_mfma_f32_16x16x4_f32 (k = 4 * (lane >> 4) + e); the four accumulators
// are walked inside the e loop, so two MFMAs on the same accumulator are four issues apart.
template <typename T>
__device__ __forceinline__ void mma_half(f32x4 (&acc)[2][2], const u32x4 (&a)[2], const u32x4 (&b)[2]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i][e]), __uint_as_float(b[j][e]), acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma_chunk<T>(acc[i][j], a[i], b[j]);
    }
}

template <typename T>
__global__ __launch_bounds__(512, 2) void conv_small_kernel(const ConvK p) {
    constexpr int BN = 64, BM = 64, NT = 512;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;                        // 128 bytes of K per tile row and step
    constexpr int STAGE = (BN + BM) * 128;
    constexpr int MAX_TAPS = CP360_SMALL_MAX_TAPS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2;                        // 0: leads (loads, then MFMAs); 1: trails by one step (MFMAs, then loads)
    const int wn = (wave >> 1) & 1, wm = wave & 1;
    // XCD-aware work mapping (as conv_igemm_kernel): every XCD takes a contiguous range of work items, ordered so that
    // neighbours share the larger operand panel through that XCD's L2.  Placement affects speed only.
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
    const int chunk = tid & 7, row0 = tid >> 3;        // this thread's 16-byte chunk of weight row row0 AND of pixel row row0

    // ---- source-offset table: element offset of tile row r's input pixel for tap t, offtab[t * 64 + r] (-1: no such output
    // pixel).  All divisions and the branchy cubepad_src() run HERE, once per (tap, row); inside the K loop a tap change is
    // one LDS read per thread.  (Computed in the loop - as conv_igemm_kernel does - the divergent code and the waits the
    // compiler merges at its join points cost more than the MFMAs of a 64 x 64 tile.)
    __shared__ int offtab[MAX_TAPS * BM];
    {
        const int ntap_all = p.ntap + (p.c_in2 > 0 ? 1 : 0);
        const CubePadGeom geom{p.h_in, p.pad, p.pad, p.pad, p.pad};
        for (int t = tid; t < ntap_all * BM; t += NT) {
            const int tp = t >> 6, r = t & 63;
            const int m = m0 + r;
            int off = -1;
            if (m < p.M) {
                const int img = m / p.hw_out, rem = m - img * p.hw_out;
                const int oy = rem / p.w_out, ox = rem - oy * p.w_out;
                if (tp >= p.ntap) {
                    off = ((img * p.h_in2 + oy * p.sy2) * p.w_in2 + ox * p.sx2) * p.pix_stride2;
                } else {
                    const int ky = tp / p.kw, kx = tp - ky * p.kw;
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
            }
            offtab[t] = off;
        }
    }
    __syncthreads();
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* in2 = reinterpret_cast<const T*>(p.in2);
    const T* src = in;                                 // tensor of the current tap (the second source for tap == ntap)
    int src_cin = p.c_in, src_cpad = p.c_pad;
    int roff;
    unsigned avoff = 0;                                // byte offset of this thread's chunk inside the tap's pixel (0 for an invalid row)
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const bool second = tap >= p.ntap;
        roff = offtab[tap * BM + row0];
        avoff = (unsigned)((roff < 0 ? 0 : roff) + chunk * EPC) * (unsigned)sizeof(T);
        src = second ? in2 : in;
        src_cin = second ? p.c_in2 : p.c_in;
        src_cpad = second ? p.c_pad2 : p.c_pad;
    };

    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
    const int first_steps = p.ntap * p.steps_per_tap;
    int tap, c0;
    if (s_begin < first_steps) {
        tap = s_begin / p.steps_per_tap;
        c0 = (s_begin - tap * p.steps_per_tap) * BK;
    } else {
        tap = p.ntap;
        c0 = (s_begin - first_steps) * BK;
    }
    const unsigned wvoff = (unsigned)(row0 * p.k_total + chunk * EPC) * (unsigned)sizeof(T);     // this thread's chunk inside the tile's weight rows
    const T* wtile = reinterpret_cast<const T*>(p.w) + (size_t)n0 * p.k_total;

    // Pipeline.  Iteration `it`, every wave: barrier -> [waves 4-7: the MFMAs of step it-1 on the fragments fetched in the
    // previous iteration] -> read the fragments of step `it` (LDS buffer it & 1, stored in the previous iteration) -> store
    // the staged global data of step it+1 (staging set (it+1) & 1) into buffer (it+1) & 1 (every wave's reads of that buffer
    // completed before the barrier: __syncthreads drains lgkmcnt, not vmcnt) -> request step it+3 from global memory into the
    // staging set just freed -> [waves 0-3: the MFMAs of step `it`].  A global load has two steps to land (activations the
    // previous launch has just written come from another XCD's L2 / the Infinity Cache) - see the staging sets below for
    // how far ahead the requests really run.  The body is unconditional: past the last step the loads repeat the last
    // step's addresses and nobody reads what they bring.
    // FOUR staging sets (this thread's weight chunk + activation chunk of a step each): what bounds this kernel at one
    // workgroup per CU is bytes in flight / latency - a step moves 16 KB into the CU and a load that misses the XCD's L2
    // (activations the previous launch wrote on another XCD, weights on their first touch) takes ~0.7 us, so two sets
    // (32 KB in flight) give 0.36 us per step for the loads ALONE and, interleaved with the MFMA phases, 0.72 us per step
    // against 0.43 of MFMA time (tools/exp_small.sh, tools/exp_small_stamps.sh)
    u32x4 sa0, sb0, sa1, sb1, sa2, sb2, sa3, sb3;
    int sm0 = 0, sm1 = 0, sm2 = 0, sm3 = 0;               // ... and whether the activation chunk is real (all ones) or must read as zero
    // The two global loads of a step: "saddr + 32-bit voffset" form, issued from inline asm.  The per-step part of both
    // addresses is wave-uniform (tile's weight rows + tap / channel offset; tensor + channel offset) and lives in SGPRs, the
    // per-thread part changes only with the tap - a step costs scalar adds and two VMEM issues instead of ~20 VALU of 64-bit
    // address arithmetic (tools/exp_small_stamps.sh: 440-670 of a step's ~1600 cycles sat there).  An invalid row (no such
    // output pixel) and the K tail load from a clamped in-bounds address and are zeroed after they land (staged()).
    // The loads are waited for by hand (s_waitcnt vmcnt(6): the six loads of the OTHER three staging sets stay in flight);
    // with compiler-visible loads the waitcnt pass, merging the loop's entry and back edges, drained them a step early.
    auto gload = [&](u32x4& sa, u32x4& sb, int& sm) __attribute__((always_inline)) {
        const T* wsb = wtile + (size_t)tap * p.c_pad + c0;                      // uniform
        const T* asb = src + c0;                                                 // uniform
        sm = ((c0 + chunk * EPC - src_cin) >> 31) & ~(roff >> 31);               // all ones: a real chunk
        const unsigned av = avoff & (unsigned)sm;
        asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %4, %5"
                     : "=&v"(sa), "=&v"(sb) : "v"(wvoff), "s"(wsb), "v"(av), "s"(asb) : "memory");
    };
    auto staged = [&](u32x4& sa, u32x4& sb, int sm) __attribute__((always_inline)) {     // this set has landed (the other may be in flight)
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(sa), "+v"(sb) :: "memory");
        sb.x &= (unsigned)sm; sb.y &= (unsigned)sm; sb.z &= (unsigned)sm; sb.w &= (unsigned)sm;
    };
    auto lds_store = [&](int buf, const u32x4& sa, const u32x4& sb) __attribute__((always_inline)) {
        unsigned char* As = lds + buf * STAGE;
        *reinterpret_cast<u32x4*>(As + lds_swz(row0, chunk)) = sa;
        *reinterpret_cast<u32x4*>(As + BN * 128 + lds_swz(row0, chunk)) = sb;
    };
    const int nloc = s_end - s_begin;
    int issued = 0;                                       // K steps requested so far
    auto next_gload = [&](u32x4& sa, u32x4& sb, int& sm) __attribute__((always_inline)) {
        if (issued > 0 && issued < nloc) {                // (uniform) move on to the next step; past the end: repeat the last
            c0 += BK;
            if (c0 >= src_cpad) {
                c0 = 0;
                ++tap;
                set_tap(tap);
            }
        }
        ++issued;
        gload(sa, sb, sm);
        // keep the requests HERE, in front of the MFMAs: left alone the scheduler sinks them to just before the next
        // barrier, where the LDS store right behind it waits out their whole latency every step
        __builtin_amdgcn_sched_barrier(0);
    };
    const int lrow = lane & 15, lchunk = half * 4 + (lane >> 4);        // this wave's 64-byte half of the step
    auto frag_load = [&](int buf, u32x4 (&a)[2], u32x4 (&b)[2]) __attribute__((always_inline)) {
        const unsigned char* As = lds + buf * STAGE;
        const unsigned char* Bs = As + BN * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const u32x4*>(As + lds_swz(wn * 32 + i * 16 + lrow, lchunk));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const u32x4*>(Bs + lds_swz(wm * 32 + j * 16 + lrow, lchunk));
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nloc > 0) {
        u32x4 fa[2], fb[2];
        set_tap(tap);
        next_gload(sa0, sb0, sm0);                       // step 0
        next_gload(sa1, sb1, sm1);                       // step 1
        next_gload(sa2, sb2, sm2);                       // step 2
        next_gload(sa3, sb3, sm3);                       // step 3
        staged(sa0, sb0, sm0);
        lds_store(0, sa0, sb0);
        next_gload(sa0, sb0, sm0);                       // step 4
        // one K step: LDS buffer `it & 1` holds it; staging set (it + 1) & 3 holds step it+1, the other three are in flight
#define CP360_SMALL_STEP(IT, BUF, SA, SB, SM)                                                        \
        {                                                                                            \
            __syncthreads();                                                                         \
            if (half && (IT) > 0) mma_half<T>(acc, fa, fb);          /* trailing waves: step IT-1 */ \
            __builtin_amdgcn_sched_barrier(0);                                                       \
            frag_load(BUF, fa, fb);                                                                  \
            staged(SA, SB, SM);                                                                      \
            lds_store((BUF) ^ 1, SA, SB);                                                            \
            next_gload(SA, SB, SM);                                  /* step IT+5 */                 \
            if (!half) mma_half<T>(acc, fa, fb);                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                       \
        }
        for (int it = 0; it < nloc; it += 4) {
            CP360_SMALL_STEP(it, 0, sa1, sb1, sm1);
            if (it + 1 >= nloc) break;
            CP360_SMALL_STEP(it + 1, 1, sa2, sb2, sm2);
            if (it + 2 >= nloc) break;
            CP360_SMALL_STEP(it + 2, 0, sa3, sb3, sm3);
            if (it + 3 >= nloc) break;
            CP360_SMALL_STEP(it + 3, 1, sa0, sb0, sm0);
        }
#undef CP360_SMALL_STEP
        if (half) mma_half<T>(acc, fa, fb);                         // trailing waves: the last step
        // the redundant loads past the last step are still in flight: their registers must not be reused before they land
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sa0), "+v"(sb0), "+v"(sa1), "+v"(sb1), "+v"(sa2), "+v"(sb2), "+v"(sa3), "+v"(sb3) :: "memory");
    }

    // ---- the trailing waves hand their sums over through LDS (the pipeline buffers are free): [pair][register][lane]
    __syncthreads();
    float* xch = reinterpret_cast<float*>(lds);
    const int pair = wave & 3;
    if (half) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) xch[((pair * 16) + (i * 2 + j) * 4 + e) * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (half) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] += xch[((pair * 16) + (i * 2 + j) * 4 + e) * 64 + lane];

    // ---- epilogue: a lane owns channels n .. n+7 of pixel m for each of its two pixel blocks
    const int ml = lane & 15;
    const int n = n0 + wn * 32 + (lane >> 4) * 8;
    if (n >= p.c_out) return;                              // c_out % 8 == 0 on this path
    float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!p.partial && p.bias) {
        const float4 t0 = *reinterpret_cast<const float4*>(p.bias + n);
        const float4 t1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bb[0] = t0.x; bb[1] = t0.y; bb[2] = t0.z; bb[3] = t0.w; bb[4] = t1.x; bb[5] = t1.y; bb[6] = t1.z; bb[7] = t1.w;
    }
    const T* res = reinterpret_cast<const T*>(p.res);
    T* outp = reinterpret_cast<T*>(p.out);
    u32x4 rr[2][sizeof(T) == 4 ? 2 : 1];
    if (!p.partial && res) {                               // both pixel blocks' residual pieces in flight together
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 32 + j * 16 + ml;
            const T* s = m < p.M ? res + (size_t)m * p.ld_res + n : reinterpret_cast<const T*>(g_zero16);
            rr[j][0] = *reinterpret_cast<const u32x4*>(s);
            if constexpr (sizeof(T) == 4) rr[j][1] = *reinterpret_cast<const u32x4*>(m < p.M ? s + 4 : s);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 32 + j * 16 + ml;
        if (m >= p.M) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = acc[0][j][e];
            v[4 + e] = acc[1][j][e];
        }
        if (p.partial) {
            float* dst = p.partial + ((size_t)split * p.M + m) * p.c_out;
            store4(dst + (p.slab_rows ? slab_col(n) : n), v);
            store4(dst + (p.slab_rows ? slab_col(n + 4) : n + 4), v + 4);
            continue;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bb[e];
        if (res) {
            float r[8];
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r[e] = __uint_as_float(rr[j][0][e]);
                    r[4 + e] = __uint_as_float(rr[j][1][e]);
                }
            } else {
                unpack8(rr[j][0], r, T());
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        T* dst = outp + (size_t)m * p.ld_out + p.out_coff + n;
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            *reinterpret_cast<u32x4*>(dst) = pack8(v, T());
        }
    }
}

// ------------------------------------------------------------------ f32 form: one wave per SIMD, fillers inside the MFMA gaps
// What the ablations of the 8-wave kernel above say ("The one-frame regime"): with EVERY load served from
// one L1-resident line it still takes 0.70 us per f32 step against 0.43 of MFMA time, and the chain without any MFMA
// (barrier, fragment reads, LDS stores, address arithmetic, load issue) takes 0.29 us - the two add.  A wave issues in order:
// its ~45 non-MFMA instructions of a step and their LDS round trip run either before or after its block of MFMAs, and the
// partner wave's MFMA stream slows them further.  In f32 an MFMA occupies the pipe for 32 cycles but the issuing wave for 8:
// here ONE wave per SIMD issues the step's 32 MFMAs and places one or two of the other instructions in each gap (the placement
// is pinned with sched_barrier; cdna_hip_programming.md, 4-wave attention structure: "budget every MFMA gap to <= 5 issues"):
// the next step's eight fragment reads, the staged chunks' LDS stores, the address update and the four global loads.
// Everything else - tile, work mapping, source-offset table, epilogue, split-K, second source - is the kernel above.
__global__ __launch_bounds__(256, 2) void conv_small_f32_kernel(const ConvK p) {
    typedef float T;
    constexpr int BN = 64, BM = 64, NT = 256;
    constexpr int EPC = 4, BK = 32;
    constexpr int STAGE = (BN + BM) * 128;
    constexpr int MAX_TAPS = CP360_SMALL_MAX_TAPS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

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
    const int chunk = tid & 7, row0 = tid >> 3;        // this thread's chunk of weight rows row0, row0 + 32 and pixel rows row0, row0 + 32

    __shared__ int offtab[MAX_TAPS * BM];              // see conv_small_kernel
    {
        const int ntap_all = p.ntap + (p.c_in2 > 0 ? 1 : 0);
        const CubePadGeom geom{p.h_in, p.pad, p.pad, p.pad, p.pad};
        for (int t = tid; t < ntap_all * BM; t += NT) {
            const int tp = t >> 6, r = t & 63;
            const int m = m0 + r;
            int off = -1;
            if (m < p.M) {
                const int img = m / p.hw_out, rem = m - img * p.hw_out;
                const int oy = rem / p.w_out, ox = rem - oy * p.w_out;
                if (tp >= p.ntap) {
                    off = ((img * p.h_in2 + oy * p.sy2) * p.w_in2 + ox * p.sx2) * p.pix_stride2;
                } else {
                    const int ky = tp / p.kw, kx = tp - ky * p.kw;
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
            }
            offtab[t] = off;
        }
    }
    __syncthreads();
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* in2 = reinterpret_cast<const T*>(p.in2);
    const T* src = in;
    int src_cin = p.c_in, src_cpad = p.c_pad;
    int roff0, roff1;
    unsigned avoff0 = 0, avoff1 = 0;
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const bool second = tap >= p.ntap;
        roff0 = offtab[tap * BM + row0];
        roff1 = offtab[tap * BM + row0 + 32];
        avoff0 = (unsigned)((roff0 < 0 ? 0 : roff0) + chunk * EPC) * 4u;
        avoff1 = (unsigned)((roff1 < 0 ? 0 : roff1) + chunk * EPC) * 4u;
        src = second ? in2 : in;
        src_cin = second ? p.c_in2 : p.c_in;
        src_cpad = second ? p.c_pad2 : p.c_pad;
    };
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
    const int first_steps = p.ntap * p.steps_per_tap;
    int tap, c0;
    if (s_begin < first_steps) {
        tap = s_begin / p.steps_per_tap;
        c0 = (s_begin - tap * p.steps_per_tap) * BK;
    } else {
        tap = p.ntap;
        c0 = (s_begin - first_steps) * BK;
    }
    const unsigned wvoff0 = (unsigned)(row0 * p.k_total + chunk * EPC) * 4u;
    const unsigned wvoff1 = wvoff0 + (unsigned)(32 * p.k_total) * 4u;
    const T* wtile = reinterpret_cast<const T*>(p.w) + (size_t)n0 * p.k_total;
    const int nloc = s_end - s_begin;
    int issued = 0;

    // four staging sets (the K steps it+2 .. it+5 at iteration it): weight chunks a0 / a1, activation chunks b0 / b1, masks
    u32x4 sA0[4], sA1[4], sB0[4], sB1[4];
    int sM0[4], sM1[4];
    const T* cur_w = wtile;                               // uniform bases of the step being requested
    const T* cur_a = src;
    auto advance = [&]() __attribute__((always_inline)) {  // (uniform) the next step to request; past the end: repeat the last
        if (issued > 0 && issued < nloc) {
            c0 += BK;
            if (c0 >= src_cpad) {
                c0 = 0;
                ++tap;
                set_tap(tap);
            }
        }
        ++issued;
        cur_w = wtile + (size_t)tap * p.c_pad + c0;
        cur_a = src + c0;
    };
#define CP360_S4_LOADW(DST, VOFF) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(DST) : "v"(VOFF), "s"(cur_w) : "memory")
#define CP360_S4_LOADA(DST, MSK, VOFF, ROFF)                                                       \
    {                                                                                              \
        MSK = ((c0 + chunk * EPC - src_cin) >> 31) & ~((ROFF) >> 31);                              \
        const unsigned av_ = (VOFF) & (unsigned)(MSK);                                             \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(DST) : "v"(av_), "s"(cur_a) : "memory"); \
    }
#define CP360_S4_REQUEST(S)                                                                        \
    {                                                                                              \
        advance();                                                                                 \
        CP360_S4_LOADW(sA0[S], wvoff0);                                                            \
        CP360_S4_LOADW(sA1[S], wvoff1);                                                            \
        CP360_S4_LOADA(sB0[S], sM0[S], avoff0, roff0);                                             \
        CP360_S4_LOADA(sB1[S], sM1[S], avoff1, roff1);                                             \
    }
    // the four chunks of staging set S have landed (the 12 loads of the other three sets may be in flight); invalid rows / K tail -> 0
#define CP360_S4_LANDED(S)                                                                         \
    {                                                                                              \
        asm volatile("s_waitcnt vmcnt(12)" : "+v"(sA0[S]), "+v"(sA1[S]), "+v"(sB0[S]), "+v"(sB1[S]) :: "memory"); \
        sB0[S].x &= (unsigned)sM0[S]; sB0[S].y &= (unsigned)sM0[S]; sB0[S].z &= (unsigned)sM0[S]; sB0[S].w &= (unsigned)sM0[S]; \
        sB1[S].x &= (unsigned)sM1[S]; sB1[S].y &= (unsigned)sM1[S]; sB1[S].z &= (unsigned)sM1[S]; sB1[S].w &= (unsigned)sM1[S]; \
    }
    const int lrow = lane & 15, lchunk = lane >> 4;
    const int st_off = lds_swz(row0, chunk), st_off1 = lds_swz(row0 + 32, chunk);
    int fa_off[2][2], fb_off[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa_off[kk][i] = lds_swz(wn * 32 + i * 16 + lrow, kk * 4 + lchunk);
            fb_off[kk][i] = BN * 128 + lds_swz(wm * 32 + i * 16 + lrow, kk * 4 + lchunk);
        }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nloc > 0) {
        u32x4 fa[2][2][2], fb[2][2][2];                   // two fragment sets [set][kk][block]
        set_tap(tap);
        CP360_S4_REQUEST(0)                               // steps 0 .. 3
        CP360_S4_REQUEST(1)
        CP360_S4_REQUEST(2)
        CP360_S4_REQUEST(3)
        CP360_S4_LANDED(0)
        *reinterpret_cast<u32x4*>(lds + st_off) = sA0[0];
        *reinterpret_cast<u32x4*>(lds + st_off1) = sA1[0];
        *reinterpret_cast<u32x4*>(lds + BN * 128 + st_off) = sB0[0];
        *reinterpret_cast<u32x4*>(lds + BN * 128 + st_off1) = sB1[0];
        CP360_S4_REQUEST(0)                               // step 4
        CP360_S4_LANDED(1)
        *reinterpret_cast<u32x4*>(lds + STAGE + st_off) = sA0[1];
        *reinterpret_cast<u32x4*>(lds + STAGE + st_off1) = sA1[1];
        *reinterpret_cast<u32x4*>(lds + STAGE + BN * 128 + st_off) = sB0[1];
        *reinterpret_cast<u32x4*>(lds + STAGE + BN * 128 + st_off1) = sB1[1];
        CP360_S4_REQUEST(1)                               // step 5
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[0][kk][i] = *reinterpret_cast<const u32x4*>(lds + fa_off[kk][i]);
                fb[0][kk][i] = *reinterpret_cast<const u32x4*>(lds + fb_off[kk][i]);
            }
        // One K step: the 32 MFMAs of step IT on fragment set CUR (LDS buffer BUF holds it), and in their gaps - pinned in this
        // order - the eight reads of step IT+1's fragments (buffer BUF ^ 1, stored in the previous iteration) into set CUR ^ 1,
        // the LDS stores of staging set S (step IT+2; its fragments of step IT left buffer BUF an iteration ago) into buffer BUF,
        // the address update and the four loads of step IT+6 into the set just stored.
        // MFMA m = (kk, e, i, j): two MFMAs on the same accumulator are four issues apart.
#define CP360_S4_MF(CUR, M)                                                                        \
        {                                                                                          \
            constexpr int kk_ = (M) >> 4, e_ = ((M) >> 2) & 3, i_ = ((M) >> 1) & 1, j_ = (M) & 1;  \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fa[CUR][kk_][i_][e_]), __uint_as_float(fb[CUR][kk_][j_][e_]), \
                                                               acc[i_][j_], 0, 0, 0);             \
        }
#define CP360_S4_PIN __builtin_amdgcn_sched_barrier(0);
#define CP360_S4_STEP(CUR, BUF, S)                                                                 \
        {                                                                                          \
            unsigned char* rd_ = lds + ((BUF) ^ 1) * STAGE;                                        \
            unsigned char* wr_ = lds + (BUF) * STAGE;                                              \
            __syncthreads();                                                                       \
            CP360_S4_PIN                                                                           \
            CP360_S4_MF(CUR, 0)  fa[(CUR) ^ 1][0][0] = *reinterpret_cast<const u32x4*>(rd_ + fa_off[0][0]); CP360_S4_PIN \
            CP360_S4_MF(CUR, 1)  fb[(CUR) ^ 1][0][0] = *reinterpret_cast<const u32x4*>(rd_ + fb_off[0][0]); CP360_S4_PIN \
            CP360_S4_MF(CUR, 2)  fa[(CUR) ^ 1][0][1] = *reinterpret_cast<const u32x4*>(rd_ + fa_off[0][1]); CP360_S4_PIN \
            CP360_S4_MF(CUR, 3)  fb[(CUR) ^ 1][0][1] = *reinterpret_cast<const u32x4*>(rd_ + fb_off[0][1]); CP360_S4_PIN \
            CP360_S4_MF(CUR, 4)  fa[(CUR) ^ 1][1][0] = *reinterpret_cast<const u32x4*>(rd_ + fa_off[1][0]); CP360_S4_PIN \
            CP360_S4_MF(CUR, 5)  fb[(CUR) ^ 1][1][0] = *reinterpret_cast<const u32x4*>(rd_ + fb_off[1][0]); CP360_S4_PIN \
            CP360_S4_MF(CUR, 6)  fa[(CUR) ^ 1][1][1] = *reinterpret_cast<const u32x4*>(rd_ + fa_off[1][1]); CP360_S4_PIN \
            CP360_S4_MF(CUR, 7)  fb[(CUR) ^ 1][1][1] = *reinterpret_cast<const u32x4*>(rd_ + fb_off[1][1]); CP360_S4_PIN \
            CP360_S4_MF(CUR, 8)  CP360_S4_LANDED(S) CP360_S4_PIN                                   \
            CP360_S4_MF(CUR, 9)  *reinterpret_cast<u32x4*>(wr_ + st_off) = sA0[S]; CP360_S4_PIN    \
            CP360_S4_MF(CUR, 10) *reinterpret_cast<u32x4*>(wr_ + st_off1) = sA1[S]; CP360_S4_PIN   \
            CP360_S4_MF(CUR, 11) *reinterpret_cast<u32x4*>(wr_ + BN * 128 + st_off) = sB0[S]; CP360_S4_PIN  \
            CP360_S4_MF(CUR, 12) *reinterpret_cast<u32x4*>(wr_ + BN * 128 + st_off1) = sB1[S]; CP360_S4_PIN \
            CP360_S4_MF(CUR, 13) advance(); CP360_S4_PIN                                           \
            CP360_S4_MF(CUR, 14) CP360_S4_LOADW(sA0[S], wvoff0); CP360_S4_PIN                      \
            CP360_S4_MF(CUR, 15) CP360_S4_LOADW(sA1[S], wvoff1); CP360_S4_PIN                      \
            CP360_S4_MF(CUR, 16) CP360_S4_LOADA(sB0[S], sM0[S], avoff0, roff0) CP360_S4_PIN        \
            CP360_S4_MF(CUR, 17) CP360_S4_LOADA(sB1[S], sM1[S], avoff1, roff1) CP360_S4_PIN        \
            CP360_S4_MF(CUR, 18) CP360_S4_MF(CUR, 19) CP360_S4_MF(CUR, 20) CP360_S4_MF(CUR, 21)    \
            CP360_S4_MF(CUR, 22) CP360_S4_MF(CUR, 23) CP360_S4_MF(CUR, 24) CP360_S4_MF(CUR, 25)    \
            CP360_S4_MF(CUR, 26) CP360_S4_MF(CUR, 27) CP360_S4_MF(CUR, 28) CP360_S4_MF(CUR, 29)    \
            CP360_S4_MF(CUR, 30) CP360_S4_MF(CUR, 31)                                              \
            CP360_S4_PIN                                                                           \
        }
        // K step s travels in staging set s & 3 (steps 0 and 1 were stored by the prologue, their sets re-requested as 4 and 5)
        for (int it = 0; it < nloc; it += 4) {
            CP360_S4_STEP(0, 0, 2)
            if (it + 1 >= nloc) break;
            CP360_S4_STEP(1, 1, 3)
            if (it + 2 >= nloc) break;
            CP360_S4_STEP(0, 0, 0)
            if (it + 3 >= nloc) break;
            CP360_S4_STEP(1, 1, 1)
        }
#undef CP360_S4_STEP
#undef CP360_S4_PIN
#undef CP360_S4_MF
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sA0[0]), "+v"(sA0[1]), "+v"(sA0[2]), "+v"(sA0[3]), "+v"(sA1[0]), "+v"(sA1[1]), "+v"(sA1[2]),
                     "+v"(sA1[3]) :: "memory");
        asm volatile("" : "+v"(sB0[0]), "+v"(sB0[1]), "+v"(sB0[2]), "+v"(sB0[3]), "+v"(sB1[0]), "+v"(sB1[1]), "+v"(sB1[2]), "+v"(sB1[3]) :: "memory");
    }
#undef CP360_S4_LANDED
#undef CP360_S4_REQUEST
#undef CP360_S4_LOADA
#undef CP360_S4_LOADW

    // ---- epilogue: a lane owns channels n .. n+7 of pixel m for each of its two pixel blocks (as conv_small_kernel)
    const int ml = lane & 15;
    const int n = n0 + wn * 32 + (lane >> 4) * 8;
    if (n >= p.c_out) return;
    float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!p.partial && p.bias) {
        const float4 t0 = *reinterpret_cast<const float4*>(p.bias + n);
        const float4 t1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bb[0] = t0.x; bb[1] = t0.y; bb[2] = t0.z; bb[3] = t0.w; bb[4] = t1.x; bb[5] = t1.y; bb[6] = t1.z; bb[7] = t1.w;
    }
    const T* res = reinterpret_cast<const T*>(p.res);
    T* outp = reinterpret_cast<T*>(p.out);
    u32x4 rr[2][2];
    if (!p.partial && res) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 32 + j * 16 + ml;
            const T* sp = m < p.M ? res + (size_t)m * p.ld_res + n : reinterpret_cast<const T*>(g_zero16);
            rr[j][0] = *reinterpret_cast<const u32x4*>(sp);
            rr[j][1] = *reinterpret_cast<const u32x4*>(m < p.M ? sp + 4 : sp);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 32 + j * 16 + ml;
        if (m >= p.M) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = acc[0][j][e];
            v[4 + e] = acc[1][j][e];
        }
        if (p.partial) {
            float* dst = p.partial + ((size_t)split * p.M + m) * p.c_out;
            store4(dst + (p.slab_rows ? slab_col(n) : n), v);
            store4(dst + (p.slab_rows ? slab_col(n + 4) : n + 4), v + 4);
            continue;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bb[e];
        if (res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += __uint_as_float(rr[j][0][e]);
                v[4 + e] += __uint_as_float(rr[j][1][e]);
            }
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        T* dst = outp + (size_t)m * p.ld_out + p.out_coff + n;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

}  // namespace

void cp360_launch_conv_small(ConvK& k, int dtype, hipStream_t st) {
    k.nt = (k.c_out + 63) / 64;
    k.mt = (k.M + 63) / 64;
    // share whichever operand panel is larger through the XCD's L2
    k.m_fast = ((long long)k.c_out * k.k_total > (long long)k.M * k.kh * k.kw * k.c_in) ? 1 : 0;
    dim3 grid((unsigned)(k.nt * k.mt * k.splits), 1, 1);
    // f32: one wave per SIMD with the fillers inside the MFMA gaps (CP360_SMALL_F32=8 keeps the 8-wave form as an A/B path)
    static const int f32_form = []() { const char* e = getenv("CP360_SMALL_F32"); return e ? atoi(e) : 4; }();
    if (dtype == CP360_F32 && f32_form != 8) hipLaunchKernelGGL(conv_small_f32_kernel, grid, dim3(256), 0, st, k);
    else if (dtype == CP360_F32) hipLaunchKernelGGL((conv_small_kernel<float>), grid, dim3(512), 0, st, k);
    else if (dtype == CP360_F16) hipLaunchKernelGGL((conv_small_kernel<f16_raw>), grid, dim3(512), 0, st, k);
    else hipLaunchKernelGGL((conv_small_kernel<bf16_raw>), grid, dim3(512), 0, st, k);
}
