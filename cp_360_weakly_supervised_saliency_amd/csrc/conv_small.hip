// K3 small-M form: implicit-GEMM convolution with a 64 (output channels) x 64 (pixels) tile per 4-wave workgroup.
//
// Why it exists (profiles/r04a_c2_fp32_static_timeline.md): the reference's static driver processes ONE frame = 6 cube
// faces per iteration (static_model/dataset_feat_extractor.py:119-162, class_activation_model.py:55-83), so layers 2-4 of
// ResNet-50-cubic (model/resnet_cubic.py:85-106) are GEMMs of M = 4704 / 1176 / 294 pixels.  The big tiles of
// conv_igemm.hip (256 x 128 ... 256 x 304, 8 waves) cut such a launch into 12-150 workgroups for 256 CUs, and in f32
// (v_mfma_f32_16x16x4_f32: 32 cycles per SIMD for 1024 MACs) a 256 x 128 tile is 3.4 us PER 128-byte K step: the
// launches were MFMA-bound on a sixth of the chip.  Here a tile is 1/8 of that, so the same launch gives 8x the
// workgroups before any split-K, and a K step is 1024 cycles per SIMD.
//
// Structure: register-staged global loads -> swizzled LDS (two buffers) -> fragments in registers one step AHEAD of the
// MFMAs that use them (the pipeline comment in the kernel), one barrier per K step, wave tile 32 x 32 = 2 x 2 MFMA blocks.  With the acc_chan row order of the
// packed weights the two row blocks of a wave give a lane EIGHT consecutive channels of one pixel: bias, residual,
// ReLU, one rounding and the store (or the split-K slab store) are 16-byte pieces straight against global memory.
// Supports everything the ring kernels do for the ResNet / CAM call sites: CubePad gather (cubepad_src) in the loader,
// stride, the stem's pixel-run taps, split-K slabs (both column orders), raw f32 sums, and the second source (the
// Bottleneck's downsample branch as one more 1x1 "tap" of conv3's K loop, resnet_cubic.py:99-100).
#include "conv_common.h"

namespace {

// One K step of a wave's 32 x 32 tile.  f32: a lane's 16-byte chunk holds its four k values of four CONSECUTIVE
// v_mfma_f32_16x16x4_f32 (k = 4 * (lane >> 4) + e); the four accumulators are walked inside the e loop so that two
// MFMAs on the same accumulator are four issues apart (dependent latency 40 cycles against an issue interval of 32).
template <typename T>
__device__ __forceinline__ void mma_step(f32x4 (&acc)[2][2], const u32x4 (&a)[2][2], const u32x4 (&b)[2][2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[kk][i][e]), __uint_as_float(b[kk][j][e]),
                                                                         acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma_chunk<T>(acc[i][j], a[kk][i], b[kk][j]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_small_kernel(const ConvK p) {
    constexpr int BN = 64, BM = 64;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;                        // 128 bytes of K per tile row and step
    constexpr int STAGE = (BN + BM) * 128;
    constexpr int MAX_TAPS = CP360_SMALL_MAX_TAPS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
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
    const int chunk = tid & 7, row0 = tid >> 3;

    // ---- source-offset table: element offset of tile row r's input pixel for tap t, offtab[t * 64 + r] (-1: no such output
    // pixel).  All divisions and the branchy cubepad_src() run HERE, once per (tap, row); inside the K loop a tap change is
    // two LDS reads per thread.  (Computed in the loop - as conv_igemm_kernel does - the divergent code and the waits the
    // compiler merges at its join points cost more than the MFMAs of a 64 x 64 tile: 0.94 us per f32 K step against 0.45.)
    __shared__ int offtab[MAX_TAPS * BM];
    {
        const int ntap_all = p.ntap + (p.c_in2 > 0 ? 1 : 0);
        const CubePadGeom geom{p.h_in, p.pad, p.pad, p.pad, p.pad};
        for (int t = tid; t < ntap_all * BM; t += 256) {
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
    int roff[2];
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const bool second = tap >= p.ntap;
        roff[0] = offtab[tap * BM + row0];
        roff[1] = offtab[tap * BM + row0 + 32];
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
    const T* wbase = reinterpret_cast<const T*>(p.w) + (size_t)(n0 + row0) * p.k_total + chunk * EPC;
    const size_t wpass = (size_t)32 * p.k_total;

    // Pipeline.  Iteration `it` computes K step `it` from fragments that are already in registers:
    //   barrier -> read the fragments of step it+1 (LDS buffer (it+1) & 1) into the OTHER fragment set -> store the staged
    //   global data of step it+2 (staging set it & 1) into buffer it & 1 (every wave's reads of that buffer completed before
    //   the barrier: __syncthreads drains lgkmcnt, not vmcnt) -> request step it+4 from global memory into the staging set
    //   just freed -> the 32 (f32) MFMAs of step it.
    // So the matrix pipe never waits for LDS (a step's fragments are requested one whole MFMA phase before their use) and a
    // global load has TWO phases to land (one was not enough: activations the previous launch has just written come from
    // another XCD's L2 / the Infinity Cache, > 1000 cycles under load); one wave per SIMD keeps its pipe busy without a
    // partner.  The body is unconditional: past the last step the loads repeat the last step's addresses and nobody reads
    // what they bring (four redundant 64-byte loads per thread against a branch-free loop).
    u32x4 ra0[2], rb0[2], ra1[2], rb1[2];                 // two staging sets: one K step of this thread's 2 + 2 tile-row chunks each
    // address of the zero page, pinned in a VGPR pair: rematerialised inside the loop it is a scalar load per step whose
    // lgkmcnt(0) wait also drains the fragment reads just issued
    unsigned long long z = (unsigned long long)reinterpret_cast<size_t>(g_zero16);
    asm volatile("" : "+v"(z));
    auto gload = [&](u32x4 (&ra)[2], u32x4 (&rb)[2]) __attribute__((always_inline)) {
        const int e = c0 + chunk * EPC;
        const size_t koff = (size_t)tap * p.c_pad + c0;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) ra[pa] = *reinterpret_cast<const u32x4*>(wbase + pa * wpass + koff);
        // invalid rows (no such output pixel) and the K tail read the 16 zero bytes of g_zero16.  The select is done with a
        // mask on the address difference, not "ok ? a : z": the compiler turned that into a divergent branch on the K-tail
        // test whose join waits for vmcnt(0) - the weight loads just issued - in front of every step's MFMAs
        const int ktail = (e - src_cin) >> 31;            // all ones while e < src_cin
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const long long mask = (long long)(ktail & ~(roff[pb] >> 31));
            const unsigned long long a = (unsigned long long)reinterpret_cast<size_t>(src + (size_t)(roff[pb] & 0x7fffffff) + e);
            // (an explicit global-address-space pointer: from a plain integer the compiler makes a FLAT load, which also
            // counts in lgkmcnt - the next barrier's LDS drain would then wait for it)
            typedef const u32x4 __attribute__((address_space(1)))* gptr_t;
            rb[pb] = *reinterpret_cast<gptr_t>(z + ((a - z) & (unsigned long long)mask));
        }
    };
    auto lds_store = [&](int buf, const u32x4 (&ra)[2], const u32x4 (&rb)[2]) __attribute__((always_inline)) {
        unsigned char* As = lds + buf * STAGE;
        unsigned char* Bs = As + BN * 128;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) *reinterpret_cast<u32x4*>(As + lds_swz(row0 + 32 * pa, chunk)) = ra[pa];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) *reinterpret_cast<u32x4*>(Bs + lds_swz(row0 + 32 * pb, chunk)) = rb[pb];
    };
    const int nloc = s_end - s_begin;
    int issued = 0;                                       // K steps requested so far
    auto next_gload = [&](u32x4 (&ra)[2], u32x4 (&rb)[2]) __attribute__((always_inline)) {
        if (issued > 0 && issued < nloc) {                // (uniform) move on to the next step; past the end: repeat the last
            c0 += BK;
            if (c0 >= src_cpad) {
                c0 = 0;
                ++tap;
                set_tap(tap);
            }
        }
        ++issued;
        gload(ra, rb);
        // keep the requests HERE, in front of the step's MFMAs: left alone the scheduler sinks them past the MFMAs to just
        // before the barrier, where the LDS store right behind it waits out their whole latency every step
        __builtin_amdgcn_sched_barrier(0);
    };
    const int lrow = lane & 15, lchunk = lane >> 4;
    auto frag_load = [&](int buf, u32x4 (&a)[2][2], u32x4 (&b)[2][2]) __attribute__((always_inline)) {
        const unsigned char* As = lds + buf * STAGE;
        const unsigned char* Bs = As + BN * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[kk][i] = *reinterpret_cast<const u32x4*>(As + lds_swz(wn * 32 + i * 16 + lrow, kk * 4 + lchunk));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[kk][j] = *reinterpret_cast<const u32x4*>(Bs + lds_swz(wm * 32 + j * 16 + lrow, kk * 4 + lchunk));
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nloc > 0) {
        u32x4 fa0[2][2], fb0[2][2], fa1[2][2], fb1[2][2];
        set_tap(tap);
        next_gload(ra0, rb0);                            // step 0
        next_gload(ra1, rb1);                            // step 1
        lds_store(0, ra0, rb0);
        next_gload(ra0, rb0);                            // step 2
        __syncthreads();
        frag_load(0, fa0, fb0);
        lds_store(1, ra1, rb1);
        next_gload(ra1, rb1);                            // step 3
        for (int it = 0; it < nloc; it += 2) {
            // even step: its fragments are in set 0; buffer 1 holds step it+1, staging set 0 step it+2, set 1 (in flight) it+3
            __syncthreads();
            frag_load(1, fa1, fb1);
            lds_store(0, ra0, rb0);
            next_gload(ra0, rb0);                        // step it+4
            mma_step<T>(acc, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 >= nloc) break;
            // odd step
            __syncthreads();
            frag_load(0, fa0, fb0);
            lds_store(1, ra1, rb1);
            next_gload(ra1, rb1);                        // step it+5
            mma_step<T>(acc, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

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

}  // namespace

void cp360_launch_conv_small(ConvK& k, int dtype, hipStream_t st) {
    k.nt = (k.c_out + 63) / 64;
    k.mt = (k.M + 63) / 64;
    // share whichever operand panel is larger through the XCD's L2
    k.m_fast = ((long long)k.c_out * k.k_total > (long long)k.M * k.kh * k.kw * k.c_in) ? 1 : 0;
    dim3 grid((unsigned)(k.nt * k.mt * k.splits), 1, 1);
    if (dtype == CP360_F32) hipLaunchKernelGGL((conv_small_kernel<float>), grid, dim3(256), 0, st, k);
    else if (dtype == CP360_F16) hipLaunchKernelGGL((conv_small_kernel<f16_raw>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((conv_small_kernel<bf16_raw>), grid, dim3(256), 0, st, k);
}
