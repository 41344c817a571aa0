// K2: CubePad as stand-alone HIP kernels (NCHW for the reference's module boundary,
// NHWC for the fused pipeline) + the host-side table export.
// Semantics: model/cube_pad.py:28-42,95-216 (see common.h: cubepad_src).
//
// NCHW kernels, in the order launch_nchw() tries them (each is bit-exact; A/B switches in launch_nchw; measurements and
// the reasoning behind the designs: profiles/r03_cubepad_plane.md):
//   cubepad_nchw_cube_kernel     faces up to 16x16 (fallback up to 32x32): (cube, channel range) items through LDS, the
//                                CubePad map as a per-workgroup table
//   cubepad_nchw_lds6_kernel     the six padded planes of a (cube, channel range) fit the LDS: assembled there, written
//                                as one linear store stream (the network's 28x28 ... 112x112 faces)
//   cubepad_nchw_band_kernel     rows of >= 256 bytes, planes larger than the LDS: the same per row band of one plane
//   cubepad_nchw_channel_kernel  (cube, channel) items, every input byte read once, pads from LDS captures
//   cubepad_nchw_plane_kernel    (plane, face) items, runs + pad stream
//   cubepad_nchw_strip_kernel    round-2 form of the plane kernel
//   cubepad_nchw_kernel          element per lane, any geometry
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <atomic>

// ---------------------------------------------------------------- NCHW
// One workgroup per (group g, channel c): it writes the 6 padded planes of that
// channel.  Lanes run along the output row (coalesced stores; the centre rows are
// coalesced loads too, the strips are short gathers from neighbouring faces that
// sit in L2 because the same workgroup reads those faces as its own centres).
// LX = lanes per output row (power of two >= Wp, capped at 64) so small faces
// (9x9 ConvLSTM tiles) still fill a wave with several rows.  Rows of at most 64 elements: corner stitches by wavefront shuffle.
template <typename T, int LOG_LX>
__global__ __launch_bounds__(256) void cubepad_nchw_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                           int C, CubePadGeom g) {
    constexpr int LX = 1 << LOG_LX;
    constexpr int ROWS = 256 / LX;
    const int n = g.n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr;
    const int plane = blockIdx.x;          // = grp * C + c
    const int grp = plane / C, c = plane - grp * C;
    const int lx = threadIdx.x & (LX - 1), ly = threadIdx.x >> LOG_LX;
    const size_t in_face = (size_t)C * n * n, out_face = (size_t)C * Hp * Wp;
    const T* xin = x + (size_t)grp * 6 * in_face + (size_t)c * n * n;
    T* yout = y + (size_t)grp * 6 * out_face + (size_t)c * Hp * Wp;
    if (Wp <= LX) {
        // A padded row sits inside one wave (LX <= 64 consecutive lanes): the corner stitch is a WAVEFRONT SHUFFLE.  A corner of
        // make_cubepad_edge (cube_pad.py:83-90,164-176) whose top / down pad is not deeper than its left / right pad repeats
        // the end element of the top / down strip along the row - the element the lane at column pl (left corners) or
        // pl + n - 1 (right corners) of the SAME row has just gathered - so the corner lanes take it from that lane instead of
        // evaluating cubepad_src() and gathering it again.  (The other case - a deeper top / down pad - repeats a left / right
        // strip element down a COLUMN, i.e. across rows that other waves own: those corners keep the gather.)
        const int j = lx;
        const bool in_l = j < g.pl, in_r = j >= g.pl + n;
        const int src_lane = (int)(threadIdx.x & 63 & ~(LX - 1)) + (in_l ? g.pl : g.pl + n - 1);
        for (int f = 0; f < 6; ++f) {
            for (int i = ly; i < Hp; i += ROWS) {
                const bool in_t = i < g.pt, in_d = i >= g.pt + n;
                const bool stitched = (in_l | in_r) && (in_t | in_d) && (in_t ? g.pt : g.pd) <= (in_l ? g.pl : g.pr);
                T v = 0;
                if (j < Wp && !stitched) {
                    const int s = cubepad_src(f, i, j, g);   // f'*n*n + i'*n + j'
                    const int sf = s / (n * n);
                    v = xin[(size_t)sf * in_face + (s - sf * n * n)];
                }
                // every lane of the row's segment executes the shuffle (rows of one segment share i, so a corner's source lane is live)
                T sv;
                if constexpr (sizeof(T) == 8) sv = (T)__shfl((long long)v, src_lane);
                else sv = (T)__shfl((int)v, src_lane);
                if (j < Wp) yout[(size_t)f * out_face + (size_t)i * Wp + j] = stitched ? sv : v;
            }
        }
        return;
    }
    for (int f = 0; f < 6; ++f) {
        for (int i = ly; i < Hp; i += ROWS) {
            for (int j = lx; j < Wp; j += LX) {
                const int s = cubepad_src(f, i, j, g);       // f'*n*n + i'*n + j'
                const int sf = s / (n * n);
                yout[(size_t)f * out_face + (size_t)i * Wp + j] = xin[(size_t)sf * in_face + (s - sf * n * n)];
            }
        }
    }
}

// ---------------------------------------------------------------- NCHW, 16-byte chunks + LDS-staged border strips
// The element-per-lane kernel above moves 1-4 bytes per lane and runs cubepad_src() for every element:
// 0.09-0.17 of the HBM peak (profiles/r02_hbm_kernels.md).  A padded plane [Hp, Wp] is one contiguous byte range,
// and all but ~2 of the 16-byte chunks of a centre row are 16 consecutive bytes of the input plane: those go
// straight through (one possibly misaligned 16-byte load - unaligned global access is on under ROCm - and
// one ALIGNED 16-byte store).  Everything a pad element can read lies within P = max pad of a face border
// (cube_pad.py:114-216 copies border strips of the neighbouring faces, corners replicate a strip's end): the
// the four pad strips of a face (output orientation) are gathered into LDS once per (plane, face) and the
// chunks that touch padding are assembled element by element from there; the <= 4 p^2 corner elements go through
// cubepad_src() directly.  One WAVE per (group, channel, face) item, 32-bit index arithmetic, divisions by
// Wp / n through a float reciprocal + correction.  ES = element bytes.
template <int ES> struct ElemOf;
template <> struct ElemOf<1> { typedef uint8_t T; };
template <> struct ElemOf<2> { typedef uint16_t T; };
template <> struct ElemOf<4> { typedef uint32_t T; };
template <> struct ElemOf<8> { typedef uint64_t T; };
typedef __attribute__((ext_vector_type(4))) unsigned int cp_u32x4;

template <int ES>
__global__ __launch_bounds__(256) void cubepad_nchw_strip_kernel(const unsigned char* __restrict__ x,
                                                                 unsigned char* __restrict__ y, int C, CubePadGeom g,
                                                                 int n_items, float rcp_wp, float rcp_n) {
    typedef typename ElemOf<ES>::T T;
    constexpr int E = 16 / ES;                               // elements per 16-byte chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char strip_raw[];
    const int n = g.n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nstrip = (g.pt + g.pd + g.pl + g.pr) * n;      // one wave's strips: top [pt][n], bottom [pd][n], left [pl][n], right [pr][n]
    T* st = reinterpret_cast<T*>(strip_raw) + (size_t)wave * nstrip;
    const T* st_t = st;
    const T* st_b = st_t + g.pt * n;
    const T* st_l = st_b + g.pd * n;
    const T* st_r = st_l + g.pl * n;
    const size_t in_face = (size_t)C * n * n, out_face = (size_t)C * Hp * Wp;
    const int plane_elems = Hp * Wp;
    // exact unsigned division by a small constant through a float reciprocal + one correction (operands < 2^24)
    auto div_by = [](int q, int d, float rcp) -> int {
        int r = (int)((float)q * rcp);
        r -= (r * d > q);
        r += ((r + 1) * d <= q);
        return r;
    };
    // a (plane, face) item per WAVE: small planes (14x14, 7x7 faces) would leave a 256-thread workgroup idle
    for (int item = blockIdx.x * 4 + wave; item < n_items; item += gridDim.x * 4) {
        const int plane = item / 6, f = item - plane * 6;
        const int grp = plane / C, c = plane - grp * C;
        const T* xin = reinterpret_cast<const T*>(x) + (size_t)grp * 6 * in_face + (size_t)c * n * n;
        const T* xf = xin + (size_t)f * in_face;
        auto src_elem = [&](int i, int j) -> T {            // any padded position through the CubePad map (global gather)
            const int s = cubepad_src(f, i, j, g);
            const int sf = div_by(s, n * n, rcp_n * rcp_n);
            return xin[(size_t)sf * in_face + (s - sf * n * n)];
        };
        // ---- stage this face's four pad strips (output orientation, corners excluded) in LDS
        for (int idx = lane; idx < nstrip; idx += 64) {
            const int k = div_by(idx, n, rcp_n), a = idx - k * n;
            int i, j;
            if (k < g.pt)                      { i = k;                              j = g.pl + a; }
            else if (k < g.pt + g.pd)          { i = g.pt + n + (k - g.pt);          j = g.pl + a; }
            else if (k < g.pt + g.pd + g.pl)   { i = g.pt + a;                       j = k - g.pt - g.pd; }
            else                               { i = g.pt + a;                       j = g.pl + n + (k - g.pt - g.pd - g.pl); }
            st[idx] = src_elem(i, j);
        }
        __builtin_amdgcn_wave_barrier();
        auto elem = [&](int i, int j) -> T {
            const bool ri = i >= g.pt && i < g.pt + n, cj = j >= g.pl && j < g.pl + n;
            if (ri && cj) return xf[(size_t)(i - g.pt) * n + (j - g.pl)];
            if (cj) return i < g.pt ? st_t[i * n + (j - g.pl)] : st_b[(i - g.pt - n) * n + (j - g.pl)];
            if (ri) return j < g.pl ? st_l[j * n + (i - g.pt)] : st_r[(j - g.pl - n) * n + (i - g.pt)];
            return src_elem(i, j);                           // corner stitch (at most 4 p^2 elements per plane)
        };
        unsigned char* yb = y + ((size_t)grp * 6 * out_face + (size_t)f * out_face + (size_t)c * Hp * Wp) * ES;
        // 16-byte-aligned chunks of the tensor's address range that overlap this plane.  Chunk ch holds plane
        // elements [ch * E - head, ch * E - head + E); the chunks fully inside a centre row's centre columns
        // ("runs": rows' chunk ranges [cs(r), ce(r))) are plain copies, everything else is assembled per element.
        const size_t a0 = reinterpret_cast<size_t>(yb), a1 = a0 + (size_t)plane_elems * ES;
        const size_t c0 = a0 & ~(size_t)15;
        const int head = (int)(a0 - c0) / ES;                // elements of chunk 0 that belong to the previous plane
        const int nchunks = (int)((a1 - c0 + 15) >> 4);
        auto run_begin = [&](int r) -> int {                 // first chunk fully inside centre row r (r = 0 .. n-1)
            return ((g.pt + r) * Wp + g.pl + head + E - 1) / E;   // (operands < 2^23: exact integer division is cheap enough per row)
        };
        auto run_end = [&](int r) -> int { return ((g.pt + r) * Wp + g.pl + n + head) / E; };
        auto slow_chunk = [&](int ch) __attribute__((always_inline)) {
            const int q0 = ch * E - head;
            unsigned char* ca = reinterpret_cast<unsigned char*>(c0 + ((size_t)ch << 4));
            if (q0 >= 0 && q0 + E <= plane_elems) {
                const int i = div_by(q0, Wp, rcp_wp), j = q0 - i * Wp;
                T tmp[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    int ie = i, je = j + e;
                    if (je >= Wp) { je -= Wp; ++ie; }        // E <= Wp (launcher)
                    tmp[e] = elem(ie, je);
                }
                cp_u32x4 v;
                __builtin_memcpy(&v, tmp, 16);
                *reinterpret_cast<cp_u32x4*>(ca) = v;
            } else {                                         // head / tail of the plane: element stores
                for (int e = 0; e < E; ++e) {
                    const int q = q0 + e;
                    if (q < 0 || q >= plane_elems) continue;
                    const int ie = div_by(q, Wp, rcp_wp), je = q - ie * Wp;
                    reinterpret_cast<T*>(yb)[q] = elem(ie, je);
                }
            }
        };
        // ---- pass 1: the runs.  cpr = chunks per row upper bound; lanes enumerate (row, chunk in run), four at a time
        const int cpr = n * ES / 16 + 1;
        const float rcp_cpr = 1.0f / (float)cpr;
        const int nfast = n * cpr;
        for (int base = lane; base < nfast; base += 64 * 4) {
            cp_u32x4 v[4];
            unsigned char* dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + 64 * u;
                dst[u] = nullptr;
                if (idx < nfast) {
                    const int r = div_by(idx, cpr, rcp_cpr), k = idx - r * cpr;
                    const int ch = run_begin(r) + k;
                    if (ch < run_end(r)) {
                        const int j = ch * E - head - (g.pt + r) * Wp;       // first column of the chunk (>= pl)
                        __builtin_memcpy(&v[u], xf + (size_t)r * n + (j - g.pl), 16);
                        dst[u] = reinterpret_cast<unsigned char*>(c0 + ((size_t)ch << 4));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u]) *reinterpret_cast<cp_u32x4*>(dst[u]) = v[u];
        }
        // ---- pass 2: what the runs leave: the pad rows above / below and the 1-2 chunks between consecutive runs
        {
            const int top_end = min(run_begin(0), nchunks);
            for (int ch = lane; ch < top_end; ch += 64) slow_chunk(ch);
            const int bot_begin = max(run_end(n - 1), top_end);
            for (int ch = bot_begin + lane; ch < nchunks; ch += 64) slow_chunk(ch);
            for (int r = 1 + lane; r < n; r += 64) {
                const int lo = max(run_end(r - 1), top_end), hi = min(run_begin(r), bot_begin);
                for (int ch = lo; ch < hi; ++ch) slow_chunk(ch);
            }
        }
        __builtin_amdgcn_wave_barrier();                     // the next item's staging overwrites the strips
    }
}


// ---------------------------------------------------------------- NCHW, large faces: run copies + pad stream (round 3)
// The strip kernel above is VALU-bound, not HBM-bound (tools/_exp timing variants, profiles/r03_cubepad_plane.md: the run
// pass alone takes 393 us on [384,64,112,112] f16 where a plain copy of the same bytes takes 250 us - ~40 instructions
// per 16-byte chunk, because the per-item values live in vector registers and every chunk pays two float-reciprocal
// divisions; the pad pass costs as much again in branchy per-element code).  Same decomposition, cheaper arithmetic:
//   * the wave index is made uniform (readfirstlane), so item / plane / face / base pointers are scalar;
//   * pass 1 (runs = chunks fully inside a centre row) steps (row, chunk-in-row) incrementally - no division;
//   * every pad element of the padded plane, in OUTPUT order, is staged in LDS once per item ("pad stream" PS: the padded
//     plane with its centre elements removed: pt full rows, then [pl left | pr right] per centre row, then pd full rows;
//     so the PS index of a pad element q is q - #centre elements before q).  Strip lines are affine in the neighbour
//     face (cubepad_src: row / col are each one of k, n-p+k, a, n-1-a): a 16-entry (base, step) table per item replaces
//     cubepad_src() per element; only the <= (pt+pd)(pl+pr) corner elements evaluate it;
//   * every chunk that is not a run - pad rows, the 1-3 chunks between consecutive runs, the head / tail chunks a plane
//     shares with its neighbours - is one branch-free routine: E consecutive output elements are [centre stream A]
//     [pads][centre stream B] (n >= 2E + pl + pr: at most one centre -> pad -> centre transition), so two possibly
//     misaligned 16-byte loads A, B positioned so that element e of the chunk is A[e] / B[e], E LDS reads PS[base + e]
//     at immediate offsets, and two compares + selects per element.
// Bit-exact like every CubePad kernel (pure copy).
template <int ES>
__global__ __launch_bounds__(256) void cubepad_nchw_plane_kernel(const unsigned char* __restrict__ x,
                                                                 unsigned char* __restrict__ y, int C, CubePadGeom g,
                                                                 int n_items, long long total_in, float rcp_wp, float rcp_n,
                                                                 float rcp_cpr, int dq, int dm, int maxc, float rcp_maxc,
                                                                 int ps_alloc, int rbw) {
    typedef typename ElemOf<ES>::T T;
    constexpr int E = 16 / ES;                               // elements per 16-byte chunk
    constexpr int LOG_E = E == 16 ? 4 : (E == 8 ? 3 : (E == 4 ? 2 : 1));
    constexpr int UNR = 4;                                   // run chunks in flight per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char strip_raw[];
    const int n = g.n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr, LR = g.pl + g.pr;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nlines = g.pt + g.pd + LR, nstrip = nlines * n;
    const int nn = n * n, plane_elems = Hp * Wp;
    const int ps_mid = g.pt * Wp;                            // PS index of centre row 0's first pad
    const int ps_bot = ps_mid + n * LR;                      // PS index of the first bottom-row element
    int* ltab = reinterpret_cast<int*>(strip_raw) + wave * 32;                  // [16 lines][base, step]
    T* ps = reinterpret_cast<T*>(strip_raw + 512) + (size_t)wave * ps_alloc + E;   // E elements of slack either side
    const int in_face = C * nn;                              // (6 * in_face < 2^31: launcher)
    const size_t out_face = (size_t)C * plane_elems;
    const int cpr = (n >> LOG_E) + 1;                        // chunks per run, upper bound
    auto div_by = [](int q, int d, float rcp) -> int {       // exact for 0 <= q < 2^24
        int r = (int)((float)q * rcp);
        r -= (r * d > q);
        r += ((r + 1) * d <= q);
        return r;
    };
    for (int item = blockIdx.x * 4 + wave; item < n_items; item += gridDim.x * 4) {
        const int plane = item / 6, f = item - plane * 6;
        const int grp = plane / C, c = plane - grp * C;
        const long long cube_off = (long long)grp * 6 * in_face + (long long)c * nn;   // elements from x
        const long long xf_off = cube_off + (long long)f * in_face;
        const T* xin = reinterpret_cast<const T*>(x) + cube_off;
        const T* xf = reinterpret_cast<const T*>(x) + xf_off;
        // ---- the strip lines of this face as (element offset from xin, step) pairs
        if (lane < nlines) {
            const int k = lane;
            int i0, j0, di = 0, dj = 0;
            if (k < g.pt)                      { i0 = k;                        j0 = g.pl; dj = 1; }
            else if (k < g.pt + g.pd)          { i0 = n + k;                    j0 = g.pl; dj = 1; }      // pt + n + (k - pt)
            else if (k < g.pt + g.pd + g.pl)   { i0 = g.pt;                     j0 = k - g.pt - g.pd; di = 1; }
            else                               { i0 = g.pt;                     j0 = n + k - g.pt - g.pd; di = 1; }   // pl + n + (k - pt - pd - pl)
            const int s0 = cubepad_src(f, i0, j0, g), s1 = cubepad_src(f, i0 + di, j0 + dj, g);
            const int sf = s0 / nn;
            ltab[2 * k] = sf * in_face + (s0 - sf * nn);
            ltab[2 * k + 1] = s1 - s0;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- pad stream: strips ...
        for (int idx = lane; idx < nstrip; idx += 64) {
            const int k = div_by(idx, n, rcp_n), a = idx - k * n;
            const int base = ltab[2 * k], step = ltab[2 * k + 1];
            int pos;
            if (k < g.pt)                      pos = k * Wp + g.pl + a;
            else if (k < g.pt + g.pd)          pos = ps_bot + (k - g.pt) * Wp + g.pl + a;
            else if (k < g.pt + g.pd + g.pl)   pos = ps_mid + a * LR + (k - g.pt - g.pd);
            else                               pos = ps_mid + a * LR + (k - g.pt - g.pd);       // pl + (k - pt - pd - pl)
            ps[pos] = xin[base + a * step];
        }
        // ---- ... and corners (make_cubepad_edge: cube_pad.py:44-93)
        const int ncorner = (g.pt + g.pd) * LR;
        for (int idx = lane; idx < ncorner; idx += 64) {
            const int ci = idx / LR, cj = idx - ci * LR;
            const int i = ci < g.pt ? ci : n + ci, j = cj < g.pl ? cj : n + cj;
            const int s = cubepad_src(f, i, j, g);
            const int sf = s / nn;
            const int pos = ci < g.pt ? i * Wp + j : ps_bot + (ci - g.pt) * Wp + j;
            ps[pos] = xin[sf * in_face + (s - sf * nn)];
        }
        __builtin_amdgcn_wave_barrier();
        unsigned char* yb = y + ((size_t)grp * 6 * out_face + (size_t)f * out_face + (size_t)c * plane_elems) * ES;
        const size_t a0 = reinterpret_cast<size_t>(yb), a1 = a0 + (size_t)plane_elems * ES;
        const size_t c0 = a0 & ~(size_t)15;
        unsigned char* c0p = reinterpret_cast<unsigned char*>(c0);
        const int head = (int)(a0 - c0) / ES;                // elements of chunk 0 that belong to the previous plane
        const int nchunks = (int)((a1 - c0 + 15) >> 4);
        const int rowA = g.pt * Wp + g.pl + head;
        auto run_begin = [&](int r) -> int { return (r * Wp + rowA + E - 1) >> LOG_E; };   // first chunk fully inside centre row r
        auto run_end = [&](int r) -> int { return (r * Wp + rowA + n) >> LOG_E; };
        auto load_vec = [&](long long goff) -> cp_u32x4 {     // 16 bytes at element goff of x; the tensor's first / last chunk may hang over its ends
            cp_u32x4 v;
            if (goff >= 0 && goff + E <= total_in) {
                __builtin_memcpy(&v, reinterpret_cast<const T*>(x) + goff, 16);
            } else {
                T tmp[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const long long t = goff + e;
                    tmp[e] = (t >= 0 && t < total_in) ? reinterpret_cast<const T*>(x)[t] : (T)0;
                }
                __builtin_memcpy(&v, tmp, 16);
            }
            return v;
        };
        auto slow_chunk = [&](int ch) __attribute__((always_inline)) {
            const int q0 = ch * E - head;
            int iA, jA;
            if (q0 < 0) { iA = -1; jA = q0 + Wp; }
            else        { iA = div_by(q0, Wp, rcp_wp); jA = q0 - iA * Wp; }
            const int a = iA - g.pt;                                           // centre row of row iA when 0 <= a < n
            const bool rowc = a >= 0 && a < n;
            const bool q0c = rowc && jA >= g.pl && jA < g.pl + n;              // q0 is a centre element
            const int cpos = a < 0 ? 0 : (a >= n ? nn : a * n + min(max(jA - g.pl, 0), n));   // centre elements before q0
            const int eA = q0c ? min(E, g.pl + n - jA) : 0;                    // leading elements from centre stream A
            const int aB = q0c ? a + 1 : (a < 0 ? 0 : (rowc ? (jA < g.pl ? a : a + 1) : n));   // centre row that starts next
            const int eB = aB < n ? min(max((g.pt + aB) * Wp + g.pl - q0, 0), E) : E;   // elements from e = eB on are its centre
            const cp_u32x4 va = load_vec(xf_off + (eA > 0 ? cpos : 0));
            const cp_u32x4 vb = load_vec(xf_off + (eB < E ? aB * n - eB : 0));
            T ta[E], tb[E], tv[E];
            __builtin_memcpy(ta, &va, 16);
            __builtin_memcpy(tb, &vb, 16);
            const T* pp = ps + (q0 - cpos - eA);                               // PS index of pad element e = (q0 - cpos) + (e - eA)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const T pv = pp[e];
                tv[e] = e < eA ? ta[e] : (e >= eB ? tb[e] : pv);
            }
            unsigned char* ca = c0p + ((size_t)ch << 4);
            if (q0 >= 0 && q0 + E <= plane_elems) {
                cp_u32x4 v;
                __builtin_memcpy(&v, tv, 16);
                *reinterpret_cast<cp_u32x4*>(ca) = v;
            } else {                                                           // chunk shared with the neighbouring plane
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (q0 + e >= 0 && q0 + e < plane_elems) reinterpret_cast<T*>(ca)[e] = tv[e];
            }
        };
        const int top_end = min(run_begin(0), nchunks);
        const int bot_begin = max(run_end(n - 1), top_end);
        // Row blocks: the runs of rows [r0, rend) (one UNR-deep body per wave), then at once the chunks around them - the
        // 128-byte lines a run leaves incomplete are completed while they are still dirty in L2 (a partial line written
        // back to HBM and completed later costs far more than its bytes: tools/_exp variants, profiles/r03_cubepad_plane.md).
        for (int r0 = 0; r0 < n; r0 += rbw) {
            const int rend = min(r0 + rbw, n);
            {
                const int rq = div_by(lane, cpr, rcp_cpr);
                int r = r0 + rq, k = lane - rq * cpr;
                while (r < rend) {
                    cp_u32x4 v[UNR];
                    unsigned doff[UNR];
                    bool ok[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int rw = r * Wp + rowA;
                        const int ch = ((rw + E - 1) >> LOG_E) + k;
                        ok[u] = r < rend && ch < ((rw + n) >> LOG_E);
                        if (ok[u]) {
                            const unsigned soff = (unsigned)(ch * E + r * (n - Wp) - rowA);   // centre element index of the chunk's first element
                            __builtin_memcpy(&v[u], xf + soff, 16);
                            doff[u] = (unsigned)ch << 4;
                        }
                        k += dm;
                        r += dq;
                        if (k >= cpr) { k -= cpr; ++r; }
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
                        if (ok[u]) *reinterpret_cast<cp_u32x4*>(c0p + doff[u]) = v[u];
                }
            }
            const int n_top = r0 == 0 ? top_end : 0, n_bot = rend == n ? nchunks - bot_begin : 0;
            const int b0 = max(r0, 1);                                          // boundaries (rr, rr + 1) with rr + 1 in [b0, rend)
            const int n_slow = n_top + n_bot + max(rend - b0, 0) * maxc;
            for (int s = lane; s < n_slow; s += 64) {
                int ch;
                if (s < n_top) ch = s;
                else if (s < n_top + n_bot) ch = bot_begin + (s - n_top);
                else {
                    const int m = s - n_top - n_bot;
                    const int mq = div_by(m, maxc, rcp_maxc);
                    const int rr = b0 - 1 + mq;
                    const int lo = max(run_end(rr), top_end), hi = min(run_begin(rr + 1), bot_begin);
                    ch = lo + (m - mq * maxc);
                    if (ch >= hi) continue;
                }
                slow_chunk(ch);
            }
        }
        __builtin_amdgcn_wave_barrier();                     // the next item's staging overwrites the pad stream
    }
}


// ---------------------------------------------------------------- NCHW, large faces, every input byte read once (round 3)
// The plane kernel's HBM reads are 2-3.2x the input (rocprofv3 FETCH_SIZE, profiles/r03_cubepad_plane.md): about half of
// a cube's 24 strips are COLUMNS of the neighbouring face (cube_pad.py:114-216), so a (plane, face) item pulls a whole
// 128-byte line per 1-4 byte pad element, and the chunks between runs re-read the first / last line of every centre row.
// Here a workgroup item is (cube, channel) - the six planes CubePad exchanges borders between - and nothing is read twice:
//   S1  for each face: stream the runs HBM -> registers -> HBM (as the plane kernel), and beside them capture into LDS
//       RE[f][row][side][E]: the first and the last E elements of every centre row (one 16-byte load each, issued in the
//       same 64-row block as the run loads of those rows, so the line is in L2), and TB[f][top|bottom][P][n]: the first /
//       last P rows.  RE + TB hold every element a pad of ANY face of this (cube, channel) can copy (a strip lies within
//       P <= E of a border; corners replicate strip ends) and every centre element of a non-run chunk (< E from a row end);
//   S2  for each face: the pad stream PS (see the plane kernel) is gathered LDS -> LDS through a line table that maps the
//       face's strip lines to affine (base, step) element offsets in RE / TB (built once per workgroup: it depends on the
//       geometry only), then every non-run chunk is assembled from LDS alone: [row end from RE][pads from PS][row start
//       from RE], E element reads at immediate offsets each, and written with one aligned 16-byte store.
// HBM traffic = input + output (+ the two shared 16-byte chunks per plane).  A workgroup is 6 (or 12) waves, one (two) per
// face: the six faces stream concurrently and independently, three workgroup barriers per item (S1 | pad streams | chunks);
// workgroups are persistent over items.
template <int ES>
__global__ __launch_bounds__(768) void cubepad_nchw_channel_kernel(const unsigned char* __restrict__ x,
                                                                   unsigned char* __restrict__ y, int C, CubePadGeom g,
                                                                   int n_items, float rcp_wp, float rcp_n, float rcp_cpr,
                                                                   int maxc, float rcp_maxc, int P, int ps_alloc, int rbw) {
    typedef typename ElemOf<ES>::T T;
    constexpr int E = 16 / ES;                               // elements per 16-byte chunk; also the width of a row-end record
    constexpr int LOG_E = E == 16 ? 4 : (E == 8 ? 3 : (E == 4 ? 2 : 1));
    constexpr int UNR = 8;                                   // loads in flight per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = g.n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr, LR = g.pl + g.pr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int parts = (int)(blockDim.x >> 6) / 6;            // waves per face: 1 or 2
    const int part = wave / 6, f = wave - part * 6;          // this wave's face and its share of the rows / strips / chunks
    const int nlines = g.pt + g.pd + LR, nstrip = nlines * n;
    const int nn = n * n, plane_elems = Hp * Wp;
    const int ps_mid = g.pt * Wp, ps_bot = ps_mid + n * LR;
    const int cpr = (n >> LOG_E) + 1;
    const int re_size = 12 * n * E, tb_face = 2 * P * n;
    int* ltab = reinterpret_cast<int*>(smem);                                   // [6][16][base, step]
    T* buf = reinterpret_cast<T*>(smem + 768) + E;                              // RE then TB, E elements of slack in front
    T* ps = buf + re_size + 6 * tb_face + E + f * ps_alloc;                     // this face's pad stream (slack either side)
    const size_t in_face = (size_t)C * nn, out_face = (size_t)C * plane_elems;
    const int dq = 64 / cpr, dm = 64 - dq * cpr;
    auto div_by = [](int q, int d, float rcp) -> int {       // exact for 0 <= q < 2^24
        int r = (int)((float)q * rcp);
        r -= (r * d > q);
        r += ((r + 1) * d <= q);
        return r;
    };
    auto lds_of = [&](int sf, int row, int col) -> int {     // RE / TB offset of element (row, col) of face sf (within E / P of a border)
        if (row < P) return re_size + sf * tb_face + row * n + col;
        if (row >= n - P) return re_size + sf * tb_face + (P + row - (n - P)) * n + col;
        if (col < E) return ((sf * n + row) * 2) * E + col;
        return ((sf * n + row) * 2 + 1) * E + col - (n - E);
    };
    // ---- line table (geometry only)
    if (tid < 96) {
        const int lf = tid >> 4, k = tid & 15;
        if (k < nlines) {
            int i0, j0, di = 0, dj = 0;
            if (k < g.pt)                      { i0 = k;     j0 = g.pl; dj = 1; }
            else if (k < g.pt + g.pd)          { i0 = n + k; j0 = g.pl; dj = 1; }
            else if (k < g.pt + g.pd + g.pl)   { i0 = g.pt;  j0 = k - g.pt - g.pd; di = 1; }
            else                               { i0 = g.pt;  j0 = n + k - g.pt - g.pd; di = 1; }
            const int s0 = cubepad_src(lf, i0, j0, g), d = cubepad_src(lf, i0 + di, j0 + dj, g) - s0;
            const int sf = s0 / nn, rem = s0 - sf * nn, row0 = rem / n, col0 = rem - row0 * n;
            int base, step;
            if (d == 1 || d == -1) {                         // a row of face sf: in TB
                base = re_size + sf * tb_face + (row0 < P ? row0 : P + row0 - (n - P)) * n + col0;
                step = d;
            } else {                                         // a column of face sf: in RE
                base = ((sf * n + row0) * 2 + (col0 < E ? 0 : 1)) * E + (col0 < E ? col0 : col0 - (n - E));
                step = d > 0 ? 2 * E : -2 * E;
            }
            ltab[(lf * 16 + k) * 2] = base;
            ltab[(lf * 16 + k) * 2 + 1] = step;
        }
    }
    const int ra = part * n / parts, rb = (part + 1) * n / parts;               // this wave's centre rows
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int grp = item / C, c = item - grp * C;
        const T* xf = reinterpret_cast<const T*>(x) + (size_t)grp * 6 * in_face + (size_t)c * nn + (size_t)f * in_face;
        unsigned char* yb = y + ((size_t)grp * 6 * out_face + (size_t)c * plane_elems + (size_t)f * out_face) * ES;
        const int head = (int)(reinterpret_cast<size_t>(yb) & 15) / ES;         // elements of chunk 0 that belong to the previous plane
        unsigned char* c0p = yb - (size_t)head * ES;
        const int rowA = g.pt * Wp + g.pl + head;
        // ================= S1: runs + capture (wave = face f, rows [ra, rb))
        {   // first / last P rows
            const int t0 = part * tb_face / parts, t1 = (part + 1) * tb_face / parts;
            for (int base = t0 + lane; base < t1; base += 64 * UNR) {
                T v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int idx = base + 64 * u;
                    if (idx < t1) v[u] = xf[idx < P * n ? idx : (n - 2 * P) * n + idx];
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (base + 64 * u < t1) buf[re_size + f * tb_face + base + 64 * u] = v[u];
            }
        }
        for (int r0 = ra; r0 < rb; r0 += rbw) {
            const int rend = min(r0 + rbw, rb);
            for (int t = lane; t < 2 * (rend - r0); t += 64) {                  // row ends of this block
                const int row = r0 + (t >> 1), side = t & 1;
                cp_u32x4 v;
                __builtin_memcpy(&v, xf + (unsigned)(row * n + (side ? n - E : 0)), 16);
                *reinterpret_cast<cp_u32x4*>(buf + ((f * n + row) * 2 + side) * E) = v;
            }
            const int rq = div_by(lane, cpr, rcp_cpr);
            int r = r0 + rq, k = lane - rq * cpr;
            while (r < rend) {
                cp_u32x4 v[UNR];
                unsigned doff[UNR];
                bool ok[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int rw = r * Wp + rowA;
                    const int ch = ((rw + E - 1) >> LOG_E) + k;
                    ok[u] = r < rend && ch < ((rw + n) >> LOG_E);
                    if (ok[u]) {
                        const unsigned soff = (unsigned)(ch * E + r * (n - Wp) - rowA);
                        __builtin_memcpy(&v[u], xf + soff, 16);
                        doff[u] = (unsigned)ch << 4;
                    }
                    k += dm;
                    r += dq;
                    if (k >= cpr) { k -= cpr; ++r; }
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (ok[u]) *reinterpret_cast<cp_u32x4*>(c0p + doff[u]) = v[u];
            }
        }
        __syncthreads();
        // ================= S2: pads and the chunks around them, from LDS
        {
            const int t0 = part * nstrip / parts, t1 = (part + 1) * nstrip / parts;
            for (int idx = t0 + lane; idx < t1; idx += 64) {
                const int k = div_by(idx, n, rcp_n), a = idx - k * n;
                const int base = ltab[(f * 16 + k) * 2], step = ltab[(f * 16 + k) * 2 + 1];
                int pos;
                if (k < g.pt)                      pos = k * Wp + g.pl + a;
                else if (k < g.pt + g.pd)          pos = ps_bot + (k - g.pt) * Wp + g.pl + a;
                else                               pos = ps_mid + a * LR + (k - g.pt - g.pd);
                ps[pos] = buf[base + a * step];
            }
            const int ncorner = part == 0 ? (g.pt + g.pd) * LR : 0;
            for (int idx = lane; idx < ncorner; idx += 64) {
                const int ci = idx / LR, cj = idx - ci * LR;
                const int i = ci < g.pt ? ci : n + ci, j = cj < g.pl ? cj : n + cj;
                const int s = cubepad_src(f, i, j, g);
                const int sf = s / nn, rem = s - sf * nn, row = rem / n;
                ps[ci < g.pt ? i * Wp + j : ps_bot + (ci - g.pt) * Wp + j] = buf[lds_of(sf, row, rem - row * n)];
            }
        }
        __syncthreads();
        {
            const int nchunks = (head + plane_elems + E - 1) >> LOG_E;
            const int top_end = min((rowA + E - 1) >> LOG_E, nchunks);                       // run_begin(0)
            const int bot_begin = max(((n - 1) * Wp + rowA + n) >> LOG_E, top_end);          // run_end(n - 1)
            const int n_tb = top_end + (nchunks - bot_begin);
            const int n_slow = n_tb + (n - 1) * maxc;
            const int s0 = part * n_slow / parts, s1 = (part + 1) * n_slow / parts;
            for (int s = s0 + lane; s < s1; s += 64) {
                int ch;
                if (s < top_end) ch = s;
                else if (s < n_tb) ch = bot_begin + (s - top_end);
                else {
                    const int m = s - n_tb;
                    const int rr = div_by(m, maxc, rcp_maxc);
                    const int lo = max((rr * Wp + rowA + n) >> LOG_E, top_end);              // run_end(rr)
                    const int hi = min(((rr + 1) * Wp + rowA + E - 1) >> LOG_E, bot_begin);  // run_begin(rr + 1)
                    ch = lo + (m - rr * maxc);
                    if (ch >= hi) continue;
                }
                const int q0 = ch * E - head;
                int iA, jA;
                if (q0 < 0) { iA = -1; jA = q0 + Wp; }
                else        { iA = div_by(q0, Wp, rcp_wp); jA = q0 - iA * Wp; }
                const int a = iA - g.pt;
                const bool rowc = a >= 0 && a < n;
                const bool q0c = rowc && jA >= g.pl && jA < g.pl + n;
                const int cpos = a < 0 ? 0 : (a >= n ? nn : a * n + min(max(jA - g.pl, 0), n));
                const int eA = q0c ? min(E, g.pl + n - jA) : 0;
                const int aB = q0c ? a + 1 : (a < 0 ? 0 : (rowc ? (jA < g.pl ? a : a + 1) : n));
                const int eB = aB < n ? min(max((g.pt + aB) * Wp + g.pl - q0, 0), E) : E;
                const T* pa = buf + (eA > 0 ? ((f * n + a) * 2 + 1) * E + (E - eA) : 0);     // row a's last eA elements at e = 0 ..
                const T* pb = buf + (eB < E ? ((f * n + aB) * 2) * E - eB : 0);               // row aB's first elements at e = eB ..
                const T* pp = ps + (q0 - cpos - eA);
                T tv[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const T va = pa[e], vb = pb[e], pv = pp[e];
                    tv[e] = e < eA ? va : (e >= eB ? vb : pv);
                }
                unsigned char* ca = c0p + ((size_t)ch << 4);
                if (q0 >= 0 && q0 + E <= plane_elems) {
                    cp_u32x4 v;
                    __builtin_memcpy(&v, tv, 16);
                    *reinterpret_cast<cp_u32x4*>(ca) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (q0 + e >= 0 && q0 + e < plane_elems) reinterpret_cast<T*>(ca)[e] = tv[e];
                }
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------- NCHW, six padded planes assembled in LDS (round 3)
// What the timing variants of the two kernels above show (profiles/r03_cubepad_plane.md): HBM rewards ONE linear sweep of
// whole 128-byte lines and punishes everything else - run chunks that leave holes, holes filled later, even whole lines
// written sparsely all land at 3-3.5 TB/s where the same structure with a linear store stream reaches 4.8-5 (= a plain
// copy).  So when the six padded planes of a (cube, channel) fit the 160 KB of LDS (n <= 114 for 2-byte elements, every
// face of the network from layer1 down), the item is assembled there and written once, linearly.  An item is (cube,
// channels c0 .. c0 + CH - 1): per face its input is CH * n^2 contiguous elements, its output CH * Hp * Wp contiguous
// elements (CH = 1 for the large faces; small faces take several channels so that an item is worth its three barriers):
//   1  load: every 16-byte chunk of the six input ranges goes to its PADDED position in LDS - face f's CH padded planes
//      start at f * PSTRIDE + (their global address & 15), so LDS and global chunks coincide; a chunk that runs over a row
//      end (n % E != 0) continues pl + pr elements later, over a plane end (pt + pd) * Wp more;
//   2  pads: LDS -> LDS element copies through the affine line table (a pad only ever reads a centre element of another
//      face - cube_pad.py:114-216 - so the copies need no ordering among themselves); corners through cubepad_src();
//   3  store: aligned 16-byte LDS reads, aligned 16-byte global stores, face after face; the first / last chunk of a
//      range that is shared with its neighbours goes element by element.
// HBM traffic = input + output, every line once.  Bit-exact (pure copy).
template <int ES>
__global__ __launch_bounds__(1024) void cubepad_nchw_lds6_kernel(const unsigned char* __restrict__ x,
                                                                 unsigned char* __restrict__ y, int C, CubePadGeom g,
                                                                 int CH, int ranges, int n_items, int pstride_b, float rcp_n,
                                                                 float rcp_nn, float rcp_cpf, float rcp_nstrip,
                                                                 float rcp_chstrip, float rcp_nchmax) {
    typedef typename ElemOf<ES>::T T;
    constexpr int E = 16 / ES;
    constexpr int LOG_E = E == 16 ? 4 : (E == 8 ? 3 : (E == 4 ? 2 : 1));
    constexpr int UNR = ES == 1 ? 2 : 4;          // (byte elements: 16 element moves per chunk - two chunks in flight fit the 128 VGPRs of a 1024-thread workgroup without spilling)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    const int n = g.n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr, LR = g.pl + g.pr;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int nlines = g.pt + g.pd + LR, nstrip = nlines * n, ncorner = (g.pt + g.pd) * LR;
    const int nn = n * n, HW = Hp * Wp;
    const int in_reg = CH * nn, out_reg = CH * HW;           // elements of a face's input / output range
    const int cpf = (in_reg + E - 1) >> LOG_E;               // 16-byte chunks per input range (the last one may be shifted back)
    const int nchmax = (out_reg + 2 * E - 2) >> LOG_E;       // chunks an output range can overlap
    int* ltab = reinterpret_cast<int*>(smem6);               // [6][16][sf << 24 | offset in the padded plane, step]
    unsigned char* planes = smem6 + 768;
    const int pstride_e = pstride_b / ES;
    const size_t in_face = (size_t)C * nn, out_face_b = (size_t)C * HW * ES;
    auto div_by = [](int q, int d, float rcp) -> int {       // exact for 0 <= q < 2^24
        int r = (int)((float)q * rcp);
        r -= (r * d > q);
        r += ((r + 1) * d <= q);
        return r;
    };
    if (tid < 96) {
        const int f = tid >> 4, k = tid & 15;
        if (k < nlines) {
            int i0, j0, di = 0, dj = 0;
            if (k < g.pt)                      { i0 = k;     j0 = g.pl; dj = 1; }
            else if (k < g.pt + g.pd)          { i0 = n + k; j0 = g.pl; dj = 1; }
            else if (k < g.pt + g.pd + g.pl)   { i0 = g.pt;  j0 = k - g.pt - g.pd; di = 1; }
            else                               { i0 = g.pt;  j0 = n + k - g.pt - g.pd; di = 1; }
            const int s0 = cubepad_src(f, i0, j0, g), d = n > 1 ? cubepad_src(f, i0 + di, j0 + dj, g) - s0 : 0;
            const int sf = s0 / nn, rem = s0 - sf * nn, row0 = rem / n, col0 = rem - row0 * n;
            ltab[(f * 16 + k) * 2] = (sf << 24) | ((g.pt + row0) * Wp + g.pl + col0);
            ltab[(f * 16 + k) * 2 + 1] = (d == 1 || d == -1) ? d : (d > 0 ? Wp : -Wp);
        }
    }
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int grp = item / ranges, c0 = (item - grp * ranges) * CH;
        const T* xc = reinterpret_cast<const T*>(x) + (size_t)grp * 6 * in_face + (size_t)c0 * nn;
        unsigned char* yc = y + (size_t)grp * 6 * out_face_b + (size_t)c0 * HW * ES;
        const size_t yc_addr = reinterpret_cast<size_t>(yc);
        auto head_b = [&](int f) -> int { return (int)((yc_addr + (size_t)f * out_face_b) & 15); };   // bytes of chunk 0 before face f's range
        // ---- 1: input chunks to their padded places
        for (int base = tid; base < 6 * cpf; base += NT * UNR) {
            cp_u32x4 v[UNR];
            int lo[UNR], wrap_at[UNR], wrap_add[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + NT * u;
                lo[u] = -1;
                if (idx < 6 * cpf) {
                    const int f = div_by(idx, cpf, rcp_cpf), ck = idx - f * cpf;
                    const int t0 = min(ck * E, in_reg - E);               // first element (last chunk: shifted back, rewrites equal values)
                    const int cc = div_by(t0, nn, rcp_nn), rem = t0 - cc * nn;
                    const int r = div_by(rem, n, rcp_n), col = rem - r * n;
                    __builtin_memcpy(&v[u], xc + (size_t)f * in_face + (unsigned)t0, 16);
                    lo[u] = f * pstride_b + head_b(f) + (cc * HW + (g.pt + r) * Wp + g.pl + col) * ES;
                    wrap_at[u] = n - col;                                  // elements from here on sit in the next row ..
                    wrap_add[u] = LR + (r + 1 == n ? (g.pt + g.pd) * Wp : 0);   // .. or the next channel's plane
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (lo[u] < 0) continue;
                T* d = reinterpret_cast<T*>(planes + lo[u]);  // element-aligned only: E element stores
                const cp_u32x4 w = v[u];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    T val;
                    if constexpr (ES == 8)      val = (T)(((unsigned long long)w[2 * e + 1] << 32) | w[2 * e]);
                    else if constexpr (ES == 4) val = (T)w[e];
                    else if constexpr (ES == 2) val = (T)(w[e >> 1] >> (16 * (e & 1)));
                    else                        val = (T)(w[e >> 2] >> (8 * (e & 3)));
                    d[e + (e >= wrap_at[u] ? wrap_add[u] : 0)] = val;
                }
            }
        }
        __syncthreads();
        // ---- 2: pads
        T* pe = reinterpret_cast<T*>(planes);
        for (int idx = tid; idx < 6 * CH * nstrip; idx += NT) {
            const int f = div_by(idx, CH * nstrip, rcp_chstrip), r1 = idx - f * CH * nstrip;
            const int cc = div_by(r1, nstrip, rcp_nstrip), rem = r1 - cc * nstrip;
            const int k = div_by(rem, n, rcp_n), a = rem - k * n;
            const int t0 = ltab[(f * 16 + k) * 2], step = ltab[(f * 16 + k) * 2 + 1];
            const int sf = t0 >> 24;
            int pos;
            if (k < g.pt)                      pos = k * Wp + g.pl + a;
            else if (k < g.pt + g.pd)          pos = (n + k) * Wp + g.pl + a;
            else if (k < g.pt + g.pd + g.pl)   pos = (g.pt + a) * Wp + (k - g.pt - g.pd);
            else                               pos = (g.pt + a) * Wp + n + (k - g.pt - g.pd);
            pe[f * pstride_e + head_b(f) / ES + cc * HW + pos] =
                pe[sf * pstride_e + head_b(sf) / ES + cc * HW + (t0 & 0xffffff) + a * step];
        }
        for (int idx = tid; idx < 6 * CH * ncorner; idx += NT) {
            const int f = idx / (CH * ncorner), r1 = idx - f * CH * ncorner;
            const int cc = r1 / ncorner, rem = r1 - cc * ncorner;
            const int ci = rem / LR, cj = rem - ci * LR;
            const int i = ci < g.pt ? ci : n + ci, j = cj < g.pl ? cj : n + cj;
            const int s = cubepad_src(f, i, j, g);
            const int sf = s / nn, r2 = s - sf * nn, row = r2 / n, col = r2 - row * n;
            pe[f * pstride_e + head_b(f) / ES + cc * HW + i * Wp + j] =
                pe[sf * pstride_e + head_b(sf) / ES + cc * HW + (g.pt + row) * Wp + g.pl + col];
        }
        __syncthreads();
        // ---- 3: linear store
        for (int base = tid; base < 6 * nchmax; base += NT * UNR) {
            cp_u32x4 v[UNR];
            int fs[UNR], chs[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + NT * u;
                fs[u] = -1;
                if (idx < 6 * nchmax) {
                    const int f = div_by(idx, nchmax, rcp_nchmax), ch = idx - f * nchmax;
                    if (ch * 16 < head_b(f) + out_reg * ES) {
                        fs[u] = f;
                        chs[u] = ch;
                        v[u] = *reinterpret_cast<const cp_u32x4*>(planes + f * pstride_b + ch * 16);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (fs[u] < 0) continue;
                const int hb = head_b(fs[u]);
                unsigned char* ca = yc + (size_t)fs[u] * out_face_b - hb + (size_t)chs[u] * 16;
                const int q0 = chs[u] * E - hb / ES;
                if (q0 >= 0 && q0 + E <= out_reg) {
                    *reinterpret_cast<cp_u32x4*>(ca) = v[u];
                } else {                                     // chunk shared with the neighbouring range
                    T tv[E];
                    __builtin_memcpy(tv, &v[u], 16);
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (q0 + e >= 0 && q0 + e < out_reg) reinterpret_cast<T*>(ca)[e] = tv[e];
                }
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------- NCHW, planes larger than the LDS: row bands (round 3)
// The lds6 idea for planes that do not fit (224x224 f32, 256x256 f16): an item is a band of R padded rows of one
// (cube, channel, face) plane - a contiguous output range - assembled in LDS and written as one linear store stream:
//   1  the band's centre rows: 16-byte chunks of the contiguous input rows to their padded places (a chunk that runs over
//      a row end continues pl + pr elements later);
//   2  its pads straight from the neighbouring faces in global memory through the affine line table (base + a * step per
//      strip line; a column strip costs one 128-byte line per row - the read amplification the channel kernel avoids,
//      ~1.2x at 896-byte rows - but nothing is written twice or out of order); corners through cubepad_src();
//   3  aligned 16-byte LDS reads -> aligned 16-byte global stores; the chunk a band shares with the previous / next band
//      goes element by element.
// Items are ordered (plane, face, band): the workgroups running at any time work on neighbouring bands.  Bit-exact.
template <int ES>
__global__ __launch_bounds__(256) void cubepad_nchw_band_kernel(const unsigned char* __restrict__ x,
                                                                unsigned char* __restrict__ y, int C, CubePadGeom g,
                                                                int n_items, int nb, int R, float rcp_n, float rcp_wp,
                                                                float rcp_lr) {
    typedef typename ElemOf<ES>::T T;
    constexpr int E = 16 / ES;
    constexpr int LOG_E = E == 16 ? 4 : (E == 8 ? 3 : (E == 4 ? 2 : 1));
    constexpr int UNR = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
    const int n = g.n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr, LR = g.pl + g.pr;
    const int tid = threadIdx.x;
    const int nlines = g.pt + g.pd + LR;
    const int nn = n * n, HW = Hp * Wp;
    const int in_face = C * nn;                              // (6 * in_face < 2^31: launcher)
    int* ltab = reinterpret_cast<int*>(smemb);               // [6][16][element offset from the cube's channel plane 0, step]
    unsigned char* band = smemb + 768;
    auto div_by = [](int q, int d, float rcp) -> int {       // exact for 0 <= q < 2^24
        int r = (int)((float)q * rcp);
        r -= (r * d > q);
        r += ((r + 1) * d <= q);
        return r;
    };
    if (tid < 96) {
        const int f = tid >> 4, k = tid & 15;
        if (k < nlines) {
            int i0, j0, di = 0, dj = 0;
            if (k < g.pt)                      { i0 = k;     j0 = g.pl; dj = 1; }
            else if (k < g.pt + g.pd)          { i0 = n + k; j0 = g.pl; dj = 1; }
            else if (k < g.pt + g.pd + g.pl)   { i0 = g.pt;  j0 = k - g.pt - g.pd; di = 1; }
            else                               { i0 = g.pt;  j0 = n + k - g.pt - g.pd; di = 1; }
            const int s0 = cubepad_src(f, i0, j0, g), s1 = cubepad_src(f, i0 + di, j0 + dj, g);
            const int sf = s0 / nn;
            ltab[(f * 16 + k) * 2] = sf * in_face + (s0 - sf * nn);
            ltab[(f * 16 + k) * 2 + 1] = s1 - s0;
        }
    }
    __syncthreads();
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int plane = item / (6 * nb), r6 = item - plane * 6 * nb;
        const int f = r6 / nb, bnd = r6 - f * nb;
        const int grp = plane / C, c = plane - grp * C;
        const T* xin = reinterpret_cast<const T*>(x) + (size_t)grp * 6 * in_face + (size_t)c * nn;
        const T* xf = xin + (size_t)f * in_face;
        const int i0 = bnd * R, i1 = min(Hp, i0 + R);
        const int nband = (i1 - i0) * Wp;                    // output elements of the band
        unsigned char* gb = y + (((size_t)grp * 6 + f) * C + c) * (size_t)HW * ES + (size_t)i0 * Wp * ES;
        const int hb = (int)(reinterpret_cast<size_t>(gb) & 15), head = hb / ES;
        T* bp = reinterpret_cast<T*>(band) + head;            // band element q (0 .. nband) lives at bp[q]
        // ---- 1: centre rows [a0, a1)
        const int a0 = max(i0, g.pt) - g.pt, a1 = min(i1, g.pt + n) - g.pt;
        if (a1 > a0) {
            const int nel = (a1 - a0) * n, nck = (nel + E - 1) >> LOG_E;
            const T* src = xf + (size_t)a0 * n;
            const int pos0 = (g.pt + a0 - i0) * Wp + g.pl;
            for (int base = tid; base < nck; base += 256 * UNR) {
                cp_u32x4 v[UNR];
                int lo[UNR], wrap_at[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int ck = base + 256 * u;
                    lo[u] = -1;
                    if (ck < nck) {
                        const int t0 = min(ck * E, nel - E);              // (last chunk shifted back: rewrites equal values)
                        const int r = div_by(t0, n, rcp_n), col = t0 - r * n;
                        __builtin_memcpy(&v[u], src + (unsigned)t0, 16);
                        lo[u] = pos0 + r * Wp + col;
                        wrap_at[u] = n - col;
                    }
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (lo[u] < 0) continue;
                    T* d = bp + lo[u];
                    const cp_u32x4 w = v[u];
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        T val;
                        if constexpr (ES == 8)      val = (T)(((unsigned long long)w[2 * e + 1] << 32) | w[2 * e]);
                        else if constexpr (ES == 4) val = (T)w[e];
                        else if constexpr (ES == 2) val = (T)(w[e >> 1] >> (16 * (e & 1)));
                        else                        val = (T)(w[e >> 2] >> (8 * (e & 3)));
                        d[e + (e >= wrap_at[u] ? LR : 0)] = val;
                    }
                }
            }
            // ---- 2a: left / right pads of those rows
            const int nside = (a1 - a0) * LR;
            for (int idx = tid; idx < nside; idx += 256) {
                const int ra = div_by(idx, LR, rcp_lr), m = idx - ra * LR, a = a0 + ra;
                const int k = g.pt + g.pd + m;
                bp[(g.pt + a - i0) * Wp + (m < g.pl ? m : n + m)] = xin[ltab[(f * 16 + k) * 2] + a * ltab[(f * 16 + k) * 2 + 1]];
            }
        }
        // ---- 2b: pad rows of the band (top: i < pt, bottom: i >= pt + n), corners included
        {
            const int ntop = max(0, min(i1, g.pt) - i0), b0 = max(i0, g.pt + n), nbot = max(0, i1 - b0);
            const int ntot = (ntop + nbot) * Wp;
            for (int base = tid; base < ntot; base += 256 * UNR) {
                T v[UNR];
                int pos[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int idx = base + 256 * u;
                    pos[u] = -1;
                    if (idx < ntot) {
                        const int ir = div_by(idx, Wp, rcp_wp), j = idx - ir * Wp;
                        const int i = ir < ntop ? i0 + ir : b0 + (ir - ntop);
                        int so;
                        if (j >= g.pl && j < g.pl + n) {
                            const int k = i < g.pt ? i : i - n;
                            so = ltab[(f * 16 + k) * 2] + (j - g.pl) * ltab[(f * 16 + k) * 2 + 1];
                        } else {
                            const int s = cubepad_src(f, i, j, g);
                            const int sf = s / nn;
                            so = sf * in_face + (s - sf * nn);
                        }
                        v[u] = xin[so];
                        pos[u] = (i - i0) * Wp + j;
                    }
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (pos[u] >= 0) bp[pos[u]] = v[u];
            }
        }
        __syncthreads();
        // ---- 3: linear store
        {
            const int nch = (head + nband + E - 1) >> LOG_E;
            unsigned char* c0p = gb - hb;
            for (int base = tid; base < nch; base += 256 * UNR) {
                cp_u32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int ch = base + 256 * u;
                    if (ch < nch) v[u] = *reinterpret_cast<const cp_u32x4*>(band + ch * 16);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int ch = base + 256 * u;
                    if (ch >= nch) continue;
                    const int q0 = ch * E - head;
                    unsigned char* ca = c0p + (size_t)ch * 16;
                    if (q0 >= 0 && q0 + E <= nband) {
                        *reinterpret_cast<cp_u32x4*>(ca) = v[u];
                    } else {                                 // chunk shared with the neighbouring band / plane
                        T tv[E];
                        __builtin_memcpy(tv, &v[u], 16);
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            if (q0 + e >= 0 && q0 + e < nband) reinterpret_cast<T*>(ca)[e] = tv[e];
                    }
                }
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------- NCHW, small faces: whole cubes through LDS (round 3)
// Faces whose rows are shorter than a cache line (28x28 / 14x14 / 7x7: layers 2-4 and the ConvLSTM) are where the
// element-per-lane kernel sits at 0.09-0.16 of the HBM peak (profiles/r02f_hbm_kernels.md): 2-4 byte accesses and one
// cubepad_src() evaluation per element.  Here a workgroup item is (cube g, channel range [c0, c0 + CH)):
//   * its input is 6 CONTIGUOUS byte ranges (face f, channels c0 .. c0+CH-1: CH * n^2 elements), read with 16-byte loads
//     into LDS - coalesced HBM reads of whole lines although a plane is only 98-1568 bytes;
//   * CubePad copies only between the 6 faces of one cube (cube_pad.py:114-216), so every output element of the item's
//     6 x CH padded planes is one of those LDS elements: a table [6][Hp][Wp] of (source face, offset) built ONCE per
//     workgroup from cubepad_src() (the face-border strips and the replicated corners of make_cubepad_edge are just table
//     entries) drives the gather;
//   * its output is 6 contiguous byte ranges too (CH * Hp * Wp elements per face), assembled 16 bytes per lane and written
//     with aligned 16-byte stores.
// Workgroups are persistent over items, so the table costs 6 Hp Wp / 256 evaluations per thread once.  Needs n <= 32,
// 16-byte aligned ranges (checked by the launcher: c0 * n^2 * ES and c0 * Hp * Wp * ES multiples of 16); anything else
// takes the kernels above.  Bit-exact like every CubePad kernel (pure copy).
template <int ES>
__global__ __launch_bounds__(256) void cubepad_nchw_cube_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y,
                                                                int C, CubePadGeom g, int CH, int n_items, int ranges) {
    typedef typename ElemOf<ES>::T T;
    constexpr int E = 16 / ES;
    extern __shared__ __attribute__((aligned(16))) unsigned char cube_lds[];
    const int n = g.n, nn = n * n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr, HW = Hp * Wp;
    unsigned short* tab = reinterpret_cast<unsigned short*>(cube_lds);                    // [6][HW]: sf << 10 | off  (n^2 <= 1024)
    const int tab_bytes = (6 * HW * 2 + 15) & ~15;
    T* in_s = reinterpret_cast<T*>(cube_lds + tab_bytes);                                // [6][CH * nn]
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 6 * HW; idx += 256) {
        const int f = idx / HW, q = idx - f * HW;
        const int i = q / Wp, j = q - i * Wp;
        const int s = cubepad_src(f, i, j, g);
        const int sf = s / nn;
        tab[idx] = (unsigned short)((sf << 10) | (s - sf * nn));
    }
    const int in_chunks = CH * nn * ES / 16, out_chunks = CH * HW * ES / 16;              // per face (launcher: exact)
    const size_t in_face = (size_t)C * nn * ES, out_face = (size_t)C * HW * ES;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int grp = item / ranges, c0 = (item - grp * ranges) * CH;
        __syncthreads();                                     // table built / previous item's gather done with in_s
        const unsigned char* xin = x + (size_t)grp * 6 * in_face + (size_t)c0 * nn * ES;
        for (int k = tid; k < 6 * in_chunks; k += 256) {
            const int f = k / in_chunks, ck = k - f * in_chunks;
            const cp_u32x4 v = *reinterpret_cast<const cp_u32x4*>(xin + (size_t)f * in_face + (size_t)ck * 16);
            *reinterpret_cast<cp_u32x4*>(reinterpret_cast<unsigned char*>(in_s) + ((size_t)f * in_chunks + ck) * 16) = v;
        }
        __syncthreads();
        unsigned char* yout = y + (size_t)grp * 6 * out_face + (size_t)c0 * HW * ES;
        for (int k = tid; k < 6 * out_chunks; k += 256) {
            const int f = k / out_chunks, ck = k - f * out_chunks;
            int e0 = ck * E;                                 // first element of the chunk inside the face's CH planes
            int ch = e0 / HW, q = e0 - ch * HW;
            const unsigned short* tf = tab + f * HW;
            T tmp[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const unsigned t = tf[q];
                tmp[e] = in_s[(t >> 10) * (CH * nn) + ch * nn + (t & 1023u)];
                if (++q == HW) { q = 0; ++ch; }
            }
            cp_u32x4 v;
            __builtin_memcpy(&v, tmp, 16);
            *reinterpret_cast<cp_u32x4*>(yout + (size_t)f * out_face + (size_t)ck * 16) = v;
        }
    }
}

// Kernels that take more than the 64 KB default of dynamic LDS: raise the limit once per (kernel instance, device) - a
// process that drives several GPUs sets it on each (thread-safe: at worst two threads set the same attribute).
template <auto Kernel>
static int ensure_big_lds() {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CP360_ERR_HIP;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess)
            return CP360_ERR_HIP;
        done.fetch_or(bit, std::memory_order_release);
    }
    return CP360_OK;
}

template <typename T>
static int launch_nchw(const void* x, void* y, int n6, int C, const CubePadGeom& g, hipStream_t st) {
    const int Wp = g.n + g.pl + g.pr;
    const int planes = (n6 / 6) * C;
    // small faces: whole cubes through LDS.  CH channels per item: the largest divisor-free choice whose input fits
    // ~24 KiB and whose per-face input / output ranges are whole 16-byte chunks at 16-byte aligned addresses.  Measured
    // against the lds6 kernel (profiles/r03zz_hbm_kernels.md): better up to 14x14 faces (29 / 12 us against 35 / 15 us
    // on [384,256,14,14] f16 / the ConvLSTM's [24,2000,7,7] f32), worse at 28x28 (59 against 42 us): it goes first for
    // n <= 16 and stays the fallback up to 32.
    auto try_cube = [&](int max_n) -> int {
        constexpr int ES = (int)sizeof(T);
        const int nn = g.n * g.n, HW = (g.n + g.pt + g.pd) * Wp;
        static const int no_cube = []() { const char* e = getenv("CP360_CUBEPAD_NOCUBE"); return e ? atoi(e) : 0; }();   // A/B switch
        if (!no_cube && g.n <= max_n && HW <= 1444 && (reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0) {
            int CH = 0;
            for (int ch = 1; ch <= C && (size_t)6 * ch * nn * ES <= 24 * 1024; ++ch)
                if (C % ch == 0 && (ch * nn * ES) % 16 == 0 && (ch * HW * ES) % 16 == 0) CH = ch;
            if (CH > 0 && ((size_t)C * nn * ES) % 16 == 0 && ((size_t)C * HW * ES) % 16 == 0) {
                const int ranges = C / CH;
                const long long items = (long long)(n6 / 6) * ranges;
                const size_t lds = (size_t)((6 * HW * 2 + 15) & ~15) + (size_t)6 * CH * nn * ES;
                long long blocks = items < 256 * 4 ? items : 256 * 4;
                hipLaunchKernelGGL((cubepad_nchw_cube_kernel<ES>), dim3((unsigned)blocks), dim3(256), lds, st,
                                   (const unsigned char*)x, (unsigned char*)y, C, g, CH, (int)items, ranges);
                CP360_CHECK_HIP();
                return CP360_OK;
            }
        }
        return 1;                                            // not taken
    };
    if (int r = try_cube(16); r <= 0) return r;
    {   // strip kernel: pads of at most 4 (every pad of the network is 1 or 3), strips within the 64 KiB LDS default
        const int P = max(max(g.pl, g.pr), max(g.pt, g.pd));
        constexpr int ES = (int)sizeof(T);
        const int Hp = g.n + g.pt + g.pd;
        const size_t lds = (size_t)4 * (g.pt + g.pd + g.pl + g.pr) * g.n * ES;      // four waves' strips
        static const int no_strip = []() {
            const char* e = getenv("CP360_CUBEPAD_ELEMENTWISE");      // A/B switch (tools/hbm_kernels.py)
            return e ? atoi(e) : 0;
        }();
        // (rows of at least 112 bytes - 56x56 f16 faces: 245 -> 149 us against the element-per-lane kernel, round 3; smaller
        //  faces take the whole-cube kernel above)
        static const int strip_min = []() { const char* e = getenv("CP360_CUBEPAD_STRIP_MIN"); return e ? atoi(e) : 112; }();   // A/B switch
        static const int strip_v1 = []() { const char* e = getenv("CP360_CUBEPAD_STRIP_V1"); return e ? atoi(e) : 0; }();   // A/B switch
        constexpr int E = 16 / ES;
        const int LR = g.pl + g.pr, nlines = g.pt + g.pd + LR;
        const long long ps_size = (long long)Hp * Wp - (long long)g.n * g.n;
        const int ps_alloc = (int)((ps_size + 2 * E + E - 1) / E * E);
        const size_t lds2 = 512 + (size_t)4 * ps_alloc * ES;
        static const int no_channel = []() { const char* e = getenv("CP360_CUBEPAD_NOCHANNEL"); return e ? atoi(e) : 0; }();   // A/B switch
        static const int channel_min = []() { const char* e = getenv("CP360_CUBEPAD_CHANNEL_MIN"); return e ? atoi(e) : 96; }();
        {   // six padded planes assembled in LDS, one linear store stream
            static const int no_lds6 = []() { const char* e = getenv("CP360_CUBEPAD_NOLDS6"); return e ? atoi(e) : 0; }();   // A/B switch
            static const int lds6_min = []() { const char* e = getenv("CP360_CUBEPAD_LDS6_MIN"); return e ? atoi(e) : 64; }();
            const long long HW = (long long)Hp * Wp, nn = (long long)g.n * g.n;
            // channels per item: one for faces whose six planes fill the LDS; for small faces the largest divisor of C that
            // keeps the six ranges within ~40 KB (>= 4 workgroups per CU) and leaves >= 512 items
            int CH = 1;
            for (int ch = 2; ch <= C && ch <= 64; ++ch) {
                if (C % ch) continue;
                const long long ps = ((ch * HW + E) * ES + 15) / 16 * 16;
                if (768 + 6 * ps > 40 * 1024 || (long long)(n6 / 6) * (C / ch) < 512) break;
                CH = ch;
            }
            const long long items = (long long)(n6 / 6) * (C / CH);
            const long long pstride = ((CH * HW + E) * ES + 15) / 16 * 16;
            const size_t lds6 = 768 + (size_t)6 * pstride;
            if (!no_strip && !no_lds6 && P >= 1 && nlines <= 16 && g.n >= E && CH * nn >= E && lds6 <= 160 * 1024 &&
                CH * HW < (1 << 22) && items >= lds6_min && items < (1ll << 31) && (reinterpret_cast<size_t>(y) % ES) == 0 &&
                (reinterpret_cast<size_t>(x) % ES) == 0 &&
                ensure_big_lds<&cubepad_nchw_lds6_kernel<ES>>() == CP360_OK) {    // (refused: on down the chain to the <= 64 KB kernels)
                static const int force_nt6 = []() { const char* e = getenv("CP360_CUBEPAD_LDS6_NT"); return e ? atoi(e) : 0; }();
                int per_cu = (int)((size_t)160 * 1024 / lds6);
                int nt = per_cu >= 4 ? 256 : (per_cu >= 2 ? 512 : 1024);
                if (force_nt6 == 256 || force_nt6 == 512 || force_nt6 == 1024) nt = force_nt6;
                if (per_cu > 2048 / nt) per_cu = 2048 / nt;
                long long blocks = items < 256ll * per_cu ? items : 256ll * per_cu;
                const int nstrip = nlines * g.n;
                const int cpf = (int)((CH * nn + E - 1) / E), nchmax = (int)((CH * HW + 2 * E - 2) / E);
                hipLaunchKernelGGL((cubepad_nchw_lds6_kernel<ES>), dim3((unsigned)blocks), dim3(nt), lds6, st,
                                   (const unsigned char*)x, (unsigned char*)y, C, g, CH, C / CH, (int)items, (int)pstride,
                                   1.0f / (float)g.n, 1.0f / (float)nn, 1.0f / (float)cpf, 1.0f / (float)(nstrip > 0 ? nstrip : 1),
                                   1.0f / (float)(CH * nstrip > 0 ? CH * nstrip : 1), 1.0f / (float)nchmax);
                CP360_CHECK_HIP();
                return CP360_OK;
            }
        }
        if (int r = try_cube(32); r <= 0) return r;
        {   // planes larger than the LDS: row bands assembled in LDS, linear stores
            static const int no_band = []() { const char* e = getenv("CP360_CUBEPAD_NOBAND"); return e ? atoi(e) : 0; }();   // A/B switch
            static const int band_min_row = []() { const char* e = getenv("CP360_CUBEPAD_BAND_MINROW"); return e ? atoi(e) : 256; }();
            int R = 16384 / (Wp * ES);                       // ~16 KB of output per item
            if (R < 2) R = 2;
            if (R > Hp) R = Hp;
            const int nb = (Hp + R - 1) / R;
            const long long items = (long long)planes * 6 * nb;
            const size_t ldsb = 768 + (size_t)((long long)R * Wp + 2 * E) * ES + 16;
            if (!no_strip && !no_band && P >= 1 && nlines <= 16 && g.n >= E && g.n * ES >= band_min_row && ldsb <= 64 * 1024 &&
                (long long)Hp * Wp < (1 << 22) && (long long)6 * C * g.n * g.n < (1ll << 31) && items < (1ll << 31) &&
                (reinterpret_cast<size_t>(y) % ES) == 0 && (reinterpret_cast<size_t>(x) % ES) == 0) {
                long long blocks = items < 256 * 8 ? items : 256 * 8;
                hipLaunchKernelGGL((cubepad_nchw_band_kernel<ES>), dim3((unsigned)blocks), dim3(256), ldsb, st,
                                   (const unsigned char*)x, (unsigned char*)y, C, g, (int)items, nb, R, 1.0f / (float)g.n,
                                   1.0f / (float)Wp, 1.0f / (float)(LR > 0 ? LR : 1));
                CP360_CHECK_HIP();
                return CP360_OK;
            }
        }
        {   // every input byte once: (cube, channel) items
            const long long items = (long long)(n6 / 6) * C;
            const size_t lds3 = 768 + ((size_t)E + (size_t)12 * g.n * E + (size_t)12 * P * g.n + (size_t)6 * ps_alloc + 2 * E) * ES;
            if (!no_strip && !strip_v1 && !no_channel && P >= 1 && P <= E && g.n * ES >= strip_min && nlines <= 16 &&
                g.n >= 2 * E + LR && g.n >= 2 * P && lds3 <= 160 * 1024 && (long long)Hp * Wp < (1 << 22) && items >= channel_min &&
                items < (1ll << 31) && (long long)g.n * g.n < (1ll << 31) && (reinterpret_cast<size_t>(y) % ES) == 0 &&
                (reinterpret_cast<size_t>(x) % ES) == 0 &&
                ensure_big_lds<&cubepad_nchw_channel_kernel<ES>>() == CP360_OK) {
                // one wave per face while that gives >= 12 waves per CU, two (768 threads) for few / large items
                static const int force_nt = []() { const char* e = getenv("CP360_CUBEPAD_CHANNEL_NT"); return e ? atoi(e) : 0; }();
                int nt = items >= 512 ? 384 : 768;
                if (force_nt == 384 || force_nt == 768) nt = force_nt;
                int per_cu = (int)((size_t)160 * 1024 / lds3);
                if (per_cu > 2048 / nt) per_cu = 2048 / nt;
                if (per_cu < 1) per_cu = 1;
                long long blocks = items < 256ll * per_cu ? items : 256ll * per_cu;
                const int cpr = g.n / E + 1, maxc = (LR + 2 * E - 2) / E;
                int rbw = 64 * 8 / cpr;                      // rows per block: one 8-deep body of run chunks per wave
                if (rbw < 4) rbw = 4;
                hipLaunchKernelGGL((cubepad_nchw_channel_kernel<ES>), dim3((unsigned)blocks), dim3(nt), lds3, st,
                                   (const unsigned char*)x, (unsigned char*)y, C, g, (int)items, 1.0f / (float)Wp,
                                   1.0f / (float)g.n, 1.0f / (float)cpr, maxc, 1.0f / (float)maxc, P, ps_alloc, rbw);
                CP360_CHECK_HIP();
                return CP360_OK;
            }
        }
        if (!no_strip && !strip_v1 && P >= 1 && g.n * ES >= strip_min && nlines <= 16 && g.n >= 2 * E + LR && lds2 <= 64 * 1024 &&
            (long long)Hp * Wp < (1 << 22) && (long long)6 * C * g.n * g.n < (1ll << 31) && (reinterpret_cast<size_t>(y) % ES) == 0 &&
            (reinterpret_cast<size_t>(x) % ES) == 0) {
            const long long items = (long long)planes * 6;
            long long blocks = (items + 3) / 4;
            if (blocks > 256 * 32) blocks = 256 * 32;
            const int cpr = g.n / E + 1, dq = 64 / cpr, dm = 64 - dq * cpr;
            const int maxc = (LR + 2 * E - 2) / E;
            hipLaunchKernelGGL((cubepad_nchw_plane_kernel<ES>), dim3((unsigned)blocks), dim3(256), lds2, st,
                               (const unsigned char*)x, (unsigned char*)y, C, g, (int)items, (long long)n6 * C * g.n * g.n,
                               1.0f / (float)Wp, 1.0f / (float)g.n, 1.0f / (float)cpr, dq, dm, maxc, 1.0f / (float)maxc, ps_alloc,
                               std::max(4, 256 / cpr));
            CP360_CHECK_HIP();
            return CP360_OK;
        }
        if (!no_strip && P >= 1 && P <= g.n && g.n * ES >= strip_min && lds <= 64 * 1024 && (long long)Hp * Wp < (1 << 22) &&
            (reinterpret_cast<size_t>(y) % ES) == 0) {
            const long long items = (long long)planes * 6;
            long long blocks = (items + 3) / 4;
            if (blocks > 256 * 32) blocks = 256 * 32;
            hipLaunchKernelGGL((cubepad_nchw_strip_kernel<ES>), dim3((unsigned)blocks), dim3(256), lds, st,
                               (const unsigned char*)x, (unsigned char*)y, C, g, (int)items, 1.0f / (float)Wp,
                               1.0f / (float)g.n);
            CP360_CHECK_HIP();
            return CP360_OK;
        }
    }
    if (Wp <= 16)
        hipLaunchKernelGGL((cubepad_nchw_kernel<T, 4>), dim3(planes), dim3(256), 0, st, (const T*)x, (T*)y, C, g);
    else if (Wp <= 32)
        hipLaunchKernelGGL((cubepad_nchw_kernel<T, 5>), dim3(planes), dim3(256), 0, st, (const T*)x, (T*)y, C, g);
    else
        hipLaunchKernelGGL((cubepad_nchw_kernel<T, 6>), dim3(planes), dim3(256), 0, st, (const T*)x, (T*)y, C, g);
    CP360_CHECK_HIP();
    return CP360_OK;
}

static int check_geom(int n6, int C, int n, int pl, int pr, int pt, int pd) {
    if (n6 <= 0 || C <= 0 || n <= 0 || pl < 0 || pr < 0 || pt < 0 || pd < 0) return CP360_ERR_BAD_SHAPE;
    if (n6 % 6 != 0) return CP360_ERR_BATCH_NOT_6N;
    if (pl > n || pr > n || pt > n || pd > n) return CP360_ERR_BAD_SHAPE;
    return CP360_OK;
}

extern "C" int cp360_cubepad_table_host(int n, int pl, int pr, int pt, int pd, int32_t* table_host) {
    if (!table_host) return CP360_ERR_NULL;
    int rc = check_geom(6, 1, n, pl, pr, pt, pd);
    if (rc) return rc;
    CubePadGeom g{n, pl, pr, pt, pd};
    const int Hp = n + pt + pd, Wp = n + pl + pr;
    for (int f = 0; f < 6; ++f)
        for (int i = 0; i < Hp; ++i)
            for (int j = 0; j < Wp; ++j) table_host[(f * Hp + i) * Wp + j] = cubepad_src(f, i, j, g);
    return CP360_OK;
}

extern "C" int cp360_cubepad_nchw(const void* x, void* y, int n6, int C, int n, int pl, int pr, int pt, int pd,
                                  int elem_size, void* stream) {
    if (!x || !y) return CP360_ERR_NULL;
    int rc = check_geom(n6, C, n, pl, pr, pt, pd);
    if (rc) return rc;
    CubePadGeom g{n, pl, pr, pt, pd};
    hipStream_t st = (hipStream_t)stream;
    switch (elem_size) {
        case 1: return launch_nchw<uint8_t>(x, y, n6, C, g, st);
        case 2: return launch_nchw<uint16_t>(x, y, n6, C, g, st);
        case 4: return launch_nchw<uint32_t>(x, y, n6, C, g, st);
        case 8: return launch_nchw<uint64_t>(x, y, n6, C, g, st);
        default: return CP360_ERR_BAD_DTYPE;
    }
}

// ---------------------------------------------------------------- NHWC
// A pixel is a contiguous channel vector: the pad is a pixel-granular gather, so
// every load and store is a full-width (4/8/16-byte per lane) contiguous access.
// VEC = 32-bit words per lane access.  One "row" of work = one output pixel.
// IT = index type: unsigned when the vector count fits 32 bits (always, in practice): the three divisions per vector are
// then 32-bit (the 64-bit form costs more instructions than cubepad_src() itself).
template <int VEC, typename IT>
__global__ __launch_bounds__(256) void cubepad_nhwc_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                           int n6, int cw /*words per in pixel*/,
                                                           int cyw /*words per out pixel*/, CubePadGeom g) {
    const int n = g.n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr;
    const int vec_per_pix = cyw / VEC;
    const IT total = (IT)((long long)n6 * Hp * Wp * vec_per_pix);
    for (IT idx = (IT)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (IT)gridDim.x * blockDim.x) {
        const IT pix = idx / (IT)vec_per_pix;
        const int v = (int)(idx - pix * (IT)vec_per_pix) * VEC;
        const int j = (int)(pix % (IT)Wp);
        const IT t = pix / (IT)Wp;
        const int i = (int)(t % (IT)Hp);
        const int img = (int)(t / (IT)Hp);
        const int grp = img / 6, f = img - grp * 6;
        const int s = cubepad_src(f, i, j, g);
        const uint32_t* src = x + ((size_t)grp * 6 * n * n + s) * cw + v;
        uint32_t* dst = y + (size_t)pix * cyw + v;
        if constexpr (VEC == 4) {
            uint4 val = (v < cw) ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(dst) = val;
        } else if constexpr (VEC == 2) {
            uint2 val = (v < cw) ? *reinterpret_cast<const uint2*>(src) : make_uint2(0, 0);
            *reinterpret_cast<uint2*>(dst) = val;
        } else {
            *dst = (v < cw) ? *src : 0u;
        }
    }
}

// Narrow pixels (1-4 words: the 3-channel f32 input of conv1 is 3): a thread owns a whole output pixel - one
// cubepad_src() and one set of 32-bit divisions per pixel instead of per word, one CW-word load and one store.
template <int CW>
__global__ __launch_bounds__(256) void cubepad_nhwc_px_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                              unsigned total_px, int cyw, CubePadGeom g) {
    const int n = g.n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr;
    for (unsigned pix = blockIdx.x * blockDim.x + threadIdx.x; pix < total_px; pix += gridDim.x * blockDim.x) {
        const int j = (int)(pix % (unsigned)Wp);
        const unsigned t = pix / (unsigned)Wp;
        const int i = (int)(t % (unsigned)Hp);
        const int img = (int)(t / (unsigned)Hp);
        const int grp = img / 6, f = img - grp * 6;
        const int s = cubepad_src(f, i, j, g);
        const uint32_t* src = x + ((size_t)grp * 6 * n * n + s) * CW;
        uint32_t* dst = y + (size_t)pix * cyw;
        uint32_t w[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) w[e] = src[e];
#pragma unroll
        for (int e = 0; e < CW; ++e) dst[e] = w[e];
        for (int e = CW; e < cyw; ++e) dst[e] = 0u;
    }
}

extern "C" int cp360_cubepad_nhwc(const void* x, void* y, int n6, int C, int Cy, int n, int pl, int pr, int pt,
                                  int pd, int elem_size, void* stream) {
    if (!x || !y) return CP360_ERR_NULL;
    int rc = check_geom(n6, C, n, pl, pr, pt, pd);
    if (rc) return rc;
    if (Cy < C || (elem_size != 1 && elem_size != 2 && elem_size != 4)) return CP360_ERR_BAD_SHAPE;
    if ((C * elem_size) % 4 != 0 || (Cy * elem_size) % 4 != 0) return CP360_ERR_ALIGN;
    const int cw = C * elem_size / 4, cyw = Cy * elem_size / 4;
    CubePadGeom g{n, pl, pr, pt, pd};
    hipStream_t st = (hipStream_t)stream;
    const int Hp = n + pt + pd, Wp = n + pl + pr;
    // widest vector that divides both pixel widths (a vector never straddles the
    // end of the input channels, so the zero-fill test is per vector)
    int vec = (cw % 4 == 0 && cyw % 4 == 0) ? 4 : ((cw % 2 == 0 && cyw % 2 == 0) ? 2 : 1);
    const long long total = (long long)n6 * Hp * Wp * (cyw / vec);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;     // ~8 workgroups per CU, grid-stride the rest
    if (blocks < 1) blocks = 1;
    const uint32_t* xi = (const uint32_t*)x;
    uint32_t* yo = (uint32_t*)y;
    static const int no_px = []() { const char* e = getenv("CP360_CUBEPAD_NHWC_NOPX"); return e ? atoi(e) : 0; }();   // A/B switch
    const long long total_px = (long long)n6 * Hp * Wp;
    long long pb = (total_px + 255) / 256;
    if (pb > 256 * 16) pb = 256 * 16;
    // narrow pixels: a thread per pixel (32-bit pixel index: the grid-stride increment must not wrap either)
    if (!no_px && vec < 4 && cw <= 4 && cyw <= 8 && total_px < (1ll << 32) - pb * 256) {
#define CP360_PX(CWV) hipLaunchKernelGGL((cubepad_nhwc_px_kernel<CWV>), dim3((unsigned)pb), dim3(256), 0, st, xi, yo, (unsigned)total_px, cyw, g)
        if (cw == 1) CP360_PX(1); else if (cw == 2) CP360_PX(2); else if (cw == 3) CP360_PX(3); else CP360_PX(4);
#undef CP360_PX
        CP360_CHECK_HIP();
        return CP360_OK;
    }
    const bool i32 = total < (1ll << 32) - (long long)blocks * 256;      // (the grid-stride increment must not wrap either)
#define CP360_NHWC(V) do { if (i32) hipLaunchKernelGGL((cubepad_nhwc_kernel<V, unsigned>), dim3((unsigned)blocks), dim3(256), 0, st, xi, yo, n6, cw, cyw, g); \
                           else hipLaunchKernelGGL((cubepad_nhwc_kernel<V, long long>), dim3((unsigned)blocks), dim3(256), 0, st, xi, yo, n6, cw, cyw, g); } while (0)
    if (vec == 4) CP360_NHWC(4);
    else if (vec == 2) CP360_NHWC(2);
    else CP360_NHWC(1);
#undef CP360_NHWC
    CP360_CHECK_HIP();
    return CP360_OK;
}

// ---------------------------------------------------------------- misc
extern "C" const char* cp360_strerror(int status) {
    switch (status) {
        case CP360_OK: return "ok";
        case CP360_ERR_BAD_SHAPE: return "bad shape";
        case CP360_ERR_BATCH_NOT_6N: return "CubePad size mismatch: batch is not a multiple of 6";
        case CP360_ERR_NOT_SQUARE: return "cube faces must be square";
        case CP360_ERR_BAD_DTYPE: return "unsupported dtype";
        case CP360_ERR_NULL: return "null pointer";
        case CP360_ERR_ALIGN: return "channel count / stride not 16-byte friendly";
        case CP360_ERR_HIP: return "HIP runtime error";
        case CP360_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown cp360 status";
    }
}

extern "C" int cp360_version(void) { return CP360_VERSION; }

static thread_local int t_launch_mode = 0;      // 0 ascending, 1 descending, 2 alternate
static thread_local int t_launch_flip = 0;
int cp360_launch_reverse() {
    if (t_launch_mode != 2) return t_launch_mode;
    t_launch_flip ^= 1;
    return t_launch_flip;
}
extern "C" int cp360_set_launch_order(int mode) {
    const int old = t_launch_mode;
    t_launch_mode = (mode == 1 || mode == 2) ? mode : 0;
    t_launch_flip = 0;
    return old;
}
extern "C" size_t cp360_conv_desc_bytes(void) { return sizeof(cp360_conv_desc); }
