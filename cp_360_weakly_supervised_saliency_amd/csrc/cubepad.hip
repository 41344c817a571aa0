// K2: CubePad as stand-alone HIP kernels (NCHW for the reference's module boundary,
// NHWC for the fused pipeline) + the host-side table export.
// Semantics: model/cube_pad.py:28-42,95-216 (see common.h: cubepad_src).
#include "common.h"
#include <stdlib.h>

// ---------------------------------------------------------------- NCHW
// One workgroup per (group g, channel c): it writes the 6 padded planes of that
// channel.  Lanes run along the output row (coalesced stores; the centre rows are
// coalesced loads too, the strips are short gathers from neighbouring faces that
// sit in L2 because the same workgroup reads those faces as its own centres).
// LX = lanes per output row (power of two >= Wp, capped at 64) so small faces
// (9x9 ConvLSTM tiles) still fill a wave with several rows.
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

template <typename T>
static int launch_nchw(const void* x, void* y, int n6, int C, const CubePadGeom& g, hipStream_t st) {
    const int Wp = g.n + g.pl + g.pr;
    const int planes = (n6 / 6) * C;
    {   // small faces: whole cubes through LDS.  CH channels per item: the largest divisor-free choice whose input fits
        // ~24 KiB and whose per-face input / output ranges are whole 16-byte chunks at 16-byte aligned addresses
        constexpr int ES = (int)sizeof(T);
        const int nn = g.n * g.n, HW = (g.n + g.pt + g.pd) * Wp;
        static const int no_cube = []() { const char* e = getenv("CP360_CUBEPAD_NOCUBE"); return e ? atoi(e) : 0; }();   // A/B switch
        if (!no_cube && g.n <= 32 && HW <= 1444 && (reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0) {
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
    }
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
template <int VEC>
__global__ __launch_bounds__(256) void cubepad_nhwc_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                           int n6, int cw /*words per in pixel*/,
                                                           int cyw /*words per out pixel*/, CubePadGeom g) {
    const int n = g.n, Hp = n + g.pt + g.pd, Wp = n + g.pl + g.pr;
    const int vec_per_pix = cyw / VEC;
    const long long total = (long long)n6 * Hp * Wp * vec_per_pix;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx / vec_per_pix;
        const int v = (int)(idx - pix * vec_per_pix) * VEC;
        const int j = (int)(pix % Wp);
        const long long t = pix / Wp;
        const int i = (int)(t % Hp);
        const int img = (int)(t / Hp);
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
    if (vec == 4)
        hipLaunchKernelGGL((cubepad_nhwc_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, st, xi, yo, n6, cw, cyw, g);
    else if (vec == 2)
        hipLaunchKernelGGL((cubepad_nhwc_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, xi, yo, n6, cw, cyw, g);
    else
        hipLaunchKernelGGL((cubepad_nhwc_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, xi, yo, n6, cw, cyw, g);
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
