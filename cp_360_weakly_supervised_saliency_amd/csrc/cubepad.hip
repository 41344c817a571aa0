// K2: CubePad as stand-alone HIP kernels (NCHW for the reference's module boundary,
// NHWC for the fused pipeline) + the host-side table export.
// Semantics: model/cube_pad.py:28-42,95-216 (see common.h: cubepad_src).
#include "common.h"

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

template <typename T>
static int launch_nchw(const void* x, void* y, int n6, int C, const CubePadGeom& g, hipStream_t st) {
    const int Wp = g.n + g.pl + g.pr;
    const int planes = (n6 / 6) * C;
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
extern "C" size_t cp360_conv_desc_bytes(void) { return sizeof(cp360_conv_desc); }
