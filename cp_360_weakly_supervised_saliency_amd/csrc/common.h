// Shared device/host helpers for libcp360 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/cp360_internal.h"

#define CP360_CHECK_HIP()                                        \
    do {                                                         \
        hipError_t e__ = hipGetLastError();                      \
        if (e__ != hipSuccess) return CP360_ERR_HIP;             \
    } while (0)

typedef unsigned short bf16_raw;   // storage type of a bf16 element
typedef _Float16 f16_raw;          // IEEE half: a distinct C++ type, so templates can tell it from bf16

__device__ __forceinline__ float bf16_to_f32(bf16_raw v) {
    return __uint_as_float(((unsigned)v) << 16);
}
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
    // plain cast -> v_cvt_pk_bf16_f32 (RNE, keeps NaN a NaN)
    __hip_bfloat16 h = __float2bfloat16(f);
    return __builtin_bit_cast(bf16_raw, h);
}

// ---------------------------------------------------------------------------------
// CubePad source map.  One inline function shared by the host table builder, the
// stand-alone CubePad kernels, the fused max-pool and the convolution tile loader.
//
// Derived from model/cube_pad.py:114-162 (strips) and :83-90,164-176 (corners).
// Faces: 0 back, 1 down, 2 front, 3 left, 4 right, 5 top.  For the four strips of
// face f the table gives (source face, row selector, column selector) with
// selectors 0: k (depth index in output order), 1: e = n - p + k, 2: a (running
// index along the edge), 3: m = n - 1 - a.
// ---------------------------------------------------------------------------------
struct CubePadGeom {
    int n, pl, pr, pt, pd;
};

// Packed strip table: one 64-bit word per side (t, d, l, r); byte f of a word is
// face f's entry = source_face | row_selector << 3 | col_selector << 5.
//   top    [k, a]: B:(T,k,m) D:(F,e,a) F:(T,e,a) L:(T,a,k) R:(T,m,e) T:(B,k,m)   cube_pad.py:114-126
//   down   [k, a]: B:(D,e,m) D:(B,e,m) F:(D,k,a) L:(D,m,k) R:(D,a,e) T:(F,k,a)   cube_pad.py:127-138
//   left   [a, k]: B:(R,a,e) D:(L,e,m) F:(L,a,e) L:(B,a,e) R:(F,a,e) T:(L,k,a)   cube_pad.py:139-150
//   right  [a, k]: B:(L,a,k) D:(R,e,a) F:(R,a,k) L:(F,a,k) R:(B,a,k) T:(R,k,m)   cube_pad.py:151-162
#define CP360_STRIP_T 0x0000603d154d4a65ULL
#define CP360_STRIP_D 0x0000423119416869ULL
#define CP360_STRIP_L 0x0000433230336b34ULL
#define CP360_STRIP_R 0x0000641012144c13ULL

// Returns flat index f'*n*n + i'*n + j' of the source element of padded position
// (face f, row i, col j), 0 <= i < n+pt+pd, 0 <= j < n+pl+pr.
__host__ __device__ __forceinline__ int cubepad_src(int f, int i, int j, const CubePadGeom& g) {
    const int n = g.n;
    const bool in_t = i < g.pt, in_d = i >= g.pt + n;
    const bool in_l = j < g.pl, in_r = j >= g.pl + n;
    if (!(in_t | in_d | in_l | in_r)) return (f * n + (i - g.pt)) * n + (j - g.pl);
    int side, a, k, p;
    if ((in_t | in_d) && !(in_l | in_r)) {            // top / down strip
        side = in_t ? 0 : 1;
        p = in_t ? g.pt : g.pd;
        k = in_t ? i : i - g.pt - n;
        a = j - g.pl;
    } else if (!(in_t | in_d)) {                      // left / right strip
        side = in_l ? 2 : 3;
        p = in_l ? g.pl : g.pr;
        k = in_l ? j : j - g.pl - n;
        a = i - g.pt;
    } else {                                          // corner (make_cubepad_edge)
        const int p_td = in_t ? g.pt : g.pd, p_lr = in_l ? g.pl : g.pr;
        const int k_td = in_t ? i : i - g.pt - n, k_lr = in_l ? j : j - g.pl - n;
        if (p_td > p_lr) {                            // replicate l/r strip row vertically
            side = in_l ? 2 : 3;
            p = p_lr;
            k = k_lr;
            a = in_t ? 0 : n - 1;
        } else {                                      // replicate t/d strip column horizontally
            side = in_t ? 0 : 1;
            p = p_td;
            k = k_td;
            a = in_l ? 0 : n - 1;
        }
    }
    const unsigned long long word = side == 0 ? CP360_STRIP_T : (side == 1 ? CP360_STRIP_D
                                   : (side == 2 ? CP360_STRIP_L : CP360_STRIP_R));
    const int code = (int)((word >> (8 * f)) & 0xff);
    const int sf = code & 7, rs = (code >> 3) & 3, cs = (code >> 5) & 3;
    const int e = n - p + k, m = n - 1 - a;
    const int row = rs == 0 ? k : (rs == 1 ? e : (rs == 2 ? a : m));
    const int col = cs == 0 ? k : (cs == 1 ? e : (cs == 2 ? a : m));
    return (sf * n + row) * n + col;
}

// Traversal order of the launch being issued by this host thread (cp360_set_launch_order, include/cp360.h): 0 = work
// items in ascending order, 1 = descending; in the alternating mode every call flips it.  Results never depend on it.
// Defined in cubepad.hip; called once per launch by the kernels that honour it.
int cp360_launch_reverse();

