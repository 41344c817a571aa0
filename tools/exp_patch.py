"""Source patches for tools/exp_build.sh (timing experiments on the convolution kernels; see there).
The patches are textual: a variant asserts if the code it rewrites has changed since it was written
(`base` always builds)."""
import sys

path, variants = sys.argv[1], sys.argv[2].split('+')
s = open(path).read()


def sub(a, b, count=None):
    global s
    assert a in s, a
    s = s.replace(a, b) if count is None else s.replace(a, b, count)


for v in variants:
    if v == 'base':
        pass
    elif v == 'nomma':
        i = s.index('template <>\n__device__ __forceinline__ void mma_chunk<bf16_raw>')
        j = s.index('}\n', i) + 2
        s = s[:i] + ('template <>\n__device__ __forceinline__ void mma_chunk<bf16_raw>(f32x4& acc, const u32x4& a, '
                     'const u32x4& b) {\n    acc[0] = __uint_as_float(__float_as_uint(acc[0]) ^ a.x ^ b.x);\n}\n') + s[j:]
    elif v in ('wonly', 'l2only'):
        sub('const bool ok = e < p.c_in && roff[pb] >= 0;', 'const bool ok = false;')
    if v in ('aonly', 'l2only'):
        sub('(size_t)(n0 + drow) * p.k_total + dchunk * EPC;', '(size_t)0 + dchunk * EPC;')
        sub('const size_t wpass = (size_t)128 * p.k_total;', 'const size_t wpass = 0;')
    if v in ('wsc1', 'wnt'):     # weight DMA bypasses L1 (sc1) / non-temporal
        mod = 'sc1' if v == 'wsc1' else 'nt'
        sub('__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst /* wave-uniform */) {',
            '__device__ __forceinline__ void glds16w(const void* gsrc, unsigned lds_dst) {\n    unsigned keep;\n'
            '    asm volatile("s_mov_b32 %0, m0\\n\\ts_mov_b32 m0, %2\\n\\ts_nop 0\\n\\t'
            'global_load_lds_dwordx4 %1, off ' + mod + '\\n\\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");\n}\n'
            '__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst /* wave-uniform */) {')
        sub('glds16(wbase + q * wpass + ((size_t)tap * p.c_pad + c0), sbase + q * 128 * 64);',
            'glds16w(wbase + q * wpass + ((size_t)tap * p.c_pad + c0), sbase + q * 128 * 64);')
    if v == 'clip_l2only':   # clip kernel: weights alias row 0, activation tiles from the zero page
        sub('    const T* wbase = reinterpret_cast<const T*>(p.w) + (size_t)(n0 + drow) * p.k_total + dchunk * EPC;\n    const size_t wpass = (size_t)128 * p.k_total;\n    const unsigned lds_base = (unsigned)(size_t)lds;\n    const unsigned lds_wave = lds_base + (unsigned)(16 * wave) * 64;\n    int aoff[3];',
            '    const T* wbase = reinterpret_cast<const T*>(p.w) + dchunk * EPC;\n    const size_t wpass = 0;\n    const unsigned lds_base = (unsigned)(size_t)lds;\n    const unsigned lds_wave = lds_base + (unsigned)(16 * wave) * 64;\n    int aoff[3];')
        sub('const bool ok = e < p.c_in && aoff[q] >= 0;', 'const bool ok = false;')
    if v == 'clip_nogather':   # clip kernel: B fragments from identity rows (no table): isolates the gather cost
        sub('return ((j & 1) ? (w >> 16) : (w & 0xffffu)) ^ cx;', 'return (unsigned)lds_swz64(wrow0 + j * 16 + lrow, lchunk) + 0u * w;')
    if v == 'narrow_noload':   # 4-wave kernels: activations from the zero page, weight rows alias row 0
        sub('const bool ok_ = kval_ && roff[pb] >= 0;', 'const bool ok_ = false;')
        sub('const T* wbase = reinterpret_cast<const T*>(p.w) + (size_t)(n0 + row0) * p.k_total + chunk * EPC;\n    const size_t wpass = (size_t)32 * p.k_total;',
            'const T* wbase = reinterpret_cast<const T*>(p.w) + chunk * EPC;\n    const size_t wpass = 0;')
    if v == 'nostore':         # LDS epilogue: skip the global stores of the output tile
        sub('if (q < BM * CPR && row < rows_valid && m < p.M && n < p.c_out)\n                    *reinterpret_cast<u32x4*>(outp',
            'if (q < BM * CPR && row < rows_valid && m < p.M && n < p.c_out && r[u].x == 0x12345678u)\n                    *reinterpret_cast<u32x4*>(outp')
    if v == 'nores':           # LDS epilogue: residual tile from the zero page
        sub('const bool ok = q < BM * CPR && row < rows_valid && m < p.M && n < p.c_out;', 'const bool ok = false;')
    if v == 'ring_noload':     # ring kernel: all operands from L1-resident lines
        sub('const T* wbase = reinterpret_cast<const T*>(p.w) + (size_t)(n0 + drow) * p.k_total + dchunk * EPC;\n    const size_t wpass = (size_t)128 * p.k_total;\n    const unsigned lds_base = (unsigned)(size_t)lds;\n    const unsigned lds_wave = lds_base + (unsigned)(16 * wave) * 64;          // this wave',
            'const T* wbase = reinterpret_cast<const T*>(p.w) + dchunk * EPC;\n    const size_t wpass = 0;\n    const unsigned lds_base = (unsigned)(size_t)lds;\n    const unsigned lds_wave = lds_base + (unsigned)(16 * wave) * 64;          // this wave')
        sub('const bool ok = e < p.c_in && roff[pb] >= 0;', 'const bool ok = false;')
    if v == 'ring_noepi':      # ring kernel: no epilogue at all (one impossible store keeps the accumulators live)
        sub('        epilogue_lds<T, BN, BM, MJ, 512, G::template lds_bytes<T>()>(p, lds, acc, n0, m0, wch0, wrow0, lane, tid);\n        return;',
            '        if (acc[0][0][0] == 1.2345e-30f) p.out[0] = 1;\n        return;')
    if v == 'ring_nok':        # ring kernel: skip the K loop (epilogue only)
        sub('    const int nloc = s_end - s_begin;\n    const int sub_per_tap = 2 * p.steps_per_tap;', '    const int nloc = 0 * (s_end - s_begin);\n    const int sub_per_tap = 2 * p.steps_per_tap;')
    if v == 'clip_nobar2':     # clip kernel: drop the second barrier of a sub-step (racy: timing only)
        sub("            if (!LAG) CP360_CLIP_HEAD(REFILL) else CP360_CLIP_TAIL()                                       \\\n            __builtin_amdgcn_sched_barrier(0);                                                             \\\n            __builtin_amdgcn_s_barrier();                                                                  \\\n",
            "            if (!LAG) CP360_CLIP_HEAD(REFILL) else CP360_CLIP_TAIL()                                       \\\n            __builtin_amdgcn_sched_barrier(0);                                                             \\\n")
    if v == 'clip_center':     # clip kernel: every tap reads the centre tap's rows (consecutive rows: no bank conflicts) - the upper bound of what a conflict-free layout of the resident tile can gain (timing only)
        sub('src = cubepad_src(f, y + t / 3, x + t % 3, geom);          // pixel index inside the clip', 'src = cubepad_src(f, y + 1, x + 1, geom);')
    if v == 'clip_notab':      # clip kernel: no per-tap table loads (entries of tap 0 throughout)
        sub('            load_ent(tap);                                                                                 \\\n        }', '        }')
    if v == 'clip_decode_head':   # clip kernel: decode the B addresses at the START of a HEAD (the round-1 placement) instead of in the TAIL
        sub("            const unsigned char* As = lds + stage * G::WSTAGE;                                             \\\n            _Pragma(\"unroll\") for (int i = 0; i < 4; ++i)                                                  \\\n                a[i] = *reinterpret_cast<const u32x4*>(As + lds_swz64(wch0 + i * 16 + lrow, lchunk));      \\\n            _Pragma(\"unroll\") for (int j = 0; j < JH; ++j)                                                 \\\n                b[j] = *reinterpret_cast<const u32x4*>(lds + ba[j]);",
            "            const unsigned char* As = lds + stage * G::WSTAGE;                                             \\\n            decode();                                                                                      \\\n            _Pragma(\"unroll\") for (int i = 0; i < 4; ++i)                                                  \\\n                a[i] = *reinterpret_cast<const u32x4*>(As + lds_swz64(wch0 + i * 16 + lrow, lchunk));      \\\n            _Pragma(\"unroll\") for (int j = 0; j < JH; ++j)                                                 \\\n                b[j] = *reinterpret_cast<const u32x4*>(lds + ba[j]);")
        sub("            decode();                                                                                      \\\n        }", "        }")
    if v == 'clip_stamps':     # clip kernel: s_memtime stamps around the two halves and the two barriers of a sub-step
        sub('__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};',
            '__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};\n'
            '__device__ unsigned long long g_stamp[8][8];\n'
            'extern "C" int cp360_debug_stamps(unsigned long long* out_host, int reset) {\n'
            '    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamp), sizeof(g_stamp)) != hipSuccess) return -7;\n'
            '    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)) != hipSuccess) return -7; }\n'
            '    return 0;\n}\n'
            '#define CP360_STAMP(K) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; '
            'asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); '
            'st_acc[K] += (unsigned)(t_ - st_prev); st_prev = t_; }')
        sub("            if (LAG) asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");                                    \\\n            __builtin_amdgcn_s_barrier();                                                                  \\\n            __builtin_amdgcn_sched_barrier(0);                                                             \\\n            if (!LAG) CP360_CLIP_HEAD(REFILL) else CP360_CLIP_TAIL()                                       \\\n            __builtin_amdgcn_sched_barrier(0);                                                             \\\n            __builtin_amdgcn_s_barrier();                                                                  \\\n            __builtin_amdgcn_sched_barrier(0);                                                             \\\n            if (!LAG) CP360_CLIP_TAIL() else CP360_CLIP_HEAD(REFILL)                                       \\\n",
            "            CP360_STAMP(0)                                                                                 \\\n            if (LAG) asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");                                    \\\n            __builtin_amdgcn_s_barrier();                                                                  \\\n            __builtin_amdgcn_sched_barrier(0);                                                             \\\n            CP360_STAMP(1)                                                                                 \\\n            if (!LAG) CP360_CLIP_HEAD(REFILL) else CP360_CLIP_TAIL()                                       \\\n            __builtin_amdgcn_sched_barrier(0);                                                             \\\n            CP360_STAMP(2)                                                                                 \\\n            __builtin_amdgcn_s_barrier();                                                                  \\\n            __builtin_amdgcn_sched_barrier(0);                                                             \\\n            CP360_STAMP(3)                                                                                 \\\n            if (!LAG) CP360_CLIP_TAIL() else CP360_CLIP_HEAD(REFILL)                                       \\\n            CP360_STAMP(4)                                                                                 \\\n")
        sub("        constexpr int YW = 2 * (G::NW - 2);\n        const int na = xw ? 3 : 2;\n        int it = 0;",
            "        constexpr int YW = 2 * (G::NW - 2);\n        const int na = xw ? 3 : 2;\n        int it = 0;\n"
            "        unsigned st_acc[5] = {0u, 0u, 0u, 0u, 0u};\n        unsigned long long st_prev;\n"
            "        asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(st_prev) :: \"memory\");")
        sub("        if (LAG) CP360_CLIP_TAIL()\n#undef CP360_CLIP_STEP",
            "        if (LAG) CP360_CLIP_TAIL()\n        if (lane == 0) {\n            for (int k = 0; k < 5; ++k) atomicAdd(&g_stamp[wave][k], (unsigned long long)st_acc[k]);\n"
            "            atomicAdd(&g_stamp[wave][7], (unsigned long long)nloc);\n        }\n#undef CP360_CLIP_STEP")
    if v == 'clip_phases':     # clip kernel: per-workgroup stamps of prologue / K loop / epilogue (s_memrealtime, 100 MHz, chip-wide)
        sub('template <int MODE> struct ClipGeom {',
            '__device__ unsigned long long g_phase[1024 * 2 * 4];\n'
            'extern "C" int cp360_debug_phases(unsigned long long* out_host, int n) {\n'
            '    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase), (size_t)n * 8) == hipSuccess ? 0 : -7;\n}\n'
            '#define CP360_PHASE(K) { if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 1024) { __builtin_amdgcn_sched_barrier(0); '
            'g_phase[(blockIdx.x * 2 + (wave >> 2)) * 4 + (K)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } }\n'
            'template <int MODE> struct ClipGeom {')
        sub("        constexpr int YW = 2 * (G::NW - 2);\n        const int na = xw ? 3 : 2;\n        int it = 0;",
            "        constexpr int YW = 2 * (G::NW - 2);\n        const int na = xw ? 3 : 2;\n        int it = 0;\n        CP360_PHASE(1)")
        sub("        if (LAG) CP360_CLIP_TAIL()\n#undef CP360_CLIP_STEP", "        if (LAG) CP360_CLIP_TAIL()\n        CP360_PHASE(2)\n#undef CP360_CLIP_STEP")
        # kernel entry / exit: stamps around the body calls
        sub("    constexpr int JHC = CLIP_JH;                       // MFMA columns issued in the load half of a sub-step (see clip_body)\n",
            "    constexpr int JHC = CLIP_JH;\n    CP360_PHASE(0)\n")
        i = s.index('        else          clip_body<T, 9, JHC, true, 0>(p, lds, n0, clip, split, wave, lane, tid, (wave - 4) * 64, 1);')
        j = s.index('\n', s.index('}', i)) + 1
        s = s[:j] + '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    CP360_PHASE(3)\n' + s[j:]
    if v == 'clip_loadprio':   # clip kernel: a wave raises its priority for its LOAD half (LDS reads, DMA issue) and drops it for COMPUTE
        sub("            const unsigned char* As = lds + stage * G::WSTAGE;                                             \\\n",
            "            const unsigned char* As = lds + stage * G::WSTAGE;                                             \\\n            __builtin_amdgcn_s_setprio(2);                                                                 \\\n")
        sub("            stage = stage == G::NW - 1 ? 0 : stage + 1;                                                    \\\n            ++tap;",
            "            __builtin_amdgcn_s_setprio(0);                                                                 \\\n            stage = stage == G::NW - 1 ? 0 : stage + 1;                                                    \\\n            ++tap;")
    if v == 'clip_lagloadprio':   # only the lagging (younger) waves raise their priority for their LOAD half
        sub("            const unsigned char* As = lds + stage * G::WSTAGE;                                             \\\n",
            "            const unsigned char* As = lds + stage * G::WSTAGE;                                             \\\n            if (LAG) __builtin_amdgcn_s_setprio(2);                                                        \\\n")
        sub("            stage = stage == G::NW - 1 ? 0 : stage + 1;                                                    \\\n            ++tap;",
            "            if (LAG) __builtin_amdgcn_s_setprio(0);                                                        \\\n            stage = stage == G::NW - 1 ? 0 : stage + 1;                                                    \\\n            ++tap;")
    if v == 'slab_f16emu':     # precision experiment: split-K partial sums rounded to fp16 before they are stored (still f32 slabs)
        sub('store4(p.partial + ((size_t)split * p.M + m) * p.c_out + (p.slab_rows ? slab_col(n) : n), v);',
            '{ for (int e_ = 0; e_ < 4; ++e_) v[e_] = (float)(_Float16)v[e_]; store4(p.partial + ((size_t)split * p.M + m) * p.c_out + (p.slab_rows ? slab_col(n) : n), v); }')
    if v == 'slab_bf16emu':
        sub('store4(p.partial + ((size_t)split * p.M + m) * p.c_out + (p.slab_rows ? slab_col(n) : n), v);',
            '{ for (int e_ = 0; e_ < 4; ++e_) v[e_] = __uint_as_float(((unsigned)f32_to_bf16(v[e_])) << 16); store4(p.partial + ((size_t)split * p.M + m) * p.c_out + (p.slab_rows ? slab_col(n) : n), v); }')
    if v == 'clip_noslab':     # ablation: the split-K slab stores of the clip kernel never execute (accumulators stay live)
        i = s.index('constexpr int CLIP_JH = 0;')
        j = s.rindex('store4(p.partial + ((size_t)split * p.M + m) * p.c_out + (p.slab_rows ? slab_col(n) : n), v);', 0, i)
        s = s[:j] + 'if (v[0] == 1.2345e-30f) ' + s[j:]
    if v.startswith('clip_jh'):   # clip kernel: MFMA columns issued in the load half (0 = pure load / compute halves)
        sub('constexpr int CLIP_JH = 0;', 'constexpr int CLIP_JH = %d;' % int(v[7:]))
    if v == 'clip_nw7':        # clip kernel: seven weight stages
        sub('static constexpr int BN = 256, BM = FACE ? 256 : HALF ? 192 : 304, NW = 6, NA = 2;', 'static constexpr int BN = 256, BM = FACE ? 256 : HALF ? 192 : 304, NW = 7, NA = 2;')
    if v == 'clip_nw5':
        sub('static constexpr int BN = 256, BM = FACE ? 256 : HALF ? 192 : 304, NW = 6, NA = 2;', 'static constexpr int BN = 256, BM = FACE ? 256 : HALF ? 192 : 304, NW = 5, NA = 2;')
    if v == 'clip_prio':       # clip kernel: static priority for the lagging half (waves 4-7)
        sub('        if (wave < 4) clip_body<T, 10, JHC, false, false>', '        if (wave >= 4) __builtin_amdgcn_s_setprio(1);\n        if (wave < 4) clip_body<T, 10, JHC, false, false>')
    if v == 'fullline':
        sub('const int drow = 16 * wave + (lane >> 2);', 'const int drow = 16 * wave + (lane >> 3);')
        sub('const int dchunk = (lane & 3) ^ ((0 - ((4 * wave + (lane >> 4)) & 3)) & 3);', 'const int dchunk = lane & 7;')
        sub('glds16(wbase + q * wpass + ((size_t)tap * p.c_pad + c0), sbase + q * 128 * 64);',
            'glds16(wbase + q * wpass + ((size_t)tap * p.c_pad + (c0 & ~(2 * BKS - 1))) + '
            '(size_t)((c0 / BKS) & 1) * 8 * p.k_total, sbase + q * 128 * 64);')
        sub('            const int e = c0 + dchunk * EPC;\n            const bool ok = e < p.c_in && roff[pb] >= 0;\n'
            '            const T* src = ok ? in + (size_t)roff[pb] + e : reinterpret_cast<const T*>(g_zero16);\n'
            '            glds16(src, sbase + BN * 64 + pb * 128 * 64);',
            '            const int e = (c0 & ~(2 * BKS - 1)) + dchunk * EPC;\n'
            '            const bool ok = e < p.c_in && roff[pb] >= 0;\n'
            '            const T* src = ok ? in + (size_t)roff[pb] + e : reinterpret_cast<const T*>(g_zero16);\n'
            '            glds16(src, sbase + BN * 64 + pb * 128 * 64);')
open(path, 'w').write(s)
