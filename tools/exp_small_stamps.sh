#!/bin/bash
# Diagnostic build of conv_small_kernel with s_memtime stamps (tools/_exp/libcp360_small_stamps.so; never the product build):
# per wave, cycles of a K step spent (1) waiting at the barrier, (2) in the trailing waves' MFMA block, (3) in the load block
# (fragment reads, staging wait, LDS stores, address arithmetic, global loads), (4) in the leading waves' MFMA block.
# Stamps go to a __device__ array of their own; read with cp360_debug_small_stamps (tools/exp_small_stamps.py).
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/tools/_exp
D=/tmp/exp_small_stamps
rm -rf $D && mkdir -p $D && cp -r $R/cp_360_weakly_supervised_saliency_amd/csrc $D/ && cd $D/csrc && rm -f conv_small.o libcp360.so
sed -i 's#"../../include/cp360_internal.h"#"'$R'/include/cp360_internal.h"#' common.h
python3 - <<'PY'
s = open('conv_small.hip').read()
def sub(a, b, n=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b) if n == 0 else s.replace(a, b, n)
sub('namespace {\n', 'namespace {\n__device__ unsigned long long g_small_stamps[8 * 8];\n'
    '#define STAMP(v) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); v = __builtin_amdgcn_s_memtime(); }\n')
# loop halves: instrument both
for buf_a, buf_b in (('0', '1'), ('1', '0')):
    pass
sub('        set_tap(tap);\n        next_gload(sa0, sb0, sm0);', '        unsigned long long T0 = __builtin_amdgcn_s_memtime(), R0 = __builtin_amdgcn_s_memrealtime();\n        set_tap(tap);\n        next_gload(sa0, sb0, sm0);')
sub('        if (half) mma_half<T>(acc, fa, fb);                         // trailing waves: the last step',
    '        if (half) mma_half<T>(acc, fa, fb);                         // trailing waves: the last step\n        st_bar = __builtin_amdgcn_s_memtime() - T0; st_mb = __builtin_amdgcn_s_memrealtime() - R0; st_n = nloc;')
sub('''    if (nloc > 0) {
        u32x4 fa[2], fb[2];''', '''    unsigned long long st_bar = 0, st_mb = 0, st_ld = 0, st_ma = 0, st_n = 0, st_f = 0, st_w = 0, st_s = 0;
    if (nloc > 0) {
        u32x4 fa[2], fb[2];''')
sub('''    // ---- the trailing waves hand their sums over through LDS''', '''    if (lane == 0) {
        atomicAdd(&g_small_stamps[wave * 8 + 0], st_bar); atomicAdd(&g_small_stamps[wave * 8 + 1], st_mb);
        atomicAdd(&g_small_stamps[wave * 8 + 2], st_ld); atomicAdd(&g_small_stamps[wave * 8 + 3], st_ma);
        atomicAdd(&g_small_stamps[wave * 8 + 7], st_n); atomicAdd(&g_small_stamps[wave * 8 + 4], st_f); atomicAdd(&g_small_stamps[wave * 8 + 5], st_w); atomicAdd(&g_small_stamps[wave * 8 + 6], st_s);
    }
    // ---- the trailing waves hand their sums over through LDS''')
s += '''
extern "C" int cp360_debug_small_stamps(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[64] = {}; return hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), z, sizeof(z)) == hipSuccess ? 0 : -7; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -7;
}
'''
s = s.replace('namespace {\n__device__ unsigned long long g_small_stamps[8 * 8];', '__device__ unsigned long long g_small_stamps[8 * 8];\nnamespace {', 1)
open('conv_small.hip', 'w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c conv_small.hip -o conv_small.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_exp/libcp360_small_stamps.so *.o
echo built stamps
