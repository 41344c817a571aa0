#!/bin/bash
# Timing ablations of conv_small_kernel (local build, run on the GPU box with CP360_LIB=...): which part of a K step costs what.
#   tools/exp_small.sh  -> tools/_exp/libcp360_small_{nomma,noglobal,nobar}.so   (most variants compute garbage)
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/tools/_exp
for V in ${VARIANTS:-nomma noglobal nolds whot ahot}; do
  D=/tmp/exp_small_$V
  rm -rf $D && mkdir -p $D && cp -r $R/cp_360_weakly_supervised_saliency_amd/csrc $D/ && cd $D/csrc && rm -f conv_small.o libcp360.so
  sed -i 's#"../../include/cp360_internal.h"#"'$R'/include/cp360_internal.h"#' common.h
  python3 - "$V" <<'PY'
import sys
v = sys.argv[1]
s = open('conv_small.hip').read()
def sub(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
if v in ('nomma', 'bare'):      # no MFMAs: the fragments are consumed by an empty asm
    sub('                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i][e]), __uint_as_float(b[j][e]), acc[i][j], 0, 0, 0);',
        '                    asm volatile("" :: "v"(a[i][e]), "v"(b[j][e]));')
if v in ('noglobal', 'bare'):   # every global load hits the 16 zero bytes of g_zero16 (one L1-resident line)
    sub('"v"(wvoff), "s"(wsb), "v"(av), "s"(asb)', '"v"(0u), "s"(reinterpret_cast<const T*>(g_zero16)), "v"(0u), "s"(reinterpret_cast<const T*>(g_zero16))')
if v == 'prio':       # the load block runs at raised priority (is the loader starved of issue slots by the partner's MFMA stream?)
    sub('            frag_load(BUF, fa, fb);                                                                  \\\n', '            __builtin_amdgcn_s_setprio(3); frag_load(BUF, fa, fb);                                   \\\n')
    sub('            next_gload(SA, SB, SM);                                  /* step IT+5 */                 \\\n', '            next_gload(SA, SB, SM); __builtin_amdgcn_s_setprio(0);                                    \\\n')
if v == 'nobar':      # no barrier inside the K loop (races: timing only)
    sub('            __syncthreads();                                                                         \\\n', '                                                                                                     \\\n')
if v == 'whot':       # weights: every tile reads the FIRST tile's 64 rows (hot in every XCD's L2); activations real
    sub('const T* wtile = reinterpret_cast<const T*>(p.w) + (size_t)n0 * p.k_total;', 'const T* wtile = reinterpret_cast<const T*>(p.w);')
if v == 'ahot':       # activations: every tile reads the FIRST tile's pixels (offset table of tile 0); weights real
    sub('            const int m = m0 + r;\n            int off = -1;', '            const int m = r;\n            int off = -1;')
if v == 'nolds':      # fragments are not re-read from LDS (stores and barriers stay)
    sub('            frag_load(0, fa, fb);\n            lds_store(1, sa1, sb1);', '            if (it == 0) frag_load(0, fa, fb);\n            lds_store(1, sa1, sb1);')
    sub('            frag_load(1, fa, fb);\n            lds_store(0, sa0, sb0);', '            lds_store(0, sa0, sb0);')
open('conv_small.hip', 'w').write(s)
PY
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c conv_small.hip -o conv_small.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_exp/libcp360_small_$V.so *.o
  echo built $V
done
