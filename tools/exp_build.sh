#!/bin/bash
# Kernel experiments on the GPU box: build variants of libcp360.so with parts of the ring kernel
# disabled or altered, to see which resource bounds it (timing only - most variants compute garbage).
# Usage: tools/exp_build.sh <variant>[+<variant>...] ; prints the .so path (use it with CP360_LIB=...).
#   base     : unmodified
#   nomma    : every bf16 MFMA replaced by one XOR on its operands (fragment reads stay live)
#   l2only   : wonly + aonly (no HBM / hardly any L2 traffic)
#   wonly    : activations come from the zero page, weights real
#   aonly    : weight rows all alias row 0, activations real
#   fullline : ring kernel DMA fetches 8 rows x 128 B per instruction instead of 16 rows x 64 B
set -e
V=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/exp_$V
rm -rf $D && mkdir -p $D/pkg/csrc $D/include
cp $R/cp_360_weakly_supervised_saliency_amd/csrc/*.hip $R/cp_360_weakly_supervised_saliency_amd/csrc/*.h $D/pkg/csrc/
cp $R/include/cp360.h $R/include/cp360_internal.h $D/include/
cd $D/pkg/csrc
python3 $R/tools/exp_patch.py conv_igemm.hip "$V"
sed -i 's#"../../include/cp360_internal.h"#"'$D'/include/cp360_internal.h"#' common.h
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c $f -o ${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so *.o
echo $D/libcp360.so
