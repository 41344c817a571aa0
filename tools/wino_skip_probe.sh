# Round-5 experiment (GPU box): what skipping the never-read (position, tile) pairs of 7x7 faces would buy the Winograd GEMM.
#   base            the shipped kernel (bare GEMM through cp360_wino_gemm_raw: every row of M is stored)
#   base + zeros    the same with those V rows zeroed (less data toggling only)
#   abl8            timing-only build: positions of the last transform row / column run 10 (8 for the corner) of the 12 column blocks
#                   and store only those
#   abl24           the same MFMA skipping, but all 12 column blocks are stored (separates the MFMAs from the slab stores)
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
for v in 8 24; do
D=/tmp/wv_abl$v; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DWINO_LAB -DWINO_ABL=$v -c $C/wino.hip -o $D/wino.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so $(ls $C/*.o | grep -v wino.o) $D/wino.o
done
for rep in 1 2 3; do
  echo "== base";            python3 $R/tools/wino_probe.py --no-check 2>&1 | grep "wino gemm  " | tail -1
  echo "== base zero rows";  python3 $R/tools/wino_probe.py --no-check --zero-unneeded 2>&1 | grep "wino gemm  " | tail -1
  echo "== abl8";            CP360_LIB=/tmp/wv_abl8/libcp360.so python3 $R/tools/wino_probe.py --no-check 2>&1 | grep "wino gemm  " | tail -1
  echo "== abl24";           CP360_LIB=/tmp/wv_abl24/libcp360.so python3 $R/tools/wino_probe.py --no-check 2>&1 | grep "wino gemm  " | tail -1
done
