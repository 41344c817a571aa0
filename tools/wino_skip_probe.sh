# Round-5 experiment (GPU box): what skipping the never-read (position, tile) pairs of 7x7 faces would buy the Winograd GEMM.
#   base            the shipped kernel
#   base + zeros    the same with those V rows zeroed (less data toggling only)
#   abl8            timing-only build: positions of the last transform row / column run 10 (8 for the corner) of the 12 column blocks
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
D=/tmp/wv_abl8; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DWINO_ABL=8 -c $C/wino.hip -o $D/wino.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so $(ls $C/*.o | grep -v wino.o) $D/wino.o
for rep in 1 2 3; do
  echo "== base";            python3 $R/tools/wino_probe.py --no-check 2>&1 | grep "wino gemm  " | tail -1
  echo "== base zero rows";  python3 $R/tools/wino_probe.py --no-check --zero-unneeded 2>&1 | grep "wino gemm  " | tail -1
  echo "== abl8";            CP360_LIB=$D/libcp360.so python3 $R/tools/wino_probe.py --no-check 2>&1 | grep "wino gemm  " | tail -1
done
