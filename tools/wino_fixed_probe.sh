# The fixed part of a Winograd GEMM launch (GPU box): time against K (sub-steps), with and without the slab stores.
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
D=/tmp/wv_nostore; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DWINO_LAB -DWINO_ABL=32 -c $C/wino.hip -o $D/wino.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so $(ls $C/*.o | grep -v wino.o) $D/wino.o
for k in 32 128 512 1024 2016 4000; do
  echo "== K $k stores:    $(python3 $R/tools/wino_probe.py --no-check --cin $k 2>&1 | grep 'wino gemm  ' | tail -1 | sed -E 's/.*x16: //' | cut -c1-50)"
  echo "== K $k no stores: $(CP360_LIB=$D/libcp360.so python3 $R/tools/wino_probe.py --no-check --cin $k 2>&1 | grep 'wino gemm  ' | tail -1 | sed -E 's/.*x16: //' | cut -c1-50)"
done
