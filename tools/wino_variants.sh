# A/B builds of csrc/wino.hip on the GPU box: each variant = extra -D flags; links the in-tree objects with the variant's wino.o.
#   bash tools/wino_variants.sh "<name>:<flags>" ...   then runs tools/wino_probe.py with CP360_LIB=<variant> (ARGS = its arguments)
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  D=/tmp/wv_$name; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DWINO_LAB $flags -c $C/wino.hip -o $D/wino.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so $(ls $C/*.o | grep -v wino.o) $D/wino.o
done
for rep in 1 2 3; do
for spec in "$@"; do
  name=${spec%%:*}
  echo "== $name"
  CP360_LIB=/tmp/wv_$name/libcp360.so python3 $R/tools/wino_probe.py $ARGS 2>&1 | grep "wino gemm" | tail -1
done
done
