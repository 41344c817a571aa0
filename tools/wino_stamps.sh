# Run ON THE GPU BOX: build the -DWINO_STAMPS diagnostic variant of csrc/wino.hip and print the in-kernel phase timeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
D=/tmp/wv_stamps; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DWINO_LAB -DWINO_STAMPS $EXTRA -c $C/wino.hip -o $D/wino.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so $(ls $C/*.o | grep -v wino.o) $D/wino.o
CP360_LIB=$D/libcp360.so python3 $R/tools/wino_stamps.py
