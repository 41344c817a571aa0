# Run ON THE GPU BOX: build the -DL1_STAMPS diagnostic variant of csrc/l1block.hip and print the in-kernel phase timeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
D=/tmp/l1_stamps; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DL1_STAMPS $EXTRA -c $C/l1block.hip -o $D/l1block.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so $(ls $C/*.o | grep -v l1block.o) $D/l1block.o
CP360_LIB=$D/libcp360.so python3 $R/tools/l1_stamps.py
