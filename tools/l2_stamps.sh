# Run ON THE GPU BOX: build the -DL2_STAMPS diagnostic variant of csrc/l2block.hip and print the in-kernel phase timeline of the LAST l2block launch
# of a static pass (layer3's fifth tail: l2block_kernel<256, 14x14>).
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
D=/tmp/l2_stamps; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DL2_STAMPS $EXTRA -c $C/l2block.hip -o $D/l2block.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so $(ls $C/*.o | grep -v l2block.o) $D/l2block.o
CP360_LIB=$D/libcp360.so python3 $R/tools/l2_stamps.py
