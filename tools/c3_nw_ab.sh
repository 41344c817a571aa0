R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
mkdir -p /tmp/v7
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DCLIP_NW=7 -c $C/conv_igemm.hip -o /tmp/v7/conv_igemm.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/v7/libcp360.so $(ls $C/*.o | grep -v conv_igemm.o) /tmp/v7/conv_igemm.o
J='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "frames/s", d["ms_per_step"], "ms", d.get("stage_ms"))'
for rep in 1 2; do
  echo "== nw6: $(python3 $R/bench.py --clips 1 --sequential --no-secondary --no-cpu-baseline --steps 20 2>&1 | grep '"metric"' | python3 -c "$J")"
  echo "== nw7: $(CP360_LIB=/tmp/v7/libcp360.so python3 $R/bench.py --clips 1 --sequential --no-secondary --no-cpu-baseline --steps 20 2>&1 | grep '"metric"' | python3 -c "$J")"
  echo "== nw6 conv: $(python3 $R/tools/bench_conv.py --clips 1 --only clstm.Conv2 --iters 50 2>&1 | tail -1)"
  echo "== nw7 conv: $(CP360_LIB=/tmp/v7/libcp360.so python3 $R/tools/bench_conv.py --clips 1 --only clstm.Conv2 --iters 50 2>&1 | tail -1)"
done
