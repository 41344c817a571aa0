R=${GRAFT_REPO_ROOT:-$(pwd)}
J='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "frames/s", d["ms_per_step"], "ms | sequential", d["config"]["sequential"]["value"], d["config"]["stage_ms"])'
for rep in 1 2; do
for fc in 0 32 16; do
  echo "== frame_chunk $fc: $(python3 $R/bench.py --frame-chunk $fc --no-secondary --no-cpu-baseline --steps 20 --warmup 3 2>&1 | grep '"metric"' | python3 -c "$J")"
done
done
