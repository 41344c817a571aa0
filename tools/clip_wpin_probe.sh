# Cache policy of the direct clip-resident kernel's weight stream (GPU box): CP360_CLIP_WPIN = sub-steps at the head of every workgroup's
# share that keep the default policy, the rest non-temporal (100000 = all default).  Literal C3 (one clip), the fp32 shard, and the
# 4-clip shard forced onto the direct kernels (CP360_WINO=0).
R=${GRAFT_REPO_ROOT:-$(pwd)}
J='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "frames/s", d["ms_per_step"], "ms", d.get("stage_ms"))'
for rep in 1 2; do
for n in 100000 0 4 8 16; do
  echo "== C3 one clip bf16, wpin $n: $(CP360_CLIP_WPIN=$n python3 $R/bench.py --clips 1 --sequential --no-secondary --no-cpu-baseline --steps 20 2>&1 | grep '"metric"' | python3 -c "$J")"
done
for n in 100000 4; do
  echo "== fp32 4 clips, wpin $n: $(CP360_CLIP_WPIN=$n python3 $R/bench.py --precision fp32 --sequential --no-secondary --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep '"metric"' | python3 -c "$J")"
  echo "== bf16 4 clips direct (CP360_WINO=0), wpin $n: $(CP360_WINO=0 CP360_CLIP_WPIN=$n python3 $R/bench.py --sequential --no-secondary --no-cpu-baseline --steps 10 2>&1 | grep '"metric"' | python3 -c "$J")"
done
done
