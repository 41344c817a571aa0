# Run ON THE GPU BOX: the round-6 evidence set (summaries -> gpurun_out/, copied to profiles/ afterwards).
#   bash tools/r06_final_profiles.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06z}
cd $R
# 1. headline workload: kernel stats + FETCH / WRITE passes, per-kernel HBM bytes, the dominant kernel's traffic file
# (--sequential: one batch at a time on one stream, so that a kernel's duration in the trace is its own - in the default pipelined
#  form launches of the two stages overlap and every duration includes the time the kernel shared the chip)
bash tools/collect_profile.sh $TAG --no-secondary --sequential > gpurun_out/collect_$TAG.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bf16 gpurun_out/traffic_bf16_b4_w7.json > gpurun_out/${TAG}_summary.log 2>&1
# 2. every launch of one static-stage pass
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/static_$TAG -- python3 $R/bench.py --static-only --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $R/gpurun_out/static_$TAG.log 2>&1
cd $R
python3 tools/static_timeline.py gpurun_out/static_$TAG gpurun_out/${TAG}_static_timeline.md --flops 3216 --peak 2500 > /dev/null
# 3. counters of the dominant kernel (+ the clock / pipe-busy pair bench.py quotes)
bash tools/pmc_wino.sh gpurun_out/r06_pmc_wino.txt > gpurun_out/pmc_wino_$TAG.log 2>&1
# 4. BASELINE config C5's per-GPU shard (one 2048x4096 clip, cube 512, fp16): kernel stats + traffic of its Winograd GEMM
bash tools/collect_profile.sh ${TAG}_c5 --no-secondary --sequential --equi 2048x4096 --cube 512 --clips 1 --precision fp16 --steps 3 --warmup 1 > gpurun_out/collect_${TAG}_c5.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_${TAG}_c5 gpurun_out/${TAG}_c5_fp16 gpurun_out/traffic_fp16_b1_w16.json > gpurun_out/${TAG}_c5_summary.log 2>&1
tail -3 gpurun_out/${TAG}_summary.log; tail -12 gpurun_out/r06_pmc_wino.txt; tail -3 gpurun_out/${TAG}_c5_summary.log
# 5. counters of every static-stage kernel (matrix pipe busy, waits, LDS conflicts, L2 hit rate, HBM bytes)
bash tools/pmc_static.sh gpurun_out/${TAG}_pmc_static.txt > gpurun_out/pmc_static_$TAG.log 2>&1
tail -30 gpurun_out/${TAG}_pmc_static.txt
