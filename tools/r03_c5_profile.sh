#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 evidence for BASELINE config C5's per-GPU shard (one 16-frame 2048x4096 clip,
# 6x512^2 faces, fp16): kernel-trace stats + FETCH / WRITE passes (tools/collect_profile.sh) and the per-launch
# timeline of one static-stage pass (tools/static_timeline.py).  Summaries -> gpurun_out/, copied to profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03_c5}
C5="--equi 2048x4096 --cube 512 --clips 1 --frames 16 --precision fp16 --no-secondary"
bash $R/tools/collect_profile.sh $TAG $C5
python3 $R/tools/summarize_profile.py $R/gpurun_out/prof_$TAG $R/gpurun_out/${TAG}_fp16 > $R/gpurun_out/${TAG}_summary.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/static_$TAG -- python3 $R/bench.py --static-only --no-cpu-baseline $C5 --steps 3 --warmup 2 > $R/gpurun_out/static_$TAG.log 2>&1
cd $R
python3 $R/tools/static_timeline.py $R/gpurun_out/static_$TAG $R/gpurun_out/${TAG}_static_timeline.md > /dev/null
tail -3 $R/gpurun_out/static_$TAG.log
