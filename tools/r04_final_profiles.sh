#!/bin/bash
# Run ON THE GPU BOX: the round-4 evidence set of the headline workload (summaries -> gpurun_out/, copied to profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04z}
cd $R
bash tools/collect_profile.sh $TAG --no-secondary > gpurun_out/collect_$TAG.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bf16 gpurun_out/traffic_bf16.json > gpurun_out/${TAG}_summary.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/static_$TAG -- python3 $R/bench.py --static-only --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $R/gpurun_out/static_$TAG.log 2>&1
cd $R
python3 tools/static_timeline.py gpurun_out/static_$TAG gpurun_out/${TAG}_static_timeline.md --flops 3216 --peak 2500 > /dev/null
tail -3 gpurun_out/${TAG}_summary.log
