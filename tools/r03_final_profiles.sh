#!/bin/bash
# Run ON THE GPU BOX: the round-3 evidence set (summaries -> gpurun_out/, copied to profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03f}
cd $R
bash tools/collect_profile.sh $TAG --no-secondary > gpurun_out/collect_$TAG.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bf16 gpurun_out/traffic_bf16.json > gpurun_out/${TAG}_summary.log 2>&1
bash tools/collect_profile.sh ${TAG}_fp32 --no-secondary --precision fp32 --steps 2 --warmup 1 > gpurun_out/collect_${TAG}_fp32.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_${TAG}_fp32 gpurun_out/${TAG}_fp32 gpurun_out/traffic_fp32_b4_w7.json > gpurun_out/${TAG}_fp32_summary.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/static_$TAG -- python3 $R/bench.py --static-only --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $R/gpurun_out/static_$TAG.log 2>&1
cd $R
python3 tools/static_timeline.py gpurun_out/static_$TAG gpurun_out/${TAG}_static_timeline.md > /dev/null
bash tools/pmc_clip_final.sh $R/gpurun_out/${TAG}_pmc_clip.txt > gpurun_out/pmc_$TAG.log 2>&1
python3 tools/hbm_kernels.py --md gpurun_out/${TAG}_hbm_kernels.md --json gpurun_out/${TAG}_hbm_kernels.json > gpurun_out/hbm_$TAG.log 2>&1
tail -3 gpurun_out/${TAG}_summary.log; tail -5 gpurun_out/${TAG}_pmc_clip.txt
