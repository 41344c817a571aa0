#!/bin/bash
# Run ON THE GPU BOX: BASELINE config C2 (one frame = 6 faces, static path only) as a per-launch timeline, fp32 and fp16.
# Usage: bash tools/r04_c2_profile.sh <tag>      -> gpurun_out/<tag>_c2_{fp32,fp16}_static_timeline.md
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04a}
cd /tmp && export TMPDIR=/tmp
for P in fp32 fp16; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c2_${TAG}_$P -- python3 $R/bench.py --static-only --clips 1 --frames 1 \
      --precision $P --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > $R/gpurun_out/c2_${TAG}_$P.log 2>&1
  PEAK=2500; [ $P = fp32 ] && PEAK=157.3
  python3 $R/tools/static_timeline.py $R/gpurun_out/c2_${TAG}_$P $R/gpurun_out/${TAG}_c2_${P}_static_timeline.md --flops 50.25 --peak $PEAK > /dev/null
  grep -h '"metric"' $R/gpurun_out/c2_${TAG}_$P.log | tail -1 | cut -c1-300
done
