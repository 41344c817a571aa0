mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c5s -- python3 $R/bench.py --static-only --no-cpu-baseline --no-secondary --steps 3 --warmup 2 --equi 2048x4096 --cube 512 --clips 1 --frames 16 --precision fp16 > $R/gpurun_out/c5s.log 2>&1
python3 $R/tools/static_timeline.py $R/gpurun_out/c5s $R/gpurun_out/c5_static_timeline.md > /dev/null 2>&1
cut -c1-105 $R/gpurun_out/c5_static_timeline.md | head -70
