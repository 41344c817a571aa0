mkdir -p gpurun_out
R=$(pwd)
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/collect_profile.sh r02d --no-secondary > gpurun_out/collect_r02d.log 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/static_prof_e -- python3 $R/bench.py --static-only --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $R/gpurun_out/static_prof_e.log 2>&1 )
python3 tools/static_timeline.py gpurun_out/static_prof_e gpurun_out/r02e_static_timeline.md > /dev/null 2>&1
python3 bench.py > gpurun_out/r02d_bench.json.log 2> gpurun_out/r02d_bench.err
tail -c 300 gpurun_out/r02d_bench.json.log
