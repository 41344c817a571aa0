mkdir -p gpurun_out/r2e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layer1_fused" > gpurun_out/r2e/tests.log 2>&1; tail -2 gpurun_out/r2e/tests.log
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2e/static_prof -- python3 $R/bench.py --static-only --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $R/gpurun_out/r2e/static.log 2>&1
cd $R; python tools/static_timeline.py gpurun_out/r2e/static_prof gpurun_out/r2e/static_timeline.md | grep -E "l1block|pass:" | cut -c1-120
