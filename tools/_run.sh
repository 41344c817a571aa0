python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stem or layer1_fused or resnet or cam" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --equi 2048x4096 --cube 512 --clips 1 --precision fp16 2>&1 | tail -1 | cut -c1-200
R=$(pwd); mkdir -p gpurun_out/r2i
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2i/static_prof -- python3 $R/bench.py --static-only --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $R/gpurun_out/r2i/static.log 2>&1
cd $R; python tools/static_timeline.py gpurun_out/r2i/static_prof gpurun_out/r2i/static_timeline.md | grep -E "l1block|stem_kernel" | cut -c1-110 | head -5
python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-160
