#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + two PMC passes for bench.py.
# Usage: tools/collect_profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{trace,fetch,write}/... (CSV).  PMC passes are collected
# on their own (no sys/hip/hsa tracing next to --pmc), FETCH_SIZE and WRITE_SIZE in
# separate passes (TCC slots: MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $OUT/write.log 2>&1
cd $R
find $OUT -name "*.csv" | head -20
grep -h '"metric"' $OUT/trace.log | tail -1
