# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the default bench workload (no secondary lines, no CPU baseline) and the top of
# the per-kernel table.   bash tools/quick_stats.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/qs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/bench.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats.csv'), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print('%-72s %5s calls  %9.3f ms  avg %8.2f us  %5s %%' % (r['Name'][:72], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage']))
PY
grep -h '"metric"' $OUT/bench.log | tail -1 | cut -c1-400
