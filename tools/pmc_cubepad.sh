#!/bin/bash
# usage: pmc_cp.sh <tag> [env assignments...]   (FETCH/WRITE + a few SQ counters; every rocprofv3 under timeout)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/pmc_cubepad_$TAG
mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/cubepad_bench.py pmc > $OUT/p$i.log 2>&1
done
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
agg = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc_cubepad_%s/p*/**/*counter_collection.csv' % sys.argv[1], recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'cubepad' not in r['Kernel_Name']:
            continue
        k = (r['Kernel_Name'][5:34], r['Grid_Size'], r['Counter_Name'])
        agg.setdefault(k, []).append(float(r['Counter_Value']))
for (kn, gs, cn), v in agg.items():
    print('%s %-30s grid=%-9s %-22s avg=%.5g' % (sys.argv[1], kn, gs, cn, sum(v) / len(v)))
PY
