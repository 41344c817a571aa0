# Same-box A/B of the static stage (GPU box):  bash tools/static_ab.sh "<name>:<-D flags>" ...
# Each variant = the sources named in FILES="a.hip b.hip" rebuilt with the flags and linked
# with the other in-tree objects; runs `bench.py --static-only --sequential` (64 frames, 20 passes) three times per variant, alternating,
# then one rocprofv3 --kernel-trace --stats pass per variant for the per-kernel averages.
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
FILES=${FILES:?set FILES="a.hip b.hip": the sources the variant flags apply to}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  D=/tmp/sv_$name; mkdir -p $D
  skip=""
  for f in $FILES; do
    src=$C/$f
    # variant "head": tools/_exp/<file>_head.hip (an older version of the source, untracked) instead of the tree's, if present
    if [ "$name" = head ] && [ -f $R/tools/_exp/${f%.hip}_head.hip ]; then cp $R/tools/_exp/${f%.hip}_head.hip $C/_head_tmp_$f; src=$C/_head_tmp_$f; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -c $src -o $D/${f%.hip}.o || exit 1
    rm -f $C/_head_tmp_$f
    skip="$skip -e /${f%.hip}.o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so $(ls $C/*.o | grep -v $skip) $D/*.o
done
for rep in 1 2 3; do
for spec in "$@"; do
  name=${spec%%:*}
  echo "== $name (rep $rep): $(CP360_LIB=/tmp/sv_$name/libcp360.so python3 $R/bench.py --static-only --sequential --no-secondary --no-cpu-baseline --steps 20 --warmup 3 2>&1 | grep '"metric"' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per 64-frame pass')")"
done
done
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}
  export CP360_LIB=/tmp/sv_$name/libcp360.so
  O=/tmp/sv_$name/prof; rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --static-only --sequential --no-secondary --no-cpu-baseline --steps 10 --warmup 3 > /tmp/sv_$name/log 2>&1
  echo "== $name per kernel"
  python3 - "$O" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats.csv'), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print('   %-60s %5s calls  avg %8.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
