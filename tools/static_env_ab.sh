# Same-box A/B of the static stage by ENVIRONMENT switches (GPU box):  bash tools/static_env_ab.sh "<name>:<VAR=val VAR2=val>" ...
# Runs `bench.py --static-only --sequential` (64 frames) REPS times per variant, alternating, then one rocprofv3 --kernel-trace --stats
# pass per variant for the per-kernel averages (top TOP kernels).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${REPS:-3}); do
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  echo "== $name (rep $rep): $(env $envs python3 $R/bench.py --static-only --sequential --no-secondary --no-cpu-baseline --steps 20 --warmup 3 2>&1 | grep '"metric"' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per 64-frame pass')")"
done
done
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  O=/tmp/se_$name/prof; rm -rf $O; mkdir -p /tmp/se_$name
  ( export $envs; rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --static-only --sequential --no-secondary --no-cpu-baseline --steps 10 --warmup 3 > /tmp/se_$name/log 2>&1 )
  echo "== $name per kernel"
  python3 - "$O" ${TOP:-16} <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats.csv'), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2])]:
    print('   %-60s %5s calls  avg %8.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
