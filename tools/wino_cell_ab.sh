# Same-box A/B of the Winograd cell's kernels (GPU box):  bash tools/wino_cell_ab.sh "<name>:<-D flags>" ...
# Each variant = csrc/wino.hip with extra flags (name "head": tools/_exp/wino_head.hip instead, if present), linked with the in-tree
# objects; runs tools/wino_cell_probe.py under rocprofv3 --kernel-trace --stats and prints the per-kernel averages + the probe's line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/cp_360_weakly_supervised_saliency_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  D=/tmp/wc_$name; mkdir -p $D
  src=$C/wino.hip
  if [ "$name" = head ] && [ -f $R/tools/_exp/wino_head.hip ]; then cp $R/tools/_exp/wino_head.hip $C/wino_head_tmp.hip; src=$C/wino_head_tmp.hip; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DWINO_LAB $flags -c $src -o $D/wino.o || exit 1
  rm -f $C/wino_head_tmp.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libcp360.so $(ls $C/*.o | grep -v wino.o) $D/wino.o
done
cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-3}); do
for spec in "$@"; do
  name=${spec%%:*}
  export CP360_LIB=/tmp/wc_$name/libcp360.so
  O=/tmp/wc_$name/prof$rep; rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/wino_cell_probe.py $ARGS > /tmp/wc_$name/log$rep 2>&1
  echo "== $name (rep $rep): $(grep 'wino cell' /tmp/wc_$name/log$rep | tail -1)"
  python3 - "$O" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats.csv'), recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'wino' in r['Name'] and 'pack' not in r['Name']:
        print('   %-40s %5s calls  avg %8.2f us' % (r['Name'].split('::')[-1][:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
done
