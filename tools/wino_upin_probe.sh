# Cache policy of the Winograd GEMM's weight stream (GPU box): CP360_WINO_UPIN = sub-steps at the head of every workgroup's U block
# that keep the default policy, the rest non-temporal (1000 = all default, as before round 5's second session); cell update times.
#   bash tools/wino_upin_probe.sh "1000 0 2 4 6 8 12"
R=${GRAFT_REPO_ROOT:-$(pwd)}
NS=${1:-"1000 0 2 4 6 8 12"}
for rep in 1 2 3; do
  for n in $NS; do echo "== upin $n"; CP360_WINO_UPIN=$n python3 $R/tools/wino_cell_probe.py --iters 40 2>&1 | grep "wino cell" | tail -1; done
done
