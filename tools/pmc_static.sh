# PMC passes over the static stage (bench.py --static-only, 64 frames) on the GPU box: per kernel MFMA busy, waits, issue, LDS conflicts, L2 hit
# rate, clock.  Separate rocprofv3 runs per counter set (--pmc with --kernel-trace only).   bash tools/pmc_static.sh [out_txt] [env VAR=val ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_static
TXT=${1:-$R/gpurun_out/pmc_static.txt}
shift
for kv in "$@"; do export $kv; done
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --static-only --sequential --no-secondary --no-cpu-baseline --steps 2 --warmup 1 > $OUT/p$i.log 2>&1
done
cd $R
python3 $R/tools/pmc_static_summary.py $OUT "$TXT"
