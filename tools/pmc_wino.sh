# PMC passes of the Winograd-domain GEMM (wino_gemm_kernel, 16 x [384 x K] x [K x 4000]) on the GPU box: MFMA busy, waits, LDS,
# L2 hit rate, HBM fetch / write.  Separate rocprofv3 runs per counter set (--pmc with --kernel-trace only).
#   bash tools/pmc_wino.sh [out_txt] [extra wino_probe.py args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_wino
TXT=${1:-$R/gpurun_out/pmc_wino.txt}
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/wino_probe.py --iters 2 "$@" > $OUT/p$i.log 2>&1
done
cd $R
python3 $R/tools/pmc_wino_summary.py $OUT "$TXT" wino_gemm $R/gpurun_out/pmc_clock_bf16_b4_w7.json
