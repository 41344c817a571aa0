#!/bin/bash
# PMC summary of the FINAL ConvLSTM kernel (conv_clip_kernel, K = 36000, M = 1176: the dominant kernel of bench.py)
# on the GPU box: MFMA busy, wait cycles, LDS bank conflicts, L2 hit rate, HBM fetch / write.  Counter passes are
# separate rocprofv3 runs (--pmc with --kernel-trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass).
#   tools/pmc_clip_final.sh [out_txt]      (SHAPE=w8.Conv2: the HALF variant at 8x8 faces instead of clstm.Conv2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=${SHAPE:-clstm.Conv2}
OUT=$R/gpurun_out/pmc_clip_final_$SHAPE
TXT=${1:-$R/gpurun_out/pmc_clip_final.txt}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_conv.py --only $SHAPE --iters 3 --clips 4 > $OUT/p$i.log 2>&1
done
cd $R
python3 $R/tools/pmc_clip_summary.py $OUT "$TXT"
